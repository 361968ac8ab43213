"""Recording stand-in for `docplex.mp.model.Model` -- TEST INFRASTRUCTURE ONLY.

Implements exactly the calls `FJSP.fluid_model` makes
(environments/class_FJSSP.py:251-274): continuous_var_dict, linear-expression
arithmetic, model.sum, model.min, maximize, add_constraints, solve,
solution.get_value_dict.  It RECORDS the model; `solve()` delegates to the
module-level SOLVE_HOOK(model) -> {key: value}, which the golden generator
points at the product's own LP solver.  See README.md.
"""

SOLVE_HOOK = None


class LinExpr(object):
    __slots__ = ("terms", "const")

    def __init__(self, terms=None, const=0.0):
        self.terms = dict(terms) if terms else {}
        self.const = float(const)

    @staticmethod
    def of(v):
        if isinstance(v, LinExpr):
            return v
        if isinstance(v, Var):
            return LinExpr({v.key: 1.0})
        return LinExpr(None, float(v))

    def __add__(self, other):
        o = LinExpr.of(other)
        t = dict(self.terms)
        for k, c in o.terms.items():
            t[k] = t.get(k, 0.0) + c
        return LinExpr(t, self.const + o.const)

    __radd__ = __add__

    def __neg__(self):
        return LinExpr({k: -c for k, c in self.terms.items()}, -self.const)

    def __sub__(self, other):
        return self + (-LinExpr.of(other))

    def __rsub__(self, other):
        return LinExpr.of(other) + (-self)

    def __mul__(self, f):
        f = float(f)
        return LinExpr({k: c * f for k, c in self.terms.items()}, self.const * f)

    __rmul__ = __mul__

    def __truediv__(self, f):
        f = float(f)
        return LinExpr({k: c / f for k, c in self.terms.items()}, self.const / f)

    def __le__(self, other):
        return Constraint(self - LinExpr.of(other), "<=")

    def __ge__(self, other):
        return Constraint(self - LinExpr.of(other), ">=")


class Var(object):
    __slots__ = ("key", "lb", "ub")

    def __init__(self, key, lb, ub):
        self.key, self.lb, self.ub = key, lb, ub

    def __mul__(self, f):
        return LinExpr({self.key: float(f)})

    __rmul__ = __mul__

    def __add__(self, other):
        return LinExpr.of(self) + other

    __radd__ = __add__


class Constraint(object):
    """expr (sense) 0, constants moved into expr.const"""
    __slots__ = ("expr", "sense")

    def __init__(self, expr, sense):
        self.expr, self.sense = expr, sense


class MinExpr(object):
    def __init__(self, exprs):
        self.exprs = [LinExpr.of(e) for e in exprs]


class Solution(object):
    def __init__(self, values):
        self.values = values

    def get_value_dict(self, var_dict):
        return {k: self.values[k] for k in var_dict}


class Model(object):
    def __init__(self, name=None):
        self.name = name
        self.var_dicts = []
        self.objective = None
        self.sense = None
        self.constraints = []

    def continuous_var_dict(self, keys, lb=0, ub=None, name=None):
        d = {k: Var(k, lb, ub) for k in sorted(keys)}
        self.var_dicts.append(d)
        return d

    def sum(self, it):
        acc = LinExpr()
        for v in it:
            acc = acc + v
        return acc

    def min(self, it):
        return MinExpr(list(it))

    def maximize(self, expr):
        self.objective, self.sense = expr, "max"

    def add_constraints(self, it):
        self.constraints.extend(list(it))

    def solve(self):
        if SOLVE_HOOK is None:
            raise RuntimeError("docplex stand-in: SOLVE_HOOK not set")
        return Solution(SOLVE_HOOK(self))

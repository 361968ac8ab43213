"""ctypes wrapper of oracle/libfjsp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The oracle takes the instance arrays and the fluid solution x as
inputs (x comes from the product's LP through an LP hook, see fjsp_oracle.h).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libfjsp_oracle.so")

SO_FJSSP = 0
SO_SFJSP = 1
MO_FJSSP_DISCRETES = 2
MO_DFJSP = 4
SO_DFJSP = 5


class _Inst(C.Structure):
    _fields_ = [("R", C.c_int), ("M", C.c_int), ("K", C.c_int), ("S", C.c_int),
                ("Jr", C.c_void_p), ("p", C.c_void_p), ("elig_n", C.c_void_p), ("elig_list", C.c_void_p),
                ("count", C.c_void_p), ("arrive", C.c_void_p), ("delivery", C.c_void_p)]


class Trace(C.Structure):
    _fields_ = [("k_sel", C.c_int), ("m_sel", C.c_int), ("job_kind", C.c_int), ("job_n", C.c_int),
                ("step_time", C.c_int), ("delay_time_sum", C.c_int64)]


LP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double))

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise ImportError("%s missing: run __graft_entry__.build() (or `make -C oracle`)" % LIB)
        L = C.CDLL(LIB)
        L.fjo_create.restype = C.c_void_p
        L.fjo_create.argtypes = [C.POINTER(_Inst), C.c_int]
        L.fjo_destroy.argtypes = [C.c_void_p]
        L.fjo_set_lp.argtypes = [C.c_void_p, LP_FN, C.c_void_p]
        L.fjo_set_rng.argtypes = [C.c_void_p, C.c_uint64]
        L.fjo_set_ddt.argtypes = [C.c_void_p, C.c_double]
        L.fjo_state_size.argtypes = [C.c_void_p]
        L.fjo_reset.argtypes = [C.c_void_p, C.c_void_p]
        L.fjo_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int),
                               C.POINTER(Trace)]
        L.fjo_step_mo.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                  C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(Trace)]
        L.fjo_set_dynamic.argtypes = [C.c_void_p] * 5
        L.fjo_step_dyn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                   C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(Trace)]
        L.fjo_energy.argtypes = [C.c_void_p]
        L.fjo_energy.restype = C.c_int64
        L.fjo_set_fixed_x.argtypes = [C.c_void_p, C.c_void_p]
        L.fjo_play_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.fjo_play_many.restype = C.c_long
        L.fjo_step_sf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                  C.POINTER(Trace)]
        for name in ("fjo_step_time", "fjo_step_count", "fjo_makespan", "fjo_completion_time"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.fjo_delay_time_sum.argtypes = [C.c_void_p]
        L.fjo_delay_time_sum.restype = C.c_int64
        L.fjo_machine_time_end.argtypes = [C.c_void_p, C.c_void_p]
        L.fjo_fluid_completed_time.argtypes = [C.c_void_p]
        L.fjo_fluid_completed_time.restype = C.c_double
        L.fjo_fluid_tables.argtypes = [C.c_void_p] * 5
        L.fjo_play.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.fjo_pyset_and_list.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def pyset_and_list(a, b):
    """CPython-3.10 order of list(set(a) & set(b)) as the oracle emulates it."""
    a = np.ascontiguousarray(a, dtype=np.int32)
    b = np.ascontiguousarray(b, dtype=np.int32)
    out = np.zeros(max(len(a), len(b), 1), np.int32)
    n = lib().fjo_pyset_and_list(a.ctypes.data, len(a), b.ctypes.data, len(b), out.ctypes.data)
    return out[:n].tolist()


class OracleEnv(object):
    """One scalar CPU environment.

    `arrays` is an object with R, M, K, S, Jr, p[K,M], elig_n, elig_list[K,M],
    count[S,R], arrive, delivery (int32 numpy).  `lp` is a callable
    (Q[K], n_now[K]) -> x[K,M] called at every reset_object_add; passing a fixed
    array x instead serves static single-order instances.
    """

    def __init__(self, arrays, lp, variant=SO_FJSSP, rng_seed=0, ddt=None):
        self.L = lib()
        a = arrays
        self._keep = [np.ascontiguousarray(v, dtype=np.int32) for v in
                      (a.Jr, a.p, a.elig_n, a.elig_list, a.count, a.arrive, a.delivery)]
        inst = _Inst(a.R, a.M, a.K, a.S, *[v.ctypes.data for v in self._keep])
        self.R, self.M, self.K, self.S = a.R, a.M, a.K, a.S
        self.h = self.L.fjo_create(C.byref(inst), variant)
        if not self.h:
            raise ValueError("fjo_create failed (unsupported variant?)")
        if callable(lp):
            self._lp_py = lp
        else:
            fixed = np.ascontiguousarray(lp, dtype=np.float64).reshape(a.K, a.M)
            self._lp_py = lambda Q, n_now: fixed
        K, M = a.K, a.M

        def _hook(user, Q, n_now, x):
            try:
                xs = np.ascontiguousarray(self._lp_py(np.ctypeslib.as_array(Q, (K,)).copy(),
                                                      np.ctypeslib.as_array(n_now, (K,)).copy()), dtype=np.float64)
                np.ctypeslib.as_array(x, (K * M,))[:] = xs.reshape(-1)
                return 0
            except Exception:  # surfaced as rc -11 from fjo_reset / fjo_step
                return 1
        self._hook = LP_FN(_hook)
        self.L.fjo_set_lp(self.h, self._hook, None)
        if not callable(lp) and a.S == 1:          # static instance: no interpreter call per reset
            self._keep.append(fixed)
            self.L.fjo_set_fixed_x(self.h, fixed.ctypes.data)
        self.L.fjo_set_rng(self.h, rng_seed & (2 ** 64 - 1))
        if ddt is not None:
            self.L.fjo_set_ddt(self.h, float(ddt))
        if variant == MO_DFJSP:
            dyn = [np.ascontiguousarray(v, dtype=np.int32) for v in (a.power, a.idle_power, a.bk_n, np.asarray(a.bk).reshape(-1))]
            if len(dyn[3]) == 0:
                dyn[3] = np.zeros(2, np.int32)
            self._keep += dyn
            self.L.fjo_set_dynamic(self.h, *[v.ctypes.data for v in dyn])
        self.state_size = self.L.fjo_state_size(self.h)
        self.done = False

    def __del__(self):
        if getattr(self, "h", None):
            self.L.fjo_destroy(self.h)
            self.h = None

    def reset(self):
        st = np.zeros(self.state_size)
        rc = self.L.fjo_reset(self.h, st.ctypes.data)
        if rc:
            raise RuntimeError("oracle reset failed rc=%d" % rc)
        self.done = False
        return st

    def step(self, action):
        st = np.zeros(self.state_size)
        rew, done, tr = C.c_double(), C.c_int(), Trace()
        rc = self.L.fjo_step(self.h, int(action[0]), int(action[1]), st.ctypes.data, C.byref(rew), C.byref(done),
                             C.byref(tr))
        if rc:
            raise RuntimeError("oracle step failed rc=%d" % rc)
        self.done = bool(done.value)
        self.trace = tr
        return st, rew.value, self.done

    def step_mo(self, action, weight_vector, completion=None, tardiness=None):
        st = np.zeros(self.state_size)
        rew, done, tr = C.c_double(), C.c_int(), Trace()
        rc = self.L.fjo_step_mo(self.h, int(action), float(weight_vector[0]), float(weight_vector[1]),
                                -1.0 if completion is None else float(completion),
                                -1.0 if tardiness is None else float(tardiness),
                                st.ctypes.data, C.byref(rew), C.byref(done), C.byref(tr))
        if rc:
            raise RuntimeError("oracle step_mo failed rc=%d" % rc)
        self.done = bool(done.value)
        self.trace = tr
        return st, rew.value, self.done

    def step_sf(self, action):
        st = np.zeros(self.state_size)
        rew, done, tr = C.c_double(), C.c_int(), Trace()
        rc = self.L.fjo_step_sf(self.h, int(action), st.ctypes.data, C.byref(rew), C.byref(done), C.byref(tr))
        if rc:
            raise RuntimeError("oracle step_sf failed rc=%d" % rc)
        self.done = bool(done.value)
        self.trace = tr
        return st, rew.value, self.done

    def step_dyn(self, action, reward_policy, completion=None, tardiness=None, energy=None):
        st = np.zeros(self.state_size)
        rew, done, tr = C.c_double(), C.c_int(), Trace()
        rc = self.L.fjo_step_dyn(self.h, int(action[0]), int(action[1]), int(reward_policy),
                                 0.0 if completion is None else float(completion),
                                 0.0 if tardiness is None else float(tardiness),
                                 0.0 if energy is None else float(energy),
                                 st.ctypes.data, C.byref(rew), C.byref(done), C.byref(tr))
        if rc:
            raise RuntimeError("oracle step_dyn failed rc=%d" % rc)
        self.done = bool(done.value)
        self.trace = tr
        return st, rew.value, self.done

    @property
    def energy_consumption(self):
        return self.L.fjo_energy(self.h)

    def play(self, actions):
        """reset + full episode in C (timing loop). Returns (steps, reward_sum)."""
        actions = np.ascontiguousarray(actions, dtype=np.uint8)
        acc = C.c_double()
        n = self.L.fjo_play(self.h, actions.ctypes.data, len(actions), C.byref(acc))
        if n < 0:
            raise RuntimeError("oracle play failed rc=%d" % n)
        return n, acc.value

    @property
    def step_time(self):
        return self.L.fjo_step_time(self.h)

    @property
    def step_count(self):
        return self.L.fjo_step_count(self.h)

    @property
    def delay_time_sum(self):
        return self.L.fjo_delay_time_sum(self.h)

    @property
    def makespan(self):
        return self.L.fjo_makespan(self.h)

    @property
    def completion_time(self):
        return self.L.fjo_completion_time(self.h)

    @property
    def fluid_completed_time(self):
        return self.L.fjo_fluid_completed_time(self.h)

    def machine_time_end(self):
        out = np.zeros(self.M, np.int32)
        self.L.fjo_machine_time_end(self.h, out.ctypes.data)
        return out

    def fluid_tables(self):
        rate = np.zeros((self.K, self.M)); arr = np.zeros((self.K, self.M))
        rs = np.zeros(self.K); ts = np.zeros(self.K)
        self.L.fjo_fluid_tables(self.h, rate.ctypes.data, arr.ctypes.data, rs.ctypes.data, ts.ctypes.data)
        return rate, arr, rs, ts


def play_many(envs_and_actions, reps):
    """Timing helper: `reps` full episodes of every (OracleEnv, actions u8[T, 2]) pair in ONE C call (the GIL is
    released for its whole duration, so host threads scale).  Returns the number of steps played."""
    n = len(envs_and_actions)
    acts = [np.ascontiguousarray(a, dtype=np.uint8) for _, a in envs_and_actions]
    T = min(len(a) for a in acts)
    hs = (C.c_void_p * n)(*[e.h for e, _ in envs_and_actions])
    ps = (C.c_void_p * n)(*[a.ctypes.data for a in acts])
    steps = lib().fjo_play_many(hs, n, ps, T, int(reps))
    if steps < 0:
        raise RuntimeError("oracle play_many failed rc=%d" % steps)
    return steps

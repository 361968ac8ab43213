"""Host-side instance sets: the array form of the reference's instance sources.

    Data(path, file_name)   environments/SO_DFJSP_instance_read.py:6-89  -> InstanceSet.load_csv
    Instance(DDT, M, S)     environments/Instance_generate.py:19-94      -> InstanceSet.generate (seeded)
    FJSP.fluid_model()      environments/class_FJSSP.py:246-280          -> InstanceSet.solve_fluid

All parsing, generation and the LP run in the native library; this module only
moves arrays across the C ABI.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import GenParams, check


def bench_10x5_params(ddt=1.0):
    """BASELINE config 2 / SURVEY 8d: R=10 kinds x 1 job, M=5, J_r~U{3..5}, p~U{1..20}, one order."""
    return GenParams(R_min=10, R_max=10, J_min=3, J_max=5, M=5, p_min=1, p_max=20,
                     N_min=1, N_max=1, S=1, DDT=ddt, t_si_min=100.0, t_si_max=200.0)


def reference_generator_params(ddt, M, S):
    """The in-env generator's own distributions (Instance_generate.py:39-66)."""
    return GenParams(R_min=3, R_max=12, J_min=3, J_max=5, M=M, p_min=40, p_max=400,
                     N_min=5, N_max=50, S=S, DDT=ddt, t_si_min=100.0, t_si_max=200.0)


class InstanceArrays(object):
    """Plain numpy view of one instance (k-major, k = koff[r] + j)."""

    def __init__(self, R, M, K, S, Jr, p, elig_n, elig_list, count, arrive, delivery, ddt, x):
        self.R, self.M, self.K, self.S = R, M, K, S
        self.Jr, self.p, self.elig_n, self.elig_list = Jr, p, elig_n, elig_list
        self.count, self.arrive, self.delivery, self.ddt, self.x = count, arrive, delivery, ddt, x

    @property
    def koff(self):
        return np.concatenate(([0], np.cumsum(self.Jr))).astype(np.int32)

    @property
    def kind_task_tuple(self):
        return tuple((r, j) for r in range(self.R) for j in range(int(self.Jr[r])))


class InstanceSet(object):
    """n problem instances owned by the native library (fjsp_instances)."""

    def __init__(self, n):
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        check(self._lib.fjsp_instances_create(int(n), C.byref(self._h)))
        self.n = int(n)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.fjsp_instances_destroy(h)
            self._h = None

    def __len__(self):
        return self.n

    @property
    def handle(self):
        return self._h

    # -- sources ---------------------------------------------------------------
    def load_csv(self, i, path, file_name):
        check(self._lib.fjsp_instances_load_csv(self._h, int(i), str(path).encode(), str(file_name).encode()))
        return self

    def generate(self, i, seed, params):
        check(self._lib.fjsp_instances_generate(self._h, int(i), int(seed) & (2 ** 64 - 1), C.byref(params)))
        return self

    def generate_range(self, seed_base, params, first=0, n=None):
        n = self.n - first if n is None else n
        for i in range(first, first + n):
            self.generate(i, seed_base + i, params)
        return self

    def set_raw(self, i, Jr, p, elig_n, elig_list, count, arrive, delivery, ddt=1.0):
        Jr = np.ascontiguousarray(Jr, dtype=np.int32)
        count = np.ascontiguousarray(count, dtype=np.int32).reshape(-1)
        arrive = np.ascontiguousarray(arrive, dtype=np.int32)
        delivery = np.ascontiguousarray(delivery, dtype=np.int32)
        R, S = len(Jr), len(arrive)
        K = int(Jr.sum())
        p = np.ascontiguousarray(p, dtype=np.int32).reshape(K, -1)
        M = p.shape[1]
        elig_n = np.ascontiguousarray(elig_n, dtype=np.int32)
        elig_list = np.ascontiguousarray(elig_list, dtype=np.int32).reshape(K, M)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._lib.fjsp_instances_set_raw(self._h, int(i), R, M, S, ptr(Jr), ptr(p), ptr(elig_n), ptr(elig_list),
                                               ptr(count), ptr(arrive), ptr(delivery), float(ddt)))
        return self

    # -- fluid LP ----------------------------------------------------------------
    def solve_fluid(self, first=0, n=None, n_threads=0):
        n = self.n - first if n is None else n
        check(self._lib.fjsp_instances_solve_fluid(self._h, int(first), int(n), int(n_threads)))
        return self

    def set_x(self, i, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        check(self._lib.fjsp_instances_set_x(self._h, int(i), x.ctypes.data_as(C.c_void_p)))
        return self

    # -- read back ---------------------------------------------------------------
    def dims(self, i):
        d = (C.c_int32 * 6)()
        check(self._lib.fjsp_instances_dims(self._h, int(i), C.byref(d)))
        return dict(R=d[0], M=d[1], K=d[2], S=d[3], jobs0=d[4], jobs=d[5])

    def arrays(self, i):
        d = self.dims(i)
        R, M, K, S = d["R"], d["M"], d["K"], d["S"]
        Jr = np.zeros(R, np.int32)
        p = np.zeros((K, M), np.int32)
        elig_n = np.zeros(K, np.int32)
        elig_list = np.zeros((K, M), np.int32)
        count = np.zeros((S, R), np.int32)
        arrive = np.zeros(S, np.int32)
        delivery = np.zeros(S, np.int32)
        x = np.zeros((K, M), np.float64)
        ddt = C.c_double()
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._lib.fjsp_instances_get(self._h, int(i), ptr(Jr), ptr(p), ptr(elig_n), ptr(elig_list), ptr(count),
                                           ptr(arrive), ptr(delivery), C.byref(ddt), ptr(x)))
        out = InstanceArrays(R, M, K, S, Jr, p, elig_n, elig_list, count, arrive, delivery, ddt.value, x)
        dd = (C.c_int32 * 2)()
        check(self._lib.fjsp_instances_dynamic_dims(self._h, int(i), C.byref(dd)))
        if dd[0]:       # MO_DFJSP_instance_read.py extras: powers + breakdown windows
            out.power = np.zeros((K, M), np.int32)
            out.idle_power = np.zeros(M, np.int32)
            out.bk_n = np.zeros(M, np.int32)
            out.bk = np.zeros((max(dd[1], 1), 2), np.int32)
            check(self._lib.fjsp_instances_get_dynamic(self._h, int(i), ptr(out.power), ptr(out.idle_power), ptr(out.bk_n),
                                                       ptr(out.bk)))
            out.bk = out.bk[:dd[1]]
        return out

    def set_dynamic(self, i, power, idle_power, bk_n, bk):
        power = np.ascontiguousarray(power, dtype=np.int32)
        idle_power = np.ascontiguousarray(idle_power, dtype=np.int32)
        bk_n = np.ascontiguousarray(bk_n, dtype=np.int32)
        bk = np.ascontiguousarray(bk, dtype=np.int32).reshape(-1)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._lib.fjsp_instances_set_dynamic(self._h, int(i), ptr(power), ptr(idle_power), ptr(bk_n),
                                                   ptr(bk) if len(bk) else None))
        return self

    def generate_machine_data(self, i, seed, max_windows=0, window_gap=(50, 400), window_len=(5, 60)):
        """Machine data for a generated instance, drawn like the reference's generator: processing power
        U{10..200} per eligible pair, idle power U{1..9} per machine (Instance_generate.py:61-66,90-91).  The
        generator has no breakdowns (that data only exists in CSV folders, MO_DFJSP_instance_read.py:56-73);
        max_windows > 0 adds up to that many sorted, disjoint windows per machine for synthetic batches."""
        a = self.arrays(i)
        rs = np.random.RandomState(int(seed) & 0x7FFFFFFF)
        power = np.where(a.p > 0, rs.randint(10, 201, a.p.shape), 0)
        idle = rs.randint(1, 10, a.M)
        bk_n = rs.randint(0, max_windows + 1, a.M) if max_windows > 0 else np.zeros(a.M, np.int64)
        bk = []
        for m in range(a.M):
            t = 0
            for _ in range(int(bk_n[m])):
                st = t + int(rs.randint(window_gap[0], window_gap[1] + 1))
                en = st + int(rs.randint(window_len[0], window_len[1] + 1))
                bk.append((st, en))
                t = en
        return self.set_dynamic(i, power, idle, bk_n, np.array(bk, np.int32).reshape(-1, 2))


def fluid_lp(Jr, p, Q, n_now):
    """One fluid LP for a live state (class_FJSSP.py:246-280). Returns (x[K,M], objective)."""
    lib = _capi.lib()
    Jr = np.ascontiguousarray(Jr, dtype=np.int32)
    K = int(Jr.sum())
    p = np.ascontiguousarray(p, dtype=np.int32).reshape(K, -1)
    M = p.shape[1]
    Q = np.ascontiguousarray(Q, dtype=np.int32)
    n_now = np.ascontiguousarray(n_now, dtype=np.int32)
    x = np.zeros((K, M), np.float64)
    obj = C.c_double()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.fjsp_fluid_lp(len(Jr), M, ptr(Jr), ptr(p), ptr(Q), ptr(n_now), ptr(x), C.byref(obj)))
    return x, obj.value

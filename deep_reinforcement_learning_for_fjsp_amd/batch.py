"""EnvBatch: N environments resident in HBM, driven through the C ABI.

PyTorch is plumbing here: it owns the device tensors handed to the library
(`.data_ptr()`) and the stream the kernels are queued on
(`torch.cuda.current_stream()`).  All environment arithmetic happens in
libfjsp_amd.so (csrc/fjsp_kernels.hip); nothing in this module computes a step
on the host.
"""
import ctypes as C

import torch

from . import _capi
from ._capi import check

VARIANT_SO_FJSSP = 0
VARIANT_SO_SFJSP = 1
VARIANT_MO_FJSSP_DISCRETES = 2
VARIANT_MO_DFJSP = 4
VARIANT_SO_DFJSP = 5

ST_BAD_TASK_RULE = 1
ST_BAD_MACHINE_RULE = 2
ST_STEP_AFTER_DONE = 4
ST_NO_EVENT = 8


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class EnvBatch(object):
    """fjsp_env handle + the device tensors it writes into."""

    def __init__(self, instances, n_envs, first=0, n_inst=None, variant=VARIANT_SO_FJSSP, device=0, rng_seed=0):
        if not torch.cuda.is_available():
            raise RuntimeError("EnvBatch needs an MI355X: the environment kernels have no CPU path")
        self._lib = _capi.lib()
        self.instances = instances
        n_inst = len(instances) - first if n_inst is None else n_inst
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self._h = C.c_void_p()
        check(self._lib.fjsp_env_create(instances.handle, int(first), int(n_inst), int(n_envs), int(variant),
                                        self.device_index, int(rng_seed) & (2 ** 64 - 1), C.byref(self._h)))
        self.N = int(n_envs)
        self.n_inst = int(n_inst)
        self.first = int(first)
        self.variant = int(variant)
        self.rng_seed = int(rng_seed)
        self.state_size = self._lib.fjsp_env_state_size(self._h)
        self.step_bytes = int(self._lib.fjsp_env_step_bytes(self._h))
        f64 = dict(dtype=torch.float64, device=self.device)
        self.state = torch.zeros(self.N, self.state_size, **f64)
        self.reward = torch.zeros(self.N, **f64)
        self.done = torch.ones(self.N, dtype=torch.uint8, device=self.device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.fjsp_env_destroy(h)
            self._h = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def env_seed(self, e):
        """random.choice stream seed of env e (matches fjsp_kernels.hip bind())."""
        return (self.rng_seed + e * 1000003) & (2 ** 64 - 1)

    # -- reset / step ------------------------------------------------------------
    def reset(self, mask=None, out=None):
        """reset(): SO_FJSSP.py:51-76 for every env (or those with mask != 0). Returns f64[N, S]."""
        out = self.state if out is None else out
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        check(self._lib.fjsp_env_reset(self._h, _ptr(mask), _ptr(out), self._stream()))
        if mask is None:
            self.done.zero_()
        else:
            self.done.masked_fill_(mask.bool(), 0)
        return out

    def step(self, actions, autoreset=False, state_out=None, reward_out=None, done_out=None, mo=None):
        """step(action): SO_FJSSP.py:168-265.  actions: uint8[N, 2] device tensor.  For the MO variant
        actions[:, 0] is the flat action and `mo` (f64[N, 4] = w0, w1, completion, tardiness; <= 0 = None)
        carries step()'s extra arguments (MO_FJSSP_discretes.py:88); for MO_DFJSP `mo` is f64[N, 4] =
        reward_policy, completion, tardiness, energy_consumption (MO_DFJSP_breakdown.py:189)."""
        if actions.dtype != torch.uint8 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.uint8).contiguous()
        state_out = self.state if state_out is None else state_out
        reward_out = self.reward if reward_out is None else reward_out
        done_out = self.done if done_out is None else done_out
        check(self._lib.fjsp_env_step(self._h, _ptr(actions), _ptr(mo), 1 if autoreset else 0, _ptr(state_out),
                                      _ptr(reward_out), _ptr(done_out), self._stream()))
        return state_out, reward_out, done_out

    def rollout(self, actions, trace=True, rewards=True, mo=None, state=True):
        """T fused steps in one launch. actions: uint8[T, N, 2]. Returns (trace_km i16[T,N,2], reward f64[T,N], state).
        state=False: no final state (the fused kernel then skips the observation; reset before stepping again)."""
        if actions.dtype != torch.uint8 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(device=self.device, dtype=torch.uint8).contiguous()
        T = actions.shape[0]
        tr = torch.full((T, self.N, 2), -1, dtype=torch.int16, device=self.device) if trace else None
        rw = torch.zeros(T, self.N, dtype=torch.float64, device=self.device) if rewards else None
        check(self._lib.fjsp_env_rollout(self._h, _ptr(actions), _ptr(mo), int(T), _ptr(tr), _ptr(rw),
                                         _ptr(self.state) if state else None, self._stream()))
        return tr, rw, (self.state if state else None)

    # -- read back -----------------------------------------------------------------
    def read(self):
        """dict of per-env attributes the reference's agents / harnesses read."""
        i32 = dict(dtype=torch.int32, device=self.device)
        out = dict(delay_time_sum=torch.zeros(self.N, dtype=torch.int64, device=self.device),
                   makespan=torch.zeros(self.N, **i32), completion_time=torch.zeros(self.N, **i32),
                   step_time=torch.zeros(self.N, **i32), step_count=torch.zeros(self.N, **i32),
                   done=torch.zeros(self.N, dtype=torch.uint8, device=self.device),
                   status=torch.zeros(self.N, dtype=torch.int32, device=self.device))
        check(self._lib.fjsp_env_read(self._h, _ptr(out["delay_time_sum"]), _ptr(out["makespan"]),
                                      _ptr(out["completion_time"]), _ptr(out["step_time"]), _ptr(out["step_count"]),
                                      _ptr(out["done"]), _ptr(out["status"]), self._stream()))
        if self.variant == VARIANT_MO_DFJSP:
            out["energy_consumption"] = torch.zeros(self.N, dtype=torch.int64, device=self.device)
            check(self._lib.fjsp_env_energy(self._h, _ptr(out["energy_consumption"]), self._stream()))
        return out

    def set_lp_threads(self, n_threads):
        """Host threads of the order-arrival LP service (0 = all cores)."""
        check(self._lib.fjsp_env_set_lp_threads(self._h, int(n_threads)))

    @property
    def lp_solves(self):
        return int(self._lib.fjsp_env_lp_solves(self._h))

    def machine_time_end(self):
        d = self.instances.dims(self.first)
        mp = max(self.instances.dims(self.first + i)["M"] for i in range(self.n_inst)) if self.n_inst > 1 else d["M"]
        out = torch.zeros(self.N, mp, dtype=torch.int32, device=self.device)
        check(self._lib.fjsp_env_machine_time_end(self._h, _ptr(out), int(mp), self._stream()))
        return out

    def fluid_tables(self, i):
        import numpy as np
        d = self.instances.dims(self.first + (i % self.n_inst))
        K, M = d["K"], d["M"]
        rate = np.zeros((K, M)); arr = np.zeros((K, M)); rs = np.zeros(K); ts = np.zeros(K)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        check(self._lib.fjsp_env_fluid_tables(self._h, int(i), p(rate), p(arr), p(rs), p(ts)))
        return rate, arr, rs, ts

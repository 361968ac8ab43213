"""EnvBatch: N environments resident in HBM, driven through the C ABI.

PyTorch is plumbing here: it owns the device tensors handed to the library
(`.data_ptr()`) and the stream the kernels are queued on
(`torch.cuda.current_stream()`).  All environment arithmetic happens in
libfjsp_amd.so (csrc/fjsp_kernels.hip); nothing in this module computes a step
on the host.
"""
import ctypes as C

import numpy as np

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


ENV_SEED_STRIDE = 1000003          # env e draws random.choice from the stream seeded rng_seed + e * ENV_SEED_STRIDE


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _splitmix64(z):
    """numpy uint64 vector form of the splitmix64 finaliser the kernels use (fjsp_kernels.hip)."""
    import numpy as np
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def global_actions(seed, first_env, n_envs, T, n_task_rules, n_machine_rules):
    """uint8[T, n_envs, 2] random rule pairs that are a pure function of (seed, GLOBAL env id, step): a shard
    [first_env, first_env + n_envs) of a larger batch draws exactly what the unsharded batch draws for those
    environments, whatever the number of GPUs."""
    import numpy as np
    g = (np.arange(n_envs, dtype=np.uint64) + np.uint64(first_env))[None, :]
    t = np.arange(T, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        u = _splitmix64(_splitmix64(np.uint64(seed) + g * np.uint64(0x632BE59BD9B4E019)) + t)
    a0 = ((u >> np.uint64(40)) % np.uint64(n_task_rules)).astype(np.uint8)
    a1 = ((u >> np.uint64(8)) % np.uint64(max(n_machine_rules, 1))).astype(np.uint8)
    return np.ascontiguousarray(np.stack([a0, a1], 2))


def _as_input(name, t, shape, dtype, device):
    """An INPUT tensor of the C ABI: converted to the dtype / device / layout the kernels read, shape enforced
    (the library takes raw pointers: a wrong extent would be an out-of-bounds device access, not an exception)."""
    if t is None:
        return None
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
    if t.dtype != dtype or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=dtype).contiguous()
    return t


def _check_output(name, t, shape, dtype, device):
    """An OUTPUT tensor of the C ABI is written in place: it cannot be converted, only checked."""
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise ValueError("%s must be a torch tensor" % name)
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device or not t.is_contiguous():
        raise ValueError("%s must be a contiguous %s tensor of shape %s on %s, got %s %s on %s%s"
                         % (name, dtype, tuple(shape), device, t.dtype, tuple(t.shape), t.device,
                            "" if t.is_contiguous() else " (not contiguous)"))
    return t


class EnvBatch(object):
    """fjsp_env handle + the device tensors it writes into."""

    def __init__(self, instances, n_envs, first=0, n_inst=None, variant=VARIANT_SO_FJSSP, device=0, rng_seed=0,
                 first_env=0):
        """first_env: GLOBAL id of this batch's environment 0 when the batch is one shard of a larger job (one
        rank of `bench.py --gpus N` / examples/train_ppo.py).  The random.choice stream of an environment is a
        function of its global id, so the traces of a sharded job equal the unsharded job's bit for bit."""
        if not torch.cuda.is_available():
            raise RuntimeError("EnvBatch needs an MI355X: the environment kernels have no CPU path")
        self._lib = _capi.lib()
        self.instances = instances
        n_inst = len(instances) - first if n_inst is None else n_inst
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self._h = C.c_void_p()
        self.first_env = int(first_env)
        # the kernels seed env e (local) with seed + e * ENV_SEED_STRIDE: shift the base by the shard's offset
        lib_seed = (int(rng_seed) + self.first_env * ENV_SEED_STRIDE) & (2 ** 64 - 1)
        check(self._lib.fjsp_env_create(instances.handle, int(first), int(n_inst), int(n_envs), int(variant),
                                        self.device_index, lib_seed, C.byref(self._h)))
        self.N = int(n_envs)
        self.n_inst = int(n_inst)
        self.first = int(first)
        self.variant = int(variant)
        self.rng_seed = int(rng_seed)
        self.state_size = self._lib.fjsp_env_state_size(self._h)
        self.step_bytes = int(self._lib.fjsp_env_step_bytes(self._h))
        # 1: stepped by the 16-lane-row kernels (csrc/fjsp_group.hip), 0: one wavefront per environment
        self.kernel_family = int(self._lib.fjsp_env_kernel_family(self._h))
        # 1: the fluid LPs of order arrivals are solved on the device (csrc/fjsp_lp_device.hip), 0: on the host
        self.lp_on_device = int(self._lib.fjsp_env_lp_on_device(self._h))
        f64 = dict(dtype=torch.float64, device=self.device)
        self.state = torch.zeros(self.N, self.state_size, **f64)
        self.reward = torch.zeros(self.N, **f64)
        self.done = torch.ones(self.N, dtype=torch.uint8, device=self.device)
        # the per-step call is launch-bound on the host (a few us): pointers of the default outputs and the shape
        # the action tensor must have are prepared once
        self._p_state, self._p_reward, self._p_done = _ptr(self.state), _ptr(self.reward), _ptr(self.done)
        self._act_shape = torch.Size((self.N, 2))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.fjsp_env_destroy(h)
            self._h = None

    def _stream(self):
        # (the raw handle of torch's CURRENT stream on this device, looked up per call: callers switch streams and
        # capture graphs; torch.cuda.current_stream() builds a Stream object first, five times the cost)
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self.device_index))

    def env_seed(self, e):
        """random.choice stream seed of (local) env e (matches open_env() in fjsp_kernels.hip)."""
        return (self.rng_seed + (self.first_env + e) * ENV_SEED_STRIDE) & (2 ** 64 - 1)

    # -- reset / step ------------------------------------------------------------
    def reset(self, mask=None, out=None):
        """reset(): SO_FJSSP.py:51-76 for every env (or those with mask != 0). Returns f64[N, S]."""
        out = self.state if out is None else _check_output("out", out, (self.N, self.state_size), torch.float64, self.device)
        if mask is not None:
            if torch.is_tensor(mask) and mask.dtype == torch.bool:
                mask = mask.to(torch.uint8)
            mask = _as_input("mask", mask, (self.N,), torch.uint8, self.device)
        check(self._lib.fjsp_env_reset(self._h, _ptr(mask), _ptr(out), self._stream()))
        if mask is None:
            self.done.zero_()
        else:
            self.done.masked_fill_(mask.bool(), 0)
        return out

    def step(self, actions, autoreset=False, state_out=None, reward_out=None, done_out=None, mo=None, state=True,
             trace_out=None):
        """step(action): SO_FJSSP.py:168-265.  actions: uint8[N, 2] device tensor.  For the MO variant
        actions[:, 0] is the flat action and `mo` (f64[N, 4] = w0, w1, completion, tardiness; <= 0 = None)
        carries step()'s extra arguments (MO_FJSSP_discretes.py:88); for MO_DFJSP `mo` is f64[N, 4] =
        reward_policy, completion, tardiness, energy_consumption (MO_DFJSP_breakdown.py:189).
        state=False: no state is returned and the kernel skips the observation (rule policies that never look at
        it); later calls that do return a state are unaffected (the library rebuilds v(t-1) first).
        trace_out: int16[N, 2] tensor that receives the chosen (operation type k, machine m) of every env."""
        if not (torch.is_tensor(actions) and actions.dtype is torch.uint8 and actions.shape == self._act_shape
                and actions.device == self.device and actions.is_contiguous()):
            actions = _as_input("actions", actions, (self.N, 2), torch.uint8, self.device)
        if actions.data_ptr() & 1:                # the kernels read an action pair as one 16-bit word
            actions = actions.clone()
        p_mo = None
        if mo is not None:
            mo = _as_input("mo", mo, (self.N, 4), torch.float64, self.device)
            p_mo = _ptr(mo)
        if state_out is None:
            state_out, p_state = self.state, self._p_state
        else:
            p_state = _ptr(_check_output("state_out", state_out, (self.N, self.state_size), torch.float64, self.device))
        if reward_out is None:
            reward_out, p_reward = self.reward, self._p_reward
        else:
            p_reward = _ptr(_check_output("reward_out", reward_out, (self.N,), torch.float64, self.device))
        if done_out is None:
            done_out, p_done = self.done, self._p_done
        else:
            p_done = _ptr(_check_output("done_out", done_out, (self.N,), torch.uint8, self.device))
        if not state:
            state_out, p_state = None, None
        if trace_out is not None:
            _check_output("trace_out", trace_out, (self.N, 2), torch.int16, self.device)
            rc = self._lib.fjsp_env_step_traced(self._h, C.c_void_p(actions.data_ptr()), p_mo, 1 if autoreset else 0, p_state,
                                                p_reward, p_done, _ptr(trace_out), self._stream())
        else:
            rc = self._lib.fjsp_env_step(self._h, C.c_void_p(actions.data_ptr()), p_mo, 1 if autoreset else 0, p_state,
                                         p_reward, p_done, self._stream())
        if rc < 0:
            check(rc)
        return state_out, reward_out, done_out

    def lp_device_solve(self, env, Q, n_now):
        """Test hook (fjsp_env_lp_device_solve): the device LP solver on one LP of env's instance; returns x[K, M] (numpy)."""
        Q = np.ascontiguousarray(Q, dtype=np.int32); n_now = np.ascontiguousarray(n_now, dtype=np.int32)
        K = Q.shape[0]
        x = np.zeros(K * 64, np.float64)
        check(self._lib.fjsp_env_lp_device_solve(self._h, int(env), Q.ctypes.data_as(C.c_void_p), n_now.ctypes.data_as(C.c_void_p),
                                                 x.ctypes.data_as(C.c_void_p)))
        return x

    def step_async(self, actions, autoreset=False, mo=None):
        """fjsp_env_step_async: like step(), but envs that reach an order arrival park (their fluid LP is solved by
        host threads in the background) while the others keep stepping.  Returns (state, reward, done, ready):
        ready[i] = 1 where this call completed a step of env i.  The action a parked step applies is the one of the
        call in which the env parked (the first ready = 0): that call ran the step up to the arrival.  While the env
        stays parked, and in the call where it comes back with ready = 1, its entry of `actions` is not looked at --
        pair the returned row with the (state, action) of the parking call.
        Call flush_arrivals() before read() / reset() / step() / rollout()."""
        actions = _as_input("actions", actions, (self.N, 2), torch.uint8, self.device)
        if actions.data_ptr() & 1:
            actions = actions.clone()
        mo = _as_input("mo", mo, (self.N, 4), torch.float64, self.device)
        if getattr(self, "ready", None) is None:
            self.ready = torch.zeros(self.N, dtype=torch.uint8, device=self.device)
        check(self._lib.fjsp_env_step_async(self._h, _ptr(actions), _ptr(mo), 1 if autoreset else 0, self._p_state, self._p_reward,
                                            self._p_done, _ptr(self.ready), self._stream()))
        self._last_mo = mo
        return self.state, self.reward, self.done, self.ready

    def flush_arrivals(self, mo=None):
        """Wait for every parked env and finish its step (rows of state / reward / done, ready = 1)."""
        mo = _as_input("mo", mo, (self.N, 4), torch.float64, self.device) if mo is not None else getattr(self, "_last_mo", None)
        if getattr(self, "ready", None) is None:
            self.ready = torch.zeros(self.N, dtype=torch.uint8, device=self.device)
        check(self._lib.fjsp_env_arrivals_flush(self._h, _ptr(mo), self._p_state, self._p_reward, self._p_done, _ptr(self.ready),
                                                self._stream()))
        return self.state, self.reward, self.done, self.ready

    @property
    def lp_cache_hits(self):
        return int(self._lib.fjsp_env_lp_cache_hits(self._h))

    @property
    def parked(self):
        return int(self._lib.fjsp_env_parked(self._h))

    def rollout(self, actions, trace=True, rewards=True, mo=None, state=True):
        """T fused steps in one launch. actions: uint8[T, N, 2]. Returns (trace_km i16[T,N,2], reward f64[T,N], state).
        state=False: no final state (the fused kernel then skips the observation)."""
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(actions)
        if actions.dim() != 3:
            raise ValueError("actions must have shape (T, %d, 2), got %s" % (self.N, tuple(actions.shape)))
        actions = _as_input("actions", actions, (actions.shape[0], self.N, 2), torch.uint8, self.device)
        if actions.data_ptr() & 1:
            actions = actions.clone()
        mo = _as_input("mo", mo, (self.N, 4), torch.float64, self.device)
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

    @property
    def lp_device_pivots(self):
        """Simplex pivots executed by the device LP service so far (0 with the host service)."""
        return int(self._lib.fjsp_env_lp_device_pivots(self._h))

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

"""ctypes binding of include/fjsp_amd.h (libfjsp_amd.so).

The library is the product: if it is missing or a symbol is absent this module
raises -- there is no Python/CPU fallback for the accelerated path.
"""
import ctypes as C
import os

from ._build import LIB_PATH

_lib = None

# name -> (restype, argtypes); keep in step with include/fjsp_amd.h
_vp, _i32, _i64, _u64, _dbl, _cp = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double, C.c_char_p
_pp = C.POINTER(C.c_void_p)


class GenParams(C.Structure):
    """fjsp_gen_params (Instance_generate.py:39-66 distributions)."""
    _fields_ = [("R_min", _i32), ("R_max", _i32), ("J_min", _i32), ("J_max", _i32), ("M", _i32),
                ("p_min", _i32), ("p_max", _i32), ("N_min", _i32), ("N_max", _i32), ("S", _i32),
                ("DDT", _dbl), ("t_si_min", _dbl), ("t_si_max", _dbl)]


class ActorParams(C.Structure):
    """fjsp_actor_params: device pointers to the f32 parameters of a state_size -> 128 -> 128 -> n_actions actor."""
    _fields_ = [("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp), ("w3", _vp), ("b3", _vp),
                ("state_size", _i32), ("hidden", _i32), ("n_actions", _i32)]


SIGNATURES = {
    "fjsp_last_error": (_cp, []),
    "fjsp_abi_version": (C.c_int, []),
    "fjsp_instances_create": (C.c_int, [_i32, _pp]),
    "fjsp_instances_destroy": (None, [_vp]),
    "fjsp_instances_count": (C.c_int, [_vp]),
    "fjsp_instances_load_csv": (C.c_int, [_vp, _i32, _cp, _cp]),
    "fjsp_instances_generate": (C.c_int, [_vp, _i32, _u64, C.POINTER(GenParams)]),
    "fjsp_instances_set_raw": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl]),
    "fjsp_instances_dims": (C.c_int, [_vp, _i32, C.POINTER(_i32 * 6)]),
    "fjsp_instances_get": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_dbl), _vp]),
    "fjsp_instances_dynamic_dims": (C.c_int, [_vp, _i32, C.POINTER(_i32 * 2)]),
    "fjsp_instances_get_dynamic": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "fjsp_instances_set_dynamic": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "fjsp_instances_solve_fluid": (C.c_int, [_vp, _i32, _i32, _i32]),
    "fjsp_instances_set_x": (C.c_int, [_vp, _i32, _vp]),
    "fjsp_fluid_lp": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, C.POINTER(_dbl)]),
    "fjsp_env_create": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _u64, _pp]),
    "fjsp_env_destroy": (None, [_vp]),
    "fjsp_env_num_envs": (C.c_int, [_vp]),
    "fjsp_env_state_size": (C.c_int, [_vp]),
    "fjsp_env_device": (C.c_int, [_vp]),
    "fjsp_env_reset": (C.c_int, [_vp, _vp, _vp, _vp]),
    "fjsp_env_step": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "fjsp_env_step_traced": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_env_step_async": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_env_arrivals_flush": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_env_parked": (_i64, [_vp]),
    "fjsp_env_lp_cache_hits": (_i64, [_vp]),
    "fjsp_env_rollout": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "fjsp_env_read": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_env_machine_time_end": (C.c_int, [_vp, _vp, _i32, _vp]),
    "fjsp_env_energy": (C.c_int, [_vp, _vp, _vp]),
    "fjsp_env_fluid_tables": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "fjsp_env_step_bytes": (_i64, [_vp]),
    "fjsp_env_kernel_family": (_i32, [_vp]),
    "fjsp_env_set_lp_threads": (C.c_int, [_vp, _i32]),
    "fjsp_env_lp_solves": (_i64, [_vp]),
    "fjsp_env_lp_on_device": (_i32, [_vp]),
    "fjsp_env_lp_device_pivots": (_i64, [_vp]),
    "fjsp_env_lp_device_solve": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "fjsp_pyset_and_order": (C.c_int, [C.c_uint32, _vp, _i32, _i32, _vp]),
    "fjsp_rollout_create": (C.c_int, [_i32, _i32, _i32, _i32, _pp]),
    "fjsp_rollout_destroy": (None, [_vp]),
    "fjsp_rollout_append": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_rollout_returns": (C.c_int, [_vp, _dbl, _vp]),
    "fjsp_rollout_returns_normalised": (C.c_int, [_vp, C.c_double, _i32, _i32, _vp, _vp]),
    "fjsp_rollout_clear": (C.c_int, [_vp]),
    "fjsp_rollout_len": (C.c_int, [_vp]),
    "fjsp_rollout_ptr": (_vp, [_vp, _i32]),
    "fjsp_actor_forward": (C.c_int, [C.POINTER(ActorParams), _vp, _i32, _vp, _vp]),
    "fjsp_env_rollout_policy": (C.c_int, [_vp, _vp, C.POINTER(ActorParams), _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_ppo_partials": (C.c_int, [_i32]),
    "fjsp_ppo_actor_loss": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_ppo_critic_loss": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "fjsp_relu_bwd_bias": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp]),
    "fjsp_adam_clip_step": (C.c_int, [_vp, _vp, _vp, _vp, _i32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "fjsp_mlp_train_groups": (C.c_int, [_i32]),
    "fjsp_mlp_train_pass": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.c_float, _vp, _i32, _vp, _vp, _vp, _vp]),
    "fjsp_mlp_train_step": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.c_float, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "fjsp_mlp_train_step_values": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, C.c_float, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                             C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "fjsp_policy_sample": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "fjsp_policy_pair_sample": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
}


class FjspError(RuntimeError):
    """A C-ABI call returned a negative FJSP_E_* code."""

    def __init__(self, code, message):
        super().__init__("libfjsp_amd error %d: %s" % (code, message))
        self.code = code


def lib():
    """Load libfjsp_amd.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP extension is the product and there is no fallback. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` from the repo root." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64; the process must hold ONE HIP runtime, the one torch's device
    # tensors and streams belong to.  Importing torch first makes the loader resolve this library's HIP
    # symbols against that runtime (loaded the other way round, the system runtime under /opt/rocm comes
    # up next to torch's and sees no device).
    import torch  # noqa: F401
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if handle.fjsp_abi_version() != 1:
        raise ImportError("libfjsp_amd.so ABI version mismatch")
    _lib = handle
    return _lib


def check(rc):
    if rc < 0:
        raise FjspError(rc, lib().fjsp_last_error().decode("utf-8", "replace"))
    return rc

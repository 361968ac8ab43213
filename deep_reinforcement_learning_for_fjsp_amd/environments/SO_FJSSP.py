"""Host-side mirror of the reference's environments/SO_FJSSP.py.

Two faces over the same kernels:

* ``BatchedSOFJSSP``       N environments stepped by one launch (the fast path);
* ``SO_FJSSP_Environment`` the reference's single-environment class, same
  constructor / reset() / step() / attributes, implemented as an
  N=1 view of a batch so the reference's agent loops run against it unchanged.

No environment arithmetic is done in Python.
"""
import random

import numpy as np
import torch

from .. import instances as _inst
from ..batch import (EnvBatch, ST_BAD_MACHINE_RULE, ST_BAD_TASK_RULE, ST_NO_EVENT, ST_STEP_AFTER_DONE,
                     VARIANT_SO_FJSSP)
from ..utilities.Utility_Class import MyError


def _raise_for_status(status):
    """Per-env status bits -> the exception the reference would have raised."""
    if status & ST_BAD_TASK_RULE:
        raise MyError("报错：未定义该工序动作规则")          # SO_FJSSP.py:297
    if status & ST_BAD_MACHINE_RULE:
        raise MyError("报错：未定义该机器分配规则。")        # SO_FJSSP.py:321
    if status & ST_STEP_AFTER_DONE:
        raise ValueError("step() called on a finished episode (reference: max() arg is an empty sequence)")
    if status & ST_NO_EVENT:
        raise ValueError("min() arg is an empty sequence")   # SO_FJSSP.py:207


class BatchedSOFJSSP(object):
    """Vectorised SO_FJSSP: reset() -> f64[N,20]; step(actions u8[N,2]) -> (state, reward, done) device tensors."""

    actions_size = [6, 5]
    action_types = "DISCRETE"
    state_size = 20
    variant = VARIANT_SO_FJSSP          # (environments/SO_DFJSP.py subclasses with its own variant)

    def __init__(self, instance_set, n_envs=None, first=0, n_inst=None, device=0, rng_seed=0, first_env=0):
        n_inst = len(instance_set) - first if n_inst is None else n_inst
        n_envs = n_inst if n_envs is None else n_envs
        self.batch = EnvBatch(instance_set, n_envs, first=first, n_inst=n_inst, variant=self.variant,
                              device=device, rng_seed=rng_seed, first_env=first_env)
        self.N = self.batch.N
        self.device = self.batch.device

    def reset(self, mask=None):
        return self.batch.reset(mask)

    def step(self, actions, autoreset=False):
        return self.batch.step(actions, autoreset=autoreset)

    def rollout(self, actions):
        return self.batch.rollout(actions)

    def read(self):
        return self.batch.read()

    def check_status(self):
        """Raise what the reference would have raised for the first env with an error bit."""
        st = self.batch.read()["status"].cpu().numpy()
        bad = np.nonzero(st)[0]
        if len(bad):
            _raise_for_status(int(st[bad[0]]))


class _MachineView(object):
    def __init__(self, time_end):
        self.time_end = time_end


class SO_FJSSP_Environment(object):
    """Drop-in for environments/SO_FJSSP.py:12 (same names, argument meaning and error behaviour).

    ``SO_FJSSP_Environment(use_instance=True, DDT=..., M=..., S=...)`` draws a random
    instance (Instance_generate.py:24; pass ``seed=`` to make it reproducible),
    ``SO_FJSSP_Environment(use_instance=False, path=..., file_name=...)`` reads a CSV folder
    (SO_DFJSP_instance_read.py:7).  The fluid LP (class_FJSSP.py:246-280) is solved by the
    library at construction; its solution is an input of the kernels.
    """
    variant = VARIANT_SO_FJSSP

    def __init__(self, use_instance=True, device=0, **kwargs):
        self._device = device
        self._set = _inst.InstanceSet(1)
        if use_instance:
            seed = kwargs.get("seed", None)
            seed = random.getrandbits(63) if seed is None else seed
            self.DDT, self.machine_count, self.order_count = kwargs["DDT"], kwargs["M"], kwargs["S"]
            self.file_name = "DDT" + str(self.DDT) + "_M" + str(self.machine_count) + "_S" + str(self.order_count)
            self._set.generate(0, seed, _inst.reference_generator_params(self.DDT, self.machine_count, self.order_count))
        else:
            self.path, self.file_name = kwargs["path"], kwargs["file_name"]
            self._set.load_csv(0, self.path, self.file_name)
        self._set.solve_fluid(0, 1, 1)
        self._finish_init(kwargs.get("rng_seed", None))

    def _finish_init(self, rng_seed):
        a = self._set.arrays(0)
        self._arrays = a
        self.kind_count, self.machine_count, self.order_count = a.R, a.M, a.S
        self.DDT = a.ddt if not hasattr(self, "DDT") else self.DDT
        self.machine_tuple = tuple(range(a.M))
        self.kind_tuple = tuple(range(a.R))
        self.order_tuple = tuple(range(a.S))
        self.kind_task_tuple = a.kind_task_tuple
        self._rng_seed = random.getrandbits(63) if rng_seed is None else rng_seed
        self._batch = EnvBatch(self._set, 1, variant=self.variant, device=self._device, rng_seed=self._rng_seed)
        # SO_FJSSP.py:17-33
        self.step_count = 0
        self.step_time = 0
        self.state = None
        self.next_state = None
        self.reward = None
        self.done = False
        self.actions_size = [6, 5]
        self.action_tuple = tuple((a1, a2) for a1 in range(6) for a2 in range(5))
        self.state_size = 20
        self.action_types = "DISCRETE"
        self.observation_space = 10
        self.reward_sum = 0
        self.delay_time_sum = 0
        self.delay_time_sum_last = 0
        self.completion_time = 0
        self._actions = torch.zeros(1, 2, dtype=torch.uint8, device=self._batch.device)

    # pickling: A3C hands whole env objects to worker processes (A3C_v5.1.py:147-156)
    def __getstate__(self):
        a = self._arrays
        return dict(arrays=(a.Jr, a.p, a.elig_n, a.elig_list, a.count, a.arrive, a.delivery, a.ddt, a.x),
                    file_name=self.file_name, device=self._device, rng_seed=self._rng_seed)

    def __setstate__(self, st):
        Jr, p, elig_n, elig_list, count, arrive, delivery, ddt, x = st["arrays"]
        self._device, self.file_name = st["device"], st["file_name"]
        self._set = _inst.InstanceSet(1).set_raw(0, Jr, p, elig_n, elig_list, count, arrive, delivery, ddt).set_x(0, x)
        self._finish_init(st["rng_seed"])

    def _refresh(self):
        r = self._batch.read()
        vals = {k: int(v.item()) for k, v in r.items()}
        self.step_time, self.step_count = vals["step_time"], vals["step_count"]
        self.delay_time_sum, self.completion_time = vals["delay_time_sum"], vals["completion_time"]
        return vals

    def reset(self):
        """SO_FJSSP.py:51-76: returns a fresh float64 array of 20."""
        st = self._batch.reset()
        self.state = st[0].cpu().numpy().copy()
        self.next_state, self.reward, self.done = None, None, False
        self.reward_sum = 0
        self.delay_time_sum_last = 0
        self._refresh()
        return self.state

    def step(self, action):
        """SO_FJSSP.py:168-265: action = indexable pair (task rule index, machine rule index)."""
        a0, a1 = int(action[0]), int(action[1])
        if not 0 <= a0 < 6:
            raise MyError("报错：未定义该工序动作规则")
        if not 0 <= a1 < 5:
            raise MyError("报错：未定义该机器分配规则。")
        self._actions[0, 0], self._actions[0, 1] = a0, a1
        st, rw, dn = self._batch.step(self._actions)
        vals = self._refresh()
        if vals["status"]:
            _raise_for_status(vals["status"])
        self.delay_time_sum_last = self.delay_time_sum
        self.next_state = st[0].cpu().numpy().copy()
        r = float(rw[0].item())
        self.reward = int(r) if r == int(r) else r      # the reference's reward is a Python int (:328)
        self.reward_sum += self.reward
        self.done = bool(dn[0].item())
        self.state = self.next_state
        return self.state, self.reward, self.done

    @property
    def machine_dict(self):
        te = self._batch.machine_time_end()[0].cpu().numpy()
        return {m: _MachineView(int(te[m])) for m in self.machine_tuple}

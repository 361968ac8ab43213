"""Host-side mirror of the reference's environments/MO_FJSSP_discretes.py (the
environment agents/MPPPO/MPPPO.py instantiates): makespan + total tardiness,
18 flat actions = 6 task rules x 3 machine rules (:26), 25-dim state = 7 static
+ 9 observed + 9 deltas (:17-21,48), step(action, weight_vector, completion,
tardiness) (:88), weighted reward (:232-244).  Same kernels as SO_FJSSP with the
variant switch (csrc/fjsp_kernels.hip); nothing is computed in Python.
"""
import random

import numpy as np
import torch

from .. import instances as _inst
from ..batch import EnvBatch, VARIANT_MO_FJSSP_DISCRETES
from ..utilities.Utility_Class import MyError
from .SO_FJSSP import _MachineView, _raise_for_status


class BatchedMOFJSSP(object):
    """Vectorised MO_FJSSP_discretes.  step(actions i64/u8[N], mo f64[N,4]) -> (state[N,25], reward[N], done[N])."""

    action_space = 18
    action_types = "DISCRETE"
    state_size = 25

    def __init__(self, instance_set, n_envs=None, first=0, n_inst=None, device=0, rng_seed=0, first_env=0):
        n_inst = len(instance_set) - first if n_inst is None else n_inst
        n_envs = n_inst if n_envs is None else n_envs
        self.batch = EnvBatch(instance_set, n_envs, first=first, n_inst=n_inst, variant=VARIANT_MO_FJSSP_DISCRETES,
                              device=device, rng_seed=rng_seed, first_env=first_env)
        self.N, self.device = self.batch.N, self.batch.device
        self._act = torch.zeros(self.N, 2, dtype=torch.uint8, device=self.device)
        self.mo = torch.zeros(self.N, 4, dtype=torch.float64, device=self.device)
        self.mo[:, 1] = 1.0
        self.mo[:, 2:] = -1.0

    def set_objective(self, weight_vector, completion=None, tardiness=None):
        """weight_vector (w_completion, w_tardiness); completion / tardiness: normalisers or None
        (scalars or per-env tensors), as MPPPO.py:161-164 passes them."""
        self.mo[:, 0], self.mo[:, 1] = float(weight_vector[0]), float(weight_vector[1])
        self.mo[:, 2] = -1.0 if completion is None else completion
        self.mo[:, 3] = -1.0 if tardiness is None else tardiness

    def reset(self, mask=None):
        return self.batch.reset(mask)

    def step(self, actions, autoreset=False):
        self._act[:, 0] = actions.to(torch.uint8)
        return self.batch.step(self._act, autoreset=autoreset, mo=self.mo)

    def read(self):
        return self.batch.read()


class MO_FJSSP_Environment(object):
    """Drop-in for environments/MO_FJSSP_discretes.py:12 (N = 1 view of the batched kernels)."""

    def __init__(self, use_instance=True, device=0, **kwargs):
        self._device = device
        self._set = _inst.InstanceSet(1)
        if use_instance:
            seed = kwargs.get("seed", None)
            seed = random.getrandbits(63) if seed is None else seed
            self.DDT = kwargs["DDT"]
            self.file_name = "DDT" + str(kwargs["DDT"]) + "_M" + str(kwargs["M"]) + "_S" + str(kwargs["S"])
            self._set.generate(0, seed, _inst.reference_generator_params(kwargs["DDT"], kwargs["M"], kwargs["S"]))
        else:
            self.path, self.file_name = kwargs["path"], kwargs["file_name"]
            self._set.load_csv(0, self.path, self.file_name)
        self._set.solve_fluid(0, 1, 1)
        a = self._set.arrays(0)
        self._arrays = a
        self.kind_count, self.machine_count, self.order_count = a.R, a.M, a.S
        self.machine_tuple = tuple(range(a.M))
        rng_seed = kwargs.get("rng_seed", None)
        self._batch = EnvBatch(self._set, 1, variant=VARIANT_MO_FJSSP_DISCRETES, device=device,
                               rng_seed=random.getrandbits(63) if rng_seed is None else rng_seed)
        self.state_size = 25
        self.action_types = "DISCRETE"
        self.action_space = 18
        self.observation_space = 9
        self.static_state_space = 7
        self.actions = tuple((t, m) for t in range(6) for m in range(3))          # :26
        self.reward_sum = 0
        self.completion_time = 0
        self.delay_time_sum = 0
        self.step_count = 0
        self.step_time = 0
        self.done = False
        self.state = None
        self._act = torch.zeros(1, 2, dtype=torch.uint8, device=self._batch.device)
        self._mo = torch.zeros(1, 4, dtype=torch.float64, device=self._batch.device)

    def _refresh(self):
        vals = {k: int(v.item()) for k, v in self._batch.read().items()}
        self.step_time, self.step_count = vals["step_time"], vals["step_count"]
        self.delay_time_sum, self.completion_time = vals["delay_time_sum"], vals["completion_time"]
        return vals

    def reset(self):
        """MO_FJSSP_discretes.py:28-53"""
        self.state = self._batch.reset()[0].cpu().numpy().copy()
        self.done, self.reward_sum = False, 0
        self._refresh()
        return self.state

    def step(self, action, weight_vector=None, completion=None, tardiness=None):
        """MO_FJSSP_discretes.py:88-174"""
        if not 0 <= int(action) < 18:
            raise IndexError("tuple index out of range")                              # self.actions[action] :92
        if weight_vector is None:
            raise TypeError("'NoneType' object is not subscriptable")                 # :240
        w0, w1 = float(weight_vector[0]), float(weight_vector[1])
        if (completion is None or tardiness is None) and w1 != 1 and w0 != 1:
            raise MyError("未定义该回报函数")                                          # :244
        self._act[0, 0] = int(action)
        self._mo[0, 0], self._mo[0, 1] = w0, w1
        self._mo[0, 2] = -1.0 if completion is None or tardiness is None else float(completion)
        self._mo[0, 3] = -1.0 if completion is None or tardiness is None else float(tardiness)
        st, rw, dn = self._batch.step(self._act, mo=self._mo)
        vals = self._refresh()
        if vals["status"]:
            _raise_for_status(vals["status"])
        self.state = st[0].cpu().numpy().copy()
        r = float(rw[0].item())
        self.reward = int(r) if (completion is None or tardiness is None) else r
        self.reward_sum += self.reward
        self.done = bool(dn[0].item())
        return self.state, self.reward, self.done

    @property
    def machine_dict(self):
        te = self._batch.machine_time_end()[0].cpu().numpy()
        return {m: _MachineView(int(te[m])) for m in self.machine_tuple}

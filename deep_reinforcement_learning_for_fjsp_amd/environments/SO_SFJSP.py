"""Host-side mirror of the reference's environments/SO_SFJSP.py (the environment
agents/DDQN/DDQN.py instantiates): makespan objective, 20 flat actions = 4 task
rules x 5 machine rules (:25), 18-dim state = 9 observed + 9 deltas (:14-17,64-83),
reward -(delta completion_time) / fluid_completed_time (:216-220).  Same kernels
as SO_FJSSP, instantiated for this variant (csrc/fjsp_kernels.hip).
"""
import random

import torch

from .. import instances as _inst
from ..batch import EnvBatch, VARIANT_SO_SFJSP
from ..utilities.Utility_Class import MyError
from .SO_FJSSP import _MachineView, _raise_for_status


class BatchedSOSFJSP(object):
    """Vectorised SO_SFJSP.  step(actions[N]) -> (state[N,18], reward[N], done[N]) device tensors."""

    action_space = 20
    action_types = "DISCRETE"
    state_size = 18

    def __init__(self, instance_set, n_envs=None, first=0, n_inst=None, device=0, rng_seed=0, first_env=0):
        n_inst = len(instance_set) - first if n_inst is None else n_inst
        n_envs = n_inst if n_envs is None else n_envs
        self.batch = EnvBatch(instance_set, n_envs, first=first, n_inst=n_inst, variant=VARIANT_SO_SFJSP,
                              device=device, rng_seed=rng_seed, first_env=first_env)
        self.N, self.device = self.batch.N, self.batch.device
        self._act = torch.zeros(self.N, 2, dtype=torch.uint8, device=self.device)

    def reset(self, mask=None):
        return self.batch.reset(mask)

    def step(self, actions, autoreset=False):
        self._act[:, 0] = actions.to(torch.uint8)
        return self.batch.step(self._act, autoreset=autoreset)

    def read(self):
        return self.batch.read()


class SO_SFJSP_Environment(object):
    """Drop-in for environments/SO_SFJSP.py:11 (N = 1 view of the batched kernels)."""

    def __init__(self, use_instance=True, device=0, **kwargs):
        self._set = _inst.InstanceSet(1)
        if use_instance:
            seed = kwargs.get("seed", None)
            seed = random.getrandbits(63) if seed is None else seed
            self.file_name = "DDT" + str(kwargs["DDT"]) + "_M" + str(kwargs["M"]) + "_S" + str(kwargs["S"])
            self._set.generate(0, seed, _inst.reference_generator_params(kwargs["DDT"], kwargs["M"], kwargs["S"]))
        else:
            self.path, self.file_name = kwargs["path"], kwargs["file_name"]
            self._set.load_csv(0, self.path, self.file_name)
        self._set.solve_fluid(0, 1, 1)
        a = self._set.arrays(0)
        self.kind_count, self.machine_count, self.order_count = a.R, a.M, a.S
        self.machine_tuple = tuple(range(a.M))
        rng_seed = kwargs.get("rng_seed", None)
        self._batch = EnvBatch(self._set, 1, variant=VARIANT_SO_SFJSP, device=device,
                               rng_seed=random.getrandbits(63) if rng_seed is None else rng_seed)
        self.state_size = 18
        self.action_types = "DISCRETE"
        self.observation_space = 9
        self.static_state_space = 0
        self.action_space = 20
        self.actions = tuple((t, m) for t in range(4) for m in range(5))          # :25
        self.reward_sum = 0
        self.completion_time = 0
        self.delay_time_sum = 0
        self.step_count = 0
        self.step_time = 0
        self.done = False
        self.state = None
        self._act = torch.zeros(1, 2, dtype=torch.uint8, device=self._batch.device)

    def _refresh(self):
        vals = {k: int(v.item()) for k, v in self._batch.read().items()}
        self.step_time, self.step_count = vals["step_time"], vals["step_count"]
        self.delay_time_sum, self.completion_time = vals["delay_time_sum"], vals["completion_time"]
        return vals

    def reset(self):
        """SO_SFJSP.py:27-52"""
        self.state = self._batch.reset()[0].cpu().numpy().copy()
        self.done, self.reward_sum = False, 0
        self._refresh()
        return self.state

    def step(self, action):
        """SO_SFJSP.py:85-167"""
        if not 0 <= int(action) < 20:
            raise IndexError("tuple index out of range")                              # self.actions[action] :87
        self._act[0, 0] = int(action)
        st, rw, dn = self._batch.step(self._act)
        vals = self._refresh()
        if vals["status"]:
            _raise_for_status(vals["status"])
        self.state = st[0].cpu().numpy().copy()
        self.reward = float(rw[0].item())
        self.reward_sum += self.reward
        self.done = bool(dn[0].item())
        return self.state, self.reward, self.done

    @property
    def machine_dict(self):
        te = self._batch.machine_time_end()[0].cpu().numpy()
        return {m: _MachineView(int(te[m])) for m in self.machine_tuple}

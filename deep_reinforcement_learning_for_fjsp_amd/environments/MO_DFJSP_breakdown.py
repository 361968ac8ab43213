"""Host-side mirror of the reference's environments/MO_DFJSP_breakdown.py (and of
environments/MO_DFJSP.py, which is the same environment without breakdown windows):
the dynamic multi-objective shop -- new orders arrive mid-episode (:281-296),
machines break down (:204-231), objectives are makespan, total tardiness and
energy (:249-256), 12 task rules x 10 machine rules (:32,357-428), 30-dim state
= 15 observed + 15 deltas (:35,94-118), step(action, reward_policy, completion,
tardiness, energy_consumption) (:189), reward policies 0..3 (:430-447).

Same kernels as SO_FJSSP with the MO_DFJSP variant switch (csrc/fjsp_kernels.hip);
an order arrival inside a step re-solves the fluid LP on the host, which makes
step() blocking for this environment.  Nothing is computed in Python.
"""
import random

import torch

from .. import instances as _inst
from ..batch import EnvBatch, VARIANT_MO_DFJSP
from ..utilities.Utility_Class import MyError
from .SO_FJSSP import _MachineView, _raise_for_status


class BatchedMODFJSP(object):
    """Vectorised MO_DFJSP_breakdown.  step(actions u8[N,2]) -> (state[N,30], reward[N], done[N])."""

    actions_size = [12, 10]
    action_types = "DISCRETE"
    state_size = 30

    def __init__(self, instance_set, n_envs=None, first=0, n_inst=None, device=0, rng_seed=0, first_env=0):
        n_inst = len(instance_set) - first if n_inst is None else n_inst
        n_envs = n_inst if n_envs is None else n_envs
        self.batch = EnvBatch(instance_set, n_envs, first=first, n_inst=n_inst, variant=VARIANT_MO_DFJSP,
                              device=device, rng_seed=rng_seed, first_env=first_env)
        self.N, self.device = self.batch.N, self.batch.device
        self.mo = torch.zeros(self.N, 4, dtype=torch.float64, device=self.device)
        self.mo[:, 0] = 1.0

    def set_objective(self, reward_policy, completion=None, tardiness=None, energy_consumption=None):
        """reward_policy 0 makespan / 1 tardiness / 2 energy / 3 normalised sum; the normalisers
        (scalars or per-env tensors) are only read by policy 3 (MO_DFJSP_breakdown.py:430-447)."""
        self.mo[:, 0] = float(reward_policy)
        self.mo[:, 1] = 0.0 if completion is None else completion
        self.mo[:, 2] = 0.0 if tardiness is None else tardiness
        self.mo[:, 3] = 0.0 if energy_consumption is None else energy_consumption

    def reset(self, mask=None):
        return self.batch.reset(mask)

    def step(self, actions, autoreset=False):
        return self.batch.step(actions, autoreset=autoreset, mo=self.mo)

    def read(self):
        return self.batch.read()


class MO_DFJSP_Environment(object):
    """Drop-in for environments/MO_DFJSP_breakdown.py:12 (N = 1 view of the batched kernels).

    ``use_instance=False, path=..., file_name=...`` reads a CSV folder with a machine_data.csv
    (MO_DFJSP_instance_read.py); ``use_instance=True, DDT=..., M=..., S=...`` draws a random instance
    with the generator's power ranges (Instance_generate.py:61-66) and no breakdown windows, i.e. what
    environments/MO_DFJSP.py plays.
    """

    def __init__(self, use_instance=True, device=0, **kwargs):
        self._set = _inst.InstanceSet(1)
        if use_instance:
            seed = kwargs.get("seed", None)
            seed = random.getrandbits(63) if seed is None else seed
            self.file_name = "DDT" + str(kwargs["DDT"]) + "_M" + str(kwargs["M"]) + "_S" + str(kwargs["S"])
            self._set.generate(0, seed, _inst.reference_generator_params(kwargs["DDT"], kwargs["M"], kwargs["S"]))
            self._set.generate_machine_data(0, seed)                            # p_rjm / p_m_idle, Instance_generate.py:61-66
        else:
            self.path, self.file_name = kwargs["path"], kwargs["file_name"]
            self._set.load_csv(0, self.path, self.file_name)
        self._set.solve_fluid(0, 1, 1)
        a = self._set.arrays(0)
        self._arrays = a
        self.DDT = a.ddt
        self.kind_count, self.machine_count, self.order_count = a.R, a.M, a.S
        self.machine_tuple = tuple(range(a.M))
        rng_seed = kwargs.get("rng_seed", None)
        self._batch = EnvBatch(self._set, 1, variant=VARIANT_MO_DFJSP, device=device,
                               rng_seed=random.getrandbits(63) if rng_seed is None else rng_seed)
        self.actions_size = [12, 10]                                                 # :32
        self.action_tuple = tuple((a1, a2) for a1 in range(12) for a2 in range(10))  # :33
        self.action_space = list(range(12))
        self.state_size = 30
        self.action_types = "DISCRETE"
        self.observation_space = 15
        self.reward_sum = 0
        self.completion_time = 0
        self.delay_time_sum = 0
        self.energy_consumption = 0
        self.step_count = 0
        self.step_time = 0
        self.done = False
        self.state = None
        self.reward = None
        self._act = torch.zeros(1, 2, dtype=torch.uint8, device=self._batch.device)
        self._mo = torch.zeros(1, 4, dtype=torch.float64, device=self._batch.device)

    def _refresh(self):
        vals = {k: int(v.item()) for k, v in self._batch.read().items()}
        self.step_time, self.step_count = vals["step_time"], vals["step_count"]
        self.delay_time_sum, self.completion_time = vals["delay_time_sum"], vals["completion_time"]
        self.energy_consumption = vals["energy_consumption"]
        return vals

    def reset(self):
        """MO_DFJSP_breakdown.py:58-90"""
        self.state = self._batch.reset()[0].cpu().numpy().copy()
        self.done, self.reward_sum = False, 0
        self._refresh()
        return self.state

    def step(self, action, reward_policy=None, completion=None, tardiness=None, energy_consumption=None):
        """MO_DFJSP_breakdown.py:189-328"""
        if len(action) == 1:
            action = self.action_tuple[action[0]]                                    # :191-192
        if reward_policy not in (0, 1, 2, 3):
            raise MyError("未定义该回报函数")                                         # :447
        if reward_policy == 3 and (completion is None or tardiness is None or energy_consumption is None):
            raise TypeError("unsupported operand type(s) for /: 'int' and 'NoneType'")
        self._act[0, 0], self._act[0, 1] = int(action[0]), int(action[1])
        self._mo[0, 0] = float(reward_policy)
        for i, v in enumerate((completion, tardiness, energy_consumption)):
            self._mo[0, 1 + i] = 0.0 if v is None else float(v)
        st, rw, dn = self._batch.step(self._act, mo=self._mo)
        vals = self._refresh()
        if vals["status"]:
            _raise_for_status(vals["status"])
        self.state = st[0].cpu().numpy().copy()
        r = float(rw[0].item())
        self.reward = int(r) if reward_policy != 3 else r                            # integer differences :433-437
        self.reward_sum += self.reward
        self.done = bool(dn[0].item())
        return self.state, self.reward, self.done

    @property
    def machine_dict(self):
        te = self._batch.machine_time_end()[0].cpu().numpy()
        return {m: _MachineView(int(te[m])) for m in self.machine_tuple}

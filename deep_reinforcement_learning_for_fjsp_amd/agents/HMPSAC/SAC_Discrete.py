"""Batched counterpart of the reference's agents/HMPSAC/SAC_Discrete.py: HMPSAC's upper-level controller,
a discrete soft actor-critic whose 3 actions pick WHICH lower-level objective policy (makespan /
tardiness / energy, trained by agents/HMPSAC/A3C.py) dispatches the next operation of a
MO_DFJSP(_breakdown) environment.

What is kept from the reference, by line:
  PolicyNet 30->3 softmax, twin CriticNet 30->3 + targets (3 x 200 hidden)   SAC_Discrete.py:85-123,151-163
  Adam(lr 3e-4, eps 1e-4) everywhere, automatic entropy tuning with
      target entropy 0.98 * log(3)                                           :164-172
  epoch structure: one episode per lower policy with reward_policy 0 gives
      the instance's three objective baselines, their minimum normalises
      reward_policy 3 of the controller's episode                            :197-240
  pick_action: uniform random until min_steps_before_learning                :248-254
  produce_action_and_action_info (log(p + 1e-8 * [p == 0]))                  :265-275
  critic / actor / alpha losses and updates                                  :308-352
  learn schedule: every update_every_n_steps env steps, 10 updates           :287-291,233-235

What changes (MI355X-first): one epoch plays the four episodes on EVERY environment of a `BatchedMODFJSP`
batch (HIP kernels), the baselines are per-environment tensors handed to the kernel's reward_policy 3
(f64[N,4] argument of fjsp_env_step), transitions go to an HBM-resident replay ring, and with
torch.distributed every network's flat gradient is all-reduced once per optimiser step.  The reference's
`load_policy_model` reads checkpoints that do not ship with it (`results/HMPSAC/policy_networks_v5.x`); here
the three lower policies are passed in (trained by `DA3C`) or loaded from files written by
`DA3C.save_actor_model` (same file names).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.distributions import Categorical
from torch.optim import Adam

from ..Base_Agent import Base_Agent
from ... import distributed as fdist
from ...utilities.data_structures.Config import Config
from ...utilities.data_structures.Replay_Buffer import DeviceReplayBuffer
from .A3C import CriticNet, MachinePolicyNet, TaskPolicyNet, _mlp


class PolicyNet(nn.Module):
    """SAC_Discrete.py:85-103"""

    def __init__(self, input_size, hidden_size, hidden_layer, output_size):
        super().__init__()
        self.name = "task_policy"
        self.layers = _mlp(input_size, hidden_size, hidden_layer, output_size)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return F.softmax(x, dim=-1)


def produce_action_and_action_info(actor, state):
    """:265-275"""
    action_probabilities = actor(state)
    max_probability_action = torch.argmax(action_probabilities, dim=-1)
    action = Categorical(action_probabilities, validate_args=False).sample()      # (the validation syncs with the host)
    z = (action_probabilities == 0.0).float() * 1e-8
    log_action_probabilities = torch.log(action_probabilities + z)
    return action, (action_probabilities, log_action_probabilities), max_probability_action


def sac_critic_losses(actor, critic_1, critic_2, target_1, target_2, alpha, batch, discount_rate):
    """:308-322"""
    state, action, reward, next_state, done = batch
    with torch.no_grad():
        _, (p, logp), _ = produce_action_and_action_info(actor, next_state)
        min_next = p * (torch.min(target_1(next_state), target_2(next_state)) - alpha * logp)
        min_next = min_next.sum(dim=1).unsqueeze(-1)
        next_q = reward + (1.0 - done) * discount_rate * min_next
    qf1 = critic_1(state).gather(1, action.long())
    qf2 = critic_2(state).gather(1, action.long())
    return F.mse_loss(qf1, next_q), F.mse_loss(qf2, next_q)


def sac_actor_loss(actor, critic_1, critic_2, alpha, state):
    """:324-333: returns (policy_loss, sum_a p log p per sample)."""
    _, (p, logp), _ = produce_action_and_action_info(actor, state)
    min_qf_pi = torch.min(critic_1(state), critic_2(state))
    inside_term = alpha * logp - min_qf_pi
    policy_loss = (p * inside_term).sum(dim=1).mean()
    return policy_loss, torch.sum(logp * p, dim=1)


class SAC_Discrete(Base_Agent, Config):
    """environment: a BatchedMODFJSP (the reference trains on its single test instance, :140-142; any batch
    works); lower_policies: {0|1|2: (TaskPolicyNet, MachinePolicyNet)} or a folder layout understood by
    load_policy_model."""

    def __init__(self, environment, lower_policies=None, hidden_size=200, hidden_layer=3, hyper=None, seed=0,
                 max_steps=4096):
        Base_Agent.__init__(self)
        Config.__init__(self)
        self.agent = "HMP_SAC"
        self.hyper_parameters = dict(self.hyper_parameters[self.agent])
        self.hyper_parameters.update(hyper or {})
        hp = self.hyper_parameters
        self.action_types = "DISCRETE"
        self.environment = environment
        self.device = environment.device
        self.state_size, self.action_size = 30, 3
        rng = torch.random.get_rng_state()
        torch.manual_seed(seed)
        mk = lambda cls, out: cls(self.state_size, hidden_size, hidden_layer, out).to(self.device)
        self.critic_local, self.critic_local_2 = mk(CriticNet, 3), mk(CriticNet, 3)
        self.critic_target, self.critic_target_2 = mk(CriticNet, 3), mk(CriticNet, 3)
        self.actor_local = mk(PolicyNet, 3)
        torch.random.set_rng_state(rng)
        lr = hp["learning_rate"]
        cap = self.device.type == "cuda"       # step counts on the device: the update is replayed from a captured graph
        self.critic_optimizer = Adam(self.critic_local.parameters(), lr=lr, eps=1e-4, capturable=cap)
        self.critic_optimizer_2 = Adam(self.critic_local_2.parameters(), lr=lr, eps=1e-4, capturable=cap)
        self.actor_optimizer = Adam(self.actor_local.parameters(), lr=lr, eps=1e-4, capturable=cap)
        self.memory = DeviceReplayBuffer(hp["buffer_size"], hp["batch_size"], self.state_size, self.device, seed=seed)
        self.automatic_entropy_tuning = hp["automatically_tune_entropy_hyper_parameter"]
        if self.automatic_entropy_tuning:
            self.target_entropy = -np.log((1.0 / self.action_size)) * 0.98                 # :166
            self.log_alpha = torch.zeros(1, requires_grad=True, device=self.device)
            self.alpha = self.log_alpha.exp().detach()
            self.alpha_optim = Adam([self.log_alpha], lr=lr, eps=1e-4, capturable=cap)
        else:
            self.alpha = hp["entropy_term_weight"]
        self.buckets = {n: fdist.FlatGradBucket(n.parameters()) for n in (self.critic_local, self.critic_local_2, self.actor_local)}
        self.objectives_policy = {"makespan": 0, "tardiness": 1, "energy": 2}               # :179
        self.action_size_dict = {"task": 12, "machine": 10}
        self.policy_dict = {0: {}, 1: {}, 2: {}}
        if isinstance(lower_policies, str):
            self.load_policy_model(lower_policies)
        elif lower_policies is not None:
            for k, (task, machine) in lower_policies.items():
                self.policy_dict[k] = {"task": task.to(self.device), "machine": machine.to(self.device)}
        self.max_steps = max_steps
        self.global_step_number = 0
        self._next_learn = hp["update_every_n_steps"]
        self.learn_sessions = 0
        self._last_losses = None
        self.use_graph = True            # replay the per-step device work and the update from captured HIP graphs (GPU only)
        self.fused_policy = True         # draw the actions of the MLP policies in one launch per (task, machine) pair (GPU only)
        self._sampler_seed = seed
        self._graphs, self._static, self._learn_graph = {}, None, None

    @property
    def last_losses(self):
        return None if self._last_losses is None else tuple(float(v) for v in self._last_losses)

    def load_policy_model(self, root):
        """:184-195; `root`/policy_networks_v5.{1,2,3}/actor_{task,machine}_model.path"""
        for objective, policy in self.objectives_policy.items():
            folder = os.path.join(root, "policy_networks_v5." + str(policy + 1))
            task = TaskPolicyNet(30, 200, 3, 12).to(self.device)
            task.load_state_dict(torch.load(os.path.join(folder, "actor_task_model.path"), weights_only=True))
            machine = MachinePolicyNet(31, 200, 3, 10).to(self.device)
            machine.load_state_dict(torch.load(os.path.join(folder, "actor_machine_model.path"), weights_only=True))
            self.policy_dict[policy] = {"task": task, "machine": machine}

    # -- acting ---------------------------------------------------------------------------------
    # -- one-launch sampling (agents/fused_policy.py, csrc/fjsp_policy_mlp.hip): GPU batches, plain MLP stacks --------------
    def _sampler(self, key):
        """PolicyPairSampler of lower policy `key` (0|1|2) or of the controller's actor ("actor"); None when the fused
        path is off or a network does not fit the kernel (the library path below is used then)."""
        if not getattr(self, "fused_policy", True) or self.device.type != "cuda":
            return None
        cache = self.__dict__.setdefault("_samplers", {})
        if key not in cache:
            from .. import fused_policy
            if key == "actor":
                layers, layers_m = self.actor_local.layers, None
            else:
                nets = self.policy_dict[int(key)]
                layers, layers_m = nets["task"].layers_1, nets["machine"].layers_2
            ok = fused_policy.supported(layers, self.device) and (layers_m is None or fused_policy.supported(layers_m, self.device))
            seed = int(getattr(self, "_sampler_seed", 0)) * 7919 + (97 if key == "actor" else 11 + int(key))
            # (the lower policies are frozen inside the controller's training: their transposed weights are taken once;
            # the actor's at every call)
            cache[key] = fused_policy.PolicyPairSampler(layers, layers_m, seed=seed, static_weights=key != "actor") if ok else None
        return cache[key]

    @torch.no_grad()
    def pick_lower_action(self, which, state, out=None):
        """:277-284 for a batch: env e follows lower policy which[e] (a scalar means everyone).  out: u8[N, 2] to write into."""
        fused = state.is_cuda and state.dtype == torch.float64 and state.is_contiguous()
        if fused and (out is None or (out.dtype == torch.uint8 and out.is_contiguous() and tuple(out.shape) == (state.shape[0], 2))):
            pair = out if out is not None else torch.empty(state.shape[0], 2, dtype=torch.uint8, device=state.device)
            if not torch.is_tensor(which):
                sm = self._sampler(int(which))
                if sm is not None:
                    sm.sample(state, pair_out=pair)           # the kernel writes the action pair itself
                    return pair
            elif which.dtype == torch.int64 and which.is_contiguous():
                sms = {k: self._sampler(k) for k in self.policy_dict}
                if all(v is not None for v in sms.values()):
                    for k, sm in sms.items():            # every lower policy proposes for every env, the controller's choice selects
                        sm.sample(state, pair_out=pair, select=which, which=int(k))
                    return pair
        if out is not None:
            out.copy_(self.pick_lower_action(which, state))
            return out
        s = state.float()
        sample = lambda nets, x: Categorical(nets(x), validate_args=False).sample()
        if not torch.is_tensor(which):
            nets = self.policy_dict[int(which)]
            a_t = sample(nets["task"], s)
            a_m = sample(nets["machine"], torch.cat([s, a_t.float().unsqueeze(1)], 1))
            return torch.stack([a_t, a_m], 1).to(torch.uint8)
        # mixed batch: every lower policy proposes for every env and the controller's choice selects -- three small
        # MLP passes and no host round trip, instead of gathering the envs of each policy (a sync per group)
        a_t = torch.zeros(s.shape[0], dtype=torch.long, device=s.device)
        a_m = torch.zeros_like(a_t)
        for k, nets in self.policy_dict.items():
            t = sample(nets["task"], s)
            m = sample(nets["machine"], torch.cat([s, t.float().unsqueeze(1)], 1))
            mine = which == k
            a_t, a_m = torch.where(mine, t, a_t), torch.where(mine, m, a_m)
        return torch.stack([a_t, a_m], 1).to(torch.uint8)

    @torch.no_grad()
    def pick_action(self, state):
        """:248-254"""
        if self.global_step_number < self.hyper_parameters["min_steps_before_learning"]:
            return torch.randint(0, self.action_size, (state.shape[0],), device=self.device)
        if state.is_cuda and state.dtype == torch.float64 and state.is_contiguous():
            sm = self._sampler("actor")
            if sm is not None:
                return sm.sample(state)[0]
        action, _, _ = produce_action_and_action_info(self.actor_local, state.float())
        return action

    # -- a vector step's device work outside the environment, replayed from two captured HIP graphs -------------
    def _step_graphs(self, which, random_phase):
        """(act, store) graphs for one kind of episode.  `act` turns the static state / done tensors into the action
        pair (controller: controller action first); `store` runs after env.step(): replay-buffer rows of the live
        envs (controller episodes), then next state / done into the static tensors.  A vector step is then two graph
        launches and the environment call instead of ~60 small launches (policy MLPs, Categorical, prefix sum and
        scatter of the replay rows), whose host cost was ~10x their device time."""
        key = (which, random_phase)
        g = self._graphs.get(key)
        if g is not None:
            return g
        env, st = self.environment, self._static
        eb = env.batch

        def act():
            st["active"].copy_(st["done"] == 0)
            if which is None:
                st["action"].copy_(self.pick_action(st["state"]))
                self.pick_lower_action(st["action"], st["state"], out=st["pair"])
            else:
                self.pick_lower_action(which, st["state"], out=st["pair"])

        def store():
            if which is None:
                self.memory.add_batch(st["state"], st["action"], eb.reward, eb.state, eb.done, st["active"])
                st["played"].add_(st["active"].sum())
            st["state"].copy_(eb.state)
            st["done"].copy_(eb.done)

        keep = {k: v.clone() for k, v in st.items()}
        mem = self.memory.snapshot_cursor() if which is None else None
        st["done"].fill_(1)                               # warm-up with no live env: nothing reaches the replay ring
        act(); store()                                    # (outside the capture: allocator, library handles)
        torch.cuda.synchronize(self.device)
        ga, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, capture_error_mode="relaxed"):
            act()
        with torch.cuda.graph(gs, capture_error_mode="relaxed"):
            store()
        for k, v in keep.items():                         # the warm-up ran on stale env outputs: undo what it changed
            st[k].copy_(v)
        if mem is not None:
            self.memory.restore_cursor(mem)
        g = self._graphs[key] = (ga, gs)
        return g

    def _episode(self, which=None, objectives=None):
        """One episode of every environment.  which = lower policy index (baseline episodes, reward_policy 0)
        or None (controller episode, reward_policy 3 normalised by `objectives` [N, 3])."""
        env, hp = self.environment, self.hyper_parameters
        if which is None:
            env.set_objective(3, objectives[:, 0], objectives[:, 1], objectives[:, 2])
        else:
            env.set_objective(0)
        graphed = self.use_graph and self.device.type == "cuda"
        if graphed and self._static is None:
            N = env.N
            self._static = dict(state=torch.zeros(N, self.state_size, dtype=torch.float64, device=self.device),
                                done=torch.zeros(N, dtype=torch.uint8, device=self.device),
                                active=torch.zeros(N, dtype=torch.bool, device=self.device),
                                action=torch.zeros(N, dtype=torch.int64, device=self.device),
                                pair=torch.zeros(N, 2, dtype=torch.uint8, device=self.device),
                                played=torch.zeros((), dtype=torch.int64, device=self.device))
        state = env.reset().clone()
        done = torch.zeros(env.N, dtype=torch.uint8, device=self.device)
        played = torch.zeros((), dtype=torch.int64, device=self.device)     # transitions since the last host visit
        if graphed:
            st = self._static
            st["state"].copy_(state); st["done"].zero_(); st["played"].zero_()
            played = st["played"]
        for t in range(self.max_steps):
            if graphed:
                random_phase = which is None and self.global_step_number < hp["min_steps_before_learning"]
                ga, gs = self._step_graphs(which, random_phase)
                ga.replay()
                if which is None:
                    env.step(st["pair"])
                    gs.replay()
                else:                                            # baseline episodes store nothing: the environment writes
                    env.batch.step(st["pair"], mo=env.mo, state_out=st["state"], done_out=st["done"])   # the static tensors itself
                done = st["done"]
            else:
                active = done == 0
                if which is None:
                    action = self.pick_action(state)
                    pair = self.pick_lower_action(action, state)
                else:
                    pair = self.pick_lower_action(which, state)
                nxt, rew, dn = env.step(pair)
                if which is None:
                    self.memory.add_batch(state, action, rew, nxt, dn, active)
                    played += active.sum()
                state, done = nxt.clone(), dn.clone()
            if t % 16 == 15:                                     # one host round trip per 16 vector steps
                if which is None:
                    self.global_step_number += int(played.item())
                    played.zero_()
                    if self.time_for_critic_and_actor_to_learn():
                        for _ in range(hp["learning_updates_per_learning_session"]):
                            self.learn()
                        self.learn_sessions += 1
                if bool((done != 0).all()):
                    break
        if which is None:
            self.global_step_number += int(played.item())
            played.zero_()
        r = env.read()
        return torch.stack([r["completion_time"].double(), r["delay_time_sum"].double(), r["energy_consumption"].double()], 1)

    def run_one_epoch(self):
        """:199-240 on the whole batch.  Returns mean (completion_time, delay_time_sum, energy_consumption) of the
        controller's episode."""
        baselines = torch.stack([self._episode(which=p) for p in self.objectives_policy.values()])   # [3, N, 3]
        objectives_value = baselines.min(0).values                                                    # :223
        out = self._episode(which=None, objectives=objectives_value)
        self.episode_number += 1
        return tuple(float(v) for v in out.mean(0))

    def run_n_episodes(self, n=None):
        n = self.hyper_parameters["num_episodes_to_run"] if n is None else n
        return [self.run_one_epoch() for _ in range(n)]

    # -- learning -------------------------------------------------------------------------------
    def time_for_critic_and_actor_to_learn(self):
        """:287-291; a vector step advances global_step_number by the number of live environments, so the
        modulo test becomes a threshold crossing."""
        hp = self.hyper_parameters
        if self.global_step_number <= hp["min_steps_before_learning"] or not self.enough_experiences_to_learn_from(self.memory, hp["batch_size"]):
            return False
        if self.global_step_number >= self._next_learn:
            self._next_learn = (self.global_step_number // hp["update_every_n_steps"] + 1) * hp["update_every_n_steps"]
            return True
        return False

    def _optimise(self, optimizer, network, loss):
        if network is not None:
            self.buckets[network].zero_()             # (gradients live in the all-reduce bucket)
        else:
            optimizer.zero_grad()
        (loss / fdist.world_size()).backward()
        if network is not None:
            self.buckets[network].all_reduce()
            torch.nn.utils.clip_grad_norm_(network.parameters(), self.hyper_parameters["gradient_clipping_norm"])
        optimizer.step()

    def learn(self):
        """:293-306,335-352.  On the GPU the update (three losses, four backward passes, four Adam steps, two soft target
        updates: ~300 small launches, 6 ms of host time for 0.5 ms of device work) is captured once into a HIP graph and
        replayed on static copies of the sampled batch; sampling itself stays outside (its size is a host value)."""
        batch = self.memory.sample()
        if not (self.use_graph and self.device.type == "cuda" and not fdist.is_distributed()):
            return self._learn_body(batch)
        g = self._learn_graph
        if g is None:                                   # first update: eager (creates the optimiser state the capture needs)
            self._learn_graph = {"graph": None}
            return self._learn_body(batch)
        if g["graph"] is None:
            g["in"] = tuple(t.clone() for t in batch)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="relaxed"):
                g["out"] = self._learn_body(g["in"])
            g["graph"] = graph
        for dst, src in zip(g["in"], batch):
            dst.copy_(src)
        g["graph"].replay()
        self._last_losses = g["out"]
        return self._last_losses

    def _learn_body(self, batch):
        hp = self.hyper_parameters
        alpha = self.alpha.detach() if torch.is_tensor(self.alpha) else self.alpha
        qf1_loss, qf2_loss = sac_critic_losses(self.actor_local, self.critic_local, self.critic_local_2, self.critic_target,
                                               self.critic_target_2, alpha, batch, hp["discount_rate"])
        self._optimise(self.critic_optimizer, self.critic_local, qf1_loss)
        self._optimise(self.critic_optimizer_2, self.critic_local_2, qf2_loss)
        self.soft_update_of_target_network(self.critic_local, self.critic_target, hp["tau"])
        self.soft_update_of_target_network(self.critic_local_2, self.critic_target_2, hp["tau"])
        policy_loss, log_pi = sac_actor_loss(self.actor_local, self.critic_local, self.critic_local_2, alpha, batch[0])
        self._optimise(self.actor_optimizer, self.actor_local, policy_loss)
        if self.automatic_entropy_tuning:
            alpha_loss = -(self.log_alpha * (log_pi + self.target_entropy).detach()).mean()      # :335-338
            self.alpha_optim.zero_grad()
            alpha_loss.backward()
            if fdist.is_distributed():
                torch.distributed.all_reduce(self.log_alpha.grad)
                self.log_alpha.grad /= fdist.world_size()
            self.alpha_optim.step()
            with torch.no_grad():
                self.alpha.copy_(self.log_alpha.exp())    # (in place: a captured graph keeps reading this tensor)
        self._last_losses = (qf1_loss.detach(), qf2_loss.detach(), policy_loss.detach())     # read lazily: no sync here
        return self._last_losses

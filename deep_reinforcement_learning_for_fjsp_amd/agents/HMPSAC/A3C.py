"""Batched counterpart of the reference's agents/HMPSAC/A3C_v5.{1,2,3}.py: the double-actor advantage
actor-critic that trains HMPSAC's three lower-level objective policies (makespan / tardiness / energy:
`reward_policy` 0 / 1 / 2 of MO_DFJSP's step, the only difference between the three files).

What is kept from the reference (A3C_v5.1.py), by line:
  TaskPolicyNet 30->12, MachinePolicyNet 31->10, CriticNet 30->1      :35-97,116-118  (3 x 200 hidden)
  the machine policy sees the state with the chosen task rule appended :268
  pick_action_and_log_prob: Categorical sample, epsilon-random override :320-337
  calculate_new_exploration: 1/(1 + episodes/denominator), scattered    :311-318
      by a factor in [1/2, 2] per worker (here: per environment)
  discounted returns, gamma 0.99, z-scored per episode (np.std, +1e-5)  :373-391
  critic loss mean (G - V)^2, advantages G - V detached                 :403-411
  actor losses -log_prob * advantage, mean over the episode             :413-418
  gradient-norm clip 1.0, Adam(lr 3e-4, eps 1e-4)                       :420-440,112-114
  test episode with the global policy (sampled, no epsilon)             :287-301

What changes (MI355X-first): the reference's parallelism is `cpu_count() - 5` worker PROCESSES, each
playing one CPU environment and pushing clipped gradients through `multiprocessing.Queue`s to an
optimiser process (:125-187).  Here one "worker episode" is one environment of a `BatchedMODFJSP` batch on
the GPU; the per-episode losses are averaged over the batch, and across GPUs the flat gradient of each
network is all-reduced once per optimiser step (RCCL) -- synchronous instead of asynchronous updates.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn
from torch.distributions import Categorical
from torch.optim import Adam

from ..Base_Agent import Base_Agent
from ..linear import run_layers
from ... import distributed as fdist
from ...utilities.data_structures.Config import Config


def _mlp(input_size, hidden_size, hidden_layer, output_size):
    layers = nn.ModuleList([nn.Linear(input_size, hidden_size), nn.ReLU()])
    for _ in range(hidden_layer - 1):
        layers.append(nn.Linear(hidden_size, hidden_size))
        layers.append(nn.ReLU())
    layers.append(nn.Linear(hidden_size, output_size))
    return layers


class TaskPolicyNet(nn.Module):
    """A3C_v5.1.py:35-53 (parameter names layers_1.* as in the reference's checkpoints)"""

    def __init__(self, input_size_1, hidden_size, hidden_layer_1, output_size_1):
        super().__init__()
        self.name = "task_policy"
        self.layers_1 = _mlp(input_size_1, hidden_size, hidden_layer_1, output_size_1)

    def forward(self, x):
        return F.softmax(run_layers(self.layers_1, x), dim=-1)


class MachinePolicyNet(nn.Module):
    """A3C_v5.1.py:57-75"""

    def __init__(self, input_size_2, hidden_size, hidden_layer_2, output_size_2):
        super().__init__()
        self.name = "machine_policy"
        self.layers_2 = _mlp(input_size_2, hidden_size, hidden_layer_2, output_size_2)

    def forward(self, x):
        return F.softmax(run_layers(self.layers_2, x), dim=-1)


class CriticNet(nn.Module):
    """A3C_v5.1.py:79-97"""

    def __init__(self, input_size, hidden_size, hidden_layer, output_size):
        super().__init__()
        self.layers = _mlp(input_size, hidden_size, hidden_layer, output_size)

    def forward(self, x):
        return run_layers(self.layers, x)


def episode_returns(rewards, valid, gamma):
    """:373-383 per environment over [T, N] f64 tensors (the reference scans Python floats)."""
    T = rewards.shape[0]
    g = torch.zeros_like(rewards[0])
    out = torch.zeros_like(rewards)
    for t in range(T - 1, -1, -1):
        g = torch.where(valid[t] > 0, rewards[t] + gamma * g, g)
        out[t] = torch.where(valid[t] > 0, g, torch.zeros_like(g))
    return out


def zscore_returns(G, valid):
    """:385-391 per environment: (G - mean) / (population std + 1e-5) over the episode's steps."""
    m = (valid > 0).to(G.dtype)
    n = m.sum(0).clamp(min=1)
    mean = (G * m).sum(0) / n
    std = ((((G - mean) ** 2) * m).sum(0) / n).sqrt()
    return (G - mean) / (std + 1e-5) * m


def a2c_losses(task_log_prob, machine_log_prob, values, returns, valid, form="reference"):
    """:403-418 averaged over the batch: every environment's episode contributes one worker's loss, the batch loss is the
    mean over environments with at least one step.

    form "reference" (default) is the arithmetic the reference ships: the critic's outputs are [1, 1] tensors, so
    `torch.cat(self.critic_outputs)` is [T, 1] and `torch.Tensor(returns) - critic_values` (:405) BROADCASTS to [T, T] --
    entry (a, b) = G_b - V_a.  The critic loss is therefore the mean over all PAIRS of (G_b - V_a)^2, and the actors'
    `-log_prob * advantages` (:414-416, [T] * [T, T]) is the mean over pairs of -log_prob_b (G_b - V_a): the baseline of
    every step is the episode's MEAN value, not the value of its own state.  Computed here without the T x T matrix:
        mean_ab (G_b - V_a)^2 = mean(G^2) - 2 mean(G) mean(V) + mean(V^2),   mean_ab -lp_b (G_b - V_a) = -mean_b lp_b (G_b - mean(V))
    (fixtures generated by the reference's own methods pin it: tests/golden/make_agent_fixtures_ref.py).
    form "per_step": the textbook advantage G_t - V(s_t) per step (what round 2 shipped)."""
    m = (valid > 0).to(values.dtype)
    n = m.sum(0)
    live = (n > 0).to(values.dtype)
    n = n.clamp(min=1)
    per_env = lambda x: ((x * m).sum(0) / n * live).sum() / live.sum().clamp(min=1)
    if form == "per_step":
        adv = (returns - values).detach()
        return per_env((returns - values) ** 2), per_env(-1.0 * task_log_prob * adv), per_env(-1.0 * machine_log_prob * adv)
    mean_t = lambda x: (x * m).sum(0) / n                                   # per environment, over its own steps
    g1, g2, v1, v2 = mean_t(returns), mean_t(returns ** 2), mean_t(values), mean_t(values ** 2)
    over_envs = lambda x: (x * live).sum() / live.sum().clamp(min=1)
    critic = over_envs(g2 - 2.0 * g1 * v1 + v2)
    adv = (returns - v1.detach().unsqueeze(0))                              # G_b - mean_a V_a
    return critic, per_env(-1.0 * task_log_prob * adv), per_env(-1.0 * machine_log_prob * adv)


class SharedAdamRule(torch.optim.Optimizer):
    """The update rule of the reference's SharedAdam (utilities/Utility_Functions.py:55-112), which is NOT torch.optim.Adam's:
    eps is added to sqrt(v) BEFORE the bias correction of the second moment,
        p -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps),
    so at the reference's eps = 1e-4 the first steps behave like Adam with an eps some 30 times larger.  (What the reference
    shares between processes -- the moments in shared memory -- is replaced by the gradient all-reduce of the batched trainer.)"""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.steps = 0
        for group in self.param_groups:
            for p in group["params"]:
                self.state[p]["exp_avg"] = torch.zeros_like(p)
                self.state[p]["exp_avg_sq"] = torch.zeros_like(p)

    @torch.no_grad()
    def step(self, closure=None):
        import math
        self.steps += 1
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            gs = [p.grad for p in ps]
            ms = [self.state[p]["exp_avg"] for p in ps]
            vs = [self.state[p]["exp_avg_sq"] for p in ps]
            b1, b2 = group["betas"]
            torch._foreach_mul_(ms, b1); torch._foreach_add_(ms, gs, alpha=1 - b1)
            torch._foreach_mul_(vs, b2); torch._foreach_addcmul_(vs, gs, gs, value=1 - b2)
            den = torch._foreach_sqrt(vs)
            torch._foreach_add_(den, group["eps"])
            step_size = group["lr"] * math.sqrt(1 - b2 ** self.steps) / (1 - b1 ** self.steps)
            torch._foreach_addcdiv_(ps, ms, den, value=-step_size)


class DA3C(Base_Agent, Config):
    """Trainer of one objective's (task policy, machine policy) pair.

    make_train_env() -> BatchedMODFJSP with fresh random instances (generated_new_environment, :248-253:
    DDT in [0.5, 1.5], M in [10, 20], S in [1, 5]); test_env: BatchedMODFJSP over the test folder(s);
    reward_policy: 0 makespan (v5.1), 1 tardiness (v5.2), 2 energy (v5.3)."""

    OBJECTIVE_KEY = {0: "completion_time", 1: "delay_time_sum", 2: "energy_consumption"}

    def __init__(self, make_train_env, test_env, reward_policy=0, hidden_size=200, hidden_layer=3, hyper=None,
                 seed=0, max_steps=4096, state_size=30, actions_size=(12, 10), loss_form="reference"):
        Base_Agent.__init__(self)
        Config.__init__(self)
        self.hp = dict(self.hyper_parameters["DA3C"])
        self.hp.update(hyper or {})
        self.make_train_env, self.environment_test = make_train_env, test_env
        self.device = test_env.device
        # reward_policy None: an environment without reward policies (agents/DA3C/DA3C_double_actor.py trains the same
        # networks at sizes 20 -> 6 / 21 -> 5 on SO_DFJSP, objective = total tardiness)
        self.reward_policy = None if reward_policy is None else int(reward_policy)
        self.state_size, self.actions_size = int(state_size), [int(actions_size[0]), int(actions_size[1])]
        rng = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.actor_task_model = TaskPolicyNet(self.state_size, hidden_size, hidden_layer, self.actions_size[0]).to(self.device)
        self.actor_machine_model = MachinePolicyNet(self.state_size + 1, hidden_size, hidden_layer, self.actions_size[1]).to(self.device)
        self.critic_model = CriticNet(self.state_size, hidden_size, hidden_layer, 1).to(self.device)
        torch.random.set_rng_state(rng)
        lr = self.hp["learning_rate"]
        self.nets = (self.actor_task_model, self.actor_machine_model, self.critic_model)
        self.loss_form = loss_form           # a2c_losses: "reference" (the arithmetic the reference ships) or "per_step"
        self.optimizers = tuple(SharedAdamRule(n.parameters(), lr=lr, eps=1e-4) for n in self.nets)      # :112-114
        self.buckets = tuple(fdist.FlatGradBucket(n.parameters()) for n in self.nets)
        self.max_steps = max_steps
        self.objective_min = float("inf")
        self.best_state = None
        self.last_losses = None

    # -- acting ---------------------------------------------------------------------------------
    @torch.no_grad()
    def _act(self, state, eps=None):
        """:266-270: task rule from the task policy, machine rule from the machine policy on [state, task rule];
        eps: per-environment exploration rates or None (test episodes)."""
        s = state.float()
        a_t = Categorical(self.actor_task_model(s), validate_args=False).sample()
        if eps is not None:
            rnd = torch.randint(0, self.actions_size[0], a_t.shape, device=s.device)
            a_t = torch.where(torch.rand(a_t.shape, device=s.device) <= eps, rnd, a_t)
        s2 = torch.cat([s, a_t.float().unsqueeze(1)], 1)
        a_m = Categorical(self.actor_machine_model(s2), validate_args=False).sample()
        if eps is not None:
            rnd = torch.randint(0, self.actions_size[1], a_m.shape, device=s.device)
            a_m = torch.where(torch.rand(a_m.shape, device=s.device) <= eps, rnd, a_m)
        return a_t, a_m

    def calculate_new_exploration(self, n):
        """:311-318, one draw per environment."""
        eps = 1.0 / (1.0 + (self.episode_number / self.hp["epsilon_decay_rate_denominator"]))
        d = self.hp["exploration_worker_difference"]
        u = torch.rand(n, device=self.device)
        return (eps / d + u * (eps * d - eps / d)).clamp(min=0.0)

    def _rollout(self, env, eps):
        if self.reward_policy is not None:
            env.set_objective(self.reward_policy)
        state = env.reset().clone()
        N = env.N
        done = torch.zeros(N, dtype=torch.uint8, device=self.device)
        S, A, R, V = [], [], [], []
        pair = torch.zeros(N, 2, dtype=torch.uint8, device=self.device)
        for t in range(self.max_steps):
            active = (done == 0)
            a_t, a_m = self._act(state, eps)
            pair[:, 0], pair[:, 1] = a_t.to(torch.uint8), a_m.to(torch.uint8)
            nxt, rew, dn = env.step(pair)
            S.append(state.float()); A.append(torch.stack([a_t, a_m], 1)); R.append(rew.clone()); V.append(active)
            state, done = nxt.clone(), dn.clone()
            if t % 16 == 15 and bool((done != 0).all()):         # (a host round trip: not every step)
                break
        return torch.stack(S), torch.stack(A), torch.stack(R), torch.stack(V)

    # -- one round ------------------------------------------------------------------------------
    def run_one_round(self):
        """One episode per environment of a fresh training batch + ONE optimiser step per network
        (the batched form of every worker pushing one gradient, :255-283), then the test episode (:285-301).
        Returns the test objective."""
        env = self.make_train_env()
        states, actions, rewards, valid = self._rollout(env, self.calculate_new_exploration(env.N))
        self.learn_from_rollout(states, actions, rewards, valid)
        self.episode_number += 1
        objective = self.run_test()
        if objective < self.objective_min:                                             # :299-301 save_actor_model
            self.objective_min = objective
            self.best_state = (copy.deepcopy(self.actor_task_model.state_dict()),
                               copy.deepcopy(self.actor_machine_model.state_dict()))
        return objective

    def learn_from_rollout(self, states, actions, rewards, valid):
        """calculate_total_loss + put_gradients_in_queue + update_shared_model (:363-437,164-187) for a batch of
        episodes: states [T, N, S] f32, actions [T, N, 2] int64, rewards [T, N] f64, valid [T, N] (1 = a step of the
        episode).  One optimiser step per network; returns (critic, task, machine) losses."""
        T, N = valid.shape
        G = zscore_returns(episode_returns(rewards, valid, self.hp["discount_rate"]), valid).float()
        flat = states.reshape(T * N, -1)
        a_t, a_m = actions[..., 0].reshape(-1), actions[..., 1].reshape(-1)
        lp_t = Categorical(self.actor_task_model(flat)).log_prob(a_t).reshape(T, N)
        lp_m = Categorical(self.actor_machine_model(torch.cat([flat, a_t.float().unsqueeze(1)], 1))).log_prob(a_m).reshape(T, N)
        values = self.critic_model(flat).reshape(T, N)
        c_loss, t_loss, m_loss = a2c_losses(lp_t, lp_m, values, G, valid, self.loss_form)
        w = fdist.world_size()
        for net, opt, bucket, loss in zip(self.nets, self.optimizers, self.buckets, (t_loss, m_loss, c_loss)):
            bucket.zero_()                            # (gradients live in the all-reduce bucket)
            (loss / w).backward()
            bucket.all_reduce()
            torch.nn.utils.clip_grad_norm_(net.parameters(), self.hp["gradient_clipping_norm"])
            opt.step()
        self.last_losses = (float(c_loss.detach()), float(t_loss.detach()), float(m_loss.detach()))
        return self.last_losses

    @torch.no_grad()
    def run_test(self):
        env = self.environment_test
        if self.reward_policy is not None:
            env.set_objective(self.reward_policy)
        state = env.reset().clone()
        pair = torch.zeros(env.N, 2, dtype=torch.uint8, device=self.device)
        for t in range(self.max_steps):
            a_t, a_m = self._act(state)
            pair[:, 0], pair[:, 1] = a_t.to(torch.uint8), a_m.to(torch.uint8)
            state, _, done = env.step(pair)
            state = state.clone()
            if t % 16 == 15 and bool((done != 0).all()):
                break
        key = self.OBJECTIVE_KEY[1 if self.reward_policy is None else self.reward_policy]
        return float(env.read()[key].double().mean())

    def save_actor_model(self, folder):
        """:241-246 file names, so SAC_Discrete.load_policy_model finds them."""
        import os
        task, machine = self.best_state or (self.actor_task_model.state_dict(), self.actor_machine_model.state_dict())
        torch.save(task, os.path.join(folder, "actor_task_model.path"))
        torch.save(machine, os.path.join(folder, "actor_machine_model.path"))

"""Batched counterpart of the reference's agents/MPPPO/MPPPO.py (on-policy update
path).  The MLPs, the return normalisation and the
clipped-PPO update stay in PyTorch-ROCm (MFMA only in the dense GEMMs); rollouts
come from the HIP environment batch and land in the C-ABI rollout buffer.

What is kept from the reference, by line:
  ActorNet / CriticNet                 MPPPO.py:31-67   (Linear+ReLU stack, softmax head)
  pick_action_and_log_prob             :272-284         (Categorical sample, epsilon-random override)
  calculate_discounted_returns         :301-312         (reverse scan, f32 element arithmetic)
  normalise to [0,1] then standardise  :258-261         (+1e-8, unbiased std)
  advantages = G - V(s)                :263
  ratio / clipped surrogate            :325-352         (exp(new) / (exp(old) + 1e-8), clip 0.2)
  Adam(lr 3e-4, eps 1e-4), grad-clip 1 :145-147,358-370
  equalise_policies                    :372-375
Two defects of the shipped script are fixed rather than replicated (SURVEY.md row
a21): the critic loss is not detached before backward (:319 makes the critic
never train), and equalise_policies copies `.data` (:375 `algorithm_means`
raises AttributeError at the end of the first episode).
Kept as the reference has them (tests/test_ppo_update.py pins both):
  exploration rate of a round          :240-241         eps = 1 / (1 + episode / denominator), then ONE draw
                                                        max(0, uniform(eps / 3, eps * 3)) per round -- from the
                                                        agent's own seeded `random.Random` (the reference uses the
                                                        unseeded global one), shared by every env of the batch
  multi_policy_update                  :192-203         policy P scores ITS OWN test objectives under every
                                                        policy's weight vector and moves towards the policy at
                                                        the arg-min index (see MPPPO.multi_policy_update)
"""
import random
import torch
import torch.nn.functional as F
from torch import nn, optim
from torch.distributions import Categorical

from ..Base_Agent import Base_Agent
from ..linear import run_layers
from ... import distributed as fdist

HYPER = {   # utilities/data_structures/Config.py "MP_PPO"
    "learning_rate": 0.0003, "discount_rate": 0.99, "clip_epsilon": 0.2, "gradient_clipping_norm": 1.0,
    "learning_iterations_per_round_actor": 10, "learning_iterations_per_round_critic": 10,
    "epsilon_decay_rate_denominator": 10, "normalized_rewards": True, "standardized_rewards": True, "tau": 0.005,
}


class ActorNet(nn.Module):
    """MPPPO.py:31-48"""

    def __init__(self, input_size, hidden_size, hidden_layer, output_size):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(input_size, hidden_size), nn.ReLU()])
        for _ in range(hidden_layer - 1):
            self.layers.append(nn.Linear(hidden_size, hidden_size))
            self.layers.append(nn.ReLU())
        self.layers.append(nn.Linear(hidden_size, output_size))

    def forward(self, x):
        return F.softmax(run_layers(self.layers, x), dim=-1)

    def log_prob(self, x, actions):
        """log of the softmax probability of `actions` -- what Categorical(self(x)).log_prob(actions) returns
        (MPPPO.py:327-328), computed as log_softmax + gather: two passes over the [samples, actions] matrix
        instead of softmax, renormalisation, clamp, log and gather."""
        return F.log_softmax(run_layers(self.layers, x), dim=-1).gather(1, actions.unsqueeze(1)).squeeze(1)


class CriticNet(nn.Module):
    """MPPPO.py:51-67"""

    def __init__(self, input_size, hidden_size, hidden_layer, output_size):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(input_size, hidden_size), nn.ReLU()])
        for _ in range(hidden_layer - 1):
            self.layers.append(nn.Linear(hidden_size, hidden_size))
            self.layers.append(nn.ReLU())
        self.layers.append(nn.Linear(hidden_size, output_size))

    def forward(self, x):
        return run_layers(self.layers, x)


# ----------------------------------------------------------------------------- math
def discounted_returns(rewards, valid, gamma):
    """MPPPO.py:301-312 for [T, N] f32 tensors: G_t = r_t + gamma * G_{t+1} per env, walking back over
    valid rows, f32 multiply then f32 add (the reference's tensor-element arithmetic)."""
    T = rewards.shape[0]
    g = torch.zeros_like(rewards[0])
    out = torch.zeros_like(rewards)
    gam = torch.tensor(gamma, dtype=rewards.dtype, device=rewards.device)
    for t in range(T - 1, -1, -1):
        g_new = rewards[t] + gam * g
        g = torch.where(valid[t] > 0, g_new, g)
        out[t] = torch.where(valid[t] > 0, g, torch.zeros_like(g))
    return out


def normalise_returns(G, valid, normalized=True, standardized=True):
    """MPPPO.py:258-261 applied per env (each env's episode is one reference episode)."""
    m = valid > 0
    if normalized:
        big = torch.finfo(G.dtype).max
        gmin = torch.where(m, G, torch.full_like(G, big)).min(0).values
        gmax = torch.where(m, G, torch.full_like(G, -big)).max(0).values
        G = (G - gmin) / (gmax - gmin + 1e-8)
    if standardized:
        n = m.sum(0).clamp(min=1).to(G.dtype)
        mean = (G * m).sum(0) / n
        var = (((G - mean) ** 2) * m).sum(0) / (n - 1).clamp(min=1)    # torch .std() is unbiased
        G = (G - mean) / (var.sqrt() + 1e-8)
    return G * m


def actor_loss_terms(new_log_prob, old_log_prob, advantages, clip_epsilon):
    """MPPPO.py:325-352 per sample (the caller takes -mean over valid samples)."""
    ratio = torch.exp(new_log_prob) / (torch.exp(old_log_prob) + 1e-8)
    return torch.min(advantages * ratio, advantages * torch.clamp(ratio, 1.0 - clip_epsilon, 1.0 + clip_epsilon))


class PPOLearner(object):
    """Networks + optimisers + one clipped-PPO learning round over a [T, N] rollout.

    Data parallel over env shards: every rank holds the same parameters, computes the
    gradient of (sum of its samples' losses) / (global sample count), the flat gradient
    bucket is all-reduced (ONE RCCL call per optimiser step) and clipped
    after the reduction so all ranks apply the identical update."""

    def __init__(self, state_size, action_size, hidden_size=128, hidden_layer=2, critic_hidden_layer=2,
                 device="cpu", seed=0, hyper=None, train_critic=True):
        self.hp = dict(HYPER)
        self.hp.update(hyper or {})
        self.device = torch.device(device)
        self.fused_learn = True            # GPU batches of >= 32 k samples train through agents/fused_mlp.py
        self.mfma_learn = True             # ... with forward + loss + backward of a network in ONE launch (csrc/fjsp_mlp_train.hip)
        self._fused = None
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)            # identical initial parameters on every rank
        self.actor_new = ActorNet(state_size, hidden_size, hidden_layer, action_size).to(self.device)
        self.actor_old = ActorNet(state_size, hidden_size, hidden_layer, action_size).to(self.device)
        self.critic = CriticNet(state_size, hidden_size, critic_hidden_layer, 1).to(self.device)
        torch.random.set_rng_state(gen_state)
        Base_Agent.copy_model_over_dict(self.actor_new, self.actor_old)
        fused = self.device.type == "cuda"       # one multi-tensor kernel per optimiser step instead of one per parameter
        adam = dict(lr=self.hp["learning_rate"], eps=1e-4, fused=fused, capturable=fused)   # capturable: step count on the device
        self.actor_optimizer = optim.Adam(self.actor_new.parameters(), **adam)
        self.critic_optimizer = optim.Adam(self.critic.parameters(), **adam)
        # which trainer owns the optimiser state -- "fused" (agents/fused_mlp.py, its own Adam moments) or "eager" (the torch
        # optimisers above) -- is decided ONCE, at the first learn(), from the smallest sample count over the ranks, and
        # never changes: a learner that switched per call would alternate between two sets of Adam moments, and ranks that
        # decided from their own shard could take different paths and drift apart
        self._path = None
        self.graph_learn = False         # replay the learning round from a HIP graph once its sample count repeats
        if self.device.type == "cuda":
            self._fused_nets()           # re-homes the parameters into flat buffers NOW: before any graph captures their addresses
        self._learn_graph = None
        self.action_size = action_size
        self.train_critic = train_critic
        self.actor_bucket = fdist.FlatGradBucket(self.actor_new.parameters())
        self.critic_bucket = fdist.FlatGradBucket(self.critic.parameters())

    @torch.no_grad()
    def act(self, states, epsilon=0.0, generator=None):
        """pick_action_and_log_prob (:272-284) for a batch of states [N, S] (f32)."""
        probs = self.actor_new(states)
        dist = Categorical(probs, validate_args=False)      # (the validation syncs with the host: illegal in a graph capture)
        action = dist.sample()
        if torch.is_tensor(epsilon) or epsilon > 0.0:
            u = torch.rand(action.shape, device=action.device, generator=generator)
            rnd = torch.randint(0, self.action_size, action.shape, device=action.device, generator=generator)
            action = torch.where(u <= epsilon, rnd, action)
        return action, dist.log_prob(action)

    def learn(self, states, actions, old_log_prob, returns, valid):
        """critic_actor_learn (:314-323) on flattened [T*N] samples; returns (critic_loss, actor_loss)."""
        hp = self.hp
        S = states.shape[-1]
        states = states.reshape(-1, S)
        actions = actions.reshape(-1)
        old_log_prob = old_log_prob.reshape(-1).detach()
        returns = returns.reshape(-1).detach()
        vmask = valid.reshape(-1) > 0
        if states.is_cuda and states.shape[0] >= (1 << 15):
            # rows of finished environments carry no sample (up to a third of a [T, N] rollout): drop them once
            # per round instead of pushing them through every one of the 20 forward/backward passes
            idx = torch.nonzero(vmask).reshape(-1)
            states, actions, old_log_prob, returns = states[idx], actions[idx], old_log_prob[idx], returns[idx]
            vmask = torch.ones(idx.shape[0], dtype=torch.bool, device=states.device)
        m = vmask.to(states.dtype)
        # big GPU batches (rows of finished environments already dropped): the fused trainer (agents/fused_mlp.py)
        if self._path is None:
            n_min = fdist.all_reduce_scalar_min(torch.tensor(float(states.shape[0]), device=states.device))
            want = self.fused_learn and states.is_cuda and self._fused_nets() is not None
            self._path = "fused" if (want and float(n_min) >= float(1 << 15)) else "eager"
        self._use_fused = self._path == "fused"
        if self.graph_learn and states.is_cuda and not fdist.is_distributed():
            return self._learn_graphed(states, actions, old_log_prob, returns, m)
        count = fdist.all_reduce_scalar_sum(m.sum())          # global number of samples
        c_loss, a_loss = self._learn_body(states, actions, old_log_prob, returns, m, count)
        return float(c_loss), float(a_loss)

    def _learn_body(self, states, actions, old_log_prob, returns, m, count):
        """The 10 critic + 10 actor iterations of critic_actor_learn (:314-323); returns the last (critic, actor) losses
        as tensors."""
        hp = self.hp
        if count is None:
            count = m.sum()                # single process: the sample count as a device scalar (no host round trip)
        if getattr(self, "_use_fused", False):
            return self._learn_body_fused(states, actions, old_log_prob, returns, count)
        with torch.no_grad():
            advantages = returns - self.critic(states).squeeze(1)                       # :263
        c_loss = a_loss = None
        for _ in range(hp["learning_iterations_per_round_critic"]):
            critic_out = self.critic(states).squeeze(1)
            c_loss = (((critic_out - returns) ** 2) * m).sum() / count                 # F.mse_loss, :318
            if self.train_critic:
                self.critic_bucket.zero_()           # (gradients live in the all-reduce bucket)
                c_loss.backward()
                self.critic_bucket.all_reduce()
                torch.nn.utils.clip_grad_norm_(self.critic.parameters(), hp["gradient_clipping_norm"])
                self.critic_optimizer.step()
            new_log_prob = self.actor_new.log_prob(states, actions)
            terms = actor_loss_terms(new_log_prob, old_log_prob, advantages, hp["clip_epsilon"])
            a_loss = -(terms * m).sum() / count                                         # -torch.mean(...), :351
            self.actor_bucket.zero_()
            a_loss.backward()
            self.actor_bucket.all_reduce()
            torch.nn.utils.clip_grad_norm_(self.actor_new.parameters(), hp["gradient_clipping_norm"])
            self.actor_optimizer.step()
        self.equalise_policies()
        return c_loss.detach(), a_loss.detach()

    def _learn_graphed(self, states, actions, old_log_prob, returns, m):
        """A learning round is ~1 200 small launches whose host cost rivals their device time.  With a fixed set
        of environments the number of valid samples (one per operation) is the same every round, so the whole
        round is captured once for that sample count and replayed: the first round of a given size runs eagerly
        (it also creates the optimiser state the capture needs), the second is captured, later ones replay."""
        n = states.shape[0]
        g = self._learn_graph
        if g is None or g["n"] != n:
            self._learn_graph = {"n": n, "graph": None}
            c, a = self._learn_body(states, actions, old_log_prob, returns, m, None)
            return float(c), float(a)
        if g["graph"] is None:
            g["in"] = [t.clone() for t in (states, actions, old_log_prob, returns, m)]
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="relaxed"):
                g["out"] = self._learn_body(*g["in"], None)
            g["graph"] = graph
        for dst, src in zip(g["in"], (states, actions, old_log_prob, returns, m)):
            dst.copy_(src)
        g["graph"].replay()
        return float(g["out"][0]), float(g["out"][1])

    def _fused_nets(self):
        """(actor trainer, critic trainer) of agents/fused_mlp.py, created on first use; None when a network has another
        shape than Linear-ReLU-Linear-ReLU-Linear or lives on the CPU."""
        from .. import fused_mlp
        if self._fused is None:
            if not (fused_mlp.supported(self.actor_new.layers, self.device) and fused_mlp.supported(self.critic.layers, self.device)):
                self._fused = False
            else:
                hp = self.hp
                mk = lambda net: fused_mlp.FusedMLP(net.layers, lr=hp["learning_rate"], eps=1e-4, max_norm=hp["gradient_clipping_norm"])
                self._fused = (mk(self.actor_new), mk(self.critic))
        return self._fused or None

    def _learn_body_fused(self, states, actions, old_log_prob, returns, count):
        """_learn_body on the fused trainer: same iteration structure (:314-323), the dense layers on the library GEMMs,
        everything between them in csrc/fjsp_ppo.hip; all samples are valid here (the caller dropped the others)."""
        hp = self.hp
        actor, critic = self._fused_nets()
        count = count.to(torch.float32).reshape(1)
        if fdist.is_distributed():
            reduce = lambda g: torch.distributed.all_reduce(g, op=torch.distributed.ReduceOp.SUM)
        else:
            reduce = None
        actions_f = actions.to(torch.float32).contiguous()
        states, returns, old_log_prob = states.contiguous(), returns.contiguous(), old_log_prob.contiguous()
        c_loss = a_loss = None
        one_launch = self.mfma_learn and actor.mfma_pass_supported() and critic.mfma_pass_supported()
        # the advantages use the critic as it is BEFORE the round's updates (:263).  The first critic iteration's forward pass
        # computes exactly those values: it hands them out and the separate forward pass (two library GEMMs) is not needed
        values_from_first_pass = one_launch and reduce is None and self.train_critic
        advantages = None
        if not values_from_first_pass:
            advantages = (returns - critic.forward(states).squeeze(1)).contiguous()                # :263
        if values_from_first_pass and getattr(self, "two_chains", True):
            # The critic's iterations and the actor's are two independent chains once the advantages exist (the actor never
            # reads the critic's new parameters inside a round, :314-323): they are issued on two streams, so the small
            # launches of one chain (gradient finish, clip + Adam: 0.28 ms of a round in sequence) run under the other's pass.
            n_it = hp["learning_iterations_per_round_critic"]
            main = torch.cuda.current_stream(states.device)
            side = self.__dict__.get("_actor_stream")
            if side is None:
                side = self._actor_stream = torch.cuda.Stream(device=states.device)
            values = torch.empty(states.shape[0], dtype=torch.float32, device=states.device)
            c_loss = critic.train_step(1, states, returns, None, None, count, values_out=values)           # F.mse_loss, :318
            advantages = returns - values                                                                  # :263
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(n_it):
                    a_loss = actor.train_step(0, states, actions_f, old_log_prob, advantages, count, hp["clip_epsilon"])   # :325-352
            for _ in range(n_it - 1):
                c_loss = critic.train_step(1, states, returns, None, None, count)
            main.wait_stream(side)
            for t in (advantages, actions_f, states, old_log_prob, count):
                t.record_stream(side)
            self.equalise_policies()
            return c_loss[0].clone(), a_loss[0].clone()
        for it in range(hp["learning_iterations_per_round_critic"]):
            if one_launch and reduce is None:
                # forward, loss, backward, clip + Adam of a network in three launches (single process: no all-reduce between)
                if self.train_critic:
                    if it == 0:
                        values = torch.empty(states.shape[0], dtype=torch.float32, device=states.device)
                        c_loss = critic.train_step(1, states, returns, None, None, count, values_out=values)   # F.mse_loss, :318
                        advantages = returns - values                                                          # :263
                    else:
                        c_loss = critic.train_step(1, states, returns, None, None, count)
                else:
                    c_loss = critic.train_pass(1, states, returns, None, None, count)
                a_loss = actor.train_step(0, states, actions_f, old_log_prob, advantages, count, hp["clip_epsilon"])   # :325-352
                continue
            if one_launch:
                c_loss = critic.train_pass(1, states, returns, None, None, count)
            else:
                critic.forward(states)
                c_loss = critic.critic_loss(returns, count)
                if self.train_critic:
                    critic.backward()
            if self.train_critic:
                critic.step(reduce)
            if one_launch:
                a_loss = actor.train_pass(0, states, actions_f, old_log_prob, advantages, count, hp["clip_epsilon"])
            else:
                actor.forward(states)
                a_loss = actor.actor_loss(actions_f, old_log_prob, advantages, hp["clip_epsilon"], count)
                actor.backward()
            actor.step(reduce)
        self.equalise_policies()
        return c_loss[0].clone(), a_loss[0].clone()

    def equalise_policies(self):
        """:372-375 with the AttributeError fixed."""
        with torch.no_grad():
            torch._foreach_copy_([p.data for p in self.actor_old.parameters()], [p.data for p in self.actor_new.parameters()])


class FusedSampler(object):
    """pick_action_and_log_prob (MPPPO.py:272-284) of a whole vector step in one launch of the library's
    `fjsp_policy_sample` (csrc/fjsp_rollout_buffer.hip) instead of ~15 small torch kernels: Categorical sample,
    epsilon override, log-probability, and the action already in the environment's u8 encoding.
    pair_div = size of the machine-rule axis for pair-action environments ([6, 5] -> 5), 0 for flat actions."""

    def __init__(self, N, T, n_actions, pair_div, device):
        from ... import _capi
        self._lib, self._check = _capi.lib(), _capi.check
        self.N, self.A, self.div = int(N), int(n_actions), int(pair_div)
        self.eps = torch.zeros((), dtype=torch.float32, device=device)
        self.seed = torch.zeros((), dtype=torch.int64, device=device)
        self.pair = torch.zeros(N, 2, dtype=torch.uint8, device=device)
        self.flat_actions = torch.zeros(T, N, dtype=torch.float32, device=device)
        self.rounds = 0

    def new_round(self, exploration):
        """Fresh random stream + exploration rate for the next rollout (device scalars: a captured graph reads them)."""
        self.rounds += 1
        self.eps.fill_(float(exploration))
        self.seed.fill_((torch.initial_seed() * 0x9E3779B1 + self.rounds * 0x85EBCA77) & 0x7FFFFFFFFFFFFFFF)

    def sample(self, probs, t, log_prob_row):
        import ctypes as C
        p = probs.contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
        ptr = lambda x: C.c_void_p(x.data_ptr())
        self._check(self._lib.fjsp_policy_sample(ptr(p), self.N, self.A, self.div, ptr(self.eps), ptr(self.seed), int(t),
                                                 ptr(self.pair), ptr(self.flat_actions[t]), ptr(log_prob_row), stream))
        return self.pair


def native_actor_params(actor):
    """fjsp_actor_params for an ActorNet of the in-kernel shape (state_size <= 32 -> 128 -> 128 -> n_actions <= 32, on
    the GPU), or None when the network has another shape: device pointers into the nn.Linear parameters themselves
    (the optimiser updates them in place, so the kernels always read the current policy)."""
    from ..._capi import ActorParams
    import ctypes as C
    lin = [l for l in actor.layers if isinstance(l, nn.Linear)]
    if len(lin) != 3 or not lin[0].weight.is_cuda:
        return None
    S, H, A = lin[0].in_features, lin[0].out_features, lin[2].out_features
    if H != 128 or lin[1].in_features != 128 or lin[1].out_features != 128 or lin[2].in_features != 128 or S > 32 or A > 32:
        return None
    tensors = [lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias]
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
        return None
    return ActorParams(*[C.c_void_p(t.data_ptr()) for t in tensors], S, H, A)


def native_actor_forward(actor, states64, out=None):
    """ActorNet.forward through the library's actor kernel (fjsp_actor_forward): f64[n, S] states -> f32[n, A]
    probabilities, the arithmetic the fused policy rollout performs inside the environment kernel."""
    from ... import _capi
    import ctypes as C
    ap = native_actor_params(actor)
    if ap is None:
        raise ValueError("the in-kernel actor is state_size (<= 32) -> 128 -> 128 -> n_actions (<= 32) on the GPU")
    states64 = states64.contiguous()
    n = states64.shape[0]
    probs = torch.empty(n, ap.n_actions, dtype=torch.float32, device=states64.device) if out is None else out
    stream = C.c_void_p(torch._C._cuda_getCurrentRawStream(states64.device.index))
    _capi.check(_capi.lib().fjsp_actor_forward(C.byref(ap), C.c_void_p(states64.data_ptr()), n, C.c_void_p(probs.data_ptr()), stream))
    return probs


def fused_policy_rollout(env, learner, memory, fused, old_log_prob, exploration, T):
    """The whole rollout loop (MPPPO.py:245-252) in ONE launch of fjsp_env_rollout_policy: the actor runs inside the
    environment kernel, rows go straight into `memory`.  Returns False when the batch / network shape is not
    supported (multi-order instances, K > 64, another network size): the caller falls back to the per-step loop."""
    from ... import _capi
    import ctypes as C
    ap = native_actor_params(learner.actor_new)
    if ap is None or ap.state_size != env.state_size:
        return False
    batch = env.batch
    mo = getattr(env, "mo", None)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    memory.clear()
    state = env.reset()
    fused.new_round(exploration)
    stream = C.c_void_p(torch._C._cuda_getCurrentRawStream(batch.device_index))
    rc = _capi.lib().fjsp_env_rollout_policy(batch._h, memory._h, C.byref(ap), p(fused.eps), p(fused.seed), fused.div, int(T), p(mo),
                                            p(state), p(fused.flat_actions), p(old_log_prob), p(batch.state), stream)
    if rc == -5:                      # FJSP_E_UNSUPPORTED
        return False
    _capi.check(rc)
    batch.done.fill_(1)               # (read() reports the per-env flags; the vector mirror is refreshed lazily)
    return True


def _rollout_body(env, learner, memory, old_log_prob, exploration, T, encode_action, pair, check_done, fused=None):
    """T vector steps: policy inference, epsilon override, HIP env step, rollout-buffer append (MPPPO.py:245-252).
    `exploration` is a float (eager) or a 0-dim device tensor (graph capture: no host branch on its value);
    with a FusedSampler the sampling chain is one library launch and the exploration rate lives in the sampler."""
    memory.clear()
    state64 = env.reset().clone()
    done = torch.zeros(env.N, dtype=torch.uint8, device=env.device)
    t = 0
    while t < T:
        active = (done == 0).to(torch.uint8)
        if fused is not None:
            if getattr(fused, "native_actor", False):
                probs = native_actor_forward(learner.actor_new, state64)      # the fused rollout's arithmetic, step by step
            else:
                with torch.no_grad():
                    probs = learner.actor_new(state64.float())
            act_pair = fused.sample(probs, t, old_log_prob[t])
            nxt, rew, dn = env.batch.step(act_pair, mo=getattr(env, "mo", None))
            memory.add_experience(state64, act_pair, rew, nxt, dn, active)
            state64, done = nxt.clone(), dn.clone()
            t += 1
            if check_done and t % 8 == 0 and bool((done != 0).all()):
                break
            continue
        a, lp = learner.act(state64.float(), exploration)
        old_log_prob[t] = lp
        nxt, rew, dn = env.step(encode_action(a))
        memory.add_experience(state64, pair, rew, nxt, dn, active)
        memory.actions[t, :, 0] = a.float()          # the flat action index is what the policy is trained on
        state64 = nxt.clone()
        done = dn.clone()
        t += 1
        if check_done and t % 8 == 0 and bool((done != 0).all()):
            break


class GraphedRollout(object):
    """The whole T-step rollout of one (environment batch, learner) pair captured ONCE into a HIP graph and
    replayed every round: a vector step is a chain of ~20 small launches (MLP, sampling, env kernel, buffer
    append), and replaying the chain from a graph removes the per-launch host cost that dominates the eager
    loop (rocprofv3: 31 ms of a 82 ms PPO round at 4096 envs).  The exploration rate lives in a device scalar
    so that it can change between replays; there is no early exit, every replay plays T steps (finished
    environments idle, their rows are masked by `valid`)."""

    def __init__(self, env, learner, memory, T, encode_action, fused=None):
        self.env, self.learner, self.memory, self.T, self.fused = env, learner, memory, T, fused
        dev = env.device
        self.eps = torch.zeros((), device=dev)
        self.old_log_prob = torch.zeros(T, env.N, device=dev)
        self.pair = torch.zeros(env.N, 2, dtype=torch.uint8, device=dev)
        body = lambda: _rollout_body(env, learner, memory, self.old_log_prob, self.eps, T, encode_action, self.pair, False,
                                     fused)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):               # warm-up outside the capture (library handles, allocator)
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="relaxed"):
            body()

    def run(self, exploration):
        self.eps.fill_(float(exploration))
        if self.fused is not None:
            self.fused.new_round(exploration)
        self.graph.replay()
        return self.memory, self.old_log_prob


def collect_and_learn(env, learner, memory_holder, exploration, max_steps, encode_action, use_graph=False, pair_div=None,
                      fused_rollout=False):
    """One batched episode on `env` with `learner`'s policy + one learning round (MPPPO.py:230-270).

    env.step(action_tensor) must accept what encode_action(flat_action) returns.  Returns
    (memory, losses): the RolloutBuffer used and (critic_loss, actor_loss)."""
    from .Buffer import RolloutBuffer
    hp = learner.hp
    N, device = env.N, env.device
    T = max_steps
    memory = memory_holder.get("memory")
    if memory is None or memory.T < T or memory.N != N or memory.S != env.state_size:
        memory = RolloutBuffer(T, N, env.state_size, device=device.index or 0)
        memory_holder["memory"] = memory
        memory_holder.pop("graph", None)
        memory_holder.pop("fused", None)
    fused = None
    if pair_div is not None and device.type == "cuda":       # one-launch sampling (library kernel)
        fused = memory_holder.get("fused")
        if fused is None or fused.N != N or fused.flat_actions.shape[0] < T or fused.A != learner.action_size or fused.div != pair_div:
            fused = FusedSampler(N, T, learner.action_size, pair_div, device)
            memory_holder["fused"] = fused
            memory_holder.pop("graph", None)
    done_in_kernel = False
    if fused_rollout and fused is not None:
        old_log_prob = memory_holder.get("old_log_prob")
        if old_log_prob is None or old_log_prob.shape != (T, N):
            old_log_prob = torch.zeros(T, N, device=device)
            memory_holder["old_log_prob"] = old_log_prob
        done_in_kernel = fused_policy_rollout(env, learner, memory, fused, old_log_prob, exploration, T)
    if done_in_kernel:
        pass
    elif use_graph:
        g = memory_holder.get("graph")
        if g is None or g.env is not env or g.learner is not learner or g.memory is not memory or g.T != T or g.fused is not fused:
            g = GraphedRollout(env, learner, memory, T, encode_action, fused)
            memory_holder["graph"] = g
        _, old_log_prob = g.run(exploration)
    else:
        old_log_prob = torch.zeros(T, N, device=device)
        pair = torch.zeros(N, 2, dtype=torch.uint8, device=device)
        if fused is not None:
            fused.new_round(exploration)
        _rollout_body(env, learner, memory, old_log_prob, exploration, T, encode_action, pair, True, fused)
    n = len(memory)
    states, actions, _, _, _ = memory.sample()
    if fused is not None:
        actions = fused.flat_actions[:n].unsqueeze(-1)
    valid = memory.valid[:n]
    if hasattr(memory, "normalised_returns"):        # device rollout buffer: scan + normalisation in one launch
        G = memory.normalised_returns(hp["discount_rate"], hp["normalized_rewards"], hp["standardized_rewards"])
    else:
        G = memory.compute_returns(hp["discount_rate"])
        G = normalise_returns(G, valid, hp["normalized_rewards"], hp["standardized_rewards"])
    losses = learner.learn(states, actions[..., 0].long(), old_log_prob[:n], G, valid)
    return memory, losses


def jittered_exploration(episode_number, denominator, rng):
    """MPPPO.py:240-241: the round's epsilon-random rate, one draw per round."""
    eps = 1.0 / (1.0 + episode_number / denominator)
    return max(0.0, rng.uniform(eps / 3.0, eps * 3.0))


class PPO(Base_Agent):
    """The agent loop of MPPPO.py:230-270 over a batch of SO_FJSSP environments (BASELINE config 3).

    `environment` is a BatchedSOFJSSP; the flat action a in [0, 30) is the rule pair
    (a // 5, a % 5) of SO_FJSSP's [6, 5] action space."""

    def __init__(self, environment, hidden_size=128, hidden_layer=2, seed=0, hyper=None, max_steps=None, use_graph=False,
                 fused_sampling=False, fused_rollout=False):
        super().__init__()
        self.environment = environment
        self.use_graph = use_graph
        self.fused_sampling = fused_sampling or fused_rollout
        # fused_rollout: the whole rollout loop in one launch with the actor inside the environment kernel
        # (fjsp_env_rollout_policy); falls back to the per-step loop for shapes the kernel does not take
        self.fused_rollout = fused_rollout
        self.learner_graph = use_graph
        self.device = environment.device
        self.state_size = environment.state_size
        self.action_size = environment.actions_size[0] * environment.actions_size[1]
        self.learner = PPOLearner(self.state_size, self.action_size, hidden_size, hidden_layer, hidden_layer,
                                  device=self.device, seed=seed, hyper=hyper)
        self.learner.graph_learn = bool(use_graph)
        self.hyper_parameters = self.learner.hp
        self.max_steps = max_steps
        self._holder = {}
        self.global_step_number = 0
        self._rng = random.Random(seed)
        self._pair = torch.zeros(environment.N, 2, dtype=torch.uint8, device=self.device)

    @property
    def memory(self):
        return self._holder.get("memory")

    def _encode(self, a):
        n1 = self.environment.actions_size[1]
        self._pair[:, 0] = (a // n1).to(torch.uint8)
        self._pair[:, 1] = (a % n1).to(torch.uint8)
        return self._pair

    def run_one_policy_network(self, exploration=None):
        """One batched episode + one learning round. Returns (mean delay_time_sum, mean makespan, losses)."""
        hp = self.hyper_parameters
        if exploration is None:                                                         # :240-241
            exploration = jittered_exploration(self.episode_number, hp["epsilon_decay_rate_denominator"], self._rng)
        memory, losses = collect_and_learn(self.environment, self.learner, self._holder, exploration,
                                           self.max_steps or 64, self._encode, use_graph=self.use_graph,
                                           pair_div=self.environment.actions_size[1] if self.fused_sampling else None,
                                           fused_rollout=self.fused_rollout)
        self.global_step_number += int(memory.valid[:len(memory)].sum().item())
        self.episode_number += 1
        r = self.environment.read()
        return float(r["delay_time_sum"].double().mean()), float(r["makespan"].double().mean()), losses


class MPPPO(Base_Agent):
    """Multi-policy PPO of agents/MPPPO/MPPPO.py:70-212 on batches of MO_FJSSP_discretes environments.

    actor_number policies share the environment; policy p is trained on the weight vector
    (1 - p/(n-1), p/(n-1)) (:111).  One epoch (:159-164): the completion policy (p = 0) and the tardiness
    policy (p = n-1) run first on the training batch and their per-environment objective values
    normalise the rewards of the weighted policies; every `evolve_every` epochs the policies are pulled
    towards the best policy for their weight vector (:192-205).

    Two defects of the shipped script are fixed, not replicated: completion_min / tardiness_min stay +inf
    there (:134-135, never updated), which zeroes every score of the evolution step; here they track the
    best test objectives seen.  `make_train_env()` is called once per epoch like
    generated_new_environment() (:149-154,160)."""

    def __init__(self, make_train_env, test_env, actor_number=5, hidden_size=200, hidden_layer=5, critic_layer=3,
                 seed=0, hyper=None, max_steps=64, evolve_every=30, fused_sampling=True):
        super().__init__()
        self.make_train_env, self.test_env = make_train_env, test_env
        self.fused_sampling = fused_sampling
        self.device = test_env.device
        self.actor_number = actor_number
        self.policy_tuple = tuple(range(actor_number))
        self.policy_completion, self.policy_tardiness = self.policy_tuple[0], self.policy_tuple[-1]
        self.policy_weight_tuple = self.policy_tuple[1:-1]
        self.weight_vector_dict = {p: (1 - 1 / (actor_number - 1) * p, 1 / (actor_number - 1) * p)
                                   for p in self.policy_tuple}                            # :111
        self.learners = {p: PPOLearner(25, 18, hidden_size, hidden_layer, critic_layer, device=self.device,
                                       seed=seed + p, hyper=hyper) for p in self.policy_tuple}
        self.hyper_parameters = self.learners[0].hp
        self.max_steps, self.evolve_every = max_steps, evolve_every
        self.completion_min = float("inf")
        self.tardiness_min = float("inf")
        self._holder = {}
        self._rng = random.Random(seed)

    def run_one_policy_network(self, environment, policy_number, completion=None, tardiness=None):
        """:230-270 for every environment of the batch. Returns per-env (delay_time_sum, completion_time)."""
        hp = self.hyper_parameters
        eps = jittered_exploration(self.episode_number, hp["epsilon_decay_rate_denominator"], self._rng)   # :240-241
        environment.set_objective(self.weight_vector_dict[policy_number], completion, tardiness)
        collect_and_learn(environment, self.learners[policy_number], self._holder, eps, self.max_steps, lambda a: a,
                          pair_div=0 if self.fused_sampling else None)
        r = environment.read()
        return r["delay_time_sum"].double(), r["completion_time"].double()

    @torch.no_grad()
    def run_one_epoch(self, environment, policy_number, completion=None, tardiness=None):
        """:214-228: evaluation episode (epsilon 0, no learning). Returns mean (delay_time_sum, completion_time)."""
        environment.set_objective(self.weight_vector_dict[policy_number], completion, tardiness)
        state = environment.reset()
        done = torch.zeros(environment.N, dtype=torch.uint8, device=self.device)
        for t in range(self.max_steps):
            a, _ = self.learners[policy_number].act(state.float(), 0.0)
            state, _, done = environment.step(a)
            if t % 8 == 7 and bool((done != 0).all()):
                break
        r = environment.read()
        return float(r["delay_time_sum"].double().mean()), float(r["completion_time"].double().mean())

    def run_n_episodes(self, n):
        """:156-190. Returns the list of per-epoch {policy: (completion, tardiness)} test objectives."""
        history = []
        for _ in range(n):
            env = self.make_train_env()
            _, completion = self.run_one_policy_network(env, self.policy_completion)
            tardiness, _ = self.run_one_policy_network(env, self.policy_tardiness)
            completion, tardiness = completion.clamp(min=1.0), tardiness.clamp(min=1.0)
            for p in self.policy_weight_tuple:
                self.run_one_policy_network(env, p, completion=completion, tardiness=tardiness)
            objs = {}
            t_c, c = self.run_one_epoch(self.test_env, self.policy_completion)
            objs[self.policy_completion] = (c, t_c)
            t, c_t = self.run_one_epoch(self.test_env, self.policy_tardiness)
            objs[self.policy_tardiness] = (c_t, t)
            for p in self.policy_weight_tuple:
                t_p, c_p = self.run_one_epoch(self.test_env, p, completion=max(c, 1.0), tardiness=max(t, 1.0))
                objs[p] = (c_p, t_p)
            self.completion_min = min(self.completion_min, min(v[0] for v in objs.values()))
            self.tardiness_min = min(self.tardiness_min, min(v[1] for v in objs.values()))
            history.append(objs)
            self.episode_number += 1
            if self.episode_number % self.evolve_every == 0:
                self.multi_policy_update(objs)
        return history

    def multi_policy_update(self, policy_objectives, selection="reference"):
        """:192-203.  selection="reference": as the reference computes it -- for policy P the list
        [w_p[0] * C_P / C_min + w_p[1] * T_P / T_min for p in policies] scores P's OWN test objectives under every
        policy's weight vector, and P moves (soft update, tau) towards the policy at the arg-min index.
        selection="own_weight": the transposed reading (every policy's objectives under P's weight vector; P moves
        towards the policy that serves P's weights best) -- arguably what was meant, not what the reference does.
        Returns {policy: index it moved towards}."""
        import copy
        actors = {p: copy.deepcopy(self.learners[p].actor_new) for p in self.policy_tuple}
        critics = {p: copy.deepcopy(self.learners[p].critic) for p in self.policy_tuple}
        tau = self.hyper_parameters["tau"]
        cmin, tmin = max(self.completion_min, 1e-9), max(self.tardiness_min, 1e-9)
        chosen = {}
        for policy in self.policy_tuple:
            if selection == "reference":
                c, t = policy_objectives[policy]
                scores = [self.weight_vector_dict[p][0] * (c / cmin) + self.weight_vector_dict[p][1] * (t / tmin)
                          for p in self.policy_tuple]
            elif selection == "own_weight":
                w = self.weight_vector_dict[policy]
                scores = [w[0] * (policy_objectives[q][0] / cmin) + w[1] * (policy_objectives[q][1] / tmin)
                          for q in self.policy_tuple]
            else:
                raise ValueError("selection must be 'reference' or 'own_weight'")
            best = scores.index(min(scores))
            chosen[policy] = best
            self.soft_update_of_target_network(actors[best], self.learners[policy].actor_new, tau)
            self.soft_update_of_target_network(critics[best], self.learners[policy].critic, tau)
        return chosen

"""Batched counterpart of the reference's agents/DDQN/DDQN.py: double deep Q-learning on batches of
SO_SFJSP environments (makespan objective, 20 flat rule-pair actions, 18-dim state).

What is kept from the reference, by line:
  ActorNet (the Q network)             DDQN.py:27-45    Linear + BatchNorm1d + ReLU stack, softmax head
  ExplorationStrategy                  :48-67           linear epsilon decay per action pick, min 0.01
  pick_action                          :150-166         network in eval mode (BatchNorm running stats), epsilon-greedy
  learn / compute_loss                 :168-209         argmax from the local net, value from the target net,
                                                        r + gamma * Q' * (1 - done), MSE, clip 5.0, soft update tau
  Adam(lr 1e-6, eps 1e-4), buffer 100k, batch 1280      Config.py "DDQN"
  step()                               :106-134         one training episode, one learning session, one greedy
                                                        test episode; best test makespan keeps the model

What changes (MI355X-first): a "training episode" is one episode of EVERY environment of a
`BatchedSOSFJSP` batch stepped by the HIP kernels, transitions go to an HBM-resident replay ring
(`DeviceReplayBuffer`) and never visit the host; under torch.distributed each rank plays its own env
shard and the flat Q-network gradient is all-reduced once per optimiser step (RCCL) before clipping.
The reference learns at most once per episode (`global_step_number % 10 == 0` at episode end, :122-124);
a batched round runs `learning_iterations * updates_per_round` updates.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn, optim

from ..Base_Agent import Base_Agent
from ... import distributed as fdist
from ...utilities.data_structures.Config import Config
from ...utilities.data_structures.Replay_Buffer import DeviceReplayBuffer


class ActorNet(nn.Module):
    """DDQN.py:27-45"""

    def __init__(self, input_size, hidden_size, hidden_layer, output_size):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(input_size, hidden_size), nn.BatchNorm1d(hidden_size), nn.ReLU()])
        for _ in range(hidden_layer - 1):
            self.layers.append(nn.Linear(hidden_size, hidden_size))
            self.layers.append(nn.BatchNorm1d(hidden_size))
            self.layers.append(nn.ReLU())
        self.layers.append(nn.Linear(hidden_size, output_size))

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return F.softmax(x, dim=-1)


class ExplorationStrategy(object):
    """DDQN.py:48-67, vectorised: one decay tick per call (the reference ticks once per action pick)."""

    def __init__(self, start_epsilon, min_epsilon, total_episodes):
        self.epsilon = start_epsilon
        self.min_epsilon = min_epsilon
        self.decay_rate = (start_epsilon - min_epsilon) / total_episodes

    def get_action(self, action_values, turn_off_exploration=False, generator=None):
        if turn_off_exploration:
            self.epsilon = self.min_epsilon
        self.epsilon = max(self.min_epsilon, self.epsilon - self.decay_rate)
        greedy = torch.argmax(action_values, dim=-1)
        n = action_values.shape[0]
        u = torch.rand(n, device=action_values.device, generator=generator)
        rnd = torch.randint(0, action_values.shape[-1], (n,), device=action_values.device, generator=generator)
        return torch.where(u < self.epsilon, rnd, greedy)


def ddqn_loss(q_local, q_target, states, actions, rewards, next_states, dones, discount_rate):
    """DDQN.py:182-209 (compute_loss and its helpers)."""
    with torch.no_grad():
        max_action_indexes = q_local(next_states).detach().argmax(1)                          # :195
        q_targets_next = q_target(next_states).gather(1, max_action_indexes.unsqueeze(1))      # :196
        q_targets = rewards + (discount_rate * q_targets_next * (1 - dones))                  # :201
    q_expected = q_local(states).gather(1, actions.long())                                    # :206
    return F.mse_loss(q_expected, q_targets)


class DDQN(Base_Agent, Config):
    """`make_train_env()` returns a fresh BatchedSOSFJSP per round (generated_new_environment, :99-104:
    random instances with M in [3, 8]); `test_env` is a BatchedSOSFJSP over the test instance(s)."""

    def __init__(self, make_train_env, test_env, hidden_size=200, hidden_layer=3, hyper=None, seed=0,
                 updates_per_round=1, max_steps=None):
        Base_Agent.__init__(self)
        Config.__init__(self)
        self.agent = "DDQN"
        self.hyper_parameters = dict(self.hyper_parameters[self.agent])
        self.hyper_parameters.update(hyper or {})
        hp = self.hyper_parameters
        self.make_train_env, self.environment_test = make_train_env, test_env
        self.device = test_env.device
        self.state_size, self.action_size = 18, 20
        self.memory = DeviceReplayBuffer(hp["buffer_size"], hp["batch_size"], self.state_size, self.device, seed=seed)
        rng = torch.random.get_rng_state()
        torch.manual_seed(seed)                      # identical initial parameters on every rank
        self.q_network_local = ActorNet(self.state_size, hidden_size, hidden_layer, self.action_size).to(self.device)
        torch.random.set_rng_state(rng)
        self.q_network_target = copy.deepcopy(self.q_network_local)
        Base_Agent.copy_model_over(from_model=self.q_network_local, to_model=self.q_network_target)
        self.q_network_optimizer = optim.Adam(self.q_network_local.parameters(), lr=hp["learning_rate"], eps=1e-4)
        self.bucket = fdist.FlatGradBucket(self.q_network_local.parameters())
        self.exploration_strategy = ExplorationStrategy(1.0, 0.01, hp["num_episodes_to_run"])
        self.turn_off_exploration = False
        self.updates_per_round = updates_per_round
        self.max_steps = max_steps
        self.environment = None
        self.global_step_number = 0
        self.completed_time = float("inf")
        self.best_state_dict = None
        self._last_loss = None

    @property
    def last_loss(self):
        return None if self._last_loss is None else float(self._last_loss)

    # -- acting ---------------------------------------------------------------------------------
    @torch.no_grad()
    def pick_action(self, state, turn_off_exploration):
        """:150-166 for a batch of states [N, 18]."""
        self.q_network_local.eval()
        action_values = self.q_network_local(state.float())
        self.q_network_local.train()
        return self.exploration_strategy.get_action(action_values, turn_off_exploration)

    def _play(self, env, greedy, learn_from=False):
        state = env.reset().clone()
        done = torch.zeros(env.N, dtype=torch.uint8, device=self.device)
        limit = self.max_steps or 100000
        played = torch.zeros((), dtype=torch.int64, device=self.device)
        t = 0
        while t < limit:
            active = done == 0
            action = self.pick_action(state, greedy)
            nxt, rew, dn = env.step(action)
            if learn_from:
                self.memory.add_batch(state, action, rew, nxt, dn, active)
                played += active.sum()
            state, done = nxt.clone(), dn.clone()
            t += 1
            if t % 16 == 0 and bool((done != 0).all()):          # one host round trip per 16 vector steps
                break
        self.global_step_number += int(played.item())
        return env.read()

    def step(self):
        """:106-134: one training round, one learning session, one greedy test episode."""
        hp = self.hyper_parameters
        self.environment = self.make_train_env()
        self._play(self.environment, self.turn_off_exploration, learn_from=True)
        if self.enough_experiences_to_learn_from(self.memory, hp["batch_size"]):
            for _ in range(hp["learning_iterations"] * self.updates_per_round):
                self.learn()
        test = self.step_test()
        self.episode_number += 1
        if test < self.completed_time:                                                # :130-133
            self.completed_time = test
            self.best_state_dict = copy.deepcopy(self.q_network_local.state_dict())
        return test

    def step_test(self):
        """:141-148: greedy episode on the test batch; returns its mean completion_time."""
        eps = self.exploration_strategy.epsilon            # a test episode must not consume the decay schedule
        r = self._play(self.environment_test, True)
        self.exploration_strategy.epsilon = eps
        return float(r["completion_time"].double().mean())

    # -- learning -------------------------------------------------------------------------------
    def learn(self, experiences=None):
        """:168-180"""
        hp = self.hyper_parameters
        states, actions, rewards, next_states, dones = self.memory.sample() if experiences is None else experiences
        loss = ddqn_loss(self.q_network_local, self.q_network_target, states, actions, rewards, next_states, dones,
                         hp["discount_rate"])
        self.bucket.zero_()                          # (gradients live in the all-reduce bucket)
        (loss / fdist.world_size()).backward()
        self.bucket.all_reduce()
        torch.nn.utils.clip_grad_norm_(self.q_network_local.parameters(), hp["gradient_clipping_norm"])
        self.q_network_optimizer.step()
        self.soft_update_of_target_network(self.q_network_local, self.q_network_target, hp["tau"])
        self._last_loss = loss.detach()          # read lazily: no sync in the update loop
        return self._last_loss

    def save_policy_network(self, path):
        torch.save(self.best_state_dict or self.q_network_local.state_dict(), path)

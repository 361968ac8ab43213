#!/usr/bin/env python3
"""CROSS-CHECK of tests/golden/make_agent_fixtures_ref.py (round 3), which generates tests/golden/agent_updates.npz by
calling the reference's own agent classes; this file is the round-2 hand transcription of the same updates and is no
longer what writes the fixtures (`make_agent_fixtures_ref.py --compare` prints the differences between the two: none for
DDQN and SAC-discrete; for A3C the transcription had (a) squeezed the critic's [1, 1] outputs, which the reference does not
-- its advantages broadcast to [T, T] -- and (b) taken SharedAdam for torch.optim.Adam, which it is not).

Original header: fixtures that pin the off-policy / actor-critic counterparts to the reference's update
arithmetic.

The reference's agent modules cannot be imported (visdom / nn_builder at import time, `D:/` paths),
so what is below is a torch-CPU TRANSCRIPTION of the three update paths, statement by statement, each citing the
reference lines it follows:

    agents/DDQN/DDQN.py:27-45,168-209           ActorNet (Linear + BatchNorm1d + ReLU stack, softmax head), learn():
                                                double-Q target, F.mse_loss, clip, Adam(eps 1e-4), soft update
    agents/HMPSAC/SAC_Discrete.py:84-138,248-352 PolicyNet / CriticNet, learn(): twin critic losses with the entropy
                                                term, actor loss, entropy-temperature loss, their optimiser steps
    agents/HMPSAC/A3C_v5.1.py:44-110,255-283,363-437 + utilities/Utility_Functions.py:55-112
                                                one worker episode: log-probabilities of the taken rule pair, returns,
                                                z-score, critic / actor losses, clipped gradients applied by SharedAdam
    agents/Base_Agent.py:73-87                  take_optimisation_step, soft_update_of_target_network

At fixed weights and a fixed batch it records: the losses, every parameter after ONE update, the BatchNorm running
statistics, the target networks after the soft update.  tests/test_gpu_agent.py loads the same weights into the
product's agents, runs their update on the MI355X and compares (f32: 1e-5 relative).  Only arrays are stored.

    python tests/golden/make_agent_fixtures.py        # rewrites tests/golden/agent_updates.npz (seconds, CPU)
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.distributions import Categorical
from torch.optim import Adam

HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ networks (transcribed)
class DDQNActorNet(nn.Module):                      # DDQN.py:27-45
    def __init__(self, input_size, hidden_size, hidden_layer, output_size):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(input_size, hidden_size), nn.BatchNorm1d(hidden_size), nn.ReLU()])
        for i in range(hidden_layer - 1):
            self.layers.append(nn.Linear(hidden_size, hidden_size))
            self.layers.append(nn.BatchNorm1d(hidden_size))
            self.layers.append(nn.ReLU())
        self.layers.append(nn.Linear(hidden_size, output_size))

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        x = F.softmax(x, dim=-1)
        return x


class ReluStack(nn.Module):                         # SAC_Discrete.py:84-138, A3C_v5.1.py:44-110 (same stack, named per net)
    def __init__(self, input_size, hidden_size, hidden_layer, output_size, softmax):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(input_size, hidden_size), nn.ReLU()])
        for i in range(hidden_layer - 1):
            self.layers.append(nn.Linear(hidden_size, hidden_size))
            self.layers.append(nn.ReLU())
        self.layers.append(nn.Linear(hidden_size, output_size))
        self.softmax = softmax

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return F.softmax(x, dim=-1) if self.softmax else x


def take_optimisation_step(optimizer, network, loss, clipping_norm=None):       # Base_Agent.py:73-82
    optimizer.zero_grad()
    loss.backward()
    if clipping_norm is not None and network is not None:
        torch.nn.utils.clip_grad_norm_(network.parameters(), clipping_norm)
    optimizer.step()


def soft_update_of_target_network(local_model, target_model, tau):              # Base_Agent.py:84-87
    for target_param, local_param in zip(target_model.parameters(), local_model.parameters()):
        target_param.data.copy_(tau * local_param.data + (1.0 - tau) * target_param.data)


def dump(prefix, module, out):
    for k, v in module.state_dict().items():
        out["%s/%s" % (prefix, k)] = v.detach().cpu().numpy().copy()


# ------------------------------------------------------------------ DDQN (DDQN.py:168-209)
def ddqn_fixture(out):
    torch.manual_seed(11)
    hp = {"learning_rate": 1e-3, "discount_rate": 1.0, "gradient_clipping_norm": 5.0, "tau": 0.005}   # Config "DDQN" but for lr
    q_local, q_target = DDQNActorNet(18, 16, 2, 20), DDQNActorNet(18, 16, 2, 20)
    opt = Adam(q_local.parameters(), lr=hp["learning_rate"], eps=1e-4)
    B = 32
    states, next_states = torch.randn(B, 18), torch.randn(B, 18)
    actions = torch.randint(0, 20, (B, 1)).float()
    rewards = -torch.rand(B, 1)
    dones = (torch.rand(B, 1) < 0.25).float()
    dump("ddqn/local0", q_local, out); dump("ddqn/target0", q_target, out)
    for k, v in (("states", states), ("next_states", next_states), ("actions", actions), ("rewards", rewards), ("dones", dones)):
        out["ddqn/" + k] = v.numpy().copy()
    out["ddqn/hyper"] = np.array([hp["learning_rate"], hp["discount_rate"], hp["gradient_clipping_norm"], hp["tau"]])
    # compute_loss :182-188
    with torch.no_grad():
        max_action_indexes = q_local(next_states).detach().argmax(1)                                  # :197
        Q_targets_next = q_target(next_states).gather(1, max_action_indexes.unsqueeze(1))              # :198
        Q_targets = rewards + (hp["discount_rate"] * Q_targets_next * (1 - dones))                     # :203
    Q_expected = q_local(states).gather(1, actions.long())                                             # :208
    loss = F.mse_loss(Q_expected, Q_targets)
    take_optimisation_step(opt, q_local, loss, hp["gradient_clipping_norm"])                           # :178
    soft_update_of_target_network(q_local, q_target, hp["tau"])                                        # :180
    out["ddqn/loss"] = np.array(float(loss.detach()))
    dump("ddqn/local1", q_local, out); dump("ddqn/target1", q_target, out)


# ------------------------------------------------------------------ SAC-discrete (SAC_Discrete.py:293-352)
def sac_fixture(out):
    torch.manual_seed(12)
    hp = {"learning_rate": 3e-3, "discount_rate": 0.99, "gradient_clipping_norm": 5.0, "tau": 0.005}
    mk = lambda sm: ReluStack(30, 16, 2, 3, sm)
    critic_local, critic_local_2, critic_target, critic_target_2, actor_local = mk(False), mk(False), mk(False), mk(False), mk(True)
    critic_optimizer = Adam(critic_local.parameters(), lr=hp["learning_rate"], eps=1e-4)
    critic_optimizer_2 = Adam(critic_local_2.parameters(), lr=hp["learning_rate"], eps=1e-4)
    actor_optimizer = Adam(actor_local.parameters(), lr=hp["learning_rate"], eps=1e-4)
    target_entropy = -np.log((1.0 / 3)) * 0.98                                                          # :166
    log_alpha = torch.tensor([0.3], requires_grad=True)
    alpha = log_alpha.exp()
    alpha_optim = Adam([log_alpha], lr=hp["learning_rate"], eps=1e-4)
    B = 24
    state_batch, next_state_batch = torch.randn(B, 30), torch.randn(B, 30)
    action_batch = torch.randint(0, 3, (B, 1)).float()
    reward_batch = -torch.rand(B, 1)
    done_batch = (torch.rand(B, 1) < 0.2).float()
    for name, net in (("critic1", critic_local), ("critic2", critic_local_2), ("target1", critic_target), ("target2", critic_target_2),
                      ("actor", actor_local)):
        dump("sac/%s0" % name, net, out)
    for k, v in (("states", state_batch), ("next_states", next_state_batch), ("actions", action_batch), ("rewards", reward_batch),
                 ("dones", done_batch)):
        out["sac/" + k] = v.numpy().copy()
    out["sac/hyper"] = np.array([hp["learning_rate"], hp["discount_rate"], hp["gradient_clipping_norm"], hp["tau"], 0.3])

    def produce_action_and_action_info(state):                                                          # :265-276
        action_probabilities = actor_local(state)
        z = action_probabilities == 0.0
        z = z.float() * 1e-8
        log_action_probabilities = torch.log(action_probabilities + z)
        return action_probabilities, log_action_probabilities

    # calculate_critic_losses :308-323
    with torch.no_grad():
        action_probabilities, log_action_probabilities = produce_action_and_action_info(next_state_batch)
        qf1_next_target = critic_target(next_state_batch)
        qf2_next_target = critic_target_2(next_state_batch)
        min_qf_next_target = action_probabilities * (torch.min(qf1_next_target, qf2_next_target) - alpha * log_action_probabilities)
        min_qf_next_target = min_qf_next_target.sum(dim=1).unsqueeze(-1)
        next_q_value = reward_batch + (1.0 - done_batch) * hp["discount_rate"] * min_qf_next_target
    qf1 = critic_local(state_batch).gather(1, action_batch.long())
    qf2 = critic_local_2(state_batch).gather(1, action_batch.long())
    qf1_loss = F.mse_loss(qf1, next_q_value)
    qf2_loss = F.mse_loss(qf2, next_q_value)
    # update_critic_parameters :340-345
    take_optimisation_step(critic_optimizer, critic_local, qf1_loss, hp["gradient_clipping_norm"])
    take_optimisation_step(critic_optimizer_2, critic_local_2, qf2_loss, hp["gradient_clipping_norm"])
    soft_update_of_target_network(critic_local, critic_target, hp["tau"])
    soft_update_of_target_network(critic_local_2, critic_target_2, hp["tau"])
    # calculate_actor_loss :325-333 (with the critics already updated, as learn() orders it :296-299)
    action_probabilities, log_action_probabilities = produce_action_and_action_info(state_batch)
    qf1_pi = critic_local(state_batch)
    qf2_pi = critic_local_2(state_batch)
    min_qf_pi = torch.min(qf1_pi, qf2_pi)
    inside_term = alpha.detach() * log_action_probabilities - min_qf_pi
    policy_loss = (action_probabilities * inside_term).sum(dim=1).mean()
    log_pi = torch.sum(log_action_probabilities * action_probabilities, dim=1)
    alpha_loss = -(log_alpha * (log_pi + target_entropy).detach()).mean()                               # :335-338
    # update_actor_parameters :347-352
    take_optimisation_step(actor_optimizer, actor_local, policy_loss, hp["gradient_clipping_norm"])
    take_optimisation_step(alpha_optim, None, alpha_loss, None)
    out["sac/losses"] = np.array([float(qf1_loss.detach()), float(qf2_loss.detach()), float(policy_loss.detach()), float(alpha_loss.detach())])
    out["sac/log_alpha1"] = log_alpha.detach().numpy().copy()
    for name, net in (("critic1", critic_local), ("critic2", critic_local_2), ("target1", critic_target), ("target2", critic_target_2),
                      ("actor", actor_local)):
        dump("sac/%s1" % name, net, out)


# ------------------------------------------------------------------ one A3C worker episode (A3C_v5.1.py:255-283,363-437)
def a3c_fixture(out):
    torch.manual_seed(13)
    rs = np.random.RandomState(13)
    lr, discount_rate, gradient_clipping_norm = 1e-3, 0.99, 5.0
    actor_task_model = ReluStack(30, 16, 2, 12, True)
    actor_machine_model = ReluStack(31, 16, 2, 10, True)
    critic_model = ReluStack(30, 16, 2, 1, False)
    # SharedAdam (Utility_Functions.py:55-112) is Adam with pre-created state and bias-corrected step size: the first
    # step from zero moments equals torch.optim.Adam's
    optimizers = [Adam(m.parameters(), lr=lr, eps=1e-4) for m in (actor_task_model, actor_machine_model, critic_model)]
    T = 14
    episode_states = [rs.randn(30) for _ in range(T)]
    episode_actions = [np.array([rs.randint(0, 12), rs.randint(0, 10)]) for _ in range(T)]
    episode_rewards = [-float(rs.randint(0, 40)) for _ in range(T)]
    for name, net in (("task", actor_task_model), ("machine", actor_machine_model), ("critic", critic_model)):
        dump("a3c/%s0" % name, net, out)
    out["a3c/states"] = np.stack(episode_states)
    out["a3c/actions"] = np.stack(episode_actions).astype(np.int64)
    out["a3c/rewards"] = np.array(episode_rewards)
    out["a3c/hyper"] = np.array([lr, discount_rate, gradient_clipping_norm])
    # the episode loop :270-283 with the recorded actions in place of the sampled ones
    episode_log_action_task_probabilities, episode_log_action_machine_probabilities, critic_outputs = [], [], []
    for state, actions in zip(episode_states, episode_actions):
        s = torch.from_numpy(state).float().unsqueeze(0)                                                # :319
        dist_task = Categorical(actor_task_model.forward(s))                                            # create_actor_distribution
        episode_log_action_task_probabilities.append(dist_task.log_prob(torch.Tensor([actions[0]])))    # :357-361
        state_add = np.append(state, actions[0])                                                        # :271
        s2 = torch.from_numpy(state_add).float().unsqueeze(0)
        dist_machine = Categorical(actor_machine_model.forward(s2))
        episode_log_action_machine_probabilities.append(dist_machine.log_prob(torch.Tensor([actions[1]])))
        critic_outputs.append(critic_model.forward(s).squeeze(0))                                       # :350-355 (one value)
    # calculate_discounted_returns :373-383
    discounted_returns = [0]
    for ix in range(len(episode_states)):
        return_value = episode_rewards[-(ix + 1)] + discount_rate * discounted_returns[-1]
        discounted_returns.append(return_value)
    discounted_returns = discounted_returns[1:]
    discounted_returns = discounted_returns[::-1]
    # normalise_discounted_returns :385-391
    discounted_returns = np.array(discounted_returns)
    mean = np.mean(discounted_returns)
    std = np.std(discounted_returns)
    discounted_returns -= mean
    discounted_returns /= (std + 1e-5)
    # calculate_critic_loss_and_advantages :403-410
    critic_values = torch.cat(critic_outputs)
    advantages = torch.Tensor(discounted_returns) - critic_values
    advantages = advantages.detach()
    critic_loss = (torch.Tensor(discounted_returns) - critic_values) ** 2
    critic_loss = critic_loss.mean()
    # calculate_actor_loss :412-417
    actor_task_loss = (-1.0 * torch.cat(episode_log_action_task_probabilities) * advantages).mean()
    actor_machine_loss = (-1.0 * torch.cat(episode_log_action_machine_probabilities) * advantages).mean()
    # put_gradients_in_queue :419-437 -> update_shared_model :164-187: clipped local gradients, one shared Adam step each
    for model, opt, loss in ((actor_task_model, optimizers[0], actor_task_loss), (actor_machine_model, optimizers[1], actor_machine_loss),
                             (critic_model, optimizers[2], critic_loss)):
        take_optimisation_step(opt, model, loss, gradient_clipping_norm)
    out["a3c/returns"] = discounted_returns.copy()
    out["a3c/losses"] = np.array([float(critic_loss.detach()), float(actor_task_loss.detach()), float(actor_machine_loss.detach())])
    for name, net in (("task", actor_task_model), ("machine", actor_machine_model), ("critic", critic_model)):
        dump("a3c/%s1" % name, net, out)


if __name__ == "__main__":
    torch.set_num_threads(1)
    out = {}
    ddqn_fixture(out)
    sac_fixture(out)
    a3c_fixture(out)
    path = os.path.join(HERE, "agent_updates.npz")
    np.savez_compressed(path, **out)
    print("%s: %d arrays, %.1f KB; torch %s" % (path, len(out), os.path.getsize(path) / 1024, torch.__version__))
    print("ddqn loss %.8f | sac losses %s | a3c losses %s" % (out["ddqn/loss"], out["sac/losses"], out["a3c/losses"]))

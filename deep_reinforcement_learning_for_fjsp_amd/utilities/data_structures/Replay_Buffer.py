"""Replay memories of the off-policy agents.

`Replay_Buffer`        the reference's utilities/data_structures/Replay_Buffer.py:7-59 (host deque of
                       namedtuples, `random.sample` without replacement, f32 device tensors out): the
                       single-environment agent loops use it unchanged.
`DeviceReplayBuffer`   the batched path: a ring of f32 rows resident in HBM; one `add_batch` appends the
                       transitions of all live environments of a vector step without leaving the device.
"""
import random
from collections import deque, namedtuple

import numpy as np
import torch


class Replay_Buffer(object):
    def __init__(self, buffer_size, batch_size, device=None):
        self.memory = deque(maxlen=buffer_size)
        self.batch_size = batch_size
        self.experience = namedtuple("Experience", field_names=["state", "action", "reward", "next_state", "done"])
        self.device = torch.device(device) if device else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")

    def add_experience(self, states, actions, rewards, next_states, dones):
        if type(dones) == list:                                                       # :20-24
            assert type(dones[0]) != list, "A done shouldn't be a list"
            self.memory.extend(self.experience(*row) for row in zip(states, actions, rewards, next_states, dones))
        else:
            self.memory.append(self.experience(states, actions, rewards, next_states, dones))

    def sample(self, num_experiences=None, separate_out_data_types=True):
        experiences = self.pick_experiences(num_experiences)
        return self.separate_out_data_types(experiences) if separate_out_data_types else experiences

    def separate_out_data_types(self, experiences):                                   # :39-47
        col = lambda f: torch.from_numpy(np.vstack([f(e) for e in experiences if e is not None])).float().to(self.device)
        return (col(lambda e: e.state), col(lambda e: e.action), col(lambda e: e.reward), col(lambda e: e.next_state),
                col(lambda e: int(e.done)))

    def pick_experiences(self, num_experiences=None):
        return random.sample(self.memory, k=self.batch_size if num_experiences is None else num_experiences)

    def __len__(self):
        return len(self.memory)


class DeviceReplayBuffer(object):
    """Ring buffer [capacity] of (state, action, reward, next_state, done) f32 rows on `device`."""

    def __init__(self, buffer_size, batch_size, state_size, device, seed=0):
        self.capacity, self.batch_size, self.device = int(buffer_size), int(batch_size), torch.device(device)
        f = dict(dtype=torch.float32, device=self.device)
        self.states = torch.zeros(self.capacity, state_size, **f)
        self.next_states = torch.zeros(self.capacity, state_size, **f)
        self.actions = torch.zeros(self.capacity, 1, **f)
        self.rewards = torch.zeros(self.capacity, 1, **f)
        self.dones = torch.zeros(self.capacity, 1, **f)
        self.size, self.head = 0, 0
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def add_batch(self, states, actions, rewards, next_states, dones, active=None):
        """Append the rows with active != 0 (all rows when None). Tensors are [N, ...] on the device."""
        if active is not None:
            idx = torch.nonzero(active.reshape(-1) != 0).reshape(-1)
            if idx.numel() == 0:
                return 0
            states, actions, rewards = states[idx], actions[idx], rewards[idx]
            next_states, dones = next_states[idx], dones[idx]
        n = states.shape[0]
        if n > self.capacity:                          # keep the newest rows, like deque(maxlen)
            states, actions, rewards = states[-self.capacity:], actions[-self.capacity:], rewards[-self.capacity:]
            next_states, dones = next_states[-self.capacity:], dones[-self.capacity:]
            n = self.capacity
        pos = (self.head + torch.arange(n, device=self.device)) % self.capacity
        self.states[pos] = states.float()
        self.next_states[pos] = next_states.float()
        self.actions[pos] = actions.float().reshape(n, 1)
        self.rewards[pos] = rewards.float().reshape(n, 1)
        self.dones[pos] = dones.float().reshape(n, 1)
        self.head = (self.head + n) % self.capacity
        self.size = min(self.capacity, self.size + n)
        return n

    def add_experience(self, states, actions, rewards, next_states, dones):
        """Single-transition form of the reference's method (host values)."""
        t = lambda v, w: torch.as_tensor(np.asarray(v, dtype=np.float32).reshape(1, w), device=self.device)
        self.add_batch(t(states, -1), t(actions, 1), t(rewards, 1), t(next_states, -1), t(float(dones), 1))

    def sample(self, num_experiences=None):
        k = self.batch_size if num_experiences is None else int(num_experiences)
        assert k <= self.size, "not enough experiences (random.sample would raise ValueError)"
        idx = torch.randperm(self.size, generator=self.gen, device=self.device)[:k]       # without replacement
        return self.states[idx], self.actions[idx], self.rewards[idx], self.next_states[idx], self.dones[idx]

    def __len__(self):
        return self.size

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
        # one scratch row past the ring takes the rows of finished environments (see add_batch)
        self._rows = [torch.zeros(self.capacity + 1, w, **f) for w in (state_size, state_size, 1, 1, 1)]
        self.states, self.next_states, self.actions, self.rewards, self.dones = (r[:self.capacity] for r in self._rows)
        self._head = torch.zeros((), dtype=torch.int64, device=self.device)
        self._size = torch.zeros((), dtype=torch.int64, device=self.device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def add_batch(self, states, actions, rewards, next_states, dones, active=None):
        """Append the rows with active != 0 (all rows when None).  Tensors are [N, ...] on the device.

        No host round trip: the ring positions of the live rows come from a prefix sum of the mask, the dead rows
        are routed to a scratch row past the ring, and head / size stay device scalars (`len()` reads them)."""
        n_rows = states.shape[0]
        if n_rows > self.capacity:
            raise ValueError("a vector step larger than the replay ring (%d > %d)" % (n_rows, self.capacity))
        if active is None:
            live = torch.ones(n_rows, dtype=torch.bool, device=self.device)
        else:
            live = active.reshape(-1) != 0
        rank = torch.cumsum(live.to(torch.int64), 0) - 1
        n_live = rank[-1] + 1
        pos = torch.where(live, (self._head + rank) % self.capacity, torch.full_like(rank, self.capacity))
        self._rows[0][pos] = states.float()
        self._rows[1][pos] = next_states.float()
        self._rows[2][pos] = actions.float().reshape(n_rows, 1)
        self._rows[3][pos] = rewards.float().reshape(n_rows, 1)
        self._rows[4][pos] = dones.float().reshape(n_rows, 1)
        self._head.copy_((self._head + n_live) % self.capacity)
        self._size.copy_(torch.clamp(self._size + n_live, max=self.capacity))

    def snapshot_cursor(self):
        """(head, size) as they are now; with restore_cursor() lets a caller undo appended rows (graph warm-ups)."""
        return self._head.clone(), self._size.clone()

    def restore_cursor(self, cursor):
        self._head.copy_(cursor[0])
        self._size.copy_(cursor[1])

    def add_experience(self, states, actions, rewards, next_states, dones):
        """Single-transition form of the reference's method (host values)."""
        t = lambda v, w: torch.as_tensor(np.asarray(v, dtype=np.float32).reshape(1, w), device=self.device)
        self.add_batch(t(states, -1), t(actions, 1), t(rewards, 1), t(next_states, -1), t(float(dones), 1))

    def sample(self, num_experiences=None):
        k = self.batch_size if num_experiences is None else int(num_experiences)
        size = len(self)
        assert k <= size, "not enough experiences (random.sample would raise ValueError)"
        idx = torch.randperm(size, generator=self.gen, device=self.device)[:k]            # without replacement
        return self.states[idx], self.actions[idx], self.rewards[idx], self.next_states[idx], self.dones[idx]

    @property
    def size(self):
        return int(self._size.item())

    def __len__(self):
        return self.size

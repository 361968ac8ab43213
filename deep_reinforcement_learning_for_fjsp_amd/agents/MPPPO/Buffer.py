"""On-policy buffers.

`Replay_Buffer`   the reference's agents/MPPPO/Buffer.py:7-58, same interface
                  (add_experience / sample / clear / __len__), for agent loops
                  that drive ONE environment exactly like the reference does.
`RolloutBuffer`   the batched counterpart behind the C ABI (fjsp_rollout_*): a
                  [T][N][...] f32 slab in HBM that the batched step output is
                  appended to on-device and that PyTorch wraps without copy.
"""
import ctypes as C
from collections import deque, namedtuple

import numpy as np
import torch

from ... import _capi
from ..._capi import check


class Replay_Buffer(object):
    def __init__(self, device=None):
        self.memory = deque()
        self.experience = namedtuple("Experience", field_names=["state", "action", "reward", "next_state", "done"])
        self.device = torch.device(device) if device else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")

    def add_experience(self, states, actions, rewards, next_states, dones):          # Buffer.py:19-28
        if type(dones) == list:
            assert type(dones[0]) != list, "A done shouldn't be a list"
            self.memory.extend(self.experience(s, a, r, n, d)
                               for s, a, r, n, d in zip(states, actions, rewards, next_states, dones))
        else:
            self.memory.append(self.experience(states, actions, rewards, next_states, dones))

    def sample(self, separate_out_data_types=True):                                  # :30-37 (returns everything, in order)
        experiences = self.pick_experiences()
        return self.separate_out_data_types(experiences) if separate_out_data_types else experiences

    def separate_out_data_types(self, experiences):                                  # :39-47
        ex = [e for e in experiences if e is not None]
        f = lambda rows: torch.from_numpy(np.vstack(rows)).float().to(self.device)
        return (f([e.state for e in ex]), f([e.action for e in ex]).squeeze(1), f([e.reward for e in ex]).squeeze(1),
                f([e.next_state for e in ex]), f([int(e.done) for e in ex]).squeeze(1))

    def pick_experiences(self):
        return self.memory

    def clear(self):
        self.memory.clear()

    def __len__(self):
        return len(self.memory)


class _DevArray(object):
    """Minimal __cuda_array_interface__ carrier so torch.as_tensor wraps library memory zero-copy."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


class RolloutBuffer(object):
    """fjsp_rollout handle; tensors are views of the library's HBM slab."""

    FIELDS = ("states", "actions", "rewards", "next_states", "dones", "valid", "returns")

    def __init__(self, T, N, state_size, device=0):
        self._lib = _capi.lib()
        self._h = C.c_void_p()
        check(self._lib.fjsp_rollout_create(int(T), int(N), int(state_size), int(device), C.byref(self._h)))
        self.T, self.N, self.S, self.device = T, N, state_size, torch.device("cuda", device)
        shapes = {"states": (T, N, state_size), "next_states": (T, N, state_size), "actions": (T, N, 2)}
        for which, name in enumerate(self.FIELDS):
            ptr = self._lib.fjsp_rollout_ptr(self._h, which)
            arr = _DevArray(ptr, shapes.get(name, (T, N)), "<f4", self)
            setattr(self, name, torch.as_tensor(arr, device=self.device))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.fjsp_rollout_destroy(h)
            self._h = None

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def add_experience(self, states, actions, rewards, next_states, dones, active=None):
        """One batched transition (f64 env outputs, u8 actions/dones) appended at the write cursor."""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(self._lib.fjsp_rollout_append(self._h, p(states), p(actions), p(rewards), p(next_states), p(dones),
                                            p(active), self._stream()))

    def compute_returns(self, gamma):
        check(self._lib.fjsp_rollout_returns(self._h, float(gamma), self._stream()))
        return self.returns[:len(self)]

    def normalised_returns(self, gamma, normalized=True, standardized=True):
        """compute_returns() + the per-episode normalisation of MPPPO.py:258-261 in one launch."""
        n = len(self)
        if getattr(self, "_norm", None) is None or self._norm.shape != self.returns.shape:
            self._norm = torch.zeros_like(self.returns)
        check(self._lib.fjsp_rollout_returns_normalised(self._h, float(gamma), 1 if normalized else 0, 1 if standardized else 0,
                                                        C.c_void_p(self._norm.data_ptr()), self._stream()))
        return self._norm[:n]

    def sample(self):
        n = len(self)
        return self.states[:n], self.actions[:n], self.rewards[:n], self.next_states[:n], self.dones[:n]

    def clear(self):
        check(self._lib.fjsp_rollout_clear(self._h))

    def __len__(self):
        return self._lib.fjsp_rollout_len(self._h)

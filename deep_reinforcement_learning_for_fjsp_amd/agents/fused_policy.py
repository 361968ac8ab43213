"""Action sampling of the HMPSAC policy networks in one launch (csrc/fjsp_policy_mlp.hip, fjsp_policy_pair_sample).

`PolicyPairSampler(task_layers, machine_layers)` wraps the Linear-ReLU-...-Linear stacks of a TaskPolicyNet /
MachinePolicyNet pair (agents/HMPSAC/A3C.py; A3C_v5.1.py:35-75) or of a single policy network (the SAC controller's
actor, SAC_Discrete.py:85-103):  sample(state) -> (a_task, a_machine) int64 tensors, drawn from
softmax(task(state.float())) and softmax(machine(cat(state.float(), a_task))) exactly as
SAC_Discrete.py:277-284 does with two Categorical objects -- same distributions, but the random numbers come from a
counter-based splitmix64 stream per row (seed, row, draws made so far) instead of torch's generator, like the
environment kernels' own streams.  The kernel reads transposed copies of the weights, refreshed at every call unless the
sampler was built with static_weights=True (frozen networks), so a call captured into a HIP graph follows later updates.

GPU only; `supported()` says whether a stack fits the kernel (<= 6 linear layers, widths <= 256, <= 64 outputs, f32).
"""
import ctypes as C

import numpy as np
import torch
from torch import nn

from .. import _capi


def _linears(layers):
    """The Linear modules of a Linear-ReLU-...-Linear stack, or None when the stack has another structure."""
    mods = list(layers)
    lin = []
    for i, m in enumerate(mods):
        if i % 2 == 0:
            if not isinstance(m, nn.Linear) or m.bias is None:
                return None
            lin.append(m)
        elif not isinstance(m, nn.ReLU):
            return None
    if not lin or len(mods) != 2 * len(lin) - 1:
        return None
    return lin


def supported(layers, device=None):
    lin = _linears(layers)
    if lin is None or len(lin) > 6:
        return False
    if any(m.weight.dtype != torch.float32 or not m.weight.is_cuda or not m.weight.is_contiguous() for m in lin):
        return False
    if device is not None and any(m.weight.device != torch.device(device) for m in lin):
        return False
    dims = [lin[0].in_features] + [m.out_features for m in lin]
    return max(dims) <= 256 and dims[-1] <= 64


class _Net:
    """Parameter pointers of one stack as the kernel wants them: the weights TRANSPOSED ([in][out]: the threads of a wave
    then read consecutive words), kept in buffers of this object.  static_weights=True: the transposes are taken once
    (networks nobody updates any more); False: at every refresh(), i.e. at every sampling call -- one small copy kernel
    per layer, also when the call is captured into a HIP graph, so replays follow in-place optimiser steps."""

    def __init__(self, layers, static_weights):
        self.lin = _linears(layers)
        self.n = len(self.lin)
        self.dims = np.array([self.lin[0].in_features] + [m.out_features for m in self.lin], dtype=np.int32)
        self.static = bool(static_weights)
        self.wt = [torch.empty(m.in_features, m.out_features, dtype=torch.float32, device=m.weight.device) for m in self.lin]
        self.w = (C.c_void_p * self.n)()
        self.b = (C.c_void_p * self.n)()
        self._filled = False

    @torch.no_grad()
    def refresh(self):
        if not (self.static and self._filled):
            for t, m in zip(self.wt, self.lin):
                t.copy_(m.weight.t())
            self._filled = True
        for i, m in enumerate(self.lin):
            self.w[i] = self.wt[i].data_ptr()
            self.b[i] = m.bias.data_ptr()


class PolicyPairSampler:
    def __init__(self, task_layers, machine_layers=None, seed=0, static_weights=False):
        if not supported(task_layers) or (machine_layers is not None and not supported(machine_layers)):
            raise ValueError("PolicyPairSampler: unsupported network (Linear-ReLU-...-Linear, f32 on a GPU, widths <= 256, outputs <= 64)")
        self._lib = _capi.lib()
        self.task = _Net(task_layers, static_weights)
        self.machine = _Net(machine_layers, static_weights) if machine_layers is not None else None
        self.device = self.task.lin[0].weight.device
        self.S = int(self.task.dims[0])
        if self.machine is not None and int(self.machine.dims[0]) != self.S + 1:
            raise ValueError("PolicyPairSampler: the machine network takes the state and the task action")
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._draws = {}            # rows -> u32[rows] draw counters

    def draws(self, rows):
        d = self._draws.get(rows)
        if d is None:
            d = self._draws[rows] = torch.zeros(rows, dtype=torch.int32, device=self.device)
        return d

    @torch.no_grad()
    def sample(self, state, out_task=None, out_machine=None, probs=False, pair_out=None, select=None, which=0):
        """state: f64[rows, S] contiguous on the networks' device.  Returns (a_task, a_machine) -- a_machine is None
        without a machine network -- and, with probs=True, also the two probability tensors.  pair_out (u8[rows, 2],
        contiguous): also receives (a_task, a_machine) in the environment's action encoding, for the rows with
        select[row] == which (select: int64[rows]; None: every row)."""
        if state.dtype != torch.float64 or not state.is_contiguous() or state.device != self.device or state.dim() != 2 or state.shape[1] != self.S:
            raise ValueError("PolicyPairSampler.sample: state must be a contiguous f64[rows, %d] tensor on %s" % (self.S, self.device))
        rows = int(state.shape[0])
        if pair_out is not None and (pair_out.dtype != torch.uint8 or tuple(pair_out.shape) != (rows, 2) or not pair_out.is_contiguous() or pair_out.device != self.device):
            raise ValueError("PolicyPairSampler.sample: pair_out must be a contiguous u8[rows, 2] tensor on the networks' device")
        if select is not None and (select.dtype != torch.int64 or select.numel() != rows or not select.is_contiguous() or select.device != self.device):
            raise ValueError("PolicyPairSampler.sample: select must be a contiguous int64[rows] tensor on the networks' device")
        a_t = out_task if out_task is not None else torch.empty(rows, dtype=torch.int64, device=self.device)
        a_m = None
        if self.machine is not None:
            a_m = out_machine if out_machine is not None else torch.empty(rows, dtype=torch.int64, device=self.device)
        p_t = torch.empty(rows, int(self.task.dims[-1]), dtype=torch.float32, device=self.device) if probs else None
        p_m = torch.empty(rows, int(self.machine.dims[-1]), dtype=torch.float32, device=self.device) if probs and self.machine is not None else None
        self.task.refresh()
        if self.machine is not None:
            self.machine.refresh()
        m = self.machine
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(self.device):
            _capi.check(self._lib.fjsp_policy_pair_sample(
                self.task.n, self.task.dims.ctypes.data, C.addressof(self.task.w), C.addressof(self.task.b),
                m.n if m else 0, m.dims.ctypes.data if m else None, C.addressof(m.w) if m else None, C.addressof(m.b) if m else None,
                state.data_ptr(), rows, self.S, self.seed, self.draws(rows).data_ptr(), a_t.data_ptr(), ptr(a_m), ptr(p_t), ptr(p_m),
                ptr(pair_out), ptr(select), int(which), torch.cuda.current_stream(self.device).cuda_stream))
        if probs:
            return a_t, a_m, p_t, p_m
        return a_t, a_m


def expected_draw(probs, seed, row, draw):
    """Host restatement of the kernel's inverse-CDF draw for one row (tests): probs f32[outputs] as the kernel returned them."""
    mask = (1 << 64) - 1
    z = (int(seed) + int(row) * 0x9E3779B97F4A7C15 + int(draw) * 1000003) & mask
    z = (z + 0x9E3779B97F4A7C15) & mask
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
    z ^= z >> 31
    u = np.float32(z >> 40) * np.float32(1.0 / 16777216.0)
    c = np.float32(0.0)
    for a, p in enumerate(np.asarray(probs, dtype=np.float32)):
        c = np.float32(c + p)
        if u < c:
            return a
    return len(probs) - 1

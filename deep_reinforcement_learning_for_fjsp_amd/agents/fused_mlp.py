"""Linear-ReLU-Linear-ReLU-Linear networks trained without autograd's swarm of small launches.

A clipped-PPO learning round (agents/MPPPO/MPPPO.py:314-370) runs 20 forward/backward passes of two such networks
over ~160 k samples.  Eager autograd spends a third of the round's device time in element-wise and reduction
launches between the GEMMs (rocprofv3: profiles/r02_ppo_round_eager_kernel_stats.csv).  `FusedMLP` keeps the dense
layers on the library GEMMs (MFMA, bias / ReLU in the GEMM epilogue) and does the rest in the library's fused
kernels (csrc/fjsp_ppo.hip): loss + its gradient in one pass, ReLU backward + bias gradient in one pass, gradient
clipping + Adam in one pass over ONE flat parameter buffer (which is also the single all-reduce bucket of SURVEY 8e).

`train_pass()` goes one step further for the reference's shapes (128 hidden units, <= 31 state features, <= 32
outputs): forward + loss + backward in ONE launch on the f32 matrix cores with the activations resident in LDS
(csrc/fjsp_mlp_train.hip; 2.1x the library-GEMM pass).  forward() / *_loss() / backward() remain for other shapes and
for the forward-only advantage pass.

The parameters stay the `nn.Parameter`s of the wrapped module -- re-homed as views of the flat buffer -- so
everything else (acting, checkpoints, equalise_policies, the in-kernel actor of the fused rollout) sees them as before.
"""
import ctypes as C

import torch
from torch import nn

from .. import _capi

SPLIT_K = 128           # chunks of the batch in the split-K weight gradient (see agents/linear.py)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def supported(module_layers, device):
    lin = [l for l in module_layers if isinstance(l, nn.Linear)]
    act = [l for l in module_layers if not isinstance(l, nn.Linear)]
    return (torch.device(device).type == "cuda" and len(lin) == 3 and len(act) == 2 and all(isinstance(a, nn.ReLU) for a in act)
            and lin[0].out_features == lin[1].in_features == lin[1].out_features == lin[2].in_features and lin[0].out_features <= 1024)


class FusedMLP(object):
    def __init__(self, module_layers, lr, eps=1e-4, betas=(0.9, 0.999), max_norm=1.0):
        self._lib = _capi.lib()
        self.lin = [l for l in module_layers if isinstance(l, nn.Linear)]
        dev = self.lin[0].weight.device
        self.device = dev
        params = []
        for l in self.lin:
            params += [l.weight, l.bias]
        self.numel = sum(p.numel() for p in params)
        self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(64, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)                 # the module's parameter now lives in the flat buffer
                self.views.append(self.grad[off:off + n].view_as(p))
                off += n
        self.lr, self.eps, self.betas, self.max_norm = float(lr), float(eps), betas, float(max_norm)
        self.H = self.lin[0].out_features
        self.out = self.lin[2].out_features
        self._buf = {}

    # -- buffers (allocated once per batch size: the learning round is replayed from a HIP graph) ----------------
    def _buffers(self, n):
        b = self._buf.get(n)
        if b is None:
            f = dict(dtype=torch.float32, device=self.device)
            nparts = min(512, max(1, n // 256))      # row bands of the two-stage column sums (bias gradients)
            b = dict(h1=torch.empty(n, self.H, **f), h2=torch.empty(n, self.H, **f), out=torch.empty(n, self.out, **f),
                     dout=torch.empty(n, self.out, **f), da=torch.empty(n, self.H, **f), db=torch.empty(n, self.H, **f),
                     partial=torch.empty(nparts, max(self.H, self.out), **f), nparts=nparts,
                     loss_partial=torch.empty(self._lib.fjsp_ppo_partials(n), **f), loss=torch.zeros(1, **f))
            self._buf = {k: v for k, v in self._buf.items() if isinstance(k, tuple)}
            self._buf[n] = b
        return b

    def _stream(self):
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))

    # -- forward ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x):
        """x f32[n, S] -> raw outputs f32[n, out] (logits / value); keeps the hidden activations for backward()."""
        b = self._buffers(x.shape[0])
        l1, l2, l3 = self.lin
        torch._addmm_activation(l1.bias, x, l1.weight.t(), out=b["h1"])            # bias + ReLU in the GEMM epilogue
        torch._addmm_activation(l2.bias, b["h1"], l2.weight.t(), out=b["h2"])
        torch.addmm(l3.bias, b["h2"], l3.weight.t(), out=b["out"])
        b["x"] = x
        return b["out"]

    # -- losses: fill b["dout"] and b["loss"] ---------------------------------------------------------------------
    @torch.no_grad()
    def actor_loss(self, actions_f32, old_log_prob, advantages, clip_epsilon, count):
        b = self._buffers(actions_f32.shape[0])
        _capi.check(self._lib.fjsp_ppo_actor_loss(_p(b["out"]), _p(actions_f32), _p(old_log_prob), _p(advantages), b["out"].shape[0], self.out,
                                                  float(clip_epsilon), _p(count), _p(b["dout"]), _p(b["loss_partial"]), _p(b["loss"]),
                                                  self._stream()))
        return b["loss"]

    @torch.no_grad()
    def critic_loss(self, returns, count):
        b = self._buffers(returns.shape[0])
        _capi.check(self._lib.fjsp_ppo_critic_loss(_p(b["out"]), _p(returns), b["out"].shape[0], _p(count), _p(b["dout"]), _p(b["loss_partial"]),
                                                   _p(b["loss"]), self._stream()))
        return b["loss"]

    # -- backward: gradients of every parameter into the flat gradient buffer -------------------------------------
    def _weight_grad(self, dz, a, out_view):
        """out_view[o, i] = sum_s dz[s, o] a[s, i] as a split-K product (the batch is the reduction dimension)."""
        S = dz.shape[0]
        per = S // SPLIT_K
        head = per * SPLIT_K
        if per == 0:
            torch.mm(dz.t(), a, out=out_view)
            return
        part = torch.bmm(dz[:head].reshape(SPLIT_K, per, -1).transpose(1, 2), a[:head].reshape(SPLIT_K, per, -1))
        torch.sum(part, 0, out=out_view)
        if head < S:
            out_view.addmm_(dz[head:].t(), a[head:])

    def _bias_grad(self, dz, h, out_view, b):
        _capi.check(self._lib.fjsp_relu_bwd_bias(_p(dz), _p(h), dz.shape[0], dz.shape[1], _p(b["partial"]), b["nparts"], _p(out_view),
                                                 self._stream()))

    @torch.no_grad()
    def backward(self):
        b = self._buf[next(k for k in self._buf if not isinstance(k, tuple))]
        l1, l2, l3 = self.lin
        gw1, gb1, gw2, gb2, gw3, gb3 = self.views
        dout, x, h1, h2 = b["dout"], b["x"], b["h1"], b["h2"]
        self._bias_grad(dout, None, gb3, b)
        self._weight_grad(dout, h2, gw3)
        torch.mm(dout, l3.weight, out=b["da"])                                     # d h2
        self._bias_grad(b["da"], h2, gb2, b)                                       # ReLU backward in place + bias gradient
        self._weight_grad(b["da"], h1, gw2)
        torch.mm(b["da"], l2.weight, out=b["db"])                                  # d h1
        self._bias_grad(b["db"], h1, gb1, b)
        self._weight_grad(b["db"], x, gw1)

    # -- forward + loss + backward in ONE launch (csrc/fjsp_mlp_train.hip) --------------------------------------------
    def mfma_pass_supported(self):
        """The one-launch training pass covers the reference's shapes: 128 hidden units, <= 31 state features, <= 32 outputs."""
        return self.H == 128 and self.lin[0].in_features <= 31 and self.out <= 32

    def train_pass_buffers(self, n):
        key = ("pass", n)
        b = self._buf.get(key)
        if b is None:
            with torch.cuda.device(self.device):                 # (the group count is the CU count of the parameters' device)
                groups = self._lib.fjsp_mlp_train_groups(n)
            f = dict(dtype=torch.float32, device=self.device)
            b = dict(groups=groups, partial=torch.empty(groups, self.numel, **f), loss_partial=torch.empty(groups, **f), loss=torch.zeros(1, **f),
                     sumsq=torch.zeros((self.numel + 63) // 64, **f))
            # keep the two most recent sample counts only (a partial buffer is groups x numel floats, ~20 MB per network)
            for old_key in [k for k in self._buf if isinstance(k, tuple) and k[0] == "pass"][:-1]:
                del self._buf[old_key]
            self._buf[key] = b
        return b

    @torch.no_grad()
    def train_pass(self, mode, x, aux0, aux1, aux2, count, clip_epsilon=0.0):
        """mode 0: actor (aux = actions f32, old log-probabilities, advantages); mode 1: critic (aux0 = returns).
        Fills the flat gradient buffer and returns the loss (f32[1])."""
        n = x.shape[0]
        b = self.train_pass_buffers(n)
        _capi.check(self._lib.fjsp_mlp_train_pass(int(mode), _p(self.flat), _p(x), n, self.lin[0].in_features, self.H, self.out, _p(aux0), _p(aux1),
                                                  _p(aux2), _p(count), float(clip_epsilon), _p(b["partial"]), b["groups"], _p(b["loss_partial"]),
                                                  _p(self.grad), _p(b["loss"]), self._stream()))
        return b["loss"]

    @torch.no_grad()
    def train_step(self, mode, x, aux0, aux1, aux2, count, clip_epsilon=0.0, values_out=None):
        """train_pass() + step() of a single process in three launches (fjsp_mlp_train_step): the gradient finish also
        produces the squared norm for the clip and advances the step count.  values_out (critic only): f32[n] that
        receives V(x) as this pass's forward computed it, i.e. under the parameters before the update."""
        n = x.shape[0]
        self.train_pass_buffers(n)
        b = self._buf[("pass", n)]
        if values_out is not None:
            if int(mode) != 1 or values_out.dtype != torch.float32 or values_out.numel() != n or not values_out.is_contiguous() or values_out.device != x.device:
                raise ValueError("train_step: values_out must be a contiguous f32[n] tensor on the samples' device (critic pass only)")
            _capi.check(self._lib.fjsp_mlp_train_step_values(
                1, _p(self.flat), _p(x), n, self.lin[0].in_features, self.H, self.out, _p(aux0), _p(aux1), _p(aux2), _p(count), float(clip_epsilon),
                _p(b["partial"]), b["groups"], _p(b["loss_partial"]), _p(self.grad), _p(b["loss"]), _p(self.exp_avg), _p(self.exp_avg_sq),
                self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps, _p(self.step_count), _p(b["sumsq"]), _p(values_out), self._stream()))
            return b["loss"]
        _capi.check(self._lib.fjsp_mlp_train_step(int(mode), _p(self.flat), _p(x), n, self.lin[0].in_features, self.H, self.out, _p(aux0),
                                                  _p(aux1), _p(aux2), _p(count), float(clip_epsilon), _p(b["partial"]), b["groups"],
                                                  _p(b["loss_partial"]), _p(self.grad), _p(b["loss"]), _p(self.exp_avg), _p(self.exp_avg_sq),
                                                  self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps, _p(self.step_count),
                                                  _p(b["sumsq"]), self._stream()))
        return b["loss"]

    # -- optimiser step --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, all_reduce=None):
        if all_reduce is not None:
            all_reduce(self.grad)                                                  # ONE collective per optimiser step
        _capi.check(self._lib.fjsp_adam_clip_step(_p(self.flat), _p(self.grad), _p(self.exp_avg), _p(self.exp_avg_sq), self.numel,
                                                  self.max_norm, self.lr, self.betas[0], self.betas[1], self.eps, _p(self.step_count),
                                                  _p(self.scratch), self._stream()))

"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm, over xGMI inside a node).

The environment path shards with no collective at all: env ids [g*N/G, (g+1)*N/G)
live on GPU g and every instance is a pure function of its global env id
(SURVEY.md 8e).  The only exchange step of the training path is ONE all-reduce
of one flat f32 gradient bucket per optimiser step (actor 2x128: 23 070 floats,
critic: 19 329 floats -> latency-bound messages far below the per-link xGMI
rate, so everything is bucketed into a single call per network and clipping
happens after the reduction).  The reference's only parallel mechanism is the
asynchronous A3C gradient queue (agents/HMPSAC/A3C_v5.1.py:125-187); this
replaces it for the on-policy path.
"""
import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_distributed() else 1


def rank():
    return dist.get_rank() if is_distributed() else 0


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        # FJSP_DIST_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than ranks
        backend = os.environ.get("FJSP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)


def shard_range(n_total, r=None, w=None):
    """Global env ids of rank r: a contiguous range, sizes differ by at most one."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n_total, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def all_reduce_scalar_sum(x):
    """Sum of a 0-dim tensor over ranks (the global sample count of a learning round)."""
    if not is_distributed():
        return x
    y = x.detach().clone().reshape(1)
    dist.all_reduce(y, op=dist.ReduceOp.SUM)
    return y[0]


def all_reduce_scalar_min(x):
    """Minimum of a 0-dim tensor over ranks (every rank must take the same trainer path: PPOLearner.learn)."""
    if not is_distributed():
        return x
    y = x.detach().clone().reshape(1)
    dist.all_reduce(y, op=dist.ReduceOp.MIN)
    return y[0]


class FlatGradBucket(object):
    """One contiguous f32 buffer that IS the gradient storage of a network: `zero_()` (in place of
    optimizer.zero_grad()) clears it and makes every parameter's `.grad` a view of its slice, so backward accumulates
    straight into the bucket and `all_reduce()` is ONE collective with no packing or unpacking.  A
    caller that still lets the optimiser drop the gradients (zero_grad(set_to_none=True)) gets the copying fallback."""

    def __init__(self, parameters):
        self.params = [p for p in parameters if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.views = None

    def _ensure(self):
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
            self.views, off = [], 0
            for p in self.params:
                n = p.numel()
                self.views.append(self.flat[off:off + n].view_as(p))
                off += n

    def zero_(self):
        self._ensure()
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def _in_place(self):
        return self.views is not None and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def all_reduce(self):
        if not is_distributed():
            return
        self._ensure()
        if self._in_place():
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            return
        for p, v in zip(self.params, self.views):          # fallback: gradients allocated by autograd, copied in and out
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(v)

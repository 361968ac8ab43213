#!/usr/bin/env python3
"""DIAGNOSTIC: device time of one training pass (forward + loss + backward) of a 2 x 128 network over n samples:
the one-launch MFMA kernel (csrc/fjsp_mlp_train.hip) against the library-GEMM trainer (agents/fused_mlp.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO import MPPPO as M
from deep_reinforcement_learning_for_fjsp_amd.agents import fused_mlp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 163840
S, A = 20, 24
dev = torch.device("cuda", 0)
torch.manual_seed(0)
actor, critic = M.ActorNet(S, 128, 2, A).to(dev), M.CriticNet(S, 128, 2, 1).to(dev)
x = torch.randn(n, S, device=dev)
actions = torch.randint(0, A, (n,), device=dev).float()
old_lp = -torch.rand(n, device=dev) * 3 - 0.2
adv = torch.randn(n, device=dev)
ret = torch.randn(n, device=dev)
count = torch.full((1,), float(n), device=dev)
ta, tc = fused_mlp.FusedMLP(actor.layers, lr=1e-3), fused_mlp.FusedMLP(critic.layers, lr=1e-3)


def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def lib_actor():
    ta.forward(x); ta.actor_loss(actions, old_lp, adv, 0.2, count); ta.backward()
def lib_critic():
    tc.forward(x); tc.critic_loss(ret, count); tc.backward()

flop = lambda out: 2.0 * n * (3 * 128 * 128 + 2 * S * 128 + 3 * out * 128)        # useful flops of one pass
for name, f, out in (("actor  one launch", lambda: ta.train_pass(0, x, actions, old_lp, adv, count, 0.2), A),
                     ("critic one launch", lambda: tc.train_pass(1, x, ret, None, None, count), 1),
                     ("actor  library   ", lib_actor, A), ("critic library   ", lib_critic, 1)):
    us = timed(f)
    print("%s n=%d: %8.1f us per pass, %6.1f TFLOP/s useful" % (name, n, us, flop(out) / us / 1e6))

#!/usr/bin/env python3
"""DIAGNOSTIC: cycles per phase of one tile of the single-crew training-pass kernel (library built with
-DFJSP_MLP_STAMPS into .diag/libfjsp_mlpstamps.so; run with FJSP_AMD_LIB pointing at it and FJSP_MLP_SINGLE_CREW=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO import MPPPO as M
from deep_reinforcement_learning_for_fjsp_amd.agents import fused_mlp
n, S, A = 163840, 20, 24
dev = torch.device("cuda", 0)
actor, critic = M.ActorNet(S, 128, 2, A).to(dev), M.CriticNet(S, 128, 2, 1).to(dev)
x = torch.randn(n, S, device=dev); actions = torch.randint(0, A, (n,), device=dev).float()
old_lp = -torch.rand(n, device=dev) * 3 - 0.2; adv = torch.randn(n, device=dev); ret = torch.randn(n, device=dev)
count = torch.full((1,), float(n), device=dev)
names = ["X write + barrier", "G1 + barrier", "G2", "G3 + barrier", "loss + barrier", "G4 G5 + barrier", "G6", "G7 + barrier", "dH1 write", "G8 + barrier"]
for label, net, mode, aux in (("actor", actor, 0, (actions, old_lp, adv)), ("critic", critic, 1, (ret, None, None))):
    tr = fused_mlp.FusedMLP(net.layers, lr=1e-3)
    for _ in range(3):
        tr.train_pass(mode, x, aux[0], aux[1], aux[2], count, 0.2)
    torch.cuda.synchronize()
    st = tr._buf[("pass", n)]["loss_partial"][16:26].cpu().tolist()
    print(label, "cycles per phase (workgroup 0, wave 0, 3rd tile); total %.0f" % sum(st))
    for nm, v in zip(names, st):
        print("   %-20s %8.0f" % (nm, v))
    lp = tr._buf[("pass", n)]["loss_partial"].cpu().tolist()
    print("   whole kernel %.0f shader cycles = %.0f ticks of 100 MHz -> %.1f us at %.0f MHz; prologue %.0f, last tile + epilogue %.0f" % (
        lp[32], lp[33], lp[33] / 100.0, lp[32] / (lp[33] / 100.0), lp[34], lp[35]))
    print("   tiles:", " ".join("%.0f" % v for v in lp[36:36 + 19]))

#!/usr/bin/env python3
"""DIAGNOSTIC helper: N envs, 100 warm-up + 500 timed autoreset step launches with whatever library
FJSP_AMD_LIB points at (used under rocprofv3 with the ablation builds of tools/ablate_step.py)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from deep_reinforcement_learning_for_fjsp_amd import instances as fi
from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
rs = np.random.RandomState(1)
acts = torch.from_numpy(np.stack([rs.randint(0, 6, (64, N)), rs.randint(0, 5, (64, N))], 2).astype(np.uint8)).cuda()
env = EnvBatch(s, N, rng_seed=3)
env.reset()
for i in range(100):
    env.step(acts[i % 64], autoreset=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(500):
    env.step(acts[i % 64], autoreset=True)
e1.record()
torch.cuda.synchronize()
print("%.2f us/launch" % (e0.elapsed_time(e1) * 2.0))

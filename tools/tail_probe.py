#!/usr/bin/env python3
"""DIAGNOSTIC: how much of a step launch is the tail of rare expensive wave paths?  Times windows of launches
(a) in steady state with autoreset (about 1/40 of the envs restart in every launch), (b) right after a
synchronised reset, where no env finishes for 25 steps, (c) with one fixed cheap rule pair for every env."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_reinforcement_learning_for_fjsp_amd import instances as fi
from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
N = 4096
s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
rs = np.random.RandomState(1)
acts = torch.from_numpy(np.stack([rs.randint(0, 6, (64, N)), rs.randint(0, 5, (64, N))], 2).astype(np.uint8)).cuda()
env = EnvBatch(s, N, rng_seed=3); env.reset()
def window(n, a=None, autoreset=True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        env.step(acts[i % 64] if a is None else a, autoreset=autoreset)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for i in range(300): env.step(acts[i % 64], autoreset=True)
torch.cuda.synchronize()
print("steady state, autoreset:            %.2f us/launch" % window(1000))
res = []
for rep in range(20):
    env.reset(); torch.cuda.synchronize()
    res.append(window(25))
print("25 launches after a reset (no env finishes): %.2f us/launch" % np.mean(res))
for pair in [(5, 4), (2, 0), (0, 0), (4, 2), (3, 3), (1, 3)]:
    a = torch.tensor(pair, dtype=torch.uint8).repeat(N, 1).cuda()
    res = []
    for rep in range(20):
        env.reset(); torch.cuda.synchronize()
        res.append(window(25, a))
    print("fixed rule pair %s, no resets:      %.2f us/launch" % (pair, np.mean(res)))

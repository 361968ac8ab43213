#!/usr/bin/env python3
"""DIAGNOSTIC: per-step time of the step kernel launched from Python vs replayed from a captured HIP graph of 64 steps."""
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
for i in range(128): env.step(acts[i % 64], autoreset=True)
torch.cuda.synchronize()
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
def eager(reps):
    for i in range(reps * 64): env.step(acts[i % 64], autoreset=True)
print("python launches: %.2f us/step" % (timed(eager, 16) / (16 * 64)))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(64): env.step(acts[i], autoreset=True)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="relaxed"):
    for i in range(64): env.step(acts[i], autoreset=True)
def replay(reps):
    for _ in range(reps): g.replay()
replay(2); torch.cuda.synchronize()
print("graph of 64 steps: %.2f us/step" % (timed(replay, 16) / (16 * 64)))
assert int((env.read()["status"] != 0).sum()) == 0

#!/usr/bin/env python3
"""DIAGNOSTIC: host cost of one step call (Python wrapper, ctypes, hipLaunchKernel) with a batch so small that the
device never is the bottleneck, and the device-side duration of the same launches at 4096 envs."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deep_reinforcement_learning_for_fjsp_amd import instances as fi
from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, global_actions, _ptr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
env = EnvBatch(s, N, rng_seed=3)
env.reset()
acts = torch.from_numpy(global_actions(1, 0, N, 64, 6, 5)).cuda()
rows = [acts[i] for i in range(64)]


def timeit(fn, n=20000):
    for i in range(500):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


print("N = %d" % N)
print("EnvBatch.step(actions[i %% 64], autoreset=True)   host %.2f us/call, incl. drain %.2f" % timeit(lambda i: env.step(acts[i % 64], autoreset=True)))
print("EnvBatch.step(rows[i %% 64], autoreset=True)      host %.2f us/call, incl. drain %.2f" % timeit(lambda i: env.step(rows[i % 64], autoreset=True)))
lib, h = env._lib, env._h
st = env._stream()
ps, pr, pd = _ptr(env.state), _ptr(env.reward), _ptr(env.done)
pa = [_ptr(r) for r in rows]
print("raw ctypes fjsp_env_step, cached arguments       host %.2f us/call, incl. drain %.2f" % timeit(lambda i: lib.fjsp_env_step(h, pa[i % 64], None, 1, ps, pr, pd, st)))
print("torch.cuda.current_stream().cuda_stream           host %.2f us/call" % timeit(lambda i: torch.cuda.current_stream(env.device).cuda_stream, 20000)[0])
print("tensor indexing acts[i %% 64]                      host %.2f us/call" % timeit(lambda i: acts[i % 64], 20000)[0])

#!/usr/bin/env python3
"""Secondary measurement: step-kernel throughput at the REFERENCE'S TRAINING DISTRIBUTION (MPPPO.py:149-154,160 with
Instance_generate.py:42-54: R 3..12 kinds, 5..50 jobs per kind, 3..5 operations per job, M 10..20 machines,
p 40..400) instead of the one-job-per-kind 10x5 workload of the headline bench: the per-(r, j) lists are long here
(hundreds of jobs per environment), which is what compute_params walks.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deep_reinforcement_learning_for_fjsp_amd import instances as fi
from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, global_actions

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--instances", type=int, default=256)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--M", type=int, default=15)
args = ap.parse_args()
N, NI = args.envs, args.instances
t0 = time.time()
insts = fi.InstanceSet(NI)
for i in range(NI):
    insts.generate(i, 5000 + i, fi.reference_generator_params(1.0, args.M, 1))
insts.solve_fluid()
prep = time.time() - t0
dims = [insts.dims(i) for i in range(NI)]
K = np.array([d["K"] for d in dims])
jobs = np.array([int(insts.arrays(i).count.sum()) for i in range(NI)])
ops = np.array([int((insts.arrays(i).count.sum(0) * insts.arrays(i).Jr).sum()) for i in range(NI)])
env = EnvBatch(insts, N, rng_seed=3)
env.reset()
acts = torch.from_numpy(global_actions(1, 0, N, 64, 6, 5)).cuda()
for i in range(30):
    env.step(acts[i % 64], autoreset=True)
torch.cuda.synchronize()
res = {}
for name, kw in (("with_state", {}), ("no_state", {"state": False})):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.steps):
        env.step(acts[i % 64], autoreset=True, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.steps
    res[name] = {"us_per_launch": us, "env_steps_per_s": N / us * 1e6}
st = env.read()["status"]
assert int((st != 0).sum()) == 0
print(json.dumps({"workload": "SO_FJSSP, reference training distribution (R 3-12, 5-50 jobs/kind, J 3-5, M %d, p 40-400), %d envs over %d "
                              "instances, random policy, per-step kernel with autoreset" % (args.M, N, NI),
                  "mean_K": float(K.mean()), "mean_jobs": float(jobs.mean()), "max_jobs": int(jobs.max()), "mean_ops_per_episode": float(ops.mean()),
                  "host_prep_s": round(prep, 2), **res}))

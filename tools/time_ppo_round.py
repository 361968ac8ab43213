#!/usr/bin/env python3
"""DIAGNOSTIC: where a PPO round of BASELINE config 3 spends its time (rollout vs learning)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_reinforcement_learning_for_fjsp_amd import instances as fi
from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSOFJSSP
from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO import MPPPO as M
N = 4096
insts = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
env = BatchedSOFJSSP(insts, rng_seed=7)
agent = M.PPO(env, hidden_size=128, hidden_layer=2, seed=1, max_steps=56, use_graph='--eager' not in sys.argv, fused_sampling='--eager' not in sys.argv,
              fused_rollout='--eager' not in sys.argv and '--per-step-rollout' not in sys.argv)
agent.run_one_policy_network()
orig_learn = agent.learner.learn
t_learn = [0.0]
def timed_learn(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_learn(*a, **k)
    torch.cuda.synchronize(); t_learn[0] += time.perf_counter() - t0
    return r
agent.learner.learn = timed_learn
torch.cuda.synchronize(); t0 = time.perf_counter()
R = 5
for _ in range(R): agent.run_one_policy_network()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("round %.1f ms: learn %.1f ms, rollout+returns %.1f ms" % (dt / R * 1e3, t_learn[0] / R * 1e3, (dt - t_learn[0]) / R * 1e3))

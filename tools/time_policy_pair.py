#!/usr/bin/env python3
"""GPU box: time of one fjsp_policy_pair_sample launch (3 x 200 task + machine networks, 4096 states; HIP events over 200 calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import MachinePolicyNet, TaskPolicyNet
from deep_reinforcement_learning_for_fjsp_amd.agents import fused_policy

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
task, machine = TaskPolicyNet(30, 200, 3, 12).cuda(), MachinePolicyNet(31, 200, 3, 10).cuda()
state = torch.rand(rows, 30, dtype=torch.float64, device="cuda")
sm = fused_policy.PolicyPairSampler(task.layers_1, machine.layers_2, seed=1, static_weights=True)
for _ in range(20):
    sm.sample(state)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    sm.sample(state)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 200
mac = rows * ((30 * 200 + 2 * 200 * 200 + 200 * 12) + (31 * 200 + 2 * 200 * 200 + 200 * 10))
print("%d rows: %.1f us per pair launch, %.1f TFLOP/s f32" % (rows, us, 2 * mac / us / 1e6))

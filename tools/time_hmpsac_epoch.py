#!/usr/bin/env python3
"""DIAGNOSTIC: where an epoch of the SAC controller (BASELINE config 5) spends its time: environment calls, the
captured act / store graphs, SAC learning sessions.  Synchronises around every piece, so the sum exceeds the real epoch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import helpers as H
from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedMODFJSP
from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import DA3C
from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.SAC_Discrete import SAC_Discrete

insts, _, _ = H.load_suite("mo_dfjsp")
industrial = H.instance_set_from([a for a in insts if not a.name.startswith("gen")])
N = 4096
env = BatchedMODFJSP(industrial, n_envs=N, rng_seed=5)
torch.manual_seed(0)
lower = {}
for policy in (0, 1, 2):
    tr = DA3C(lambda: BatchedMODFJSP(industrial, n_envs=64, rng_seed=policy), BatchedMODFJSP(industrial, rng_seed=11), reward_policy=policy, seed=policy, max_steps=64)
    lower[policy] = (tr.actor_task_model, tr.actor_machine_model)
sac = SAC_Discrete(env, lower_policies=lower, seed=1, max_steps=4096,
                   hyper={"min_steps_before_learning": 4 * N, "update_every_n_steps": 16 * N, "buffer_size": 1 << 20, "batch_size": 4096})
sac.run_one_epoch()
acc = {"env.step": 0.0, "learn": 0.0, "graph replay": 0.0}
cnt = {"env.step": 0, "learn": 0, "graph replay": 0}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[key] += time.perf_counter() - t0; cnt[key] += 1
        return r
    setattr(obj, name, g)
wrap(env, "step", "env.step")
wrap(sac, "learn", "learn")
for key, (ga, gs) in list(sac._graphs.items()):
    class R(object):
        def __init__(self, g): self.g = g
        def replay(self):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            self.g.replay()
            torch.cuda.synchronize(); acc["graph replay"] += time.perf_counter() - t0; cnt["graph replay"] += 1
    sac._graphs[key] = (R(ga), R(gs))
torch.cuda.synchronize(); t0 = time.perf_counter()
sac.run_one_epoch()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("epoch %.2f s (instrumented)" % dt)
for k in acc:
    print("  %-14s %6.2f s in %5d calls (%.3f ms each)" % (k, acc[k], cnt[k], acc[k] / max(cnt[k], 1) * 1e3))
print("  other          %6.2f s" % (dt - sum(acc.values())))

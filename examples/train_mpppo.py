#!/usr/bin/env python3
"""agents/MPPPO/MPPPO.py of the reference, batched: five policies with weight vectors (1,0) .. (0,1) on
MO_FJSSP_discretes environments (makespan + tardiness), a fresh batch of random instances per epoch
(generated_new_environment, MPPPO.py:149-154), the single-objective policies' results normalising the
rewards of the weighted ones (:159-164), periodic evolution towards the best policy per weight vector
(:192-205).

    python examples/train_mpppo.py --envs 1024 --epochs 3
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--epochs", type=int, default=3)
    args = ap.parse_args()
    import torch
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedMOFJSSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.MPPPO.MPPPO import MPPPO

    test_env = BatchedMOFJSSP(fi.InstanceSet(64).generate_range(900000, fi.bench_10x5_params()).solve_fluid(), rng_seed=1)
    epoch = [0]

    def make_train_env():
        epoch[0] += 1
        s = fi.InstanceSet(args.envs).generate_range(10_000_000 * epoch[0], fi.bench_10x5_params()).solve_fluid()
        return BatchedMOFJSSP(s, rng_seed=epoch[0])

    torch.manual_seed(0)
    agent = MPPPO(make_train_env, test_env, actor_number=5, hidden_size=200, hidden_layer=5, critic_layer=3, max_steps=56,
                  evolve_every=2)
    agent.run_n_episodes(1)                      # warm-up epoch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = agent.run_n_episodes(args.epochs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "MPPPO, 5 policies, %d MO_FJSSP_discretes 10x5 envs per epoch" % args.envs,
                      "epochs": args.epochs, "s_per_epoch": dt / args.epochs,
                      "test_objectives_last_epoch": {str(p): {"completion_time": c, "tardiness": t} for p, (c, t) in hist[-1].items()},
                      "completion_min": agent.completion_min, "tardiness_min": agent.tardiness_min}))


if __name__ == "__main__":
    main()

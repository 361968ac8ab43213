#!/usr/bin/env python3
"""agents/DDQN/DDQN.py of the reference, batched: double DQN (BatchNorm Q-network 3 x 200, replay ring in HBM)
on SO_SFJSP environments (makespan reward, 20 flat actions); every round plays one episode of a fresh batch of
random instances (generated_new_environment, DDQN.py:99-104: M in [3, 8]) and one greedy test episode.

    python examples/train_ddqn.py --envs 1024 --rounds 5
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
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    import torch
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedSOSFJSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.DDQN.DDQN import DDQN

    test_env = BatchedSOSFJSP(fi.InstanceSet(64).generate_range(800000, fi.bench_10x5_params()).solve_fluid(), rng_seed=1)
    rounds = [0]

    def make_train_env():
        rounds[0] += 1
        s = fi.InstanceSet(args.envs)
        for i in range(args.envs):
            seed = 5_000_000 * rounds[0] + i
            s.generate(i, seed, fi.GenParams(R_min=3, R_max=12, J_min=3, J_max=5, M=3 + seed % 6, p_min=40, p_max=400, N_min=1,
                                             N_max=2, S=1, DDT=0.5 + (seed % 11) / 10.0, t_si_min=100.0, t_si_max=200.0))
        return BatchedSOSFJSP(s.solve_fluid(), rng_seed=rounds[0])

    torch.manual_seed(0)
    agent = DDQN(make_train_env, test_env, updates_per_round=8, hyper={"learning_rate": 1e-4})
    agent.step()                                  # warm-up round
    torch.cuda.synchronize()
    t0, n0 = time.perf_counter(), agent.global_step_number
    tests = [agent.step() for _ in range(args.rounds)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "DDQN on %d SO_SFJSP envs per round" % args.envs, "rounds": args.rounds,
                      "train_env_steps_per_s": (agent.global_step_number - n0) / dt, "s_per_round": dt / args.rounds,
                      "test_makespan_per_round": tests, "best_test_makespan": agent.completed_time, "last_loss": agent.last_loss,
                      "replay_size": len(agent.memory), "epsilon": agent.exploration_strategy.epsilon}))


if __name__ == "__main__":
    main()

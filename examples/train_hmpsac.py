#!/usr/bin/env python3
"""BASELINE configs[4]: HMPSAC on the dynamic environment (MO_DFJSP_breakdown: order arrivals + machine
breakdowns + energy), everything batched on one MI355X:

  1. three lower-level objective policies (makespan / tardiness / energy) trained by the batched
     double-actor advantage actor-critic (agents/HMPSAC/A3C.py) on freshly generated dynamic instances;
  2. the SAC-discrete controller (agents/HMPSAC/SAC_Discrete.py) choosing among them, trained on N copies of
     the reference's industrial instances (data/industrial/DDT0.5_M20_S{1,3,5} + two data/HMPSAC folders as
     stored in tests/golden/mo_dfjsp.npz, replicated round-robin, different random streams per env).

    python examples/train_hmpsac.py --envs 4096 --lower-rounds 2 --epochs 2

Prints one JSON line: env-steps/s of the controller loop (policy inference + HIP env + host LP service at
order arrivals + SAC updates) and the objectives of the last epoch."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--lower-envs", type=int, default=512)
    ap.add_argument("--lower-rounds", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=2)
    args = ap.parse_args()
    import torch
    from tests import helpers as H
    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.environments import BatchedMODFJSP
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.A3C import DA3C
    from deep_reinforcement_learning_for_fjsp_amd.agents.HMPSAC.SAC_Discrete import SAC_Discrete

    insts, _, _ = H.load_suite("mo_dfjsp")
    industrial = H.instance_set_from([a for a in insts if not a.name.startswith("gen")])
    test_env = BatchedMODFJSP(industrial, rng_seed=11)
    rounds = [0]

    def make_train_env():      # generated_new_environment (A3C_v5.1.py:248-253) at reduced job counts + random windows
        rounds[0] += 1
        n = args.lower_envs
        s = fi.InstanceSet(n)
        for i in range(n):
            seed = 100000 * rounds[0] + i
            prm = fi.GenParams(R_min=3, R_max=8, J_min=3, J_max=5, M=10 + seed % 11, p_min=40, p_max=400, N_min=2, N_max=8,
                               S=1 + seed % 5, DDT=0.5 + (seed % 11) / 10.0, t_si_min=100.0, t_si_max=200.0)
            s.generate(i, seed, prm)
            while not (s.arrays(i).p > 0).any(axis=0).all():     # a machine nothing can run on: the reference's
                seed += 7919                                     # Machine.gap_ave divides by zero there; redraw
                s.generate(i, seed, prm)
            s.generate_machine_data(i, seed, max_windows=3, window_gap=(200, 3000), window_len=(20, 300))
        return BatchedMODFJSP(s.solve_fluid(), rng_seed=rounds[0])

    torch.manual_seed(0)
    lower, t0 = {}, time.perf_counter()
    lower_obj = {}
    for policy in (0, 1, 2):
        tr = DA3C(make_train_env, test_env, reward_policy=policy, seed=policy, max_steps=4096)
        lower_obj[policy] = [tr.run_one_round() for _ in range(args.lower_rounds)]
        lower[policy] = (tr.actor_task_model, tr.actor_machine_model)
    t_lower = time.perf_counter() - t0

    env = BatchedMODFJSP(industrial, n_envs=args.envs, rng_seed=5)
    sac = SAC_Discrete(env, lower_policies=lower, seed=1, max_steps=4096,
                       hyper={"min_steps_before_learning": 4 * args.envs, "update_every_n_steps": 16 * args.envs,
                              "buffer_size": 1 << 20, "batch_size": 4096})
    sac.run_one_epoch()        # warm-up epoch (allocations)
    torch.cuda.synchronize()
    steps0, lp0 = sac.global_step_number, env.batch.lp_solves
    t0 = time.perf_counter()
    out = None
    for _ in range(args.epochs):
        out = sac.run_one_epoch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctrl_steps = sac.global_step_number - steps0
    print(json.dumps({
        "workload": "HMPSAC on %d MO_DFJSP_breakdown envs (industrial instances), 1 GPU" % args.envs,
        "epochs": args.epochs, "s_per_epoch": dt / args.epochs,
        "env_steps_per_s": 4 * ctrl_steps / dt,      # an epoch = 3 baseline episodes + the controller episode
        "controller_transitions": ctrl_steps, "order_arrival_lps": env.batch.lp_solves - lp0,
        "learn_sessions": sac.learn_sessions, "losses": sac.last_losses, "alpha": float(sac.alpha.detach()),
        "last_epoch_mean_objectives": {"completion_time": out[0], "delay_time_sum": out[1], "energy_consumption": out[2]},
        "lower_policy_training_s": t_lower, "lower_policy_test_objectives": lower_obj}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Secondary measurement (BASELINE configs[4]): MO_DFJSP_breakdown at 4096 envs on one MI355X.

    python tools/bench_dynamic.py [--envs 4096] [--steps 300] [--lp-threads 0]

Workload: the reference's `data/industrial/DDT0.5_M20_S{1,3,5}` and two `data/HMPSAC` instances (K = 31..45,
M = 10..20, 63 breakdown windows, 1-5 orders) as stored in tests/golden/mo_dfjsp.npz, replicated round-robin
to N envs that differ by random-choice stream and action sequence; random policy over the 12 x 10 rule
pairs; one bench step = fjsp_env_step over all envs with autoreset (per-step kernel + the host LP service
for the envs that hit an order arrival in that step + arrival_kernel).  Prints one JSON line; this is NOT
the headline metric (bench.py), it sizes the dynamic environment and its LP service.
"""
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
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--lp-threads", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--instances", choices=["all", "industrial"], default="all",
                    help="industrial: only data/industrial (K 31, M 20) -- every order-arrival LP fits a CU's LDS and is solved on the device")
    ap.add_argument("--blocking", action="store_true", help="fjsp_env_step (every call waits for its order-arrival LPs) instead of "
                                                            "the asynchronous arrival service (parked envs wait, the others step)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from tests import helpers as H
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch, VARIANT_MO_DFJSP

    insts, _, _ = H.load_suite("mo_dfjsp")
    insts = [a for a in insts if not a.name.startswith("gen")]
    if args.instances == "industrial":
        insts = [a for a in insts if a.name.startswith("industrial")]
    s = H.instance_set_from(insts)
    N = args.envs
    env = EnvBatch(s, N, variant=VARIANT_MO_DFJSP, rng_seed=77)
    env.set_lp_threads(args.lp_threads)
    Tbuf = 64
    rs = np.random.RandomState(99)
    actions_h = np.stack([rs.randint(0, 12, (Tbuf, N)), rs.randint(0, 10, (Tbuf, N))], 2).astype(np.uint8)
    actions = torch.from_numpy(actions_h).cuda()
    mo = torch.zeros(N, 4, dtype=torch.float64, device="cuda")
    mo[:, 0] = 1.0
    env.reset()
    if args.blocking:
        for i in range(args.warmup):
            env.step(actions[i % Tbuf], autoreset=True, mo=mo)
        torch.cuda.synchronize()
        lp0 = env.lp_solves
        t0 = time.perf_counter()
        for i in range(args.steps):
            env.step(actions[i % Tbuf], autoreset=True, mo=mo)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lps = env.lp_solves - lp0
        env_steps = N * args.steps
    else:
        # asynchronous service: a call completes a step of every env that is not parked at an order arrival; completed
        # steps are counted on the device (sum of the ready flags), parked envs catch up inside the timed region's flush
        for i in range(args.warmup):
            env.step_async(actions[i % Tbuf], autoreset=True, mo=mo)
        env.flush_arrivals(mo)
        torch.cuda.synchronize()
        lp0 = env.lp_solves
        count = torch.zeros(N, dtype=torch.int32, device="cuda")
        t0 = time.perf_counter()
        for i in range(args.steps):
            _, _, _, ready = env.step_async(actions[i % Tbuf], autoreset=True, mo=mo)
            count.add_(ready)
        env.flush_arrivals(mo)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lps = env.lp_solves - lp0
        env_steps = int(count.sum().item())
    st = env.read()["status"]
    assert int((st != 0).sum().item()) == 0, "an environment reported an error status"

    # CPU oracle on one core, same instances (LP through the product's host simplex, like the kernels)
    cpu = None
    if args.cpu_seconds > 0:
        steps = 0
        tc = time.perf_counter()
        e = 0
        while time.perf_counter() - tc < args.cpu_seconds:
            a = insts[e % len(insts)]
            got = H.play_oracle(a, a.x, np.tile(actions_h[:, e % N], (40, 1)), env.env_seed(e), variant=4, mo=(1, 0, 0, 0))
            steps += got["T"]
            e += 1
        cpu = {"value": steps / (time.perf_counter() - tc), "unit": "env-steps/s", "cores": 1, "kind": "port",
               "sample": "%d full random-policy episodes on oracle/fjsp_oracle.c (python-driven, one core)" % e}
    print(json.dumps({
        "metric": "env-steps/sec (batched MO_DFJSP_breakdown, industrial instances)", "value": env_steps / dt,
        "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "dtype": "f64", "data": "reference data/industrial + data/HMPSAC instances replicated to %d envs" % N,
        "config": {"workload": "BASELINE configs[4] environment side: %d MO_DFJSP_breakdown envs, random 12x10 rule "
                               "policy, per-step kernel + host LP service at order arrivals" % N,
                   "arrival_service": "blocking (fjsp_env_step)" if args.blocking else "asynchronous (fjsp_env_step_async)",
                   "env_steps_completed": env_steps,
                   "instances": [a.name for a in insts], "lp_threads": min(args.lp_threads or 16, os.cpu_count() or 1),
                   "lp_on_device": int(env.lp_on_device)},
        "order_arrival_lps": lps, "lps_per_step": lps / args.steps, "lp_cache_hits_total": env.lp_cache_hits, "lp_device_pivots_total": env.lp_device_pivots, "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()

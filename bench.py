#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: batched SO_FJSSP env steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1], per GPU): 4096 generated 10-job x 5-machine
SO_FJSSP instances (generator seeds 1000 + global env id, fluid LP solved on the
host before the timed region), random policy from a pre-generated action tensor
resident in HBM, ONE bench step = ONE launch of the per-step HIP kernel over all
envs through the C ABI (fjsp_env_step, autoreset on, so every launch advances
every env by exactly one environment step).

Prints ONE JSON line (rank 0): metric env-steps/s (whole job), plus
  roofline      the step kernel's algorithmic HBM bytes per launch / its mean launch
                duration from HIP events, against the 8 TB/s HBM peak
  cpu_baseline  the C oracle (a scalar CPU port of the reference) timed on one host
                core on a bounded sample of the same instances and actions
  fused         the T-steps-per-launch rollout kernel on the same workload

For N > 1 the driver launches this under torch.distributed.run, one rank per GPU;
envs are sharded by global env id, there is no data-path collective (the path
shards: SURVEY.md 8e), the timed region is bracketed by barriers and the max over
ranks is reported ("weak" scaling: per-GPU work is fixed).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# reference Python step() measured in the build container (tests/golden/GENERATION_REPORT.txt):
# 1 core, synthetic 10x5 instances, reference env under oracle/ref_shim
REFERENCE_PYTHON_STEPS_PER_S = 1700.0


def csrc_hash():
    """sha256 over the sources the environment kernels are compiled from: ties a committed PMC figure to the code it
    was measured on (the agents' kernels -- fjsp_ppo.hip, fjsp_mlp_train.hip, fjsp_policy_mlp.hip, fjsp_rollout_buffer.hip -- are not part of the measured
    launch)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(REPO, "deep_reinforcement_learning_for_fjsp_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) and f not in ("fjsp_ppo.hip", "fjsp_mlp_train.hip", "fjsp_policy_mlp.hip", "fjsp_rollout_buffer.hip"):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the K timed launches from a captured HIP graph (auto: when K <= 256, where a region is launch-bound at its edges)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                         % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path for the environment kernels")
    # FJSP_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks then share devices); the driver's runs use the default: nccl == RCCL, one GPU per rank
    backend = os.environ.get("FJSP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from deep_reinforcement_learning_for_fjsp_amd import instances as fi
    from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch

    N = args.envs_per_gpu
    first_env = rank * N
    t0 = time.time()
    insts = fi.InstanceSet(N).generate_range(1000 + first_env, fi.bench_10x5_params()).solve_fluid()
    t_prep = time.time() - t0
    K = np.array([insts.dims(i)["K"] for i in range(N)])
    Tbuf = 64
    # instances, actions and choice streams are functions of the GLOBAL env id: the job's traces do not depend on
    # how many GPUs it is sharded over (tests/test_gpu_parity.py::test_shards_concatenate_to_the_unsharded_batch)
    from deep_reinforcement_learning_for_fjsp_amd.batch import global_actions
    actions_h = global_actions(4242, first_env, N, Tbuf, 6, 5)
    actions = torch.from_numpy(actions_h).cuda(local_rank)
    env = EnvBatch(insts, N, device=local_rank, rng_seed=20260, first_env=first_env)
    env.reset()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        env.step(actions[i % Tbuf], autoreset=True)
    sync_all()

    # The timed region is EXACTLY `steps` launches between barrier + synchronize on both sides.  A region of a few
    # launches lasts a fraction of a millisecond -- launch ramp-up and clock noise dominate it -- so the region is
    # repeated (at least 25 times and until 50 ms have been timed, at most 2 000 times) and the MEDIAN region is
    # reported; every region is the max over ranks.
    # Short regions (the driver's --steps 20) are launch-bound at their edges: the K launches are captured once into a HIP
    # graph and a region replays it -- the same K step launches on the same stream, one host call instead of K.
    graph = None
    if args.graph == "on" or (args.graph == "auto" and args.steps <= 256):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(3):
                env.step(actions[i % Tbuf], autoreset=True)
        torch.cuda.current_stream().wait_stream(side)
        sync_all()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(args.steps):
                env.step(actions[i % Tbuf], autoreset=True)
        sync_all()

    def timed_region(k, with_events=True):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if with_events:
            ev0.record()
        if graph is not None:
            graph.replay()
        else:
            for i in range(k):
                env.step(actions[(step_ctr[0] + i) % Tbuf], autoreset=True)
        if with_events:
            ev1.record()
        sync_all()
        dt = time.perf_counter() - t0
        step_ctr[0] += k
        return dt, (ev0.elapsed_time(ev1) if with_events else None)

    step_ctr = [args.warmup]
    regions = []
    local_dts = []
    total_t = 0.0
    # Short regions: the two event records around the K launches are stream operations of their own (a few microseconds of a
    # 0.2 ms region), so the wall-clock regions run without them and the HIP-event figure of the roofline comes from a second
    # set of regions of the same K launches, timed by events only.
    split_events = graph is not None
    while len(regions) < 25 or (total_t < 0.05 and len(regions) < 2000):
        dt, ev_ms = timed_region(args.steps, with_events=not split_events)
        local_dts.append(dt)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        regions.append((dt, ev_ms))
        total_t += dt
    if split_events:
        ev_regions = [timed_region(args.steps, with_events=True)[1] for _ in range(max(25, min(len(regions), 200)))]
    else:
        ev_regions = [r[1] for r in regions]
    regions.sort(key=lambda r: r[0])
    elapsed_local = sorted(local_dts)[len(local_dts) // 2]       # this rank's own median region (before the max over ranks)
    elapsed, _ = regions[len(regions) // 2]
    region_ms = sorted(ev_regions)[len(ev_regions) // 2]
    status = env.read()["status"]
    assert int((status != 0).sum().item()) == 0, "an environment reported an error status during the bench"

    # ---- roofline leg: per-launch HIP events on the launch stream ------------------
    n_ev = min(args.steps, 400)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ev)]
    for i, (a, b) in enumerate(evs):
        a.record()
        env.step(actions[i % Tbuf], autoreset=True)
        b.record()
    torch.cuda.synchronize()
    per_launch_us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    kern_us = float(np.mean(per_launch_us[n_ev // 10: n_ev - n_ev // 10]))     # trimmed mean
    bytes_per_launch = env.step_bytes * N

    # ---- fused rollout kernel on the same workload ----------------------------------
    T = int(K.max())
    fused = None
    if rank == 0:
        env2 = EnvBatch(insts, N, device=local_rank, rng_seed=1)
        reps = 20
        env2.reset(); env2.rollout(actions[:T], trace=False, rewards=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for _ in range(reps):
            env2.reset()
            e0.record(); env2.rollout(actions[:T], trace=False, rewards=False); e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        steps_per_rollout = int(K.sum())
        tot2 = 0.0
        for _ in range(reps):                 # rule-sweep form: no state handed back, the kernel skips the observation
            env2.reset()
            e0.record(); env2.rollout(actions[:T], trace=False, rewards=False, state=False); e1.record()
            torch.cuda.synchronize()
            tot2 += e0.elapsed_time(e1)
        # The fused kernel keeps the environment in registers + LDS: it moves the state once per EPISODE, so an HBM
        # figure says nothing about it.  Its bound is instruction issue: instructions per wave-step (rocprofv3 PMC of
        # this tree, profiles/traffic_step_kernel.json "fused_insts_per_wave_step") x wave-steps / (1024 SIMDs x
        # 2.4 GHz, one instruction per cycle and SIMD).
        fused = {"kernel": "grollout_kernel" if env2.kernel_family == 1 else "rollout_kernel", "env_steps_per_launch": steps_per_rollout,
                 "env_steps_per_s_no_state": steps_per_rollout / (tot2 / reps * 1e-3),
                 "ms_per_launch": tot / reps, "env_steps_per_s": steps_per_rollout / (tot / reps * 1e-3),
                 "bytes_per_env_episode": None, "issue_roofline": None}

    # ---- cpu baseline: the oracle on one host core, bounded sample -------------------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import pyoracle
        n_s = 256
        envs = []
        for i in range(n_s):
            a = insts.arrays(i)
            envs.append((pyoracle.OracleEnv(a, a.x, rng_seed=env.env_seed(i)), np.ascontiguousarray(actions_h[:, i])))
        steps = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 10.0:
            for oe, act in envs:
                n, _ = oe.play(act)
                steps += n
        dt = time.perf_counter() - t0
        cpu = {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
               "sample": "oracle/fjsp_oracle.c (scalar C port of the reference, bit-exact to it) replaying full "
                         "random-policy episodes of the first %d instances of this workload for %.1f s on one host "
                         "core (%d steps); reference Python itself: ~%.0f env-steps/s on one core (build container, "
                         "tests/golden/GENERATION_REPORT.txt)" % (n_s, dt, steps, REFERENCE_PYTHON_STEPS_PER_S)}

    # the same port on the host cores the box gives one GPU (16): one oracle env per thread, ctypes releases the GIL
    cpu_mt = None
    if cpu is not None:
        from concurrent.futures import ThreadPoolExecutor
        n_thr = min(16, os.cpu_count() or 1)
        per = (len(envs) + n_thr - 1) // n_thr           # contiguous shards: neighbouring oracle envs share cache lines
        shards = [envs[i * per:(i + 1) * per] for i in range(n_thr) if envs[i * per:(i + 1) * per]]

        def worker(my, deadline):
            n_steps = 0
            while time.perf_counter() < deadline:
                n_steps += pyoracle.play_many(my, 8)      # 8 episodes of each env per C call, GIL released
            return n_steps
        t0 = time.perf_counter()
        with ThreadPoolExecutor(len(shards)) as ex:
            done_steps = sum(ex.map(lambda my: worker(my, t0 + 5.0), shards))
        dt = time.perf_counter() - t0
        cpu_mt = {"value": done_steps / dt, "unit": "env-steps/s", "cores": len(shards), "kind": "port",
                  "sample": "the same %d oracle environments spread over %d host threads (fjo_play_many, 8 episodes per C call) for %.1f s" % (n_s, n_thr, dt)}

    # ---- multi-rank evidence: how many ranks the process group really has, and each rank's own rate (stragglers) ----
    ranks_seen, per_rank_value = 1, None
    my_value = N * args.steps / elapsed_local
    if world > 1:
        dev = "cuda" if backend == "nccl" else "cpu"
        one = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one)                                   # (RCCL over xGMI with the nccl backend)
        ranks_seen = int(one.item())
        mine = torch.tensor([my_value], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank_value = [float(g.item()) for g in gathered]
    else:
        per_rank_value = [my_value]

    # the kernel that steps this batch: 16-lane-row family (csrc/fjsp_group.hip; its variant that requests gap_ave's rows ahead
    # of time for per-step launches of at most 5120 envs) or one wavefront per environment (csrc/fjsp_kernels.hip)
    if env.kernel_family == 1:
        step_kernel_name = "grp::gstep_kernel<0, 5, %s>" % ("true" if (N <= 5120 and os.environ.get("FJSP_GROUP_EARLY") != "0") or os.environ.get("FJSP_GROUP_EARLY") == "1" else "false")
    else:
        step_kernel_name = "step_kernel<1, 0, true>"
    if rank == 0:
        total_steps = N * world * args.steps
        # average launch duration = HIP events bracketing the K timed launches on the launch stream / K
        # (agrees with rocprofv3's per-kernel average within ~1 %, profiles/README.md; events around every
        # single launch read ~2 us high, reported as launch_us_per_launch_events)
        region_us = region_ms * 1e3 / args.steps
        achieved = bytes_per_launch / (region_us * 1e-6) / 1e9
        # roofline.traffic is a PMC figure (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
        # tools/profile_bench.sh); PMC cannot be collected from inside the run, so the committed figure is used only
        # when it was taken on THIS kernel source (hash of csrc/), otherwise traffic is null
        traffic, traffic_note = None, "no PMC profile of this kernel source is committed (tools/profile_bench.sh)"
        tpath = os.path.join(REPO, "profiles", "traffic_step_kernel.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            fresh = tj.get("csrc_sha256") == csrc_hash()
            if fresh and N != tj.get("envs", 4096) and str(N) in tj.get("by_envs", {}):
                traffic = tj["by_envs"][str(N)]["traffic_bytes_per_launch"]
                traffic_note = tj["note"] + " (this batch size: %s)" % tj.get("by_envs_source", "tools/traffic_sizes.sh")
            elif fresh and N == tj.get("envs", 4096):
                traffic, traffic_note = tj["traffic_bytes_per_launch"], tj["note"]
                if fused is not None and tj.get("fused_insts_per_env_step"):
                    ips = tj["fused_insts_per_env_step"]
                    peak = 1024 * 2.4e9 / ips                 # env-steps/s if every SIMD issued one instruction per cycle
                    fused["issue_roofline"] = {"insts_per_env_step": ips, "peak_env_steps_per_s": peak,
                                               "frac": fused["env_steps_per_s"] / peak,
                                               "note": "1024 SIMDs x 2.4 GHz / wave instructions per env-step (SQ_INSTS_VALU + SALU + LDS + "
                                                       "SMEM + VMEM of the fused kernel over the env-steps of its launch, PMC); a wave alone "
                                                       "on its SIMD issues a dependent instruction every ~10 cycles, so the reachable "
                                                       "fraction at 4096 environments is ~0.1"}
                    fused["bytes_per_env_episode"] = tj.get("fused_bytes_per_env_episode")
            elif fresh:
                traffic_note = "no PMC passes at this batch size are committed for this kernel source (tools/traffic_sizes.sh)"
            else:
                traffic_note = "profiles/traffic_step_kernel.json was taken on another kernel source (%s): stale, not reported" \
                               % tj.get("commit", "?")
        out = {
            "metric": "env-steps/sec (batched SO_FJSSP 10x5)",
            "value": total_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "per_rank_value": per_rank_value,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "timed_regions": len(regions),
            "region_ms_min_median_max": [regions[0][0] * 1e3, elapsed * 1e3, regions[-1][0] * 1e3],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d parallel SO_FJSSP 10x5 generated instances per GPU "
                                   "(seeds 1000+i), random policy, per-step HIP kernel (%s) with autoreset, "
                                   "one launch = one env step of every env" % (N, step_kernel_name),
                       "envs_per_gpu": N, "mean_ops_per_instance": float(K.mean()), "sharding": "env id range per rank, no collective",
                       "launch": ("the K step launches of a timed region replayed from one captured HIP graph" if graph is not None
                                  else "K step launches from the host per timed region"),
                       "host_prep_s": round(t_prep, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": step_kernel_name, "bytes_per_env_step": env.step_bytes,
                         "bytes_per_launch": bytes_per_launch, "launch_us_hip_events": region_us,
                         "hip_event_regions": ("%d further regions of the same K launches, bracketed by HIP events on the launch stream (the "
                                               "wall-clock regions of a short run carry no event records)" % len(ev_regions)) if split_events
                                              else "the wall-clock regions themselves, bracketed by HIP events on the launch stream",
                         "launch_us_per_launch_events": kern_us},
            "cpu_baseline": cpu,
            "cpu_baseline_multicore": cpu_mt,
            "fused": fused,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

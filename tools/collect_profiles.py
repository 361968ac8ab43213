#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/<tag>_* and refresh
profiles/traffic_step_kernel.json (tools/update_traffic.py).  Run in the build container after the gpurun call."""
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
names = {
    "bench.json": "bench.json", "bench_under_rocprof.json": "bench_under_rocprofv3.json",
    "kernel_stats.csv": "rocprofv3_kernel_stats_bench_steps500.csv", "pmc_means.json": "rocprofv3_pmc_per_dispatch_means.json",
    "bench_envs32768.json": "bench_envs32768.json", "bench_envs262144.json": "bench_envs262144.json",
    "kernel_stats_envs32768.txt": "rocprofv3_kernel_stats_envs32768.txt", "kernel_stats_envs262144.txt": "rocprofv3_kernel_stats_envs262144.txt",
    "fetch_calibration.json": "fetch_calibration.json", "bench_mo_dfjsp_blocking.json": "bench_mo_dfjsp_blocking.json",
    "bench_mo_dfjsp_async.json": "bench_mo_dfjsp_async.json", "bench_training_distribution.json": "bench_training_distribution.json",
    "train_ppo.json": "train_ppo.json", "train_ppo_per_step_rollout.json": "train_ppo_per_step_rollout.json",
    "ppo_round_split.txt": "ppo_round_split.txt", "ppo_round_kernel_stats.csv": "ppo_round_rocprofv3_kernel_stats.csv",
    "host_overhead.txt": "host_overhead.txt", "train_hmpsac.json": "train_hmpsac.json", "hmpsac_epoch_split.txt": "hmpsac_epoch_split.txt",
    "mlp_pass_timing.txt": "mlp_pass_timing.txt",
}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_%s" % (tag, b)))
        print("copied", a)
    else:
        print("MISSING", a)
subprocess.run([sys.executable, os.path.join(REPO, "tools", "update_traffic.py"), src], check=True)

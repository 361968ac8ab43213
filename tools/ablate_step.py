#!/usr/bin/env python3
"""DIAGNOSTIC ONLY: time step_kernel with phases compiled out (-DFJSP_ABLATE=n; results are wrong by
construction) to see where a launch's time goes.  0 full | 1 no emit_state | 2 no observe/emit |
3 loads+compute_params+stores | 4 loads+stores | 5 empty kernel | 6 full without the column gather of machine_select |
7 stop after task_select | 8 stop after machine_select | 9 stop after dispatch_and_advance."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
csrc = os.path.join(REPO, "deep_reinforcement_learning_for_fjsp_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("fjsp_kernels.hip", "fjsp_env.hip", "fjsp_rollout_buffer.hip", "fjsp_ppo.hip", "fjsp_mlp_train.hip", "fjsp_policy_mlp.hip", "fjsp_group.hip", "fjsp_lp_device.hip",
                                         "fjsp_instance.cpp", "fjsp_lp.cpp")]
BUILD_ONLY = "--build-only" in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 4096
child = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, %r)
from deep_reinforcement_learning_for_fjsp_amd import instances as fi
from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch
N = %d
s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
rs = np.random.RandomState(1)
acts = torch.from_numpy(np.stack([rs.randint(0, 6, (64, N)), rs.randint(0, 5, (64, N))], 2).astype(np.uint8)).cuda()
env = EnvBatch(s, N, rng_seed=3); env.reset()
for i in range(100): env.step(acts[i %% 64], autoreset=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(1000): env.step(acts[i %% 64], autoreset=True)
e1.record(); torch.cuda.synchronize()
print("%%.2f us/launch" %% (e0.elapsed_time(e1)))
''' % (REPO, N)
LEVELS = [int(a.split('=')[1]) for a in sys.argv if a.startswith('--level=')] or [5, 4, 3, 7, 8, 9, 2, 1, 0]
for level in LEVELS:
    out = os.path.join(REPO, "gpurun_out", "libfjsp_ablate%d.so" % level)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-ffp-contract=off", "-DFJSP_ABLATE=%d" % level, "-Wno-unused-function",
                    "-I", os.path.join(REPO, "include"), "-I", csrc] + srcs + ["-o", out, "-lpthread"], check=True)
    if BUILD_ONLY:
        continue
    env = dict(os.environ, FJSP_AMD_LIB=out)
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    print("ablate level", level, r.stdout.strip(), r.stderr.strip()[-200:] if r.returncode else "")
    os.remove(out)

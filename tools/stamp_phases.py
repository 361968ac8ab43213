#!/usr/bin/env python3
"""DIAGNOSTIC ONLY: build libfjsp_amd with -DFJSP_STAMPS into gpurun_out/ and print the share of
wave-cycles each phase of step_kernel takes (s_memtime deltas summed by lane 0 of every wave).
Never used for timing claims: the stamps' own waits perturb the schedule; read the SHARES."""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
out = os.path.join(REPO, ".diag", "libfjsp_amd_stamps.so")
csrc = os.path.join(REPO, "deep_reinforcement_learning_for_fjsp_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("fjsp_kernels.hip", "fjsp_env.hip", "fjsp_rollout_buffer.hip", "fjsp_ppo.hip", "fjsp_mlp_train.hip",
                                         "fjsp_instance.cpp", "fjsp_lp.cpp")]
os.makedirs(os.path.dirname(out), exist_ok=True)
if not (os.path.exists(out) and "--no-build" in sys.argv):
  subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                "-ffp-contract=off", "-DFJSP_STAMPS", "-Wno-unused-function", "-I", os.path.join(REPO, "include"),
                "-I", csrc] + srcs + ["-o", out, "-lpthread"], check=True)
os.environ["FJSP_AMD_LIB"] = out
if "--build-only" in sys.argv:
    sys.exit(0)
import numpy as np
import torch
from deep_reinforcement_learning_for_fjsp_amd import instances as fi, _capi
from deep_reinforcement_learning_for_fjsp_amd.batch import EnvBatch

_a = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(_a[0]) if _a else 4096
s = fi.InstanceSet(N).generate_range(1000, fi.bench_10x5_params()).solve_fluid()
rs = np.random.RandomState(1)
acts = torch.from_numpy(np.stack([rs.randint(0, 6, (64, N)), rs.randint(0, 5, (64, N))], 2).astype(np.uint8)).cuda()
env = EnvBatch(s, N, rng_seed=3)
env.reset()
lib = C.CDLL(out)
buf = (C.c_ulonglong * 16)()
for i in range(50):
    env.step(acts[i % 64], autoreset=True)
torch.cuda.synchronize()
lib.fjsp_debug_read_stamps(buf, 1)
for i in range(200):
    env.step(acts[i % 64], autoreset=True)
torch.cuda.synchronize()
lib.fjsp_debug_read_stamps(buf, 0)
names = ["open_env (loads)", "compute_params #1 / autoreset", "task_select", "machine_select", "dispatch+advance",
         "compute_params #2", "observe_prepare", "barrier 1 (wait for the workgroup)", "tail pass 1 + barrier 2",
         "deviations + barrier 3", "tail pass 2 + barrier 4", "finish + emit_state", "outputs + store_dynamic"]
waves = buf[15]
tot = sum(buf[i] for i in range(13))
print("waves", waves, "mean stamped shader cycles per wave", tot / waves)
for i, n in enumerate(names):
    print("%-32s %8.0f ticks/wave  %5.1f %%" % (n, buf[i] / waves, 100.0 * buf[i] / tot))

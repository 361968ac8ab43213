#!/usr/bin/env python3
"""DIAGNOSTIC ONLY: build libfjsp_amd with -DFJSP_GSTAMPS into .diag/ and print the shader cycles each phase of the group
step kernel takes per wave (s_memtime deltas summed by lane 0 of every wave, waits included).  Never used for timing
claims: the stamps perturb the schedule; read the shares.  Usage: tools/stamp_group.py [--build-only|--no-build] [N]"""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
out = os.path.join(REPO, ".diag", "libfjsp_gstamps.so")
csrc = os.path.join(REPO, "deep_reinforcement_learning_for_fjsp_amd", "csrc")
srcs = [os.path.join(csrc, f) for f in ("fjsp_kernels.hip", "fjsp_group.hip", "fjsp_env.hip", "fjsp_rollout_buffer.hip", "fjsp_ppo.hip",
                                         "fjsp_mlp_train.hip", "fjsp_instance.cpp", "fjsp_lp.cpp")]
os.makedirs(os.path.dirname(out), exist_ok=True)
if "--no-build" not in sys.argv:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                    "-DFJSP_GSTAMPS", "-Wno-unused-function", "-I", os.path.join(REPO, "include"), "-I", csrc] + srcs +
                   ["-o", out, "-lpthread"], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["FJSP_AMD_LIB"] = out
import numpy as np
import torch
from deep_reinforcement_learning_for_fjsp_amd import instances as fi
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
lib.fjsp_debug_read_gstamps(buf, 1)
for i in range(200):
    env.step(acts[i % 64], autoreset=True)
torch.cuda.synchronize()
lib.fjsp_debug_read_gstamps(buf, 0)
names = ["open (loads in)", "autoreset", "current-op gather", "task_select", "gap_ave rows", "machine_select (+ walk)",
         "dispatch + event loop", "statistics + observation", "emit + reward + store"]
waves = buf[15]
tot = sum(buf[i] for i in range(9))
print("waves", waves, "mean stamped shader cycles per wave", tot / waves)
for i, n in enumerate(names):
    print("%-28s %8.0f cycles/wave  %5.1f %%" % (n, buf[i] / waves, 100.0 * buf[i] / tot))

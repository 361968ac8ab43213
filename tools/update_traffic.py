#!/usr/bin/env python3
"""profiles/traffic_step_kernel.json from a tools/profile_bench.sh output directory (gpurun_out/prof_<tag>/):
the PMC traffic of step_kernel<1, 0> that bench.py reports as roofline.traffic, stamped with the hash of the
kernel sources it was measured on (bench.py drops the figure when the sources have changed since), plus the
fused kernel's instruction count per wave-step for its issue roofline.

FETCH_SIZE correction: rocprofv3 reports FETCH_SIZE in KiB; on gfx950 it counts HALF the bytes of coalesced row
reads at 4, 8 and 16 bytes per lane alike (tools/fetch_calib.sh, profiles/r02_r_fetch_calibration.json: reported /
true = 0.5000 for all three), so it is doubled.  WRITE_SIZE is taken as reported."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import csrc_hash  # noqa: E402

src = sys.argv[1]
tag = os.path.basename(src.rstrip("/")).replace("prof_", "")
pm = json.load(open(os.path.join(src, "pmc_means.json")))
skey = [k for k in pm if "gstep_kernel" in k or "step_kernel<1, 0, true>" in k][0]
sk = pm[skey]
fetch = sk["FETCH_SIZE"] * 1024 * 2.0
write = sk["WRITE_SIZE"] * 1024
bench = json.load(open(os.path.join(src, "bench.json")))
out = {
    "kernel": skey,
    "envs": 4096,
    "fetch_size_kib_per_launch_reported": sk["FETCH_SIZE"],
    "write_size_kib_per_launch_reported": sk["WRITE_SIZE"],
    "traffic_bytes_per_launch": fetch + write,
    "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
    "source": "profiles/%s_rocprofv3_pmc_per_dispatch_means.json" % tag,
    "csrc_sha256": csrc_hash(),
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip(),
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_bench.sh), per-dispatch means over "
            "%d dispatches; FETCH_SIZE doubled (gfx950 counts half the bytes of coalesced 4/8/16 B-per-lane row reads: "
            "profiles/r02_r_fetch_calibration.json), WRITE_SIZE as reported; the ~18 MB a launch touches stay in the 256 MB "
            "Infinity Cache between launches, whose hits these fabric-side counters include" % sk.get("dispatches_pmc_fetch", 0),
    "insts_per_wave_step": {k: sk[k] / sk["SQ_WAVES"] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
                                                                "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if k in sk},
}
rk = pm.get([k for k in pm if "rollout_kernel" in k][0]) if [k for k in pm if "rollout_kernel" in k] else None
if rk and bench.get("fused"):
    steps_per_wave = bench["fused"]["env_steps_per_launch"] / rk["SQ_WAVES"]
    insts = sum(rk.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD",
                                         "SQ_INSTS_VMEM_WR"))
    # (the profiled run alternates rollouts with and without a final state; the mean is over both)
    # (instructions of all waves / env-steps of the launch: a wave of the row kernels steps four environments at once)
    out["fused_insts_per_env_step"] = insts / rk["SQ_WAVES"] / steps_per_wave
    out["fused_bytes_per_env_episode"] = (rk["FETCH_SIZE"] * 1024 * 2.0 + rk["WRITE_SIZE"] * 1024) / rk["SQ_WAVES"]
envs_per_wave = 4096.0 / sk["SQ_WAVES"]
out["envs_per_wave"] = envs_per_wave
out["insts_per_env_step"] = sum(out["insts_per_wave_step"].values()) / envs_per_wave
# other batch sizes: a tools/traffic_sizes.sh summary (second argument), same corrections
if len(sys.argv) > 2:
    by = {}
    for n, v in json.load(open(sys.argv[2])).items():
        k = [k for k in v if "gstep_kernel" in k or "step_kernel<" in k]
        if not k:
            continue
        c = v[k[0]]
        by[n] = {"kernel": k[0], "traffic_bytes_per_launch": c["FETCH_SIZE"] * 1024 * 2.0 + c["WRITE_SIZE"] * 1024,
                 "fetch_bytes_per_env_step": c["FETCH_SIZE"] * 1024 * 2.0 / int(n), "write_bytes_per_env_step": c["WRITE_SIZE"] * 1024 / int(n),
                 "read_requests_128B_per_env_step": c.get("TCC_EA0_RDREQ_sum", 0.0) / int(n), "dispatches": c["dispatches"]}
    out["by_envs"] = by
    out["by_envs_source"] = os.path.relpath(sys.argv[2], REPO)
json.dump(out, open(os.path.join(REPO, "profiles", "traffic_step_kernel.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

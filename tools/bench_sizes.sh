#!/bin/bash
# GPU box: bench.py at several batch sizes for one step implementation (FJSP_STEP_IMPL), one summary line per size
IMPL=${1:-group}; shift || true
for n in ${@:-4096 32768 262144}; do
  FJSP_STEP_IMPL=$IMPL python bench.py --envs-per-gpu $n --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/b_${IMPL}_$n.json 2> gpurun_out/b_${IMPL}_$n.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/b_${IMPL}_$n.json").read().strip().splitlines()[-1])
f=d.get("fused") or {}
print("$IMPL", $n, round(d["value"]/1e6,1), "M/s", round(d["roofline"]["launch_us_hip_events"],2), "us frac", round(d["roofline"]["frac"],3), "fused", round(f.get("env_steps_per_s",0)/1e6,1), round(f.get("env_steps_per_s_no_state",0)/1e6,1))
PY
done

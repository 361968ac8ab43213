#!/bin/bash
# GPU box: per-wave instruction mix and cycles of the step kernel of the in-tree library (or FJSP_AMD_LIB) at N envs.
# Usage: tools/pmc_step.sh <tag> [N] [kernel substring]
TAG=${1:-run}; N=${2:-4096}; KN=${3:-step_kernel}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcstep_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d $OUT/a -- python3 $ROOT/tools/run_steps.py $N > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 $ROOT/tools/run_steps.py $N > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python3 $ROOT/tools/run_steps.py $N > /dev/null 2>&1
python3 - "$OUT" "$KN" <<'PY' | tee $OUT/summary.txt
import collections, csv, glob, sys
out, kn = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/[ab]/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kn in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1.0)
print("waves per launch", w)
print("per wave:", {k: round(v / w, 1) for k, v in sorted(m.items()) if k != "SQ_WAVES"})
for f in glob.glob(out + "/st/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if kn in row["Name"]:
            print(row["Name"].split("(")[0], "calls", row["Calls"], "avg_ns", row["AverageNs"], "min", row["MinNs"], "max", row["MaxNs"])
PY

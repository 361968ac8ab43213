#!/bin/bash
# DIAGNOSTIC (GPU box): per-wave instruction counts and kernel durations of step_kernel for the ablation
# builds of tools/ablate_step.py (levels in $LEVELS), to attribute a step's instructions to its phases.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LEVELS=${LEVELS:-"4 3 7 8 9 2 0"}
mkdir -p $ROOT/gpurun_out/ablpmc
cd $ROOT && python3 tools/ablate_step.py --build-only $(for l in $LEVELS; do echo -n "--level=$l "; done)
cd /tmp && export TMPDIR=/tmp
for l in $LEVELS; do
  export FJSP_AMD_LIB=$ROOT/gpurun_out/libfjsp_ablate$l.so
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $ROOT/gpurun_out/ablpmc/pmc$l -- python3 $ROOT/tools/run_steps.py 4096 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ablpmc/st$l -- python3 $ROOT/tools/run_steps.py 4096 > /dev/null 2>&1
  echo "level $l: $(python3 $ROOT/tools/pmc_means.py $ROOT/gpurun_out/ablpmc/pmc$l step_kernel 4096) $(python3 $ROOT/tools/kstat.py $ROOT/gpurun_out/ablpmc/st$l step_kernel)"
  rm -f $FJSP_AMD_LIB
done

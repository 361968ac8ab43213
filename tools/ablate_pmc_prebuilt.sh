#!/bin/bash
# DIAGNOSTIC (GPU box): like ablate_pmc.sh, for ablation libraries built beforehand into .diag/
# (tools/build_diag.sh, in the build container: the box's minutes go to measuring, not compiling).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LEVELS=${LEVELS:-"4 3 7 8 9 2 1 0"}
OUT=$ROOT/gpurun_out/ablpmc_${TAG:-run}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for l in $LEVELS; do
  if [ "$l" = "0" ]; then unset FJSP_AMD_LIB; else export FJSP_AMD_LIB=$ROOT/.diag/libfjsp_ablate$l.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS SQ_WAIT_ANY SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc$l -- python3 $ROOT/tools/run_steps.py 4096 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$l -- python3 $ROOT/tools/run_steps.py 4096 > /dev/null 2>&1
  echo "level $l: $(python3 $ROOT/tools/pmc_means.py $OUT/pmc$l step_kernel 4096) $(python3 $ROOT/tools/kstat.py $OUT/st$l step_kernel)" | tee -a $OUT/summary.txt
done

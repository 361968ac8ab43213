#!/bin/bash
# GPU box: PMC + kernel time of the group step kernel per ablation level (libraries from tools/build_diag_group.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-run}; N=${2:-4096}
for l in ${LEVELS:-1 2 3 4 5 0}; do
  if [ "$l" = "0" ]; then unset FJSP_AMD_LIB; else export FJSP_AMD_LIB=$ROOT/.diag/libfjsp_gablate$l.so; fi
  echo "level $l: $(bash $ROOT/tools/pmc_step.sh ${TAG}_l$l $N step_kernel | tail -2 | tr '\n' ' ')"
done | tee $ROOT/gpurun_out/ablate_group_$TAG.txt

#!/bin/bash
# Build container: the ablation libraries of tools/ablate_step.py (-DFJSP_ABLATE=n) into .diag/ so that the GPU box's
# minutes go to measuring, not compiling (tools/ablate_pmc_prebuilt.sh reads them there).  .diag/ is git-ignored and
# listed in .gpurunignore: remove its line there for the one gpurun call that needs the libraries.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/deep_reinforcement_learning_for_fjsp_amd/csrc
LEVELS=${LEVELS:-"1 2 3 4 7 8 9"}
mkdir -p $ROOT/.diag
for l in $LEVELS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DFJSP_ABLATE=$l -Wno-unused-function \
    -I $ROOT/include -I $CSRC $CSRC/fjsp_kernels.hip $CSRC/fjsp_env.hip $CSRC/fjsp_rollout_buffer.hip $CSRC/fjsp_ppo.hip $CSRC/fjsp_mlp_train.hip $CSRC/fjsp_policy_mlp.hip $CSRC/fjsp_instance.cpp $CSRC/fjsp_lp.cpp \
    -o $ROOT/.diag/libfjsp_ablate$l.so -lpthread &
done
wait
ls -la $ROOT/.diag/

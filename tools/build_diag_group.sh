#!/bin/bash
# Build container: ablation libraries of the group step kernel (-DFJSP_GABLATE=n, fjsp_group.hip g_step) into .diag/:
# 1 state in / state out | 2 + update_parameter + task_select | 3 + machine_select | 4 + dispatch + event loop |
# 5 + second update_parameter + integer statistics + reward (no observation) | (0 = the product library)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/deep_reinforcement_learning_for_fjsp_amd/csrc
mkdir -p $ROOT/.diag
for l in ${LEVELS:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DFJSP_GABLATE=$l -Wno-unused-function \
    -I $ROOT/include -I $CSRC $CSRC/fjsp_kernels.hip $CSRC/fjsp_group.hip $CSRC/fjsp_lp_device.hip $CSRC/fjsp_env.hip $CSRC/fjsp_rollout_buffer.hip $CSRC/fjsp_ppo.hip $CSRC/fjsp_mlp_train.hip $CSRC/fjsp_policy_mlp.hip \
    $CSRC/fjsp_instance.cpp $CSRC/fjsp_lp.cpp -o $ROOT/.diag/libfjsp_gablate$l.so -lpthread &
done
wait

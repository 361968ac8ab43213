#!/bin/bash
# VGPRs / scratch of every kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage); exits 1 if any kernel
# uses scratch: the wave state must stay in registers (see the note at pick<KC>() in csrc/fjsp_kernels.hip).
# Exception: rollout_policy_kernel (16 environments + the actor's weights per workgroup: 1024 threads, 128 VGPRs) spills
# a few dozen bytes of SGPR-overflow lanes; measured cost nil next to its ~2 000 instructions per step.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/deep_reinforcement_learning_for_fjsp_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../include -I . -c fjsp_kernels.hip -o /tmp/fjsp_kernels_check.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize" | paste - - - | \
    sed 's/.*Function Name: //' | awk '{print $1, "VGPRs", $6, "scratch", $(NF-1)}' | tee /tmp/fjsp_resources.txt | grep -v "scratch 0" | grep -v "rollout_policy_kernel" && { echo "kernels with scratch above"; exit 1; }
echo "$(wc -l < /tmp/fjsp_resources.txt) kernels, none uses scratch"

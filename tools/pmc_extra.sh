#!/bin/bash
# DIAGNOSTIC (GPU box): further per-wave counters of step_kernel (branches, total instructions, scalar / vector /
# LDS issue-active cycles, instruction-fetch wait) for the library FJSP_AMD_LIB points at (default: the in-tree one).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcx_${TAG:-run}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 $ROOT/tools/run_steps.py 4096 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 --output-format csv -d $OUT/b -- python3 $ROOT/tools/run_steps.py 4096 > /dev/null 2>&1
echo "a: $(python3 $ROOT/tools/pmc_means.py $OUT/a step_kernel 4096)" | tee $OUT/summary.txt
echo "b: $(python3 $ROOT/tools/pmc_means.py $OUT/b step_kernel 4096)" | tee -a $OUT/summary.txt

#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + PMC passes of bench.py, summaries into gpurun_out/prof_<tag>/.
# Usage: tools/profile_bench.sh <tag>
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/summarize_prof.py $OUT

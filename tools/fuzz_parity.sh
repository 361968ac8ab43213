#!/bin/bash
# GPU box: sweep the randomised differential test (HIP kernels vs the C oracle) over more random draws.
# Usage: tools/fuzz_parity.sh <first_seed> <last_seed>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
fail=0
for seed in $(seq ${1:-1} ${2:-10}); do
  if FJSP_FUZZ_SEED=$seed python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k randomised > gpurun_out/fuzz_$seed.log 2>&1; then
    echo "seed $seed ok: $(tail -1 gpurun_out/fuzz_$seed.log)"
  else
    echo "seed $seed FAILED"; tail -25 gpurun_out/fuzz_$seed.log; fail=1
  fi
done
exit $fail

#!/bin/bash
# GPU box: every measurement DESIGN.md quotes, into gpurun_out/prof_<tag>/ (copied into profiles/ by tools/collect_profiles.py).
#   headline bench + rocprofv3 kernel stats + PMC passes (tools/profile_bench.sh), the same at 32 768 / 262 144 envs per GPU with
#   kernel stats, FETCH_SIZE calibration, the dynamic environment (blocking and asynchronous arrival service), the reference's
#   training distribution, a PPO training run with the kernel stats of its rounds.
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
bash $ROOT/tools/profile_bench.sh $TAG
cd /tmp && export TMPDIR=/tmp
for n in 32768 262144; do
  python3 $ROOT/bench.py --envs-per-gpu $n --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_envs$n.json 2> $OUT/bench_envs$n.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_envs$n -- python3 $ROOT/bench.py --envs-per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/trace_envs$n.err
  python3 $ROOT/tools/kstat.py $OUT/trace_envs$n kernel > $OUT/kernel_stats_envs$n.txt
done
bash $ROOT/tools/fetch_calib.sh > $OUT/fetch_calib.txt 2>&1 && cp $ROOT/gpurun_out/fetch_calib/fetch_calibration.json $OUT/
python3 $ROOT/tools/bench_dynamic.py --steps 1000 --blocking > $OUT/bench_mo_dfjsp_blocking.json 2> $OUT/bench_mo_dfjsp_blocking.err
python3 $ROOT/tools/bench_dynamic.py --steps 1000 --cpu-seconds 0 > $OUT/bench_mo_dfjsp_async.json 2> $OUT/bench_mo_dfjsp_async.err
python3 $ROOT/tools/bench_training_dist.py > $OUT/bench_training_distribution.json 2> $OUT/bench_training_distribution.err
python3 $ROOT/examples/train_ppo.py --rounds 10 > $OUT/train_ppo.json 2> $OUT/train_ppo.err
python3 $ROOT/examples/train_ppo.py --rounds 10 --per-step-rollout > $OUT/train_ppo_per_step_rollout.json 2> $OUT/train_ppo_per_step_rollout.err
python3 $ROOT/tools/time_ppo_round.py > $OUT/ppo_round_split.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ppo -- python3 $ROOT/tools/time_ppo_round.py > /dev/null 2> $OUT/trace_ppo.err
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace_ppo/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[1] + "/ppo_round_kernel_stats.csv", "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows[:60])
PY
python3 $ROOT/tools/host_overhead.py 8 > $OUT/host_overhead.txt 2>&1
python3 $ROOT/examples/train_hmpsac.py --envs 4096 --lower-envs 256 --lower-rounds 2 --epochs 1 > $OUT/train_hmpsac.json 2> $OUT/train_hmpsac.err
python3 $ROOT/tools/time_hmpsac_epoch.py > $OUT/hmpsac_epoch_split.txt 2>&1
python3 $ROOT/tools/time_mlp_pass.py > $OUT/mlp_pass_timing.txt 2>&1
echo done

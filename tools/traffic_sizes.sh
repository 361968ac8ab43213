#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE (separate --pmc passes, program directly after `--`) of the step kernel at several batch
# sizes, with the raw TCC request counters beside them, into gpurun_out/traffic_<tag>/summary.json.
# Usage: tools/traffic_sizes.sh <tag> [sizes...]
set -e
TAG=${1:-run}; shift || true
SIZES=${@:-"4096 32768 262144"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
for n in $SIZES; do
  st=100; [ $n -ge 100000 ] && st=20
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$n -- python3 $ROOT/bench.py --envs-per-gpu $n --steps $st --warmup 10 --no-cpu-baseline > $OUT/bench_fetch_$n.json 2> $OUT/fetch_$n.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$n -- python3 $ROOT/bench.py --envs-per-gpu $n --steps $st --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/write_$n.err
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/req_$n -- python3 $ROOT/bench.py --envs-per-gpu $n --steps $st --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/req_$n.err || true
  echo "envs $n done"
done
python3 - "$OUT" $SIZES <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
res = {}
for n in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for tag in ("fetch", "write", "req"):
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, tag, n), recursive=True):
            for row in csv.DictReader(open(f)):
                if "fjsp::" in row["Kernel_Name"]:
                    short = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res[n] = {k: dict({c: sum(x) / len(x) for c, x in v.items()}, dispatches=len(next(iter(v.values())))) for k, v in acc.items()}
    try:
        res[n]["bench"] = json.loads(open("%s/bench_fetch_%s.json" % (out, n)).read().strip().splitlines()[-1])["roofline"]
    except Exception as e:
        res[n]["bench"] = str(e)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY

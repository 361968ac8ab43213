#!/bin/bash
# GPU box: FETCH_SIZE calibration for 4 / 8 / 16 B-per-lane row reads (tools/ubench/fetch_calib.hip, built into .diag/).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_calib
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- $ROOT/.diag/fetch_calib > $OUT/stdout.txt 2> $OUT/err.txt
cat $OUT/stdout.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = {}
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rows<" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
            out[row["Kernel_Name"].split("(")[0]] = float(row["Counter_Value"])
true = {"HIP_vector_type": 65536 * 16 * 1024, "uint4": 65536 * 16 * 1024, "unsigned long": 65536 * 16 * 512, "unsigned int": 65536 * 16 * 256}
res = {}
for k, v in out.items():
    t = next(tv for tk, tv in true.items() if tk in k)
    res[k] = {"FETCH_SIZE_KiB": v, "true_bytes": t, "ratio_reported_over_true": v * 1024 / t}
    print(k, res[k])
json.dump(res, open(sys.argv[1] + "/fetch_calibration.json", "w"), indent=1)
PY

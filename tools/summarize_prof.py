#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into two small files:
kernel_stats.csv (rocprofv3 --stats rows of our kernels) and pmc_means.json (per-dispatch means)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fjsp::" in row["Name"]:
            rows.append(row)
with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
pm = {}
for tag in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            if "fjsp::" in row["Kernel_Name"]:
                short = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            pm.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
            pm[k]["dispatches_" + tag] = len(next(iter(v.values())))
json.dump(pm, open(os.path.join(out, "pmc_means.json"), "w"), indent=1)
for r in rows:
    print(r["Name"].split("(")[0][-30:], r["Calls"], r["AverageNs"])
for k, sk in pm.items():
    if "step_kernel" in k:
        print(k, "FETCH_SIZE KiB", sk.get("FETCH_SIZE"), "WRITE_SIZE KiB", sk.get("WRITE_SIZE"))

#!/usr/bin/env python3
"""Per-dispatch means of rocprofv3 --pmc counters for kernels matching a substring, divided by a wave count."""
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
div = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
print({k: round(sum(v) / len(v) / div, 1) for k, v in sorted(acc.items())})

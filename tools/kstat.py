#!/usr/bin/env python3
"""Print Calls / AverageNs / MinNs / MaxNs of kernels matching a substring from a rocprofv3 *_kernel_stats.csv tree."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Name"]:
            print(row["Name"].split("(")[0][-40:], "calls", row["Calls"], "avg_ns", row["AverageNs"], "min", row["MinNs"], "max", row["MaxNs"])

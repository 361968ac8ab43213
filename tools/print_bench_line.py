#!/usr/bin/env python3
"""One summary line of a bench.py JSON file (A/B runs).  Usage: tools/print_bench_line.py <label> <file.json>"""
import json
import sys

d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
f = d.get("fused") or {}
print(sys.argv[1], round(d["value"] / 1e6, 1), "M/s", round(d["roofline"]["launch_us_hip_events"], 3), "us frac", round(d["roofline"]["frac"], 3),
      "fused", round(f.get("env_steps_per_s", 0) / 1e6, 1), round(f.get("env_steps_per_s_no_state", 0) / 1e6, 1))

#!/usr/bin/env python3
"""Per-env-step read / write bytes of the step kernels from a tools/traffic_sizes.sh summary.  Usage: tools/print_traffic.py <summary.json>"""
import json
import sys

d = json.load(open(sys.argv[1]))
for n, v in d.items():
    for k, c in v.items():
        if "gstep" in k or "grollout" in k or "step_kernel<" in k:
            N = int(n)
            rd, wr = c["FETCH_SIZE"] * 1024 * 2 / N, c["WRITE_SIZE"] * 1024 / N
            print(n, k.split("(")[0][-40:], "read lines/env-step", round(c.get("TCC_EA0_RDREQ_sum", 0.0) / N, 2), "read B", round(rd, 1), "write B", round(wr, 1),
                  "traffic / 1966 B", round((rd + wr) / 1966.0, 3))

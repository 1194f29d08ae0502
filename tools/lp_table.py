#!/usr/bin/env python3
"""Per-kernel table of a rocprofv3 --kernel-trace --stats --output-format csv run of tools/latency_profile.py enc B L N:
  python tools/lp_table.py KERNEL_STATS_CSV FORWARDS     (FORWARDS = N + 3 warm-ups)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = 0.0
for r in rows:
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    per = calls / n * avg
    tot += per
    print(f"{r['Name'][:64]:64s} launches per forward {calls / n:6.2f}  avg {avg:7.2f} us  per forward {per:7.1f} us")
print(f"sum of kernel time per forward: {tot:.1f} us")

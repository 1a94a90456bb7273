#!/usr/bin/env python
"""Print a rocprofv3 kernel_stats.csv as a short table: python tools/kstats.py <csv> [rows] [passes]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
passes = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms ({tot / 1e6 / passes:.1f} ms per pass)")
for r in rows[:top]:
    t = int(r["TotalDurationNs"])
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:9.1f} us "
          f"per pass {t / 1e6 / passes:7.2f} ms {100 * t / tot:5.1f}%")

"""Top kernels of a rocprofv3 --stats kernel_stats.csv: python tools/stats_top.py <kernel_stats.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:8.1f} us = {int(r['TotalDurationNs']) / 1e6:7.2f} ms  {r['Name'][:120]}")
print("total", tot / 1e6, "ms")

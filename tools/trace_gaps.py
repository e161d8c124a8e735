"""Per-kernel durations and inter-kernel gaps from a rocprofv3 --kernel-trace CSV (in-graph replays): python tools/trace_gaps.py <kernel_trace.csv> [first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stat = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    key = (name, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
    st = stat[key]
    st[0] += 1; st[1] += (e - s) / 1e3
    if prev_end is not None and 0 <= s - prev_end < 50000: st[2] += (s - prev_end) / 1e3
    prev_end = e
tot = sum(v[1] + v[2] for v in stat.values())
print(f"total kernel+gap time {tot/1e3:.2f} ms over {len(rows)} dispatches")
for k, v in sorted(stat.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{v[0]:7d} x  dur {v[1]/v[0]:7.2f} us  gap-before {v[2]/v[0]:6.2f} us  share {(v[1]+v[2])/tot*100:5.1f}%  grid {k[1]:>8} wg {k[2]:>5}  {k[0]}")

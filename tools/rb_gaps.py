"""Idle gaps of the GPU in the steady state of a row-batched generate(): python tools/rb_gaps.py <kernel_trace.csv>.
Takes the last 60 % of the trace, merges the kernels' [start, end] intervals over all streams and lists the largest idle gaps with the
kernels on either side, plus busy / wall."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.4):]
t0 = int(rows[0]["Start_Timestamp"])
iv = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]) for r in rows]
gaps, busy, cur_end, last = [], 0, iv[0][0], iv[0][2]
for s, e, n in iv:
    if s > cur_end:
        gaps.append((s - cur_end, cur_end, last, n))
        busy += e - s
        cur_end, last = e, n
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end, last = e, n
wall = cur_end - iv[0][0]
print(f"{len(iv)} kernels, wall {wall / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / wall:.1f} %), idle {(wall - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
import collections
by = collections.Counter()
cnt = collections.Counter()
for g, at, a, b in gaps:
    by[(a, b)] += g; cnt[(a, b)] += 1
print("idle time by (kernel before -> kernel after):")
for (a, b), g in by.most_common(14):
    print(f"  {g / 1e3:9.1f} us in {cnt[(a, b)]:4d} gaps (avg {g / cnt[(a, b)] / 1e3:6.1f})  {a}  ->  {b}")

"""Per generate() call of tools/rb_run.py's kernel trace: wall time, hardware queues that carried conv-tail kernels, and the share of the
wall time during which kernels of two or more queues ran at once.  python tools/rb_modes.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sin" in r["Kernel_Name"]]
qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
print("columns:", [c for c in rows[0].keys()][:14])
for ci in range(len(marks) - 1):
    seg = rows[marks[ci] + 1: marks[ci + 1]]
    seg = seg[len(seg) // 3:]                      # steady state: skip voice encode + prefill + graph capture
    t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    conv_q = collections.Counter(r[qcol] for r in seg if "ffn_in" in r["Kernel_Name"] or "block1d" in r["Kernel_Name"])
    ev = []
    for r in seg:
        ev.append((int(r["Start_Timestamp"]), 1, r[qcol])); ev.append((int(r["End_Timestamp"]), -1, r[qcol]))
    ev.sort()
    active = collections.Counter()
    last, multi, busy = t0, 0, 0
    for t, d, q in ev:
        nq = sum(1 for v in active.values() if v > 0)
        if nq >= 1: busy += t - last
        if nq >= 2: multi += t - last
        last = t
        active[q] += d
    print(f"call {ci}: steady part {1e-6 * (t1 - t0):7.2f} ms, {len(seg)} kernels, busy {100 * busy / (t1 - t0):5.1f} %, >= 2 queues at once {100 * multi / (t1 - t0):5.1f} %, conv-tail kernels per queue {dict(conv_q)}")

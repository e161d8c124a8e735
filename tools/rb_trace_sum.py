"""Per-kernel summary of tools/rb_trace.py's kernel trace: the segments between the sin markers are 20 replays of graph A and 20 of graph H."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sin" in r["Kernel_Name"]]
marks = marks[-3:]
for name, (a, b) in zip(("graph A: LLM decode step, 8 rows + tail", "graph H: diffusion sampling, 8 rows"), ((marks[0], marks[1]), (marks[1], marks[2]))):
    seg = rows[a + 1: b]
    per = len(seg) // 20
    one = seg[-per * 10:]                     # the last 10 replays
    span = (int(one[-1]["End_Timestamp"]) - int(one[0]["Start_Timestamp"])) / 1e3 / 10
    acc = collections.OrderedDict()
    for r in one:
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""),
             r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        v = acc.setdefault(k, [0, 0.0]); v[0] += 1; v[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"== {name}: {per} kernels per replay, {span:.1f} us per replay (last 10 of 20)")
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {v[0] / 10:6.1f} x {v[1] / v[0]:7.2f} us = {v[1] / 10:8.1f} us  grid {k[1]:>8s} wg {k[2]:>4s}  {k[0]}")

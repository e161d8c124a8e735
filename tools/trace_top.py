"""Top kernels (by name + grid) of the last of N repetitions in a rocprofv3 kernel trace: python tools/trace_top.py <csv> <reps>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
mark = [i for i, n in enumerate(names) if "affine" in n or "copy" in n.lower()]
reps = int(sys.argv[2])
n = len(rows)
# last repetition = the trailing 1/reps of the kernels after the last torch RNG kernel
last = max(i for i, nm in enumerate(names) if "distribution" in nm or "normal" in nm)
rows = rows[last + 1:]
per = len(rows) // reps
one = rows[-per:]
acc = collections.OrderedDict()
for r in one:
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48], r["Grid_Size_X"], r["Grid_Size_Y"])
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("one repetition:", round((int(one[-1]["End_Timestamp"]) - int(one[0]["Start_Timestamp"])) / 1e3), "us,", len(one), "kernels")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"{v[0]:4d} x {v[1]/v[0]:8.1f} us = {v[1]:8.0f} us  grid {k[1]}x{k[2]}  {k[0]}")

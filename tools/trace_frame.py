"""Kernel-by-kernel timeline of the last graph replay in a rocprofv3 --kernel-trace CSV: python tools/trace_frame.py <csv> <replays>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
last_torch = max(i for i, n in enumerate(names) if "at::native" in n)
rows = rows[last_torch + 1:]
per = len(rows) // int(sys.argv[2])
one = rows[-per:]
t0 = int(one[0]["Start_Timestamp"])
print(f"one replay: {(int(one[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, {len(one)} kernels")
prev = t0
for r in one:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
    print(f"{(s - t0) / 1e3:8.1f}  gap {(s - prev) / 1e3:5.1f}  dur {(e - s) / 1e3:6.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}x{int(r['Grid_Size_Y']):<3d} wg {r['Workgroup_Size_X']:>4}  {n}")
    prev = e

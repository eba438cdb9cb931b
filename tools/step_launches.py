"""Durations of every launch of the kernels whose name contains a given substring inside ONE train step (between the last two adamw_flat_kernel
launches of a rocprofv3 kernel trace), in launch order, with the grid size: python tools/step_launches.py trace.csv substring"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
groups = [(a, b) for a, b in zip(marks[:-1], marks[1:]) if b - a > 10]
a, b = groups[-1]
tot = 0.0
for r in rows[a:b]:
    if sys.argv[2] in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        print("%9.1f us  grid %s x %s  wg %s" % (d, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X")))
print("total %.1f us" % tot)

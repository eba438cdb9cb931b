"""Median kernel duration per configuration of tools/conv_ablate.py from a rocprofv3 kernel trace (205 launches per config)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "conv_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
names = ["full", "no loop(32)", "no halo(4)", "no store(8)", "no halo,loop(36)", "no store,loop(40)", "no halo,store(12)", "skeleton(44)", "return(16)"]
for i, nm in enumerate(names):
    d = sorted(x[1] for x in rows[i * 205 + 5:(i + 1) * 205])
    if d:
        print(f"{nm:20s} median {d[len(d)//2]/1e3:7.2f} us  min {d[0]/1e3:7.2f} us")

"""Step period from a kernel trace: time between successive adamw_flat_kernel launches, and the GPU-busy share in between."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
for a, b in zip(marks[:-1], marks[1:]):
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b])
    print("period %.2f ms, %d dispatches, GPU busy %.2f ms" % ((t1 - t0) / 1e6, b - a, busy / 1e6))

"""Reads a rocprofv3 kernel trace CSV of tools/bench_conv.py and prints the median / min duration of every run of `reps`
consecutive launches of the conv kernels (the benchmark launches them in a fixed order)."""
import csv, statistics, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_ws" in r["Kernel_Name"] or "ksplit" in r["Kernel_Name"] or "igemm_kernel" in r["Kernel_Name"]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
labels = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for i in range(0, len(rows), reps):
    seg = rows[i:i + reps]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in seg]
    name = "ws" if "conv_ws" in seg[0]["Kernel_Name"] else ("ksplit" if "ksplit" in seg[0]["Kernel_Name"] else "igemm")
    lab = labels[i // reps] if labels and i // reps < len(labels) else ""
    print(f"{lab:28s} {name:7s} grid {seg[0]['Grid_Size_X']:>8s}x{seg[0]['Grid_Size_Y']} n={len(seg):3d} median {statistics.median(d):7.1f} us  min {min(d):7.1f} us")

"""Reads the kernel trace of `rocprofv3 --kernel-trace -- python tools/bench_ln.py 20` and prints the median DEVICE duration of each timed
group (the event timing of bench_ln.py contains the host's per-call cost, which exceeds the forward kernel).  python tools/ln_prof_read.py <dir> [reps]"""
import csv, glob, sys, statistics
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "layernorm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
names = ["forward", "backward", "backward + add", "backward, five gradients + add"]
d = d[1:]  # the first forward call (outputs for the backward)
per = 3 + reps
for i, n in enumerate(names):
    g = d[i * per + 3:(i + 1) * per]
    if g:
        print("%-32s median %6.1f us  min %6.1f  (%d launches)" % (n, statistics.median(g), min(g), len(g)))

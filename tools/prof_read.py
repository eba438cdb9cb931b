"""Median device duration per kernel name, in launch order groups, from a rocprofv3 kernel trace:  python tools/prof_read.py <dir> <name substring> [group size]
Launches of kernels whose name contains the substring are taken in stream order and cut into groups of `group size` (default 23 = the 3 warm-up +
20 timed calls of the tools/bench_*.py scripts); prints the median of each group without its first 3."""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 23
rows = sorted((r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "launches")
for i in range(0, len(d), per):
    g = d[i + 3:i + per]
    if g:
        print("group %2d: median %7.1f us  min %7.1f  (%d)" % (i // per, statistics.median(g), min(g), len(g)))

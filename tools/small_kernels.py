"""Where do the small torch kernels of a step come from?  For every dispatch shorter than `max_us` whose name contains one of the
given substrings, count (previous kernel, kernel, next kernel) triples over the last `window_ms` of a rocprofv3 kernel trace:
python tools/small_kernels.py trace.csv [window_ms] [max_us]"""
import csv, sys, collections

path = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 70.0
max_us = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - win * 1e6]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("at::native::", "")
    for a, b in (("vectorized_elementwise_kernel", "vec"), ("elementwise_kernel_manual_unroll", "unroll"), ("_ZN12_GLOBAL__N_1", "")):
        n = n.replace(a, b)
    return n[:70]


cnt = collections.Counter()
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"]
    if d <= max_us and ("at::native" in n or "rocclr" in n):
        p = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        q = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        cnt[(p, short(n), q)] += 1
for (p, n, q), c in cnt.most_common(60):
    print("%4d  %-70s <- %-50s -> %s" % (c, n, p[:50], q[:50]))

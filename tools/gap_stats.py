"""Idle time between consecutive kernels of the last `window_ms` of a rocprofv3 kernel trace: histogram of the gaps and the
(previous kernel -> next kernel) pairs that own most of the idle time.   python tools/gap_stats.py trace.csv [window_ms]"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 57.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = max(int(r["End_Timestamp"]) for r in rows)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - win * 1e6]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("at::native::", "").replace("_ZN12_GLOBAL__N_1", "")
    return n[:44]


hist = collections.Counter()
pairs = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
last_end = None
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None:
        g = max(0.0, (s - last_end) / 1e3)
        tot += g
        b = 0 if g < 1 else 1 if g < 2 else 2 if g < 4 else 3 if g < 8 else 4 if g < 16 else 5 if g < 50 else 6
        hist[b] += 1
        key = (short(prev), short(r["Kernel_Name"]))
        pairs[key][0] += 1
        pairs[key][1] += g
    last_end = max(last_end or 0, e)
    prev = r["Kernel_Name"]
print("total idle %.2f ms over %d gaps" % (tot / 1e3, len(rows) - 1))
names = ["<1", "1-2", "2-4", "4-8", "8-16", "16-50", ">50"]
print("gap us: " + "  ".join("%s: %d" % (names[b], hist[b]) for b in range(7)))
for (a, b), (c, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%7.1f us  x%4d  avg %5.1f   %-44s -> %s" % (g, c, g / c, a, b))

import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
groups = [(a, b) for a, b in zip(marks[:-1], marks[1:]) if b - a > 10]
a, b = groups[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for i in range(a, b):
    r = rows[i]
    n = r["Kernel_Name"]
    if "at::native" in n and "elementwise" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (n[:400], r.get("Grid_Size_X"))
        agg[key][0] += 1; agg[key][1] += d
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%4d x %8.1f us total  grid %s\n      %s" % (c, t, g, n))

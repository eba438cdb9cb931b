"""Which kernels run right before / after the launches of a given kernel?  python tools/kernel_context.py <kernel_trace.csv> <name substring>
(finds the origin of anonymous launches -- runtime copy / fill kernels -- from their neighbours in stream order)"""
import collections, csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
ctx = collections.Counter()
grid = collections.Counter()
for i, r in enumerate(rows):
    if pat in r["Kernel_Name"]:
        prev = rows[i - 1]["Kernel_Name"][:60] if i else "-"
        nxt = rows[i + 1]["Kernel_Name"][:60] if i + 1 < len(rows) else "-"
        ctx[(prev, nxt)] += 1
        grid[(r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")))] += 1
for (p, n), c in ctx.most_common(25):
    print("%5d  after %-60s before %s" % (c, p, n))
print("grids:", grid.most_common(8))

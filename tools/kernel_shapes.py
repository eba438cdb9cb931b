"""Launches of the LAST step in a rocprofv3 kernel trace of bench.py, grouped by (kernel name, grid size): count, average and total duration.
python tools/kernel_shapes.py <kernel_trace.csv> [name substring]   (the last step = the launches after the last adamw_flat_kernel but one)"""
import collections, csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
ad = [i for i, r in enumerate(rows) if "adamw_flat_kernel" in r["Kernel_Name"]]
ends = [i for k, i in enumerate(ad) if k + 1 == len(ad) or ad[k + 1] != i + 1]  # last launch of each step's optimizer (one launch per parameter group)
lo = ends[-2] + 1 if len(ends) >= 2 else 0
hi = ends[-1] + 1 if ends else len(rows)
agg = collections.defaultdict(list)
for r in rows[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    if pat in n:
        g = r.get("Grid_Size", "") or "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        w = r.get("Workgroup_Size", "") or r.get("Workgroup_Size_X", "")
        agg[(n[:90], g, w)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%4d x %8.1f us = %8.2f ms  grid %-14s wg %-5s %s" % (len(v), sum(v) / len(v), sum(v) / 1e3, g, w, n))

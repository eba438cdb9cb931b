"""Average PMC counter values per (kernel, grid) from a rocprofv3 --pmc counter_collection CSV."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in agg.items():
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} n={len(v):4d} avg {sum(v)/len(v):16.1f}")

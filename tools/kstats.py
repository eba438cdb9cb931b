"""Print per-kernel stats (calls, avg/min us, grid, LDS, VGPR) grouped by (name, grid) from a rocprofv3 kernel trace CSV."""
import collections, csv, re, sys
agg = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if len(sys.argv) > 2 and sys.argv[2] not in n:
            continue
        key = (n[:90], r["Grid_Size_X"], r["Grid_Size_Y"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c = agg.setdefault(key, [])
        c.append(d)
for k, v in agg.items():
    v2 = sorted(v)[len(v) // 10:]  # drop the fastest 10% / keep rest; report median and min
    print(f"calls {len(v):5d} med {sorted(v)[len(v)//2]/1e3:8.1f} us min {min(v)/1e3:8.1f} us  grid {k[1]:>7}x{k[2]:<3} lds {k[3]:>6} vgpr {k[4]:>3}+{k[5]:<3} {k[0]}")

"""Summarise a rocprofv3 --kernel-trace CSV over the LAST `window_ms` of kernel activity (= the last timed step of
bench.py), so MIOpen's first-call search kernels and warm-up do not pollute the per-kernel averages.
usage: python tools/prof_summary.py <kernel_trace.csv> <window_ms> [out.txt]"""
import collections
import csv
import re
import sys


def main():
    path, window_ms = sys.argv[1], float(sys.argv[2])
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"] + "x" + r["Grid_Size_Y"],
                         r["VGPR_Count"], r["Accum_VGPR_Count"]))
    rows.sort()
    t_end = rows[-1][1]
    win = [r for r in rows if r[0] >= t_end - int(window_ms * 1e6)]
    agg = collections.OrderedDict()
    for s, e, n, lds, v, a in win:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        key = (n[:140], lds, v, a)
        c = agg.setdefault(key, [0, 0, 1 << 62, 0])
        c[0] += 1
        c[1] += e - s
        c[2] = min(c[2], e - s)
        c[3] = max(c[3], e - s)
    tot = sum(v[1] for v in agg.values())
    print(f"# window {window_ms} ms before the last kernel end: {len(win)} dispatches, span {(win[-1][1] - win[0][0]) / 1e6:.2f} ms, "
          f"GPU busy {tot / 1e6:.2f} ms", file=out)
    print("# pct  calls  avg_us  min_us  max_us  grid(threads)  VGPR  AGPR  kernel   [one line per (kernel, grid)]", file=out)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{100 * v[1] / tot:6.2f} {v[0]:6d} {v[1] / v[0] / 1e3:9.1f} {v[2] / 1e3:8.1f} {v[3] / 1e3:8.1f} {k[1]:>10} {k[2]:>4} {k[3]:>4}  {k[0]}", file=out)


if __name__ == "__main__":
    main()

"""Kernel time of ONE train step (between the last two adamw_flat_kernel launches of a rocprofv3 kernel trace) by kernel name.
python tools/step_kernels.py trace.csv [N]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
groups = [(a, b) for a, b in zip(marks[:-1], marks[1:]) if b - a > 10]
a, b = groups[-1]
seg = rows[a:b]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    n = re.sub(r"\(anonymous namespace\)::|at::native::|_ZN12_GLOBAL__N_1\d*", "", r["Kernel_Name"])
    if "elementwise" in n or "Functor" in n:  # torch: keep the functor name
        m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|direct_copy\w*|\w+_kernel_impl\w*)<?(c10::BFloat16|float|double)?", n.split("<", 1)[1] if "<" in n else n)
        n = "torch " + (m.group(0) if m else n[:60]) + "  [%s threads]" % r.get("Grid_Size_X", "?")
    else:
        n = re.sub(r"<.*", "", n)[:48]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[n][0] += 1
    agg[n][1] += d
tot = sum(v[1] for v in agg.values())
period = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6
print("step: %d dispatches, kernel time %.2f ms, period %.2f ms" % (len(seg), tot / 1e3, period))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%6.2f%% %8.2f ms %5d x %7.1f us  %s" % (100 * t / tot, t / 1e3, c, t / c, n))

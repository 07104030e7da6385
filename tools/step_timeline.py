"""One step of the bench (hipGraph replay) as rocprofv3 saw it: python tools/step_timeline.py <kernel_trace.csv> [--full]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "sg_adam" in n]
# a step = three Adam launches (D, G, G); take a window in the middle of the run
k = (len(idx) // 2) // 3 * 3
while k + 3 < len(idx) and idx[k + 1] - idx[k] < idx[k + 2] - idx[k + 1]:      # align on the D update (the longest gap follows it)
    k += 1
start, end = idx[k] + 1, idx[k + 3] + 1
t0 = int(rows[start]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in rows[start:end]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "--full" in sys.argv:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d:6.1f} {n}")
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += d
tot = (int(rows[end - 1]["End_Timestamp"]) - t0) / 1e3
print(f"step: {end - start} kernels, {tot:.1f} us")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:8.1f} us {c:3d} x {d / c:6.1f}  {n}")

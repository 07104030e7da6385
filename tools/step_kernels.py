"""rocprofv3 kernel_stats.csv of `bench.py --steps K --warmup W --no_kernel_profile` -> launches and microseconds per training step, by kernel
(tuning instrument).  python tools/step_kernels.py <kernel_stats.csv> <steps incl. warm-up and capture steps>"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
agg = {}
for r in rows:
    k = short(r["Name"])[:58]
    a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot_c = sum(a[0] for a in agg.values()); tot_t = sum(a[1] for a in agg.values())
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:58s} {c / n:6.1f} launches/step {t / c / 1e3:7.1f} us avg {t / n / 1e3:8.1f} us/step {100 * t / tot_t:5.1f}%")
print(f"TOTAL {tot_c / n:.1f} launches/step, {tot_t / n / 1e3:.1f} us of kernel time per step")

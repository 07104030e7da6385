"""gpurun_out/r01/* (rocprofv3 CSVs) -> profiles/r01_* (small, committed)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pat):
    fs = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
    return fs[-1] if fs else None

for tag, sub in (("bench", "stats"), ("bench_eager_1stream", "stats_eager")):
    f = newest(f"{sub}/runc/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{R}_{tag}_kernel_stats.csv"))
for name in ("bench.json", "bench_eager.json", "bench_skip.json", "bench_nug1.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{R}_{name}"))

def pmc(counter):
    f = newest(f"pmc_{counter}/runc/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not f:
        return agg
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
    return agg

fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
out = {}
for n in fetch:
    f_kb = fetch[n][1] / fetch[n][0]
    w_kb = write[n][1] / write[n][0] if n in write and write[n][0] else 0.0
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950 -> x2; WRITE_SIZE exact
    out[n] = {"launches": fetch[n][0], "FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
              "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024)}
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/prof_step.py --steps 2 --no_d_streams",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB (FETCH_SIZE under-reports 16B/lane streams by 2x on gfx950)",
           "kernels": out}, open(os.path.join(dst, f"{R}_pmc_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))

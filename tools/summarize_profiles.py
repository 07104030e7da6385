"""gpurun_out/<round>/* (bench lines + rocprofv3 CSVs) -> profiles/<round>_* (small, committed).   python tools/summarize_profiles.py r02"""
import collections, csv, glob, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = os.path.join(ROOT, "gpurun_out", R), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def newest(pat):
    fs = sorted(glob.glob(os.path.join(src, pat)), key=os.path.getmtime)
    return fs[-1] if fs else None


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # noqa: E402


for tag, sub in (("bench", "stats"), ("bench_eager_1stream", "stats_eager"), ("bench_cgan_eager_1stream", "stats_cgan_eager"),
                 ("bench_twostage_eager_1stream", "stats_twostage_eager")):
    f = newest(f"{sub}/*/*_kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{R}_{tag}_kernel_stats.csv"))
for name in ("bench.json", "bench_eager.json", "bench_skip.json", "bench_nug1.json", "bench_cgan.json", "bench_twostage.json"):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{R}_{name}"))


def pmc(sub, counter):
    f = newest(f"{sub}/*/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not f:
        return agg
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = short(r["Kernel_Name"])
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
    return agg


fetch, write = pmc("pmc_FETCH_SIZE", "FETCH_SIZE"), pmc("pmc_WRITE_SIZE", "WRITE_SIZE")
out = {}
for n in fetch:
    f_kb = fetch[n][1] / fetch[n][0]
    w_kb = write[n][1] / write[n][0] if n in write and write[n][0] else 0.0
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950 -> x2; WRITE_SIZE exact
    out[n] = {"launches": fetch[n][0], "FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
              "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024)}
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/prof_step.py --steps 2 --no_d_streams",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB (FETCH_SIZE under-reports 16B/lane streams by 2x on gfx950; "
                         "Infinity-Cache hits are counted, MI355X_MICROARCH.md 'HBM')",
           "kernels": out}, open(os.path.join(dst, f"{R}_pmc_traffic.json"), "w"), indent=1)
busy, tot = pmc("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES"), pmc("pmc_mfma", "SQ_BUSY_CYCLES")
mf = {n: {"launches": busy[n][0], "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": round(busy[n][1] / busy[n][0]),
          "SQ_BUSY_CYCLES_per_launch": round(tot[n][1] / tot[n][0]) if n in tot and tot[n][0] else None,
          "mfma_busy_over_sq_busy": round(busy[n][1] / tot[n][1], 4) if n in tot and tot[n][1] else None} for n in busy}
if mf:
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python tools/prof_step.py --steps 2 --no_d_streams",
               "note": "raw counter sums over all SEs/XCDs as rocprofv3 reports them; use as a ratio between kernels",
               "kernels": mf}, open(os.path.join(dst, f"{R}_pmc_mfma.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))

"""Kernel-only timing (hipGraph replay of back-to-back launches) of the conv forward / backward-data / backward-weight launches
of the fcgan step, per arithmetic mode and tile shape.  Tuning instrument for sgan_igemm3.hip / the split-bf16 wgrad.

    python tools/bench_igemm3.py [fwd|dgrad|wgrad ...] [--tiles auto,64x64,128x64,128x128] [--only D3x6,G2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from supervised_gan_amd import _lib, ops  # noqa: E402
from hip_utils import derived_copies  # noqa: E402

# name, kind, k, s, p, cin, cout, [input sizes of the grouped problems]
D_SIZES = {1: (257, 129, 65), 2: (129, 65, 33), 3: (65, 33, 17)}
LAUNCHES = [
    ("D1x6", "conv", 4, 2, 2, 32, 64, [257, 129, 65] * 2),
    ("D2x6", "conv", 4, 2, 2, 64, 128, [129, 65, 33] * 2),
    ("D3x6", "conv", 4, 1, 2, 128, 256, [65, 33, 17] * 2),
    ("D3x3", "conv", 4, 1, 2, 128, 256, [65, 33, 17]),
    ("D2x3", "conv", 4, 2, 2, 64, 128, [129, 65, 33]),
    ("D1x3", "conv", 4, 2, 2, 32, 64, [257, 129, 65]),
    ("G1", "convT", 4, 2, 1, 256, 256, [16]),
    ("G2", "convT", 4, 2, 1, 256, 128, [32]),
    ("G3", "convT", 4, 2, 1, 128, 64, [64]),
    ("G4", "convT", 4, 2, 1, 64, 32, [128]),
    ("U512", "conv", 4, 2, 1, 512, 512, [32]),         # cgan unet_256 inner levels
    ("U256", "conv", 4, 2, 1, 256, 512, [64]),
    ("C64", "conv", 3, 1, 1, 64, 64, [512]),           # CRN 64 -> 64 @ 512^2
    ("D64", "conv", 4, 2, 2, 64, 128, [257] * 2),      # cgan D ndf 64
    ("Dc3", "conv", 4, 1, 2, 256, 512, [65] * 2),
]


def graph_time(fn, n=20, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ops", nargs="*", default=["fwd", "dgrad", "wgrad"])
    ap.add_argument("--tiles", default="auto")
    ap.add_argument("--only", default="")
    ap.add_argument("--modes", default="f32,bf16x3")
    a = ap.parse_args()
    lib = _lib.lib()
    only = [s for s in a.only.split(",") if s]
    tiles = a.tiles.split(",")
    print(f"{'launch':6s} {'op':5s} {'GFLOP':>7s} | " + " | ".join(f"{m + ':' + t:>16s}" for m in a.modes.split(",") for t in (tiles if m == "bf16x3" else ["-"])))
    for name, kind, k, s, p, cin, cout, sizes in LAUNCHES:
        if only and name not in only:
            continue
        tr = kind == "convT"
        w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
        wm, wt = derived_copies(w, k, cout, cin)
        b = torch.randn(cout, device="cuda")
        jobs_f, jobs_d, jobs_w, keep = [], [], [], []
        gf = 0.0
        for H in sizes:
            Ho = (H - 1) * s - 2 * p + k if tr else (H + 2 * p - k) // s + 1
            x = torch.randn(H, H, cin, device="cuda")
            y = torch.empty(Ho, Ho, cout, device="cuda")
            r = torch.randn(Ho, Ho, cout, device="cuda")
            dx = torch.empty(H, H, cin, device="cuda")
            dw = torch.zeros_like(w)
            db = torch.zeros(cout, device="cuda")
            st_in = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
            st_in[cin:] = H * H
            st_out = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
            sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
            nrm = ops.norm_desc(st_in, None, None, H * H, 1e-5, 2, 0.2)
            desc = ops.conv_desc(1 if tr else 0, k, s, p, H, H, cin, Ho, Ho, cout)
            gf += 2.0 * (H * H if tr else Ho * Ho) * cin * cout * k * k / 1e9
            jobs_f.append((desc, x, nrm, wm, b, y, st_out))
            jobs_d.append((desc, r, wt, dx, x, nrm, sums, 0, False, True))
            jobs_w.append((desc, x, nrm, r, dw, db))
            keep.append((x, y, r, dx, dw, db, st_in, st_out, sums, nrm, desc))
        fns = {"fwd": lambda: ops.conv_fwd_grouped(jobs_f), "dgrad": lambda: ops.conv_dgrad_grouped(jobs_d),
               "wgrad": lambda: ops.conv_wgrad_grouped(jobs_w)}
        for op in a.ops:
            cells = []
            for m in a.modes.split(","):
                ops.set_math(m)
                for t in (tiles if m == "bf16x3" else ["-"]):
                    os.environ.pop("SGAN_TILE3", None)
                    os.environ.pop("SGAN_IGEMM3P", None)
                    if t == "patch":
                        os.environ["SGAN_IGEMM3P"] = "1"
                    elif t not in ("auto", "-"):
                        os.environ["SGAN_TILE3"] = t
                        os.environ["SGAN_IGEMM3P"] = "0"
                    us = graph_time(fns[op])
                    kn = lib.sgan_last_kernel().decode().replace("sg_", "").replace("_kernel", "")
                    cells.append(f"{us:7.1f} {gf / us * 1e3:5.0f}TF")
                    if t in ("auto", "-", "patch"):
                        cells[-1] += f" {kn[-14:]}"
            print(f"{name:6s} {op:5s} {gf:7.3f} | " + " | ".join(cells), flush=True)
    os.environ.pop("SGAN_TILE3", None)
    os.environ.pop("SGAN_IGEMM3P", None)


main()

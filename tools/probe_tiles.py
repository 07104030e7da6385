"""Forward of the deep PatchGAN layers on the patch kernel (64 x 64 tile) against sg_igemm3_kernel with larger tiles (tuning instrument).
Run once per configuration: SGAN_IGEMM3P=0 SGAN_TILE3=128x64 python tools/probe_tiles.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from supervised_gan_amd import ops, _lib
from hip_utils import derived_copies
from probe_head import timeit
ops.set_math("bf16x3")
for (cin, cout, s, sizes) in [(128, 256, 1, [65, 33, 17] * 2), (128, 256, 1, [65, 33, 17]), (64, 128, 2, [129, 65, 33] * 2)]:
    k, p = 4, 2
    w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
    wm, wt = derived_copies(w, k, cout, cin)
    b = torch.randn(cout, device="cuda")
    jobs, keep = [], []
    for H in sizes:
        Ho = (H + 2 * p - k) // s + 1
        desc = ops.conv_desc(0, k, s, p, H, H, cin, Ho, Ho, cout)
        x = torch.randn(H, H, cin, device="cuda"); y = torch.empty(Ho, Ho, cout, device="cuda")
        st = torch.zeros(2 * cin, dtype=torch.float64, device="cuda"); st[cin:] = H * H
        so = ops.stat_arena(2 * cout, "cuda")
        nd = ops.norm_desc(st, None, None, H * H, 1e-5, 2, 0.2)
        jobs.append((desc, x, nd, wm, b, y, so, 0, ops.stat_rep(so))); keep.append((x, y, st, so, nd, desc))
    gf = sum(2.0 * j[0].Hout * j[0].Wout * cin * cout * 16 for j in jobs) / 1e9
    t = timeit(lambda: ops.conv_fwd_grouped(jobs))
    print(f"{cin}->{cout} s{s} n={len(sizes)} {gf:.2f} GF forward {t:.1f} us ({gf / t * 1e-3 * 1e3:.0f} TF useful) {_lib.lib().sgan_last_kernel().decode()}  [{os.environ.get('SGAN_IGEMM3P', '')} {os.environ.get('SGAN_TILE3', '')}]")

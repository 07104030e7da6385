"""The two halves of a fused backward launch apart and together (tuning instrument): D 128 -> 256 k4 s1 (six problems) and 64 -> 128 k4 s2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from supervised_gan_amd import ops, _lib
from hip_utils import derived_copies
from bench_thin import timeit
ops.set_math("bf16x3")
CASES = [(128, 256, 1, [65, 33, 17] * 2, True), (64, 128, 2, [129, 65, 33] * 2, True), (128, 256, 1, [65, 33, 17], True),
         (32, 64, 2, [257, 129, 65] * 2, False), (32, 64, 2, [257, 129, 65], False), (64, 128, 2, [129, 65, 33], True)]
for (cin, cout, s, sizes, normed) in CASES:
    sizes = sorted(sizes, reverse=True)      # as chain.multi_forward orders a group: largest problem first
    k, p = 4, 2
    w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
    wm, wt = derived_copies(w, k, cout, cin)
    dj, wj, keep = [], [], []
    dw = torch.zeros_like(w); db = torch.zeros(cout, device="cuda")
    for H in sizes:
        Ho = (H + 2 * p - k) // s + 1
        desc = ops.conv_desc(0, k, s, p, H, H, cin, Ho, Ho, cout)
        x = torch.randn(H, H, cin, device="cuda"); dy = torch.randn(Ho, Ho, cout, device="cuda"); din = torch.empty(H, H, cin, device="cuda")
        st = torch.zeros(2 * cin, dtype=torch.float64, device="cuda"); st[cin:] = H * H
        sums = ops.stat_arena(2 * cin, "cuda") if normed else None
        nd = ops.norm_desc(st if normed else None, None, None, H * H, 1e-5, 2, 0.2)
        dj.append((desc, dy, wt, din, x, nd, sums, 0, False, True, ops.stat_rep(sums) if normed else 0))
        wj.append((desc, x, nd, dy, dw, db))
        keep.append((x, dy, din, st, sums, nd, desc))
    gf = sum(2.0 * d[0].Hout * d[0].Wout * cin * cout * 16 for d in dj) / 1e9
    td = timeit(lambda: ops.conv_dgrad_grouped(dj)); kd = _lib.lib().sgan_last_kernel().decode()
    tw = timeit(lambda: ops.conv_wgrad_grouped(wj)); kw = _lib.lib().sgan_last_kernel().decode()
    tf = timeit(lambda: ops.conv_bwd_grouped(dj, wj)); kf = _lib.lib().sgan_last_kernel().decode()
    print(f"{cin}->{cout} s{s} n={len(sizes)} {gf:.2f} GF per half: dgrad {td:.1f} us ({gf / td * 1e-3:.0f} TF, {kd}) | wgrad {tw:.1f} us ({gf / tw * 1e-3:.0f} TF, {kw}) | fused {tf:.1f} us ({2 * gf / tf * 1e-3:.0f} TF, {kf})")

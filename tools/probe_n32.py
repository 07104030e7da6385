"""Backward-data of the second PatchGAN layer (64 -> 32 channels, 4 phases of 2 x 2 taps): 128 x 32 tile vs the patch kernel (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from supervised_gan_amd import ops, _lib
from hip_utils import derived_copies
from bench_thin import timeit
ops.set_math("bf16x3")
cin, cout, k = 32, 64, 4
w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
wm, wt = derived_copies(w, k, cout, cin)
jobs = []
for H in [257, 129, 65] * 2:
    Ho = (H + 4 - 4) // 2 + 1
    desc = ops.conv_desc(0, 4, 2, 2, H, H, cin, Ho, Ho, cout)
    dy = torch.randn(Ho, Ho, cout, device="cuda"); din = torch.empty(H, H, cin, device="cuda"); x = torch.randn(H, H, cin, device="cuda")
    nd = ops.norm_desc(None, None, None, H * H, 0.0, 2, 0.2)
    jobs.append((desc, dy, wt, din, x, nd, None, 0, False, True, 0))
for force in ("0", "1"):
    os.environ["SGAN_IGEMM3P"] = force
    t = timeit(lambda: ops.conv_dgrad_grouped(jobs))
    print(f"SGAN_IGEMM3P={force}: {t:.1f} us  {_lib.lib().sgan_last_kernel().decode()}")

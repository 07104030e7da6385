"""Run ONE grouped conv launch a few times, eagerly (for rocprofv3 --pmc passes).
   python tools/pmc_one.py D3x6 fwd bf16x3 128x128 [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
name, op, mode, tile = sys.argv[1:5]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 5
if tile == "patch":
    os.environ["SGAN_IGEMM3P"] = "1"
elif tile != "auto":
    os.environ["SGAN_TILE3"] = tile
    os.environ["SGAN_IGEMM3P"] = "0"
import torch  # noqa: E402

from supervised_gan_amd import ops  # noqa: E402
from hip_utils import derived_copies  # noqa: E402

SH = {"D3x6": ("conv", 4, 1, 2, 128, 256, [65, 33, 17] * 2), "D3x3": ("conv", 4, 1, 2, 128, 256, [65, 33, 17]),
      "Dc3": ("conv", 4, 1, 2, 256, 512, [65] * 2), "C64": ("conv", 3, 1, 1, 64, 64, [512]), "G3": ("convT", 4, 2, 1, 128, 64, [64]),
      "D2x6": ("conv", 4, 2, 2, 64, 128, [129, 65, 33] * 2), "BIG": ("conv", 4, 1, 2, 128, 256, [257])}
kind, k, s, p, cin, cout, sizes = SH[name]
tr = kind == "convT"
ops.set_math(mode)
w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
wm, wt = derived_copies(w, k, cout, cin)
b = torch.randn(cout, device="cuda")
jf, jd, jw, keep = [], [], [], []
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
    jf.append((desc, x, nrm, wm, b, y, st_out))
    jd.append((desc, r, wt, dx, x, nrm, sums, 0, False, True))
    jw.append((desc, x, nrm, r, dw, db))
    keep.append((x, y, r, dx, dw, db, st_in, st_out, sums, nrm, desc))
fn = {"fwd": lambda: ops.conv_fwd_grouped(jf), "dgrad": lambda: ops.conv_dgrad_grouped(jd), "wgrad": lambda: ops.conv_wgrad_grouped(jw)}[op]
for _ in range(n):
    fn()
torch.cuda.synchronize()
print("done")

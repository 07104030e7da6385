"""Diagnostics: per-workgroup phase times of sg_igemm3p_kernel from a -DSG3P_STAMP build of the library (tools/stamp3p.sh)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from supervised_gan_amd import _lib, ops  # noqa: E402
from hip_utils import derived_copies  # noqa: E402

op, nprob = sys.argv[1], int(sys.argv[2])
k, s, p, cin, cout = 4, 1, 2, 128, 256
sizes = [65, 33, 17] * (nprob // 3)
os.environ["SGAN_IGEMM3P"] = "1"
ops.set_math("bf16x3")
w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
wm, wt = derived_copies(w, k, cout, cin)
b = torch.randn(cout, device="cuda")
jf, jd, keep = [], [], []
for H in sizes:
    Ho = H + 1
    x = torch.randn(H, H, cin, device="cuda"); y = torch.empty(Ho, Ho, cout, device="cuda"); r = torch.randn(Ho, Ho, cout, device="cuda")
    dx = torch.empty(H, H, cin, device="cuda")
    st_in = torch.zeros(2 * cin, dtype=torch.float64, device="cuda"); st_in[cin:] = H * H
    st_out = torch.zeros(2 * cout, dtype=torch.float64, device="cuda"); sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
    nrm = ops.norm_desc(st_in, None, None, H * H, 1e-5, 2, 0.2)
    desc = ops.conv_desc(0, k, s, p, H, H, cin, Ho, Ho, cout)
    jf.append((desc, x, nrm, wm, b, y, st_out)); jd.append((desc, r, wt, dx, x, nrm, sums, 0, False, True))
    keep.append((x, y, r, dx, st_in, st_out, sums, nrm, desc))
if op == "gfwd":      # the generator's second layer: ConvT k4 s2 p1 256 -> 256, 16x16 -> 32x32, BatchNorm + ReLU on load (64 workgroups)
    k, s, p, cin, cout, H = 4, 2, 1, 256, 256, 16
    w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
    wm, wt = derived_copies(w, k, cout, cin)
    x = torch.randn(H, H, cin, device="cuda"); y = torch.empty(2 * H, 2 * H, cout, device="cuda")
    st_in = ops.stat_arena(2 * cin, "cuda"); st_in[cin:] = H * H
    st_out = ops.stat_arena(2 * cout, "cuda")
    gam, bet = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")
    nrm = ops.norm_desc(st_in, gam, bet, H * H, 1e-5, 1, 0.0, 0, ops.stat_rep(st_in))
    desc = ops.conv_desc(1, k, s, p, H, H, cin, 2 * H, 2 * H, cout)
    jf = [(desc, x, nrm, wm, torch.randn(cout, device="cuda"), y, st_out, 0, ops.stat_rep(st_out))]
    keep.append((x, y, st_in, st_out, gam, bet, nrm, desc))
if op in ("s2fwd1", "s2fwd2"):      # the stride-2 PatchGAN layers (32 -> 64 from 257^2, 64 -> 128 from 129^2), parity-plane patch kernel
    k, s, p = 4, 2, 2
    cin, cout, sizes = (32, 64, [257, 129, 65]) if op == "s2fwd1" else (64, 128, [129, 65, 33])
    sizes = sizes * (nprob // 3)
    w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
    wm, wt = derived_copies(w, k, cout, cin)
    b = torch.randn(cout, device="cuda")
    jf = []
    for H in sizes:
        Ho = (H + 2 * p - k) // s + 1
        x = torch.randn(H, H, cin, device="cuda"); y = torch.empty(Ho, Ho, cout, device="cuda")
        st_in = torch.zeros(2 * cin, dtype=torch.float64, device="cuda"); st_in[cin:] = H * H
        st_out = ops.stat_arena(2 * cout, "cuda")
        nrm = ops.norm_desc(st_in, None, None, H * H, 1e-5, 2, 0.2)
        desc = ops.conv_desc(0, k, s, p, H, H, cin, Ho, Ho, cout)
        jf.append((desc, x, nrm, wm, b, y, st_out, 0, ops.stat_rep(st_out))); keep.append((x, y, st_in, st_out, nrm, desc))
if op == "d1dgrad":      # backward-data INTO the first PatchGAN layer's output (32 channels, LeakyReLU, no norm) on the patch kernel: SGAN_PATCH_N32=1
    k, s, p, cin, cout = 4, 2, 2, 32, 64
    sizes = [257, 129, 65] * (nprob // 3)
    w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
    wm, wt = derived_copies(w, k, cout, cin)
    jd = []
    for H in sizes:
        Ho = (H + 2 * p - k) // s + 1
        x = torch.randn(H, H, cin, device="cuda"); r = torch.randn(Ho, Ho, cout, device="cuda"); dx = torch.empty(H, H, cin, device="cuda")
        nrm = ops.norm_desc(None, None, None, H * H, 1e-5, 2, 0.2)
        desc = ops.conv_desc(0, k, s, p, H, H, cin, Ho, Ho, cout)
        jd.append((desc, r, wt, dx, x, nrm, None, 0, False, True)); keep.append((x, r, dx, nrm, desc))
fn = (lambda: ops.conv_fwd_grouped(jf)) if op in ("fwd", "gfwd", "s2fwd1", "s2fwd2") else (lambda: ops.conv_dgrad_grouped(jd))
for _ in range(5):
    fn()
torch.cuda.synchronize()
buf = np.zeros(8 * 4096, dtype=np.uint64)
lib = _lib.lib()
lib.sgan_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.sgan_debug_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4096, 8)
st = st[st[:, 0] > 0].astype(np.int64)
print(op, "workgroups stamped:", len(st))
d = np.diff(st[:, :5], axis=1)
names = ["setup (tap table, scale/shift, barrier)", "first patch + weight tile", "tap loop", "epilogue"]
for i, n in enumerate(names):
    print(f"  {n:42s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f} cycles")
if (st[:, 5] > 0).all():
    print(f"  of the set-up: kernel arguments + tile decode (up to the tap table) median {np.median(st[:, 5] - st[:, 0]):9.0f} cycles")
tot = st[:, 4] - st[:, 0]
print(f"  workgroup total median {np.median(tot):.0f} cycles")
rt0 = st[:, 6].min()
start = (st[:, 6] - rt0) / 100.0
end = (st[:, 7] - rt0) / 100.0
print(f"  start times (us): p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f};  end max {end.max():.1f};  WG duration us median {np.median(end - start):.1f}")
late = start > 3.0
print(f"  workgroups starting later than 3 us after the first: {late.sum()} (second round)")


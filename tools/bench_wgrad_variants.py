"""Backward-weight kernel per layer, with / without the normalise-on-load prologue (tuning instrument)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from supervised_gan_amd import _lib, ops  # noqa: E402
from supervised_gan_amd.ops import pad4  # noqa: E402

LAYERS = [("G1", "convT", 4, 2, 1, 256, 256, 16), ("G2", "convT", 4, 2, 1, 256, 128, 32), ("G3", "convT", 4, 2, 1, 128, 64, 64),
          ("G4", "convT", 4, 2, 1, 64, 32, 128), ("D0c1", "conv", 4, 2, 2, 32, 64, 257), ("D0c2", "conv", 4, 2, 2, 64, 128, 129),
          ("D0c3", "conv", 4, 1, 2, 128, 256, 65), ("BIGc3", "conv", 4, 1, 2, 128, 256, 257), ("BIGc2", "conv", 4, 2, 2, 64, 128, 513)]
if os.environ.get("ONLY"):
    LAYERS = [l for l in LAYERS if l[0] in os.environ["ONLY"].split(",")]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


lib = _lib.lib()
for name, kind, k, s, p, cin, cout, H in LAYERS:
    tr = kind == "convT"
    Ho = (H - 1) * s - 2 * p + k if tr else (H + 2 * p - k) // s + 1
    ci, co = pad4(cin), pad4(cout)
    x = torch.randn(H, H, ci, device="cuda")
    r = torch.randn(Ho, Ho, co, device="cuda")
    dw = torch.zeros(k * k * co * ci, device="cuda")
    db = torch.zeros(co, device="cuda")
    st_in = torch.zeros(2 * ci, dtype=torch.float64, device="cuda")
    st_in[ci:] = H * H
    nrm = ops.norm_desc(st_in, None, None, H * H, 1e-5, 2, 0.2)
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, H, ci, Ho, Ho, co)
    pix = H * H if tr else Ho * Ho
    gf = 2.0 * pix * cin * cout * k * k / 1e9
    row = []
    for pro in (True, False):
        t = timeit(lambda: ops.conv_wgrad(desc, x, nrm if pro else None, r, dw, db))
        row.append(f"pro={int(pro)}: {t:6.1f} us {gf / t * 1e3:5.1f} TF")
    print(f"{name:5s} {gf:6.3f} GF {lib.sgan_last_kernel().decode():32s} | " + " | ".join(row))

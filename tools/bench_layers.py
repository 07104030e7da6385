"""Per-layer timing of the conv kernels at the BASELINE fcgan shapes (tuning instrument)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from supervised_gan_amd import _lib, ops  # noqa: E402
from supervised_gan_amd.ops import pad4  # noqa: E402

LAYERS = [  # name, kind, k, s, p, cin, cout, H
    ("G0", "convT", 4, 2, 1, 8, 256, 8), ("G1", "convT", 4, 2, 1, 256, 256, 16), ("G2", "convT", 4, 2, 1, 256, 128, 32),
    ("G3", "convT", 4, 2, 1, 128, 64, 64), ("G4", "convT", 4, 2, 1, 64, 32, 128), ("G5", "convT", 4, 2, 1, 32, 2, 256),
    ("D0c0", "conv", 4, 2, 2, 2, 32, 512), ("D0c1", "conv", 4, 2, 2, 32, 64, 257), ("D0c2", "conv", 4, 2, 2, 64, 128, 129),
    ("D0c3", "conv", 4, 1, 2, 128, 256, 65), ("D0c4", "conv", 4, 1, 2, 256, 1, 66),
    ("D1c0", "conv", 4, 2, 2, 2, 32, 256), ("D1c1", "conv", 4, 2, 2, 32, 64, 129), ("D1c2", "conv", 4, 2, 2, 64, 128, 65),
    ("D1c3", "conv", 4, 1, 2, 128, 256, 33), ("D1c4", "conv", 4, 1, 2, 256, 1, 34),
    ("D2c0", "conv", 4, 2, 2, 2, 32, 128), ("D2c1", "conv", 4, 2, 2, 32, 64, 65), ("D2c2", "conv", 4, 2, 2, 64, 128, 33),
    ("D2c3", "conv", 4, 1, 2, 128, 256, 17), ("D2c4", "conv", 4, 1, 2, 256, 1, 18),
]


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


def main():
    only = sys.argv[1:]
    lib = _lib.lib()
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'layer':6s} {'GFLOP':>7s} | {'fwd us':>8s} {'TF':>6s} {'kernel':34s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
    for name, kind, k, s, p, cin, cout, H in LAYERS:
        if only and not any(name.startswith(o) for o in only):
            continue
        tr = kind == "convT"
        Ho = (H - 1) * s - 2 * p + k if tr else (H + 2 * p - k) // s + 1
        ci, co = pad4(cin), pad4(cout)
        x = torch.randn(H, H, ci, device="cuda")
        w = torch.randn(k * k * co * ci, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        y = torch.empty(Ho, Ho, co, device="cuda")
        r = torch.randn(Ho, Ho, co, device="cuda")
        dx = torch.empty(H, H, ci, device="cuda")
        dw = torch.zeros_like(w)
        db = torch.zeros(co, device="cuda")
        st_in = torch.zeros(2 * ci, dtype=torch.float64, device="cuda")
        st_in[ci:] = H * H
        st_out = torch.zeros(2 * co, dtype=torch.float64, device="cuda")
        sums = torch.zeros(2 * ci, dtype=torch.float64, device="cuda")
        nrm = ops.norm_desc(st_in, None, None, H * H, 1e-5, 2, 0.2) if cin > 8 else None
        desc = ops.conv_desc(1 if tr else 0, k, s, p, H, H, ci, Ho, Ho, co)
        pix = H * H if tr else Ho * Ho
        gf = 2.0 * pix * cin * cout * k * k / 1e9
        t_f = timeit(lambda: ops.conv_fwd(desc, x, nrm, w, b, y, 0, st_out if cout > 4 else None))
        kf = lib.sgan_last_kernel().decode()
        t_d = timeit(lambda: ops.conv_dgrad(desc, r, w, dx, x if nrm is not None else None, nrm, sums if nrm is not None else None))
        t_w = timeit(lambda: ops.conv_wgrad(desc, x, nrm, r, dw, db))
        tot["fwd"] += t_f
        tot["dgrad"] += t_d
        tot["wgrad"] += t_w
        print(f"{name:6s} {gf:7.3f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} {kf:34s} | {t_d:8.1f} {gf / t_d * 1e3:6.1f} | {t_w:8.1f} {gf / t_w * 1e3:6.1f}")
    print("totals us:", {k: round(v, 1) for k, v in tot.items()})


main()

"""Repeat-run probe of the fused backward launch: the same (backward-data, backward-weight) pair 300 times; the input gradient must
repeat bit for bit, the weight gradient (fp32 atomics) within 1e-5 relative L2 of the first run.  python tools/race_probe_fused.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from supervised_gan_amd import ops, _lib
from hip_utils import master_weight, pad_vec, stats_of, to_buf
ops.set_math("bf16x3")
g = torch.Generator().manual_seed(5)
bad = 0
for kind, k, s, p, cin, cout, H, W in (("conv", 3, 1, 1, 64, 64, 64, 64), ("conv", 4, 1, 2, 128, 256, 65, 65), ("conv", 4, 2, 2, 64, 128, 129, 129),
                                       ("convT", 4, 2, 1, 128, 64, 32, 32)):
    tr = kind == "convT"
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), generator=g) * 0.05
    wm = master_weight(w, tr)
    Ho, Wo = ((H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k) if tr else ((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, W, cin, Ho, Wo, cout)
    nd = ops.norm_desc(stats_of(x), None, None, H * W, 1e-5, 1, 0.0)
    xb, dy = to_buf(x), to_buf(torch.randn(1, cout, Ho, Wo, generator=g))
    first = None
    for it in range(300):
        din = torch.full((H, W, cin), float("nan"), device="cuda")
        sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
        dw, db = torch.zeros_like(wm), torch.zeros(pad_vec(torch.zeros(cout)).numel(), device="cuda")
        fused = ops.conv_bwd_grouped([(desc, dy, wm._sgan_wt, din, xb, nd, sums, 0, False, True, 0)], [(desc, xb, nd, dy, dw, db)])
        if first is None:
            first = (din.clone(), dw.clone(), db.clone(), sums.clone())
            name = _lib.lib().sgan_last_kernel().decode()
        else:
            e_w = float((dw - first[1]).norm() / first[1].norm())
            e_s = float((sums - first[3]).abs().max() / first[3].abs().max())
            if not torch.equal(din, first[0]) or e_w > 1e-5 or e_s > 1e-9:
                bad += 1
                print("run", it, "differs: din equal", torch.equal(din, first[0]), "dW rel L2", e_w, "sums", e_s)
    torch.cuda.synchronize()
    print(kind, cin, cout, H, W, "fused" if fused else "apart", name, "mismatching runs so far:", bad)
print("OK" if bad == 0 else "RACE")

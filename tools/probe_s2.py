"""Stride-2 patch kernel against sg_igemm3_kernel and fp64 on the cgan / U-Net shapes (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from supervised_gan_amd import ops, _lib
from hip_utils import master_weight, pad_vec, stats_of, to_buf, from_buf
ops.set_math("bf16x3")
g = torch.Generator().manual_seed(5)
for (cin, cout, H, W, p, norm, ld_extra) in [(64, 128, 257, 257, 2, False, 0), (64, 128, 256, 256, 1, False, 64), (128, 256, 128, 128, 1, True, 128), (32, 64, 257, 257, 2, False, 0), (128, 256, 129, 129, 2, True, 0)]:
    x = torch.randn(1, cin, H, W, generator=g) * 1.3 + 0.2
    w = torch.randn(cout, cin, 4, 4, generator=g) * 0.05
    b = torch.randn(cout, generator=g) * 0.1
    a = x.double()
    if norm:
        a = F.instance_norm(a, eps=1e-5)
    a = F.leaky_relu(a, 0.2)
    ref = F.conv2d(a, w.double(), b.double(), stride=2, padding=p)
    Ho, Wo = ref.shape[2:]
    desc = ops.conv_desc(0, 4, 2, p, H, W, cin, Ho, Wo, cout)
    xb_full = torch.zeros(H, W, cin + ld_extra, device="cuda"); xb_full[..., :cin] = to_buf(x)
    xb = xb_full[..., :cin]
    nd = ops.norm_desc(stats_of(x) if norm else None, None, None, H * W, 1e-5, 2, 0.2)
    wm, bb = master_weight(w, False), pad_vec(b)
    outs = {}
    for force in ("1", "0"):
        os.environ["SGAN_IGEMM3P"] = force
        ob_full = torch.full((Ho, Wo, cout + ld_extra), float("nan"), device="cuda"); ob = ob_full[..., :cout]
        ost = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
        ops.conv_fwd(desc, xb, nd, wm, bb, ob, 0, ost)
        torch.cuda.synchronize()
        outs[force] = (ob.clone(), ost.clone(), _lib.lib().sgan_last_kernel().decode())
    r = ref[0].permute(1, 2, 0)
    e1 = float((outs["1"][0].double().cpu() - r).abs().max() / r.abs().max())
    e0 = float((outs["0"][0].double().cpu() - r).abs().max() / r.abs().max())
    d = (outs["1"][0] - outs["0"][0]).abs()
    st_ref = torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])
    s1 = float((outs["1"][1].cpu() - st_ref).abs().max() / st_ref.abs().max()); s0 = float((outs["0"][1].cpu() - st_ref).abs().max() / st_ref.abs().max())
    print(f"{cin}->{cout} {H}x{W} p{p} norm={norm} ld+{ld_extra}: {outs['1'][2]} err {e1:.2e} stats {s1:.2e} | {outs['0'][2]} err {e0:.2e} stats {s0:.2e} | max diff {float(d.max()):.2e} at {tuple(int(v) for v in (d == d.max()).nonzero()[0])}")

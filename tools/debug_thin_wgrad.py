"""Thin wgrad kernel vs torch on a few shapes; prints the error pattern per (co, tap, ci)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from supervised_gan_amd import ops
from hip_utils import to_buf, from_master
from supervised_gan_amd.ops import pad4

def run(cin, cout, H, W, k=4, s=2, p=2):
    torch.manual_seed(0)
    x = torch.randn(1, cin, H, W)
    w = (torch.randn(cout, cin, k, k) * 0.1).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    Ho, Wo = y.shape[2:]
    desc = ops.conv_desc(0, k, s, p, H, W, pad4(cin), Ho, Wo, pad4(cout))
    dw = torch.zeros(k * k * pad4(cout) * pad4(cin), device="cuda")
    db = torch.zeros(pad4(cout), device="cuda")
    ops.conv_wgrad(desc, to_buf(x), None, to_buf(r), dw, db)
    torch.cuda.synchronize()
    got = from_master(dw, k, cin, cout, False)
    err = (got - w.grad).abs()
    print(f"cin {cin} cout {cout} {H}x{W} s{s} p{p}: out {Ho}x{Wo} max err {float(err.max()):.3e} (scale {float(w.grad.abs().max()):.3e})")
    if err.max() > 1e-3 * w.grad.abs().max():
        bad_co = (err.amax((1, 2, 3)) > 1e-3).nonzero().flatten().tolist()
        bad_tap = (err.amax((0, 1)) > 1e-3).nonzero().tolist()
        print("   bad co:", bad_co[:40], " bad taps (ky,kx):", bad_tap[:20])

for a in [(2, 32, 37, 41), (2, 32, 64, 64), (2, 8, 33, 33), (2, 8, 64, 64), (3, 8, 128, 128), (2, 8, 128, 128), (3, 64, 64, 64)]:
    run(*a)
run(2, 8, 256, 256, 4, 2, 1)

torch.manual_seed(0)
cin, cout, H, W, k, s, p = 2, 8, 33, 33, 4, 2, 2
x = torch.randn(1, cin, H, W)
w = (torch.randn(cout, cin, k, k) * 0.1).requires_grad_(True)
y = F.conv2d(x, w, None, s, p); r = torch.randn_like(y); (y * r).sum().backward()
Ho, Wo = y.shape[2:]
desc = ops.conv_desc(0, k, s, p, H, W, 4, Ho, Wo, 8)
dw = torch.zeros(k * k * 8 * 4, device="cuda"); db = torch.zeros(8, device="cuda")
ops.conv_wgrad(desc, to_buf(x), None, to_buf(r), dw, db)
got = from_master(dw, k, cin, cout, False)
print("expected co0..3, ci0, tap(0,0..3):\n", w.grad[:4, 0, 0, :])
print("got:\n", got[:4, 0, 0, :])
print("db got", db.cpu(), "exp", r.sum((0, 2, 3)))

print("---- thin Cout (swap) cases")
def run_t(kind, cin, cout, H, k=4, s=2, p=1, norm=True):
    torch.manual_seed(1)
    x = torch.randn(1, cin, H, H) * 1.5 + 0.3
    xa = F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2) if norm else x
    if kind:
        w = (torch.randn(cin, cout, k, k) * 0.1).requires_grad_(True); b = torch.zeros(cout, requires_grad=True)
        y = F.conv_transpose2d(xa, w, b, s, p)
    else:
        w = (torch.randn(cout, cin, k, k) * 0.1).requires_grad_(True); b = torch.zeros(cout, requires_grad=True)
        y = F.conv2d(xa, w, b, s, p)
    r = torch.randn_like(y); (y * r).sum().backward()
    Ho = y.shape[2]
    from hip_utils import stats_of
    desc = ops.conv_desc(kind, k, s, p, H, H, pad4(cin), Ho, Ho, pad4(cout))
    dw = torch.zeros(k * k * pad4(cout) * pad4(cin), device="cuda"); db = torch.zeros(pad4(cout), device="cuda")
    nrm = ops.norm_desc(stats_of(x), None, None, H * H, 1e-5, 2, 0.2) if norm else None
    ops.conv_wgrad(desc, to_buf(x), nrm, to_buf(r), dw, db)
    torch.cuda.synchronize()
    got = from_master(dw, k, cin, cout, bool(kind))
    err = float((got - w.grad).abs().max()); sc = float(w.grad.abs().max())
    eb = float((db[:cout].cpu() - b.grad).abs().max()); sb = float(b.grad.abs().max())
    from supervised_gan_amd import _lib
    print(f"kind {kind} {cin}->{cout} H{H} k{k}s{s}p{p} norm={norm}: dW err {err:.2e}/{sc:.2e}  db err {eb:.2e}/{sb:.2e}  {_lib.lib().sgan_last_kernel().decode()}")

run_t(1, 32, 2, 16)
run_t(1, 32, 2, 33)
run_t(1, 128, 1, 20)
run_t(0, 256, 1, 10, 4, 1, 2)
run_t(0, 256, 1, 34, 4, 1, 2)
run_t(0, 64, 1, 13, 3, 1, 1)
run_t(1, 32, 2, 16, norm=False)

"""Repeat-run determinism probe of the patch kernel on small forward launches (fewer workgroups than CUs; 4 / 8 / 5 / 7 channel blocks):
the same launch 300 times (RACE_REPEATS), every result compared bit for bit with the first and with an fp64 reference.  python tools/race_probe_patch.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from supervised_gan_amd import ops, _lib
from hip_utils import master_weight, pad_vec, stats_of, to_buf
ops.set_math("bf16x3")
g = torch.Generator().manual_seed(3)
bad = 0
for kind, k, s, p, cin, cout, H, W in (("conv", 4, 1, 2, 128, 256, 33, 29), ("convT", 4, 2, 1, 256, 256, 16, 16), ("conv", 3, 1, 1, 128, 64, 24, 24),
                                       ("conv", 3, 1, 1, 160, 64, 24, 24), ("conv", 4, 1, 2, 224, 128, 20, 20)):      # 5 and 7 channel blocks: uneven groups
    tr = kind == "convT"
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), generator=g) * 0.05
    wm, bb = master_weight(w, tr), pad_vec(torch.randn(cout, generator=g) * 0.1)
    Ho, Wo = ((H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k) if tr else ((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, W, cin, Ho, Wo, cout)
    nd = ops.norm_desc(stats_of(x), None, None, H * W, 1e-5, 1, 0.0)
    xb = to_buf(x)
    first = None
    for it in range(int(os.environ.get("RACE_REPEATS", 300))):
        ob = torch.full((Ho, Wo, cout), float("nan"), device="cuda")
        ost = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
        ops.conv_fwd(desc, xb, nd, wm, bb, ob, 0, ost)
        if first is None:
            first = ob.clone()
            name = _lib.lib().sgan_last_kernel().decode()
        elif not torch.equal(ob, first):
            bad += 1
            print("run", it, "differs: max abs", float((ob - first).abs().max()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv_transpose2d(torch.relu(torch.nn.functional.instance_norm(x.double())), w.double(), None, stride=s, padding=p) if tr else \
        torch.nn.functional.conv2d(torch.relu(torch.nn.functional.instance_norm(x.double())), w.double(), None, stride=s, padding=p)
    got = first[..., :cout].permute(2, 0, 1).unsqueeze(0).double().cpu() - bb[:cout].double().cpu().view(1, -1, 1, 1)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(kind, cin, cout, H, W, name, "mismatching runs so far:", bad, "rel err vs fp64:", f"{err:.2e}")
    assert err < 5e-6, err
print("OK" if bad == 0 else "RACE")

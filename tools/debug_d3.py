import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch, torch.nn.functional as F
import sgan_oracle as O
from supervised_gan_amd import networks as N, ops
from hip_utils import to_buf, from_buf

def run(ndf, nl, s, H, nc=2, seed=3):
    sd = O.init_nlayer_d(seed, nc, ndf, nl, s)
    for v in sd.values(): v.requires_grad_(True)
    x = O.np_uniform(900 + seed, (1, nc, H, H)).requires_grad_(True)
    taps = {}
    p = O.nlayer_d_forward(sd, x, nl, s, True, taps=taps)
    for t in taps.values(): t.retain_grad()
    O.gan_loss(p, True).backward()
    x3 = taps["conv3"].detach(); dX4 = taps["conv4"].grad; dX3_ref = taps["conv3"].grad
    W4 = sd["model.11.weight"].detach()
    # CPU reference pieces in double
    x3d = x3.double()
    mean = x3d.mean((2, 3), keepdim=True); var = x3d.var((2, 3), unbiased=False, keepdim=True)
    rstd = 1 / torch.sqrt(var + 1e-5)
    xhat = (x3d - mean) * rstd
    dA = F.conv_transpose2d(dX4.double(), W4.double(), None, stride=1, padding=2)
    dY = dA * torch.where(xhat > 0, 1.0, 0.2)
    s1 = dY.sum((0, 2, 3)); s2 = (dY * xhat).sum((0, 2, 3))
    M = x3.shape[2] * x3.shape[3]
    dX3 = rstd * (dY - s1.view(1, -1, 1, 1) / M - xhat * s2.view(1, -1, 1, 1) / M)
    print("cpu double vs autograd dX3:", O.rel_err(dX3, dX3_ref))
    print("var min/max", float(var.min()), float(var.max()), "mean absmax", float(mean.abs().max()))
    print("|dY| max", float(dY.abs().max()), "|dX3| max", float(dX3.abs().max()))
    # GPU pieces
    D = N.define_D(nc, ndf, "n_layers", n_layers_D=nl, norm="instance", use_sigmoid=True, scale_factor=s, gpu_ids=[0])
    D.load_state_dict({k: v.detach() for k, v in sd.items()})
    xb = D._prepare_input(x.detach().cuda())
    outs, stats = D.run_forward(xb["chain_in"])
    torch.cuda.synchronize()
    st = stats[3].cpu()
    C = 256
    gmean = st[:C] / M; gvar = st[C:] / M - gmean * gmean
    print("gpu mean err", float((gmean - mean.view(-1)).abs().max()), "gpu var rel err", float(((gvar - var.view(-1)).abs() / var.view(-1)).max()))
    geo = D._geometry(*xb["chain_in"].shape[:2])
    desc = geo[4][0]
    L4 = D.layers[4]
    wt, _ = D._wb(L4)
    din = torch.empty((18, 18, 256), device="cuda")
    sums = torch.zeros(512, dtype=torch.float64, device="cuda")
    in_norm = D._norm_of(3, stats, 18 * 18)
    ops.conv_dgrad(desc, to_buf(dX4), wt, din, outs[3], in_norm, sums)
    torch.cuda.synchronize()
    print("dY rel", O.rel_err(from_buf(din, 256), dY), " s1 rel", O.rel_err(sums[:256], s1), " s2 rel", O.rel_err(sums[256:], s2))
    # plain dgrad (no dact)
    din2 = torch.empty((18, 18, 256), device="cuda")
    ops.conv_dgrad(desc, to_buf(dX4), wt, din2, None, None, None)
    print("dA rel", O.rel_err(from_buf(din2, 256), dA))
    ops.norm_bwd_apply(din, outs[3], in_norm, sums)
    torch.cuda.synchronize()
    print("dX3 rel", O.rel_err(from_buf(din, 256), dX3_ref))
    # which channels are bad
    err = (from_buf(din, 256).double() - dX3).abs().amax((0, 2, 3)) / dX3.abs().max()
    bad = torch.nonzero(err > 1e-3).view(-1)
    print("bad channels", bad.tolist()[:20], "n", bad.numel())
    for c in bad.tolist()[:5]:
        print(c, "var", float(var.view(-1)[c]), "mean", float(mean.view(-1)[c]), "s1", float(s1[c]), float(sums[c]), "s2", float(s2[c]), float(sums[256 + c]))

run(32, 3, 4, 512)

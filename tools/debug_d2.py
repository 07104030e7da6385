"""Per-layer backward comparison (dX of every conv) + determinism check."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import sgan_oracle as O
from supervised_gan_amd import networks as N, ops

def run(ndf, nl, s, H, nc=2, seed=3, tag=""):
    sd = O.init_nlayer_d(seed, nc, ndf, nl, s)
    for v in sd.values(): v.requires_grad_(True)
    x = O.np_uniform(900 + seed, (1, nc, H, H)).requires_grad_(True)
    taps = {}
    p = O.nlayer_d_forward(sd, x, nl, s, True, taps=taps)
    for t in taps.values(): t.retain_grad()
    O.gan_loss(p, True).backward()
    D = N.define_D(nc, ndf, "n_layers", n_layers_D=nl, norm="instance", use_sigmoid=True, scale_factor=s, gpu_ids=[0])
    D.load_state_dict({k: v.detach() for k, v in sd.items()})
    D.fuse_sigmoid_into_loss = True
    rec = []
    orig = ops.conv_wgrad
    def spy(desc, xx, in_norm, dout, dw, db):
        rec.append(dout.clone())
        return orig(desc, xx, in_norm, dout, dw, db)
    ops.conv_wgrad = spy
    try:
        res = []
        for rep in range(2):
            rec.clear()
            D.zero_grad_flat()
            xg = x.detach().cuda().requires_grad_(True)
            N.GANLoss(use_lsgan=False)(D.forward(xg), True).backward()
            torch.cuda.synchronize()
            line = []
            for i, d in enumerate(reversed(rec)):   # rec is last layer first
                ref = taps[f"conv{i}"].grad
                got = d[..., :ref.shape[1]].permute(2, 0, 1).unsqueeze(0)
                line.append(f"dX{i} {O.rel_err(got, ref):.1e}")
            line.append(f"dx {O.rel_err(xg.grad, x.grad):.1e}")
            line.append(f"w8 {O.rel_err(dict(D.named_parameters())['model.8.weight'].grad, sd['model.8.weight'].grad):.1e}")
            print(tag, f"ndf={ndf} s={s} H={H} rep{rep}:", "  ".join(line))
    finally:
        ops.conv_wgrad = orig

if __name__ == "__main__":
    order = sys.argv[1] if len(sys.argv) > 1 else "a"
    if order == "a":
        run(32, 3, 4, 512); run(32, 3, 2, 256); run(32, 3, 4, 512)
    else:
        run(32, 3, 2, 256); run(32, 3, 4, 512); run(32, 3, 2, 256)

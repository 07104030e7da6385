"""Per-layer HIP-vs-oracle comparison of one NLayerDiscriminator (debug aid, GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import sgan_oracle as O
from supervised_gan_amd import networks as N, ops

def run(ndf, nl, s, H, nc=2, seed=3):
    sd = O.init_nlayer_d(seed, nc, ndf, nl, s)
    for v in sd.values(): v.requires_grad_(True)
    x = O.np_uniform(900 + seed, (1, nc, H, H)).requires_grad_(True)
    taps = {}
    p = O.nlayer_d_forward(sd, x, nl, s, True, taps=taps)
    for t in taps.values(): t.retain_grad()
    loss = O.gan_loss(p, True)
    loss.backward()
    D = N.define_D(nc, ndf, "n_layers", n_layers_D=nl, norm="instance", use_sigmoid=True, scale_factor=s, gpu_ids=[0])
    D.load_state_dict({k: v.detach() for k, v in sd.items()})
    xg = x.detach().cuda().requires_grad_(True)
    xb = D._prepare_input(xg)
    outs, stats = D.run_forward(xb["chain_in"])
    print(f"--- ndf={ndf} nl={nl} s={s} H={H}")
    for i, o in enumerate(outs):
        ref = taps[f"conv{i}"]
        got = o[..., :ref.shape[1]].permute(2, 0, 1).unsqueeze(0)
        print(f"fwd conv{i} {tuple(ref.shape)} rel {O.rel_err(got, ref):.2e}")
    D.fuse_sigmoid_into_loss = True
    crit = N.GANLoss(use_lsgan=False)
    l = crit(D.forward(xg), True)
    l.backward()
    torch.cuda.synchronize()
    print("loss", float(l), float(loss))
    print(f"dx rel {O.rel_err(xg.grad, x.grad):.2e}")
    for k, prm in D.named_parameters():
        if k.startswith("model."):
            print(f"grad {k} rel {O.rel_err(prm.grad, sd[k].grad):.2e}  max {float(sd[k].grad.abs().max()):.3e}")

if __name__ == "__main__":
    run(32, 3, 4, 512)
    run(32, 3, 1, 128)
    run(8, 3, 4, 512)
    run(32, 3, 2, 256)

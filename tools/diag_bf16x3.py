"""Where does the split-bf16 error of the first discriminator layer's weight gradient come from?  One PatchGAN D (ndf 32, 512^2):
every parameter gradient in both math modes against an fp64 CPU evaluation of the same net."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import sgan_oracle as O  # noqa: E402
from supervised_gan_amd import networks as N, ops  # noqa: E402

torch.manual_seed(0)
H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sd = O.init_nlayer_d(2, 2, 32, 3, 1)
x = O.np_uniform(7000, (1, 3, H, H))[:, :2].contiguous()
if len(sys.argv) > 2 and sys.argv[2] == "tanh":
    x = torch.tanh(3 * x)
# fp64 truth
w = {k: v.double().requires_grad_(True) for k, v in sd.items() if k.startswith("model.")}
a = x.double()
a = F.leaky_relu(F.conv2d(a, w["model.0.weight"], w["model.0.bias"], stride=2, padding=2), 0.2)
for i, s in ((2, 2), (5, 2), (8, 1)):
    a = F.conv2d(a, w[f"model.{i}.weight"], w[f"model.{i}.bias"], stride=s, padding=2)
    a = F.leaky_relu(F.instance_norm(a, eps=1e-5), 0.2)
y = F.conv2d(a, w["model.11.weight"], w["model.11.bias"], stride=1, padding=2)
loss = F.binary_cross_entropy(torch.sigmoid(y), torch.ones_like(y))
loss.backward()
truth = {k: v.grad for k, v in w.items()}
for mode in ("f32", "bf16x3"):
    ops.set_math(mode)
    D = N.define_D(2, 32, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=1).cuda()
    D.load_state_dict(sd)
    crit = N.GANLoss(use_lsgan=False)
    D.fuse_sigmoid_into_loss = True
    xi = x.cuda()
    l = crit(D.forward(xi), True)
    l.backward()
    torch.cuda.synchronize()
    print(mode, "loss err", abs(float(l) - float(loss)))
    for k, p in D.named_parameters():
        if k in truth:
            t = truth[k]
            e = float((p.grad.double().cpu() - t).abs().max() / t.abs().max())
            print(f"   {k:18s} max|g| {float(t.abs().max()):.3e}  rel err {e:.2e}")

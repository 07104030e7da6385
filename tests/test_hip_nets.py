"""Network-level parity on the MI355X: the drop-in `define_G` / `define_D` modules against golden
vectors produced by the real reference (tests/golden, see oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import sgan_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def N():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd import _lib, networks
    _lib.lib()
    return networks


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def rel(a, b):
    a = a.detach() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
    b = b.detach() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b))
    return O.rel_err(a, b)


def test_fcgan_g_small(N, golden_dir):
    g = load(golden_dir, "fcgan_g_small.npz")
    G = N.define_G(2, 0, 8, "fcgan", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8, gpu_ids=[0])
    sd = O.init_fcgan_g(11, 8, 2, 8, 5)
    G.load_state_dict(sd)
    z = O.np_normal(101, (1, 8, 2, 2)).cuda().requires_grad_(True)
    r = O.np_normal(102, (1, 2, 128, 128)).cuda()
    y = G.forward(z)
    assert y.shape == (1, 2, 128, 128)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL
    assert rel(z.grad, g["dz"]) < TOL
    params = dict(G.named_parameters())
    undet = O.norm_cancelled_keys_g(5)
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            if name in undet:
                scale = np.abs(g["grad/" + name.replace(".bias", ".weight")]).max()
                assert np.abs(params[name].grad.cpu().numpy() - g[k]).max() < TOL * scale, name
            else:
                assert rel(params[name].grad, g[k]) < TOL, name
        if k.startswith("buf/"):
            assert rel(G.state_dict()[k[4:]].double(), g[k].astype(np.float64)) < TOL, k


def test_fcgan_g_dropout_small(N, golden_dir):
    """--which_model_netG fcgan without --no_dropout (models/networks.py:513-521: ConvT -> BatchNorm -> Dropout(0.5) -> ReLU above the
    first block) against the reference golden made with the same injected masks: the masked BatchNorm output (affine included) is
    materialised by sgan_norm_apply_fwd, its backward runs mask -> sums -> norm backward with the affine's gradients."""
    g = load(golden_dir, "fcgan_g_dropout_small.npz")
    G = N.define_G(2, 0, 8, "fcgan", "instance", True, n_layers_G=5, use_fcn=True, noise_nc=8, gpu_ids=[0])
    sd = O.init_fcgan_g(11, 8, 2, 8, 5, use_dropout=True)
    assert list(G.state_dict().keys()) == list(sd.keys())       # nn.Sequential indices with the Dropout modules counted
    G.load_state_dict(sd)
    G.mask_source = lambda li, shape: O.dropout_mask_np(70 + li, (1, shape[2], shape[0], shape[1]))[0].permute(1, 2, 0).contiguous().cuda()
    z = O.np_normal(101, (1, 8, 2, 2)).cuda().requires_grad_(True)
    r = O.np_normal(102, (1, 2, 128, 128)).cuda()
    y = G.forward(z)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL and rel(z.grad, g["dz"]) < TOL
    params = dict(G.named_parameters())
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            wname = name.replace(".bias", ".weight")
            if name.endswith(".bias") and params[wname].dim() == 4:      # conv bias in front of a BatchNorm: analytically zero
                scale = np.abs(g["grad/" + wname]).max()
                assert np.abs(params[name].grad.cpu().numpy() - g[k]).max() < TOL * scale, name
            else:
                assert rel(params[name].grad, g[k]) < TOL, name
        if k.startswith("buf/") and not k.endswith("running_mean"):      # running_mean tracks the undetermined conv bias
            assert rel(G.state_dict()[k[4:]].double(), g[k].astype(np.float64)) < TOL, k
    G.mask_source = None      # own Philox masks: a fresh set per forward
    y1, y2 = G.forward(z.detach()), G.forward(z.detach())
    torch.cuda.synchronize()
    assert float((y1 - y2).abs().max()) > 0 and float(y1.abs().max()) <= 1.0


@pytest.mark.parametrize("s", [1, 2, 4])
def test_nlayer_d_small(N, golden_dir, s):
    g = load(golden_dir, f"nlayer_d_small_s{s}.npz")
    D = N.define_D(2, 8, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=s, gpu_ids=[0])
    D.load_state_dict(O.init_nlayer_d(20 + s, 2, 8, 3, s))
    x = O.np_uniform(200 + s, (1, 2, 128, 128)).cuda().requires_grad_(True)
    crit = N.GANLoss(use_lsgan=False)
    p = D.forward(x)
    assert p.shape == g["p"].shape
    l_real = crit(p, True)
    D.fuse_sigmoid_into_loss = True        # second call exercises the logits-tagged path
    l_fake = crit(D.forward(x), False)
    (l_real * 0.7 + l_fake * 0.3).backward()
    torch.cuda.synchronize()
    assert rel(p, g["p"]) < TOL
    assert abs(float(l_real) - float(g["l_real"])) < 1e-4
    assert abs(float(l_fake) - float(g["l_fake"])) < 1e-4
    assert rel(x.grad, g["dx"]) < TOL
    params = dict(D.named_parameters())
    undet = O.norm_cancelled_keys_d(2, 8, 3)
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            if name in undet:
                scale = np.abs(g["grad/" + name.replace(".bias", ".weight")]).max()
                assert np.abs(params[name].grad.cpu().numpy() - g[k]).max() < TOL * scale, name
            else:
                assert rel(params[name].grad, g[k]) < TOL, name


@pytest.mark.parametrize("s", [1, 2])
def test_nlayer_d_sep_small(N, golden_dir, s):
    """`--which_model_netD n_layers_sep`: the two stems as one block-diagonal chain (discriminators.NLayerDiscriminatorSep) against
    the reference golden -- output, loss, image gradient, every parameter gradient under the reference's keys; the cross blocks of
    the stem weights stay exactly zero in the parameters and in the gradients."""
    g = load(golden_dir, f"nlayer_d_sep_small_s{s}.npz")
    D = N.define_D(3, 8, "n_layers_sep", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=s, gpu_ids=[0])
    D.load_state_dict(O.init_nlayer_d_sep(40 + s, 8, 3, s))
    x = O.np_uniform(240 + s, (1, 3, 128, 128)).cuda().requires_grad_(True)
    crit = N.GANLoss(use_lsgan=False)
    p = D.forward(x)
    assert p.shape == g["p"].shape
    loss = crit(p, True) * 0.6 + crit(D.forward(x), False) * 0.4
    loss.backward()
    torch.cuda.synchronize()
    assert rel(p, g["p"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    assert rel(x.grad, g["dx"]) < TOL
    params = dict(D.named_parameters())
    assert {k[5:] for k in g.files if k.startswith("grad/")} <= set(params)
    wscale = {}
    for k in g.files:
        if k.startswith("grad/") and k.endswith(".weight"):
            wscale[k[5:-7]] = np.abs(g[k]).max()
            assert rel(params[k[5:]].grad, g[k]) < TOL, k
    for k in g.files:      # biases: those in front of an InstanceNorm have an analytically zero gradient (compared on the weight's scale)
        if k.startswith("grad/") and k.endswith(".bias"):
            assert np.abs(params[k[5:]].grad.cpu().numpy() - g[k]).max() < TOL * max(wscale[k[5:-5]], np.abs(g[k]).max()), k
    for L in D.layers[:2]:
        for flat in (D._flat, D._gflat):
            m = flat[L.w_off: L.w_off + 16 * L.cout_s * L.cin_s].view(16, L.cout_s, L.cin_s)
            h, ca = L.cout // 2, (2 if L.key == "stem.0" else L.cin // 2)
            assert float(m[:, :h, ca:].abs().max()) == 0.0 and float(m[:, h:, :ca].abs().max()) == 0.0
    # a second pass (gradients accumulate), and an optimizer over ALL parameters keeps the block structure
    from supervised_gan_amd.optim import FusedAdam
    opt = FusedAdam(D.parameters() if s == 1 else [p_ for n_, p_ in D.named_parameters() if not n_.startswith("gauss")], lr=1e-3, betas=(0.5, 0.999))
    opt.step()
    L = D.layers[0]
    m = D._flat[L.w_off: L.w_off + 16 * L.cout_s * L.cin_s].view(16, L.cout_s, L.cin_s)
    assert float(m[:, :8, 2:].abs().max()) == 0.0 and float(m[:, 8:, :2].abs().max()) == 0.0 and float(m[:, :8, :2].abs().max()) > 0


def test_nlayer_d_n4_lsgan(N, golden_dir):
    g = load(golden_dir, "nlayer_d_small_n4_lsgan.npz")
    D = N.define_D(3, 8, "n_layers", n_layers_D=4, norm="instance", use_sigmoid=False, scale_factor=1, gpu_ids=[0])
    D.load_state_dict(O.init_nlayer_d(29, 3, 8, 4, 1))
    x = O.np_uniform(209, (1, 3, 128, 128)).cuda().requires_grad_(True)
    p = D.forward(x)
    loss = N.GANLoss(use_lsgan=True)(p, True)
    loss.backward()
    torch.cuda.synchronize()
    assert rel(p, g["p"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    assert rel(x.grad, g["dx"]) < TOL
    params = dict(D.named_parameters())
    assert rel(params["model.0.weight"].grad, g["grad/model.0.weight"]) < TOL
    assert rel(params["model.11.weight"].grad, g["grad/model.11.weight"]) < TOL


def test_skip_param_grads_and_grad_reattach(N):
    """G-step mode (no discriminator weight gradients) and torch-style zero_grad(set_to_none=True)."""
    D = N.define_D(2, 8, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=2, gpu_ids=[0])
    x = torch.rand(1, 2, 64, 64, device="cuda").requires_grad_(True)
    crit = N.GANLoss(use_lsgan=False)
    D.compute_param_grads = False
    crit(D.forward(x), True).backward()
    assert float(D._gflat.abs().max()) == 0.0 and x.grad is not None and float(x.grad.abs().max()) > 0
    D.compute_param_grads = True
    for p in D.model.parameters():
        p.grad = None
    crit(D.forward(x.detach()), True).backward()
    g1 = D._gflat.clone()
    assert all(p.grad is not None for p in D.model.parameters()) and float(g1.abs().max()) > 0
    crit(D.forward(x.detach()), True).backward()       # accumulates like autograd does
    assert O.rel_err(D._gflat, 2 * g1) < 1e-4


def test_cpu_tensors_fail_loudly(N):
    from supervised_gan_amd._lib import SganError
    D = N.define_D(2, 8, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True)
    with pytest.raises(SganError):
        D.forward(torch.rand(1, 2, 64, 64))


def test_grouped_chains_equal_individual_chains(N):
    """The three multi-scale discriminators on two inputs as ONE launch per layer (multi_forward) must give what
    six separate network calls give: outputs, input gradients, accumulated weight gradients."""
    Ds = [N.define_D(2, 8, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=s, gpu_ids=[0]) for s in (1, 2, 4)]
    for i, d in enumerate(Ds):
        d.load_state_dict(O.init_nlayer_d(40 + i, 2, 8, 3, (1, 2, 4)[i]))
        d.fuse_sigmoid_into_loss = True
    assert N.can_group(Ds + Ds)
    xa = O.np_uniform(300, (1, 2, 160, 160)).cuda().requires_grad_(True)
    xb = O.np_uniform(301, (1, 2, 160, 160)).cuda().requires_grad_(True)
    crit = N.GANLoss(use_lsgan=False)
    lam = [0.5, 0.4, 0.1, 0.2, 0.3, 0.6]

    def run(grouped):
        for d in Ds:
            d.zero_grad_flat()
        xa.grad = xb.grad = None
        jobs = [(d, xa) for d in Ds] + [(d, xb) for d in Ds]
        preds = N.multi_forward(jobs) if grouped else [d.forward(x) for d, x in jobs]
        loss = sum(crit(p, i % 2 == 0) * l for i, (p, l) in enumerate(zip(preds, lam)))
        loss.backward()
        torch.cuda.synchronize()
        return ([p.detach().clone() for p in preds], xa.grad.clone(), xb.grad.clone(), [d._gflat.clone() for d in Ds], float(loss))

    pa, ga, gb, wa, la = run(False)
    pg, gga, ggb, wg, lg = run(True)
    assert abs(la - lg) < 1e-6
    # not bit-identical: the single-problem path splits deep reductions over K, the grouped one does not
    for a, b in zip(pa, pg):
        assert O.rel_err(b, a) < 1e-4
    assert O.rel_err(gga, ga) < 1e-3 and O.rel_err(ggb, gb) < 1e-3
    for a, b in zip(wa, wg):
        assert O.rel_err(b, a) < 1e-3
    # G-step mode: no weight gradients
    for d in Ds:
        d.zero_grad_flat()
        d.compute_param_grads = False
    preds = N.multi_forward([(d, xa) for d in Ds])
    sum(crit(p, True) for p in preds).backward()
    assert all(float(d._gflat.abs().max()) == 0.0 for d in Ds)
    for d in Ds:
        d.compute_param_grads = True


def test_begin_step_clears_the_gradients_zero_grad_was_about_to(N):
    """ops.begin_step(also_zero=FusedAdam.take_zeroing()): the step's one zeroing launch clears the optimizer's gradient arena and the
    zero_grad() that follows does nothing (once); an optimizer whose step() already zeroed its gradients hands over nothing."""
    from supervised_gan_amd import ops
    from supervised_gan_amd.optim import FusedAdam
    D = N.define_D(2, 8, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=1, gpu_ids=[0])
    opt = FusedAdam(list(D.model.parameters()), lr=2e-4, betas=(0.5, 0.999))
    D._gflat.fill_(3.0)
    bufs = opt.take_zeroing()
    assert bufs and sum(b.numel() for b in bufs) == D._gflat.numel()      # the padded flat arena, not just the logical parameters
    ops.begin_step(bufs)
    torch.cuda.synchronize()
    assert float(D._gflat.abs().max()) == 0.0
    D._gflat.fill_(5.0)
    opt.zero_grad()          # the one the caller took over: a no-op
    assert float(D._gflat.min()) == 5.0
    opt.zero_grad()          # the next one clears again
    torch.cuda.synchronize()
    assert float(D._gflat.abs().max()) == 0.0
    G = N.define_G(2, 0, 8, "fcgan", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8, gpu_ids=[0])
    og = FusedAdam(list(G.parameters()), lr=2e-4, betas=(0.5, 0.999), zero_grads_in_step=True)
    G._gflat.fill_(1e-3)
    og.step()                # zeroes what it consumed
    torch.cuda.synchronize()
    assert float(G._gflat.abs().max()) == 0.0 and og.take_zeroing() == []


def test_forward_pair_refills_a_kept_forward(N):
    """chain.forward_pair: G(za) and G(zb) as one two-problem pass, G(zb) written into the buffers of an earlier kept forward and handed
    back under a fresh autograd node == two separate calls (outputs, parameter gradients, BatchNorm running statistics in call order)."""
    from supervised_gan_amd import chain, ops

    def make():
        G = N.define_G(2, 0, 8, "fcgan", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8, gpu_ids=[0])
        G.load_state_dict(O.init_fcgan_g(11, 8, 2, 8, 5))
        return G

    def latent(seed):
        buf = torch.zeros(2, 2, 8, device="cuda")
        v = ops.logical_view(buf, 8)
        v.copy_(O.np_normal(seed, (1, 8, 2, 2)).cuda())
        return v

    r = O.np_normal(102, (1, 2, 128, 128)).cuda()
    Gr, Gp = make(), make()
    # reference: separate calls  z0 | za, zb (+ backward through G(zb)) | zc, zd (+ backward through G(zd))
    Gr.forward(latent(1))
    ref = []
    for sa, sb in ((2, 3), (4, 5)):
        ya = Gr.forward(latent(sa)).detach().clone()
        Gr.zero_grad_flat()
        yb = Gr.forward(latent(sb))
        (yb * r).sum().backward()
        torch.cuda.synchronize()
        ref.append((ya, yb.detach().clone(), Gr._gflat.clone(), {k: v.clone() for k, v in Gr.state_dict().items() if "running" in k or "tracked" in k}))
    # kept forward + two refills of its buffers
    zb = latent(1)
    Gp._keep_next = True
    y0 = Gp.forward(zb)
    kept_out = Gp._kept["outs"][-1]
    for (sa, sb), (ra, rb, rg, rbuf) in zip(((2, 3), (4, 5)), ref):
        zb.copy_(O.np_normal(sb, (1, 8, 2, 2)).cuda())          # the kept call's input buffer, new latent
        Gp.zero_grad_flat()
        ya, yb = chain.forward_pair(Gp, latent(sa), zb)
        assert yb.data_ptr() == kept_out.data_ptr() and ya.grad_fn is None and yb.grad_fn is not None
        (yb * r).sum().backward()
        torch.cuda.synchronize()
        assert rel(ya, ra) < 1e-4 and rel(yb, rb) < 1e-4
        assert O.rel_err(Gp._gflat, rg) < 1e-3
        for k, v in rbuf.items():
            assert rel(Gp.state_dict()[k].double(), v.double()) < 1e-5, k
    assert y0.data_ptr() == kept_out.data_ptr()


# ------------------------------------------------------------------------------------------------
# U-Net generator (models/networks.py:318-419)
# ------------------------------------------------------------------------------------------------
UNET_SMALL = {"skipall_dropout": dict(use_dropout=True, num_skips=-1, add_gaussian_noise=False),
              "skip4_noise": dict(use_dropout=False, num_skips=4, add_gaussian_noise=True),
              "residual": dict(use_dropout=False, num_skips=-1, add_gaussian_noise=False, use_residual=True, out_nc=2),
              "batchnorm": dict(use_dropout=True, num_skips=-1, add_gaussian_noise=False, norm="batch")}


def inject_unet_random(G, H, mask_seed, noise_seed):
    """Give the HIP U-Net the numpy-seeded dropout masks / Gaussian noise the golden vectors were made with."""
    masks, noises = {}, {}
    for l in range(1, G.n):
        h = H >> l
        shape = (1, G.c[l - 1], h, h)
        if G.drop[l]:
            masks[l] = O.dropout_mask_np(mask_seed, shape)[0].permute(1, 2, 0).contiguous().cuda()
        if G.add_gauss:
            noises[l] = O.gauss_noise_np(noise_seed, shape)[0].permute(1, 2, 0).contiguous().cuda()
    G.mask_override = masks if G.use_dropout else None
    G.noise_override = noises if G.add_gauss else None


@pytest.mark.parametrize("tag", list(UNET_SMALL))
def test_unet_small(N, golden_dir, tag):
    g = load(golden_dir, f"unet_small_{tag}.npz")
    kw = UNET_SMALL[tag]
    onc = kw.get("out_nc", 1)      # "residual": --use_residual, tanh(x + y) on 2 -> 2 channels (models/networks.py:367)
    norm = kw.get("norm", "instance")      # "batchnorm": --norm batch (affine + running statistics, networks.py:43-50,387-389)
    G = N.define_G(2, onc, 8, "unet_128", norm, kw["use_dropout"], n_layers_G_skip=kw["num_skips"],
                   add_gaussian_noise=kw["add_gaussian_noise"], gaussian_sigma=0.1, use_residual=kw.get("use_residual", False), gpu_ids=[0])
    sd = O.init_unet(31, 7, 2, onc, 8, kw["num_skips"], norm=norm)
    assert set(G.state_dict().keys()) == set(sd.keys())
    assert list(G.state_dict().keys())[:3] == ['model.0.weight', 'model.0.bias', 'model.1.model.1.weight']     # nn.Sequential order
    assert list(G.state_dict().keys())[-2:] == ['model.3.weight', 'model.3.bias']
    G.load_state_dict(sd)
    inject_unet_random(G, 256, 40, 50)
    x = O.np_uniform(301, (1, 2, 256, 256)).cuda().requires_grad_(True)
    r = O.np_normal(302, (1, onc, 256, 256)).cuda()
    y = G.forward(x)
    assert y.shape == (1, onc, 256, 256)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL
    assert rel(x.grad, g["dx"]) < TOL
    params = dict(G.named_parameters())
    undet = O.norm_cancelled_keys_unet(7, 8, kw["num_skips"])
    for k in g.files:
        if not k.startswith("grad/"):
            continue
        name = k[5:]
        if name in undet:
            scale = np.abs(params[name.replace(".bias", ".weight")].grad.cpu().numpy()).max()
            assert np.abs(params[name].grad.cpu().numpy()).max() < TOL * scale, name
        else:
            assert rel(params[name].grad, g[k]) < TOL, name
    for k in g.files:      # --norm batch: the running statistics the forward left behind
        if k.startswith("buf/"):
            assert rel(G.state_dict()[k[4:]].double(), g[k].astype(np.float64)) < TOL, k


@pytest.mark.parametrize("tag,which,nb,drop,res", [("6", "resnet_6blocks", 6, False, False), ("9_dropout", "resnet_9blocks", 9, True, False),
                                                   ("6_residual", "resnet_6blocks", 6, False, True), ("6_batchnorm", "resnet_6blocks", 6, True, False)])
def test_resnet_small(N, golden_dir, tag, which, nb, drop, res):
    """--which_model_netG resnet_6blocks / resnet_9blocks (models/networks.py:221-311) on the HIP path against the reference golden:
    reflection padding (materialised gather), 49-tap k7 layers, stride-2 convs, residual blocks with dropout, ConvT k3 s2 with
    output padding; state_dict keys and order as the reference's nn.Sequential."""
    g = load(golden_dir, f"resnet_small_{tag}.npz")
    onc = 2 if res else 1      # --use_residual: no Tanh module at the end of the Sequential, forward() = tanh(x + y) (:258-268)
    norm = "batch" if "batchnorm" in tag else "instance"      # --norm batch: BatchNorm2d (affine, running statistics) behind every conv but the last
    G = N.define_G(2, onc, 8, which, norm, drop, use_residual=res, gpu_ids=[0])
    sd = O.init_resnet(41, 2, onc, 8, nb, drop, norm=norm)
    assert list(G.state_dict().keys()) == list(sd.keys())
    G.load_state_dict(sd)
    G.mask_source = lambda i, shape: O.dropout_mask_np(60 + i, (1, shape[2], shape[0], shape[1]))[0].permute(1, 2, 0).contiguous().cuda()
    x = O.np_uniform(311, (1, 2, 64, 64)).cuda().requires_grad_(True)
    r = O.np_normal(312, (1, onc, 64, 64)).cuda()
    y = G.forward(x)
    assert y.shape == (1, onc, 64, 64)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL
    assert rel(x.grad, g["dx"]) < TOL
    params = dict(G.named_parameters())
    last = f"model.{17 + nb}.bias"
    for k in g.files:
        if not k.startswith("grad/"):
            continue
        name = k[5:]
        if name.endswith(".bias") and name != last and params[name.replace(".bias", ".weight")].dim() == 4:
            scale = np.abs(params[name.replace(".bias", ".weight")].grad.cpu().numpy()).max()
            assert np.abs(params[name].grad.cpu().numpy()).max() < TOL * scale, name
        else:
            assert rel(params[name].grad, g[k]) < TOL, name
    for k in g.files:      # --norm batch: the running statistics the forward left behind (running_mean follows the undetermined conv bias: skipped)
        if k.startswith("buf/") and not k.endswith("running_mean"):
            assert rel(G.state_dict()[k[4:]].double(), g[k].astype(np.float64)) < TOL, k
    if drop:      # without injected masks: Philox masks, a fresh set per forward, still a valid tanh image
        G.mask_source = None
        y1, y2 = G.forward(x.detach()), G.forward(x.detach())
        torch.cuda.synchronize()
        assert float((y1 - y2).abs().max()) > 0 and float(y1.abs().max()) <= 1.0


def test_unet_own_dropout_and_noise(N):
    """Without injected tensors the masks come from the Philox kernel: half the entries kept (scaled by 2), a fresh
    mask per forward."""
    from supervised_gan_amd import ops
    m = torch.empty(64, 64, 32, device="cuda")
    off = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.dropout_mask(m, 0.5, 7, off)
    m2 = torch.empty_like(m)
    ops.dropout_mask(m2, 0.5, 7, off)
    torch.cuda.synchronize()
    assert set(m.unique().tolist()) == {0.0, 2.0}
    assert abs(float(m.mean()) - 1.0) < 0.02
    assert float((m != m2).float().mean()) > 0.4
    G = N.define_G(2, 1, 8, "unet_128", "instance", True, add_gaussian_noise=True, gpu_ids=[0])
    x = O.np_uniform(1, (1, 2, 256, 256)).cuda()
    y1, y2 = G.forward(x), G.forward(x)
    assert torch.isfinite(y1).all() and float((y1 - y2).detach().abs().max()) > 0


# ------------------------------------------------------------------------------------------------
# Cascaded refinement network (models/networks.py:642-794)
# ------------------------------------------------------------------------------------------------
CRN_SMALL = {"convt_b1": ("convt", 1), "bilinear_b2": ("bilinear", 2), "bilinear_b2_batchnorm": ("bilinear", 2)}


@pytest.mark.parametrize("tag", list(CRN_SMALL))
def test_crn_small(N, golden_dir, tag):
    g = load(golden_dir, f"crn_small_{tag}.npz")
    mode, nlb = CRN_SMALL[tag]
    norm = "batch" if "batchnorm" in tag else "instance"      # --norm batch (the bilinear block's norm is child 2 of its Sequential; shared label norm)
    G = N.define_G(2, 1, 8, "crn", norm, False, n_layers_G=5, noise_nc=8, upsample_mode=mode, n_layers_CRN_block=nlb,
                   share_label_weights=True, gpu_ids=[0])
    sd = O.init_crn(41, 2, 1, 8, 8, mode, nlb, True, norm=norm)
    assert list(G.state_dict().keys()) == list(sd.keys())          # the reference's module order, no `model.` prefix
    G.load_state_dict(sd)
    label = O.np_uniform(401, (1, 2, 128, 128)).cuda().requires_grad_(True)
    z = O.np_normal(402, (1, 8, 2, 2)).cuda().requires_grad_(True)
    r = O.np_normal(403, (1, 1, 128, 128)).cuda()
    y = G.forward(label, z)
    assert y.shape == (1, 1, 128, 128)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL
    assert rel(label.grad, g["dlabel"]) < TOL
    assert rel(z.grad, g["dz"]) < TOL
    params = dict(G.named_parameters())
    undet = O.norm_cancelled_keys_crn(2, 1, 8, 8, mode, nlb, True)
    for k in g.files:
        if not k.startswith("grad/"):
            continue
        name = k[5:]
        if name in undet:
            scale = np.abs(params[name.replace(".bias", ".weight")].grad.cpu().numpy()).max()
            assert np.abs(params[name].grad.cpu().numpy()).max() < TOL * scale, name
        else:
            assert rel(params[name].grad, g[k]) < TOL, name
    for k in g.files:      # --norm batch: running statistics (running_mean follows the undetermined conv bias only where there is one ... it is determined here: first step)
        if k.startswith("buf/"):
            assert rel(G.state_dict()[k[4:]].double(), g[k].astype(np.float64)) < TOL, k


def test_crn_small_noise(N, golden_dir):
    """`--add_gaussian_noise` of the CRN: norm(u) + sigma * noise is materialised (sgan_norm_apply_fwd) in front of the inter block of
    stages 5..1; the reference's noise tensors are injected.  Without injection two calls differ (own Philox draws)."""
    g = load(golden_dir, "crn_small_noise.npz")
    G = N.define_G(2, 1, 8, "crn", "instance", False, n_layers_G=5, noise_nc=8, upsample_mode="convt", n_layers_CRN_block=2,
                   share_label_weights=True, add_gaussian_noise=True, gaussian_sigma=0.1, gpu_ids=[0])
    G.load_state_dict(O.init_crn(43, 2, 1, 8, 8, "convt", 2, True))
    G.noise_override = {s: O.gauss_noise_np(80, (1, 8, 128 >> s, 128 >> s))[0].permute(1, 2, 0).contiguous().cuda() for s in range(1, 6)}
    label = O.np_uniform(411, (1, 2, 128, 128)).cuda().requires_grad_(True)
    z = O.np_normal(412, (1, 8, 2, 2)).cuda().requires_grad_(True)
    r = O.np_normal(413, (1, 1, 128, 128)).cuda()
    y = G.forward(label, z)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL and rel(label.grad, g["dlabel"]) < TOL and rel(z.grad, g["dz"]) < TOL
    params = dict(G.named_parameters())
    undet = O.norm_cancelled_keys_crn(2, 1, 8, 8, "convt", 2, True)
    for k in g.files:
        if k.startswith("grad/") and k[5:] not in undet:
            assert rel(params[k[5:]].grad, g[k]) < TOL, k
    G.noise_override = None
    with torch.no_grad():
        y1, y2 = G.forward(label.detach(), z.detach()), G.forward(label.detach(), z.detach())
    assert float((y1 - y2).abs().max()) > 0


def test_autoencoder_small(N, golden_dir):
    """`--which_model_netG autoencoder` (models/networks.py:421-490), a plain chain on the same kernels."""
    g = load(golden_dir, "autoencoder_small.npz")
    G = N.define_G(2, 1, 8, "autoencoder", "instance", False, n_layers_G=3, gpu_ids=[0])
    sd = O.init_autoencoder(61, 2, 1, 3, 8)
    assert list(G.state_dict().keys()) == list(sd.keys())
    G.load_state_dict(sd)
    x = O.np_uniform(601, (1, 2, 128, 128)).cuda().requires_grad_(True)
    r = O.np_normal(602, (1, 1, 128, 128)).cuda()
    y = G.forward(x)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL and rel(x.grad, g["dx"]) < TOL
    params = dict(G.named_parameters())
    for k in g.files:
        if k.startswith("grad/") and k.endswith(".weight"):
            assert rel(params[k[5:]].grad, g[k]) < TOL, k


def test_autoencoder_dropout(N, golden_dir):
    """`--which_model_netG autoencoder` with dropout: the masked normalised tensor is materialised (sgan_norm_apply_fwd) and its
    backward runs mask -> sums -> norm backward (sgan_norm_apply_bwd_sums, sgan_norm_bwd_apply); the reference's masks are injected.
    Eval mode drops nothing."""
    g = load(golden_dir, "autoencoder_dropout.npz")
    G = N.define_G(2, 1, 8, "autoencoder", "instance", True, n_layers_G=3, gpu_ids=[0])
    sd = O.init_autoencoder(62, 2, 1, 3, 8, True)
    assert list(G.state_dict().keys()) == list(sd.keys())
    G.load_state_dict(sd)
    order = [li for li, L in enumerate(G.layers) if L.drop > 0]
    ps = {li: G.layers[li].drop for li in order}
    assert [ps[li] for li in order] == [0.2, 0.2, 0.5, 0.5]
    G.mask_source = lambda li, shape: O.dropout_mask_np(70 + order.index(li), (1, shape[2], shape[0], shape[1]), ps[li])[0].permute(1, 2, 0).contiguous().cuda()
    x = O.np_uniform(611, (1, 2, 128, 128)).cuda().requires_grad_(True)
    r = O.np_normal(612, (1, 1, 128, 128)).cuda()
    y = G.forward(x)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL and rel(x.grad, g["dx"]) < TOL
    params = dict(G.named_parameters())
    for k in g.files:
        if k.startswith("grad/") and k.endswith(".weight"):
            assert rel(params[k[5:]].grad, g[k]) < TOL, k
    G.mask_source = None
    y1 = G.forward(x.detach())          # own Philox masks: a different draw every call, same statistics
    y2 = G.forward(x.detach())
    assert float((y1 - y2).abs().max()) > 0
    G.eval()
    with torch.no_grad():
        ye = G.forward(x.detach())
    ref = O.autoencoder_forward({k: v for k, v in sd.items()}, x.detach().cpu(), 3, 8, True, mask_seed=None)
    assert rel(ye, ref) < TOL


def test_dcgan_small(N, golden_dir):
    """`--which_model_netG dcgan` / `--which_model_netD dcgan` (models/networks.py:1015-1129): k4 s1 p0 ConvT on a 1x1 latent, BatchNorm
    chains without biases, a k4 s1 p0 logits conv on a 4x4 map."""
    g = load(golden_dir, "dcgan_small.npz")
    nz, nc, ngf, ndf = 8, 2, 8, 8
    G = N.define_G(nc, 0, ngf, "dcgan", "batch", False, noise_nc=nz, gpu_ids=[0])
    sd = O.init_dcgan_g(71, nz, nc, ngf)
    assert list(G.state_dict().keys()) == list(sd.keys())
    G.load_state_dict(sd)
    z = O.np_normal(701, (1, nz, 1, 1)).cuda().requires_grad_(True)
    y = G.forward(z)
    (y * O.np_normal(702, tuple(y.shape)).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert tuple(y.shape) == (1, nc, 128, 128)
    assert rel(y, g["G/y"]) < TOL and rel(z.grad, g["G/dz"]) < TOL
    params, bufs = dict(G.named_parameters()), G.state_dict()
    for k in g.files:
        if k.startswith("G/grad/") and k.endswith(".weight"):
            assert rel(params[k[7:]].grad, g[k]) < TOL, k
        if k.startswith("G/buf/"):
            assert rel(bufs[k[6:]], g[k]) < 1e-4, k
    D = N.define_D(nc, ndf, "dcgan", gpu_ids=[0])
    sdd = O.init_dcgan_d(72, nc, ndf)
    assert list(D.state_dict().keys()) == list(sdd.keys())
    D.load_state_dict(sdd)
    x = O.np_uniform(703, (1, nc, 128, 128)).cuda().requires_grad_(True)
    p = D.forward(x)
    assert tuple(p.shape) == (1,)
    loss = torch.nn.functional.binary_cross_entropy(p, torch.ones_like(p))
    loss.backward()
    torch.cuda.synchronize()
    assert rel(p, g["D/p"]) < TOL and abs(float(loss) - float(g["D/loss"])) < 1e-4 and rel(x.grad, g["D/dx"]) < TOL
    dparams = dict(D.named_parameters())
    for k in g.files:
        if k.startswith("D/grad/") and k.endswith(".weight"):
            assert rel(dparams[k[7:]].grad, g[k]) < TOL, k


def test_fcgan_star_small(N, golden_dir):
    """`--which_model_netG fcgan_star` (models/networks.py:543-640): two deconv chains, chain b fed the concatenation of both
    (written side by side, never concatenated here); output, latent gradient, every weight / BN gradient, running statistics."""
    g = load(golden_dir, "fcgan_star_small.npz")
    nz, ngf = 8, 4
    G = N.define_G(2, 0, ngf, "fcgan_star", "batch", False, n_layers_G=5, use_fcn=True, noise_nc=nz, gpu_ids=[0])
    sd = O.init_fcgan_star(81, nz, ngf)
    assert list(G.state_dict().keys()) == list(sd.keys())
    G.load_state_dict(sd)
    z = O.np_normal(801, (1, nz, 2, 2)).cuda().requires_grad_(True)
    y = G.forward(z)
    (y * O.np_normal(802, tuple(y.shape)).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert tuple(y.shape) == (1, 2, 128, 128)
    assert rel(y, g["y"]) < TOL and rel(z.grad, g["dz"]) < TOL
    params, bufs = dict(G.named_parameters()), G.state_dict()
    n = 0
    for k in g.files:
        if k.startswith("grad/"):
            # BN beta / gamma gradients of the deepest levels are sums of a few hundred terms that nearly cancel
            assert rel(params[k[5:]].grad, g[k]) < (TOL if k.endswith(".0.weight") else 5 * TOL), k
            n += 1
        if k.startswith("buf/"):
            assert rel(bufs[k[4:]].double(), g[k]) < 1e-4, k
    assert n == 32
    # G-step use inside a trainer: no parameter gradients wanted, second forward advances the running statistics again
    G.zero_grad_flat()
    G.compute_param_grads = False
    y2 = G.forward(z.detach().requires_grad_(True))
    y2.sum().backward()
    assert float(G._gflat.abs().max()) == 0.0
    assert int(bufs["conv0a.1.num_batches_tracked"]) == 2


def test_fcgan_g_noisesize1(N, golden_dir):
    """FCGANGenerator with --noiseSize 1 (use_fcn False, models/networks.py:503-504): first ConvT k4 s1 p0 on a 1x1 latent."""
    g = load(golden_dir, "fcgan_g_nofcn_small.npz")
    G = N.define_G(2, 0, 8, "fcgan", "instance", False, n_layers_G=5, use_fcn=False, noise_nc=8, gpu_ids=[0])
    G.load_state_dict(O.init_fcgan_g(12, 8, 2, 8, 5))
    z = O.np_normal(111, (1, 8, 1, 1)).cuda().requires_grad_(True)
    y = G.forward(z)
    assert y.shape == (1, 2, 128, 128)
    (y * O.np_normal(112, tuple(y.shape)).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, g["y"]) < TOL and rel(z.grad, g["dz"]) < TOL
    params = dict(G.named_parameters())
    for k in g.files:
        if k.startswith("grad/") and k.endswith(".weight") and params[k[5:]].dim() == 4:
            assert rel(params[k[5:]].grad, g[k]) < TOL, k


def test_generators_with_a_callers_output_activation(N):
    """The reference's generators take `activation=` (models/networks.py:362,535,708; the segmentation trainer passes the identity,
    segm_model.py:155): the fused tanh is switched off for that call and the callable runs on the raw output.  fcgan, U-Net and CRN
    against the oracle's `tanh=False` forward on the CPU: output, input gradient, the first and last weight gradients."""
    ident = lambda t: t                                                    # noqa: E731

    def grads_of(sd):
        return {k: v for k, v in sd.items() if v.is_floating_point() and "running" not in k}

    # fcgan generator, softsign as the caller's activation
    sd = O.init_fcgan_g(11, 8, 2, 8, 5)
    G = N.define_G(2, 0, 8, "fcgan", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8, gpu_ids=[0])
    G.load_state_dict(sd)
    z = O.np_normal(101, (1, 8, 2, 2))
    r = O.np_normal(102, (1, 2, 128, 128))
    zc = z.clone().requires_grad_(True)
    for v in grads_of(sd).values():
        v.requires_grad_(True)
    yo = torch.nn.functional.softsign(O.fcgan_g_forward(sd, zc, 5, tanh=False))
    (yo * r).sum().backward()
    zg = z.cuda().requires_grad_(True)
    y = G.forward(zg, activation=torch.nn.functional.softsign)
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, yo) < TOL and rel(zg.grad, zc.grad) < TOL
    P = dict(G.named_parameters())
    assert rel(P["model.0.weight"].grad, sd["model.0.weight"].grad) < TOL and rel(P["model.15.weight"].grad, sd["model.15.weight"].grad) < TOL
    y_t = G.forward(zg.detach())                                            # the next plain call is a tanh call again
    assert rel(y_t, torch.tanh(O.fcgan_g_forward(O.init_fcgan_g(11, 8, 2, 8, 5), z, 5, tanh=False, update_running=False))) < TOL

    # U-Net (no dropout / noise), identity
    sd = O.init_unet(31, 7, 2, 1, 8, -1)
    G = N.define_G(2, 1, 8, "unet_128", "instance", False, gpu_ids=[0])
    G.load_state_dict(sd)
    x = O.np_uniform(301, (1, 2, 256, 256))
    r = O.np_normal(302, (1, 1, 256, 256))
    xc = x.clone().requires_grad_(True)
    for v in grads_of(sd).values():
        v.requires_grad_(True)
    yo = O.unet_forward(sd, xc, 7, 8, tanh=False)
    (yo * r).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = G.forward(xg, activation=ident)
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert float(y.abs().max()) > 1.0 or rel(y, yo) < TOL                  # raw logits are not confined to (-1, 1)
    assert rel(y, yo) < TOL and rel(xg.grad, xc.grad) < TOL
    P = dict(G.named_parameters())
    assert rel(P["model.0.weight"].grad, sd["model.0.weight"].grad) < TOL and rel(P["model.3.weight"].grad, sd["model.3.weight"].grad) < TOL

    # CRN (bilinear, 2-layer blocks), identity
    sd = O.init_crn(41, 2, 1, 8, 8, "bilinear", 2, True)
    G = N.define_G(2, 1, 8, "crn", "instance", False, n_layers_G=5, noise_nc=8, upsample_mode="bilinear", n_layers_CRN_block=2,
                   share_label_weights=True, gpu_ids=[0])
    G.load_state_dict(sd)
    label, zz, r = O.np_uniform(401, (1, 2, 128, 128)), O.np_normal(402, (1, 8, 2, 2)), O.np_normal(403, (1, 1, 128, 128))
    lc, zc = label.clone().requires_grad_(True), zz.clone().requires_grad_(True)
    for v in grads_of(sd).values():
        v.requires_grad_(True)
    yo = O.crn_forward(sd, lc, zc, 8, "bilinear", 2, True, tanh=False)
    (yo * r).sum().backward()
    lg, zg = label.cuda().requires_grad_(True), zz.cuda().requires_grad_(True)
    y = G.forward(lg, zg, activation=ident)
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert rel(y, yo) < TOL and rel(lg.grad, lc.grad) < TOL and rel(zg.grad, zc.grad) < TOL
    last = [k for k in sd if k.endswith(".weight")][-1]
    assert rel(dict(G.named_parameters())[last].grad, sd[last].grad) < TOL


@pytest.mark.parametrize("math", ["bf16x3", "f32"])
def test_derived_weight_copies_follow_every_write(math):
    """The kernels read three copies made from the master weights (transposed fp32, split-bf16 forward / backward).  Whatever
    writes the parameters -- torch.optim.Adam on the Parameter views, load_state_dict after a backward, param.copy_,
    net.apply(weights_init), FusedAdam -- the next forward / backward must see the new weights: results equal those after an
    explicit invalidate_derived() (a stale copy would keep the old weights in y or in dx)."""
    from supervised_gan_amd import networks as N
    from supervised_gan_amd import ops
    from supervised_gan_amd.optim import FusedAdam
    prev = ops.get_math()
    ops.set_math(math)
    try:
        torch.manual_seed(0)
        D = N.define_D(2, 16, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=False, scale_factor=1).cuda()
        D.apply(N.weights_init)
        x = torch.randn(1, 2, 96, 96, device="cuda")

        def run():
            xi = x.clone().requires_grad_(True)
            y = D.forward(xi)
            y.square().sum().backward()
            torch.cuda.synchronize()
            return y.detach().clone(), xi.grad.detach().clone()

        def check(what):
            y1, dx1 = run()
            D.invalidate_derived()
            y2, dx2 = run()
            assert torch.equal(y1, y2) and float((dx1 - dx2).abs().max()) <= 1e-6 * float(dx2.abs().max()), what
            return y1, dx1

        y0, dx0 = check("initial")
        opt = torch.optim.Adam(D.model.parameters(), lr=1e-2)
        opt.step()                                        # gradients of run() are in place
        y1, dx1 = check("torch.optim.Adam step")
        assert float((y1 - y0).abs().max()) > 0 and float((dx1 - dx0).abs().max()) > 0
        sd = {k: v.clone() + 0.01 for k, v in D.state_dict().items()}
        D.load_state_dict(sd)
        y2, _ = check("load_state_dict after a backward")
        assert float((y2 - y1).abs().max()) > 0
        with torch.no_grad():
            for p in D.model.parameters():
                p.copy_(p * 0.5)
        y3, _ = check("param.copy_")
        assert float((y3 - y2).abs().max()) > 0
        D.apply(N.weights_init)
        y4, _ = check("net.apply(weights_init)")
        assert float((y4 - y3).abs().max()) > 0
        fa = FusedAdam(D.model.parameters(), lr=1e-2, betas=(0.5, 0.999))
        fa.step()
        y5, _ = check("FusedAdam step")
        assert float((y5 - y4).abs().max()) > 0
    finally:
        ops.set_math(prev)

"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Runs only in the build container (the reference never travels).  Container-only shims,
none of which touch arithmetic:
  * `skimage` stub module (util/util.py:9 imports it at module top; not installed);
  * `scale_factor` passed as an int subclass whose `/` floor-divides -- the reference is
    Python-2 code (models/networks.py:127-129, :808-811 compute sigma = scale_factor / 2);
  * `--which_model_netG fcgan` (README's `deconv` alias is rejected by define_G);
  * noise tensors are injected (Tensor.normal_ is patched to pop from a numpy-seeded queue
    for the latent shape) so the vectors do not depend on torch's RNG stream; for the U-Net the
    Dropout(0.5) masks and the optional Gaussian noise are injected the same way (F.dropout /
    Tensor.normal_ patched to numpy-seeded tensors keyed on the shape).

fp64 arbitration vectors (`python oracle/make_golden.py f64`): the SAME reference trainers, weights, inputs and injected
noise with every tensor of the model object converted to double after initialize() (dtype only, no arithmetic of the
reference is touched): `<name>_f64.npz` holds the strided gradient samples / summaries / losses of the full-size cases, so
that the GPU tests can tell an error of the HIP path from the reference's own fp32 rounding (tests/test_oracle_golden.py
`check_grads(..., f64=...)`).

Usage:  python oracle/make_golden.py            (writes tests/golden/)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import sgan_oracle as O  # noqa: E402  (only for the numpy-seeded init helpers / configs)

for name in ("skimage", "skimage.measure"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["skimage"].measure = sys.modules["skimage.measure"]

import models.networks as RN  # noqa: E402


class Py2Int(int):
    """int whose true-division floor-divides (Python-2 semantics the reference relies on)."""
    def __truediv__(self, other):
        return Py2Int(int(self) // int(other))

    def __rmul__(self, other):
        return Py2Int(int(other) * int(self))

    def __mul__(self, other):
        return Py2Int(int(self) * int(other))

    def __add__(self, other):
        return Py2Int(int(self) + int(other))

    __radd__ = __add__


def to_np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


F64 = False      # "f64" target: run the reference in double and keep only what the arbitration needs


def to_double(model):
    """Every tensor / module / tensor-type attribute of a reference trainer object -> double (after initialize())."""
    if not F64:
        return model

    def conv(v):
        if isinstance(v, torch.nn.Module):
            v.double()
            if getattr(v, "Tensor", None) is torch.FloatTensor:
                v.Tensor = torch.DoubleTensor          # GANLoss builds its target tensors with it (GANLossMultiClass keeps its LongTensor)
            for attr in ("real_label_var", "fake_label_var"):
                if getattr(v, attr, None) is not None:
                    setattr(v, attr, None)
            return v
        if torch.is_tensor(v) and v.is_floating_point():
            return v.double()
        if isinstance(v, (list, tuple)):
            return type(v)(conv(x) for x in v)
        return v
    for k, v in list(model.__dict__.items()):
        model.__dict__[k] = conv(v)
    model.Tensor = torch.DoubleTensor
    return model


def f64_exists(name):
    return F64 and os.path.exists(os.path.join(OUT, name.replace(".npz", "_f64.npz")))


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    if F64:
        name = name.replace(".npz", "_f64.npz")
        arrs = {k: v for k, v in arrs.items() if ("/sample/" in k or "/summary/" in k or "loss" in k) and not k.startswith("summary/")}
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


def ref_define_D(input_nc, ndf, n_layers, scale_factor, use_sigmoid=True):
    sf = Py2Int(scale_factor) if scale_factor > 1 else scale_factor
    return RN.define_D(input_nc, ndf, "n_layers", n_layers_D=n_layers, norm="instance",
                       use_sigmoid=use_sigmoid, scale_factor=sf, gpu_ids=[])


def ref_define_G(input_nc, ngf, n_layers, noise_nc):
    return RN.define_G(input_nc, 0, ngf, "fcgan", "instance", False, n_layers_G=n_layers,
                       use_fcn=True, noise_nc=noise_nc, gpu_ids=[])


def load_sd(net, sd):
    net.load_state_dict({k: v.detach().clone() for k, v in sd.items()})


# ---------------------------------------------------------------------------------------
def golden_gauss():
    arrs = {}
    for s in (2, 4):
        for nc in (2, 3):
            d = ref_define_D(nc, 8, 3, s)
            arrs[f"s{s}_nc{nc}"] = d.gauss_filter[0].weight.detach().numpy()
            arrs[f"s{s}_nc{nc}_pad"] = np.asarray(d.gauss_filter[0].padding)
    save("gauss.npz", **arrs)


def golden_g_small():
    ngf, nl, nz, out_nc, zs = 8, 5, 8, 2, 2
    sd = O.init_fcgan_g(11, nz, out_nc, ngf, nl)
    g = ref_define_G(out_nc, ngf, nl, nz)
    load_sd(g, sd)
    z = O.np_normal(101, (1, nz, zs, zs)).requires_grad_(True)
    r = O.np_normal(102, (1, out_nc, zs * 64, zs * 64))
    # raw conv outputs via hooks (ConvTranspose2d modules only)
    taps = {}
    convs = [m for m in g.model if m.__class__.__name__ == "ConvTranspose2d"]
    for i, m in enumerate(convs):
        m.register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(f"conv{i}", out.detach().numpy().copy()))
    y = g.forward(z)
    loss = (y * r).sum()
    loss.backward()
    arrs = {"y": y.detach().numpy(), "dz": z.grad.numpy(), "loss": np.float64(loss.item())}
    for k, p in g.named_parameters():
        arrs["grad/" + k] = p.grad.numpy()
    for k, v in g.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs["buf/" + k] = v.numpy()
    for k, v in taps.items():
        if k in ("conv0", "conv5"):
            arrs["tap/" + k] = v
    save("fcgan_g_small.npz", **arrs)


def golden_g_dropout_small():
    """FCGANGenerator with use_dropout (ConvT -> BatchNorm -> Dropout(0.5) -> ReLU in the blocks above the first, networks.py:513-521):
    the i-th dropout call of the pass gets O.dropout_mask_np(70 + i + 1, shape) (block i + 1)."""
    ngf, nl, nz, out_nc, zs = 8, 5, 8, 2, 2
    sd = O.init_fcgan_g(11, nz, out_nc, ngf, nl, use_dropout=True)
    g = RN.define_G(out_nc, 0, ngf, "fcgan", "instance", True, n_layers_G=nl, use_fcn=True, noise_nc=nz, gpu_ids=[])
    load_sd(g, sd)
    z = O.np_normal(101, (1, nz, zs, zs)).requires_grad_(True)
    r = O.np_normal(102, (1, out_nc, zs * 64, zs * 64))
    with SeqDropoutInjector(71):
        y = g.forward(z)
    loss = (y * r).sum()
    loss.backward()
    arrs = {"y": y.detach().numpy(), "dz": z.grad.numpy(), "loss": np.float64(loss.item())}
    for k, p in g.named_parameters():
        arrs["grad/" + k] = p.grad.numpy()
    for k, v in g.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs["buf/" + k] = v.numpy()
    save("fcgan_g_dropout_small.npz", **arrs)


def golden_d_small():
    ndf, nl, nc, hw = 8, 3, 2, 128
    for s in (1, 2, 4):
        sd = O.init_nlayer_d(20 + s, nc, ndf, nl, s)
        d = ref_define_D(nc, ndf, nl, s)
        load_sd(d, sd)
        x = O.np_uniform(200 + s, (1, nc, hw, hw)).requires_grad_(True)
        crit = RN.GANLoss(use_lsgan=False)
        p = d.forward(x)
        l_real = crit(p, True)
        l_fake = crit(d.forward(x), False)
        loss = l_real * 0.7 + l_fake * 0.3
        loss.backward()
        arrs = {"p": p.detach().numpy(), "l_real": np.float64(l_real.item()),
                "l_fake": np.float64(l_fake.item()), "dx": x.grad.numpy()}
        for k, prm in d.named_parameters():
            if k.startswith("model."):
                arrs["grad/" + k] = prm.grad.numpy()
        save(f"nlayer_d_small_s{s}.npz", **arrs)
    # n_layers=4 (cgan's second D), 3 input channels, lsgan head as well
    sd = O.init_nlayer_d(29, 3, ndf, 4, 1)
    d = ref_define_D(3, ndf, 4, 1, use_sigmoid=False)
    load_sd(d, sd)
    x = O.np_uniform(209, (1, 3, hw, hw)).requires_grad_(True)
    crit = RN.GANLoss(use_lsgan=True)
    p = d.forward(x)
    loss = crit(p, True)
    loss.backward()
    arrs = {"p": p.detach().numpy(), "loss": np.float64(loss.item()), "dx": x.grad.numpy()}
    for k, prm in d.named_parameters():
        arrs["grad/" + k] = prm.grad.numpy()
    save("nlayer_d_small_n4_lsgan.npz", **arrs)


def golden_d_sep():
    """NLayerDiscriminatorSep (`--which_model_netD n_layers_sep`).  ONE call of the reference is patched, like Py2Int a shim that
    restores what the code evidently means: its CPU branch runs `y_B = self.netA(x_B)` (models/networks.py:940) -- the 2-channel
    stem on the 1-channel half -- and raises; the data_parallel branch three lines above uses netB.  The patched forward below is
    that method with `self.netB(x_B)`; everything else (constructor, layers, init) is the reference's."""
    import functools
    ndf, nl, hw = 8, 3, 128

    def forward_netB(self, x):
        if self.gauss_filter is not None:
            x = self.gauss_filter(x)
        return self.model(torch.cat([self.netA(x.narrow(1, 0, 2)), self.netB(x.narrow(1, 2, 1))], dim=1))
    for s in (1, 2):
        sf = Py2Int(s) if s > 1 else s
        d = RN.define_D(3, ndf, "n_layers_sep", n_layers_D=nl, norm="instance", use_sigmoid=True, scale_factor=sf, gpu_ids=[])
        d.forward = functools.partial(forward_netB, d)
        sd = O.init_nlayer_d_sep(40 + s, ndf, nl, s)
        load_sd(d, sd)
        x = O.np_uniform(240 + s, (1, 3, hw, hw)).requires_grad_(True)
        crit = RN.GANLoss(use_lsgan=False)
        p = d.forward(x)
        loss = crit(p, True) * 0.6 + crit(d.forward(x), False) * 0.4
        loss.backward()
        arrs = {"p": p.detach().numpy(), "loss": np.float64(loss.item()), "dx": x.grad.numpy()}
        for k, prm in d.named_parameters():
            if not k.startswith("gauss"):
                arrs["grad/" + k] = prm.grad.numpy()
        save(f"nlayer_d_sep_small_s{s}.npz", **arrs)


class NoiseInjector:
    """Patch torch.Tensor.normal_ so latents of `shape` come from a numpy-seeded queue."""
    def __init__(self, shape, seed0):
        self.shape, self.seed, self.n = tuple(shape), seed0, 0
        self._orig = torch.Tensor.normal_

    def __enter__(self):
        inj = self

        def patched(t, mean=0.0, std=1.0, *a, **k):
            if tuple(t.shape) == inj.shape and mean == 0 and std == 1:
                t.copy_(O.np_normal(inj.seed + inj.n, inj.shape))
                inj.n += 1
                return t
            return inj._orig(t, mean, std, *a, **k)
        torch.Tensor.normal_ = patched
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self._orig


def build_ref_fcgan(cfg: O.FCGANConfig, seed: int, tmpdir: str):
    from options.train_options import TrainOptions
    from models.fcgan_model import FCGANModel
    argv = ["x", "--dataroot", "/nonexistent", "--name", "golden", "--model", "fcgan", "--which_direction", "A",
            "--dataset_mode", "single", "--fineSize", str(cfg.fineSize), "--batchSize", "1",
            "--input_nc", str(cfg.input_nc), "--which_model_netG", "fcgan", "--n_layers_G", str(cfg.n_layers_G),
            "--ngf", str(cfg.ngf), "--which_model_netD", "n_layers", "--n_layers_D", *map(str, cfg.n_layers_D),
            "--ndf", str(cfg.ndf), "--scale_factor", *map(str, cfg.scale_factor),
            "--lambda_D", *map(str, cfg.lambda_D), "--noise_nc", str(cfg.noise_nc), "--noiseSize", str(cfg.noiseSize),
            "--norm", "instance", "--no_dropout", "--n_update_G", str(cfg.n_update_G), "--no_lsgan",
            "--which_channel", "rg", "--gpu_ids", "-1", "--display_id", "0", "--checkpoints_dir", tmpdir,
            "--pool_size", str(cfg.pool_size)]
    old = sys.argv
    sys.argv = argv
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = old
    opt.scale_factor = [Py2Int(s) if s > 1 else s for s in opt.scale_factor]
    model = FCGANModel()
    model.initialize(opt)
    # numpy-seeded weights (same as FCGANOracle(seed))
    load_sd(model.netG, O.init_fcgan_g(seed + 1, cfg.noise_nc, cfg.input_nc, cfg.ngf, cfg.n_layers_G))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D, cfg.scale_factor)):
        load_sd(model.netD[i], O.init_nlayer_d(seed + 2 + i, cfg.input_nc, cfg.ndf, nl, sf))
    return to_double(model)


def grad_sample_idx(n, k=512):
    """Indices of the elementwise-checked sample of a flattened gradient (shared with tests)."""
    return np.unique(np.linspace(0, n - 1, num=min(n, k)).astype(np.int64))


def capture_grads(arrs, prefix, net):
    for k, p in net.named_parameters():
        if p.grad is None or not k.startswith("model."):
            continue
        gflat = p.grad.detach().reshape(-1)
        arrs[f"{prefix}/summary/{k}"] = np.asarray(O.tensor_summary(gflat))
        arrs[f"{prefix}/sample/{k}"] = gflat[torch.from_numpy(grad_sample_idx(gflat.numel()))].numpy()


def golden_step(name, cfg: O.FCGANConfig, seed: int, nsteps: int, full_params: bool):
    """Step 1 is driven through the reference model's own methods in optimize_parameters' order
    (models/fcgan_model.py:178-193) so that pre-Adam quantities can be captured; steps 2.. call
    optimize_parameters() itself."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        zshape = (1, cfg.noise_nc, cfg.noiseSize, cfg.noiseSize)
        with NoiseInjector(zshape, 5000) as inj:
            # fixed_noiseA/B draws in initialize() also pop from the queue (2 draws)
            model = build_ref_fcgan(cfg, seed, tmp)
            arrs = {"n_init_noise_draws": np.int64(inj.n)}
            losses = []
            for step in range(nsteps):
                real3 = O.np_uniform(7000 + step, (1, 3, cfg.fineSize, cfg.fineSize))
                model.set_input({"A": real3, "A_paths": ["synthetic"]})
                if step > 0:
                    model.optimize_parameters()
                else:
                    assert cfg.n_update_D == 1
                    model.forward()
                    fake1 = model.fake.detach()
                    arrs["step1/fake_summary"] = np.asarray(O.tensor_summary(fake1))
                    arrs["step1/fake_crop"] = fake1[:, :, :64, :64].numpy().copy()
                    model.optimizer_D.zero_grad()
                    model.backward_D()
                    for i, d in enumerate(model.netD):
                        capture_grads(arrs, f"step1/gradD_{i}", d)
                    arrs["step1/loss_D"] = np.asarray([float(model.loss_D_real), float(model.loss_D_fake)])
                    model.optimizer_D.step()
                    for it in range(cfg.n_update_G):
                        model.optimizer_G.zero_grad()
                        model.backward_G()
                        if it == 0:
                            capture_grads(arrs, "step1/gradG", model.netG)
                            arrs["step1/loss_G"] = np.float64(float(model.loss_G))
                        model.optimizer_G.step()
                        if cfg.n_update_G > 1:
                            model.sample_noise()
                losses.append([float(model.loss_G), float(model.loss_D_real), float(model.loss_D_fake)])
            arrs["losses"] = np.asarray(losses, dtype=np.float64)
            arrs["n_noise_draws"] = np.int64(inj.n)
        # probe: G-step gradients through the INITIAL discriminators (fresh model, same seeds, the
        # reference's own forward() + backward_G()).  Step 1's real G step runs after D's first Adam
        # update, which is sign(g)-like and therefore not reproducible to 1e-3 even by the reference.
        with NoiseInjector(zshape, 5000):
            probe = build_ref_fcgan(cfg, seed, tmp)
            probe.set_input({"A": O.np_uniform(7000, (1, 3, cfg.fineSize, cfg.fineSize)), "A_paths": ["synthetic"]})
            probe.forward()
            probe.optimizer_G.zero_grad()
            probe.optimizer_D.zero_grad()
            probe.backward_G()
            capture_grads(arrs, "probeG/gradG", probe.netG)
            for i, d in enumerate(probe.netD):
                capture_grads(arrs, f"probeG/gradD_{i}", d)      # the "wasted" D gradients of the G step
            arrs["probeG/loss_G"] = np.float64(float(probe.loss_G))
        nets = {"G": model.netG}
        for i, d in enumerate(model.netD):
            nets[f"D_{i}"] = d
        for label, net in nets.items():
            for k, v in net.state_dict().items():
                if not v.is_floating_point():
                    arrs[f"buf/{label}/{k}"] = v.numpy()
                    continue
                if full_params:
                    arrs[f"param/{label}/{k}"] = v.numpy()
                arrs[f"summary/{label}/{k}"] = np.asarray(O.tensor_summary(v))
        fake = model.fake.detach()
        arrs["fake_crop"] = fake[:, :, :64, :64].numpy() if not full_params else fake.numpy()
        arrs["fake_summary"] = np.asarray(O.tensor_summary(fake))
        save(name, **arrs)


# ---------------------------------------------------------------------------------------
# U-Net generator / cgan
# ---------------------------------------------------------------------------------------
class UnetRandomInjector:
    """Dropout masks and Gaussian noise of the U-Net from numpy (same generators as the oracle).  `count_forwards(net)`
    makes the seeds advance by 100 with every call of net.forward (the oracle's per-forward seeding)."""
    def __init__(self, mask_seed, noise_seed):
        self.mask_seed, self.noise_seed = mask_seed, noise_seed
        self.nfwd = 0
        self._orig_dropout = torch.nn.functional.dropout
        self._orig_normal = torch.Tensor.normal_

    def count_forwards(self, net):
        inner = net.forward
        m0, n0 = self.mask_seed, self.noise_seed

        def fwd(*a, **k):
            self.mask_seed, self.noise_seed = m0 + 100 * self.nfwd, n0 + 100 * self.nfwd
            self.nfwd += 1
            return inner(*a, **k)
        net.forward = fwd

    def __enter__(self):
        inj = self

        def dropout(x, p=0.5, training=True, inplace=False):
            assert p == 0.5 and training
            return x * O.dropout_mask_np(inj.mask_seed, x.shape)

        def normal_(t, mean=0.0, std=1.0, *a, **k):
            if t.dim() == 4 and t.shape[2] > 1 and mean == 0 and std == 1:
                t.copy_(O.gauss_noise_np(inj.noise_seed, t.shape))
                return t
            return inj._orig_normal(t, mean, std, *a, **k)
        torch.nn.functional.dropout = dropout
        torch.Tensor.normal_ = normal_
        return self

    def __exit__(self, *exc):
        torch.nn.functional.dropout = self._orig_dropout
        torch.Tensor.normal_ = self._orig_normal


def golden_unet_small(only=()):
    """unet_128 (7 downs) at 256x256, ngf=8: all skips + dropout; 4 skips + Gaussian noise, no dropout; --use_residual (2 -> 2 channels)."""
    for tag, kw in (("skipall_dropout", dict(use_dropout=True, num_skips=-1, add_gaussian_noise=False)),
                    ("skip4_noise", dict(use_dropout=False, num_skips=4, add_gaussian_noise=True)),
                    ("residual", dict(use_dropout=False, num_skips=-1, add_gaussian_noise=False, use_residual=True, out_nc=2)),
                    ("batchnorm", dict(use_dropout=True, num_skips=-1, add_gaussian_noise=False, norm="batch"))):
        if only and tag not in only:
            continue
        ngf, in_nc, out_nc, hw = 8, 2, kw.get("out_nc", 1), 256
        sd = O.init_unet(31, 7, in_nc, out_nc, ngf, kw["num_skips"], norm=kw.get("norm", "instance"))
        g = RN.define_G(in_nc, out_nc, ngf, "unet_128", kw.get("norm", "instance"), kw["use_dropout"], n_layers_G_skip=kw["num_skips"],
                        add_gaussian_noise=kw["add_gaussian_noise"], gaussian_sigma=0.1, use_residual=kw.get("use_residual", False), gpu_ids=[])
        load_sd(g, sd)
        x = O.np_uniform(301, (1, in_nc, hw, hw)).requires_grad_(True)
        r = O.np_normal(302, (1, out_nc, hw, hw))
        with UnetRandomInjector(40, 50):
            y = g.forward(x)
        loss = (y * r).sum()
        loss.backward()
        arrs = {"y": y.detach().numpy(), "dx": x.grad.numpy(), "loss": np.float64(loss.item())}
        for k, p in g.named_parameters():
            arrs["grad/" + k] = p.grad.numpy()
        for k, v in g.state_dict().items():
            if "running" in k or "num_batches" in k:
                arrs["buf/" + k] = v.numpy()
        save(f"unet_small_{tag}.npz", **arrs)


class SeqDropoutInjector:
    """F.dropout -> numpy-seeded masks: the i-th dropout call of the scope gets O.dropout_mask_np(seed + i, shape) (ResnetBlocks
    all drop tensors of one shape, so the mask is keyed on the call order)."""
    def __init__(self, seed):
        self.seed, self.n = seed, 0
        self._orig = torch.nn.functional.dropout

    def __enter__(self):
        inj = self

        def dropout(x, p=0.5, training=True, inplace=False):
            assert training
            m = O.dropout_mask_np(inj.seed + inj.n, x.shape, p)
            inj.n += 1
            return x * m
        torch.nn.functional.dropout = dropout
        return self

    def __exit__(self, *exc):
        torch.nn.functional.dropout = self._orig


def golden_resnet_small(only=()):
    """resnet_6blocks without dropout and resnet_9blocks with dropout at 64x64, ngf 8, 2 -> 1 channels (models/networks.py:221-311)."""
    for tag, which, nb, drop, res in (("6", "resnet_6blocks", 6, False, False), ("9_dropout", "resnet_9blocks", 9, True, False),
                                      ("6_residual", "resnet_6blocks", 6, False, True), ("6_batchnorm", "resnet_6blocks", 6, True, False)):
        if only and tag not in only:
            continue
        norm = "batch" if "batchnorm" in tag else "instance"
        ngf, in_nc, out_nc, hw = 8, 2, 2 if res else 1, 64
        sd = O.init_resnet(41, in_nc, out_nc, ngf, nb, drop, norm=norm)
        g = RN.define_G(in_nc, out_nc, ngf, which, norm, drop, use_residual=res, gpu_ids=[])
        load_sd(g, sd)
        x = O.np_uniform(311, (1, in_nc, hw, hw)).requires_grad_(True)
        r = O.np_normal(312, (1, out_nc, hw, hw))
        with SeqDropoutInjector(60):
            y = g.forward(x)
        loss = (y * r).sum()
        loss.backward()
        arrs = {"y": y.detach().numpy(), "dx": x.grad.numpy(), "loss": np.float64(loss.item())}
        for k, p in g.named_parameters():
            arrs["grad/" + k] = p.grad.numpy()
        for k, v in g.state_dict().items():
            if "running" in k or "num_batches" in k:
                arrs["buf/" + k] = v.numpy()
        save(f"resnet_small_{tag}.npz", **arrs)


def golden_autoencoder_small():
    in_nc, out_nc, nl, ngf, hw = 2, 1, 3, 8, 128
    sd = O.init_autoencoder(61, in_nc, out_nc, nl, ngf)
    g = RN.define_G(in_nc, out_nc, ngf, "autoencoder", "instance", False, n_layers_G=nl, gpu_ids=[])
    assert list(g.state_dict().keys()) == list(sd.keys()), (list(g.state_dict().keys()), list(sd.keys()))
    load_sd(g, sd)
    x = O.np_uniform(601, (1, in_nc, hw, hw)).requires_grad_(True)
    r = O.np_normal(602, (1, out_nc, hw, hw))
    y = g.forward(x)
    loss = (y * r).sum()
    loss.backward()
    arrs = {"y": y.detach().numpy(), "dx": x.grad.numpy(), "loss": np.float64(loss.item())}
    for k, p in g.named_parameters():
        arrs["grad/" + k] = p.grad.numpy()
    save("autoencoder_small.npz", **arrs)


def golden_autoencoder_dropout():
    """The autoencoder with use_dropout (Dropout(0.2) in the encoder, Dropout(0.5) in the decoder, between norm and ReLU;
    models/networks.py:441-447,468-474), masks injected in call order."""
    in_nc, out_nc, nl, ngf, hw = 2, 1, 3, 8, 128
    sd = O.init_autoencoder(62, in_nc, out_nc, nl, ngf, True)
    g = RN.define_G(in_nc, out_nc, ngf, "autoencoder", "instance", True, n_layers_G=nl, gpu_ids=[])
    assert list(g.state_dict().keys()) == list(sd.keys()), (list(g.state_dict().keys()), list(sd.keys()))
    load_sd(g, sd)
    x = O.np_uniform(611, (1, in_nc, hw, hw)).requires_grad_(True)
    r = O.np_normal(612, (1, out_nc, hw, hw))
    with SeqDropoutInjector(70):
        y = g.forward(x)
    loss = (y * r).sum()
    loss.backward()
    arrs = {"y": y.detach().numpy(), "dx": x.grad.numpy(), "loss": np.float64(loss.item())}
    for k, p in g.named_parameters():
        arrs["grad/" + k] = p.grad.numpy()
    save("autoencoder_dropout.npz", **arrs)


def golden_g_nofcn_small():
    """FCGANGenerator with --noiseSize 1 (use_fcn False): first ConvT k4 s1 p0, 1x1 latent -> 4x4 (models/networks.py:503-504)."""
    ngf, nl, nz, out_nc = 8, 5, 8, 2
    sd = O.init_fcgan_g(12, nz, out_nc, ngf, nl)
    g = RN.define_G(out_nc, 0, ngf, "fcgan", "instance", False, n_layers_G=nl, use_fcn=False, noise_nc=nz, gpu_ids=[])
    load_sd(g, sd)
    z = O.np_normal(111, (1, nz, 1, 1)).requires_grad_(True)
    y = g.forward(z)
    r = O.np_normal(112, tuple(y.shape))
    (y * r).sum().backward()
    arrs = {"y": y.detach().numpy(), "dz": z.grad.numpy()}
    for k, p in g.named_parameters():
        arrs["grad/" + k] = p.grad.numpy()
    save("fcgan_g_nofcn_small.npz", **arrs)


def golden_dcgan_small():
    """`--which_model_netG dcgan` / `--which_model_netD dcgan` (128x128): G(z) and D(x) with all gradients and BN running stats."""
    nz, nc, ngf, ndf = 8, 2, 8, 8
    g = RN.define_G(nc, 0, ngf, "dcgan", "batch", False, noise_nc=nz, gpu_ids=[])
    sd = O.init_dcgan_g(71, nz, nc, ngf)
    assert list(g.state_dict().keys()) == list(sd.keys()), (list(g.state_dict().keys()), list(sd.keys()))
    load_sd(g, sd)
    z = O.np_normal(701, (1, nz, 1, 1)).requires_grad_(True)
    y = g.forward(z)
    r = O.np_normal(702, tuple(y.shape))
    (y * r).sum().backward()
    arrs = {"G/y": y.detach().numpy(), "G/dz": z.grad.numpy()}
    for k, p in g.named_parameters():
        arrs["G/grad/" + k] = p.grad.numpy()
    for k, v in g.state_dict().items():
        if "running" in k:
            arrs["G/buf/" + k] = v.numpy().copy()
    d = RN.define_D(nc, ndf, "dcgan", gpu_ids=[])
    sdd = O.init_dcgan_d(72, nc, ndf)
    assert list(d.state_dict().keys()) == list(sdd.keys()), (list(d.state_dict().keys()), list(sdd.keys()))
    load_sd(d, sdd)
    x = O.np_uniform(703, (1, nc, 128, 128)).requires_grad_(True)
    p = d.forward(x)
    loss = torch.nn.functional.binary_cross_entropy(p, torch.ones_like(p))
    loss.backward()
    arrs.update({"D/p": p.detach().numpy(), "D/dx": x.grad.numpy(), "D/loss": np.float64(loss.item())})
    for k, q in d.named_parameters():
        arrs["D/grad/" + k] = q.grad.numpy()
    save("dcgan_small.npz", **arrs)


def golden_fcgan_star_small():
    """`--which_model_netG fcgan_star` at 2x128x128 (z 8 x 2 x 2, ngf 4): output, latent gradient, all parameter gradients,
    BN running statistics after one forward."""
    nz, ngf = 8, 4
    g = RN.define_G(2, 0, ngf, "fcgan_star", "batch", False, n_layers_G=5, use_fcn=True, noise_nc=nz, gpu_ids=[])
    sd = O.init_fcgan_star(81, nz, ngf)
    assert list(g.state_dict().keys()) == list(sd.keys()), (list(g.state_dict().keys()), list(sd.keys()))
    load_sd(g, sd)
    z = O.np_normal(801, (1, nz, 2, 2)).requires_grad_(True)
    y = g.forward(z)
    r = O.np_normal(802, tuple(y.shape))
    (y * r).sum().backward()
    arrs = {"y": y.detach().numpy(), "dz": z.grad.numpy()}
    for k, p in g.named_parameters():
        arrs["grad/" + k] = p.grad.numpy()
    for k, v in g.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs["buf/" + k] = v.numpy().copy()
    save("fcgan_star_small.npz", **arrs)


def golden_crn_small(only=()):
    """crn at 128x128 (label 2 ch, noise 8 x 2 x 2), ngf 8: ConvTranspose upsampling with 1-layer blocks, and the README's
    bilinear upsampling with 2-layer blocks; shared label block."""
    for tag, mode, nlb in (("convt_b1", "convt", 1), ("bilinear_b2", "bilinear", 2), ("bilinear_b2_batchnorm", "bilinear", 2)):
        if only and tag not in only:
            continue
        norm = "batch" if "batchnorm" in tag else "instance"
        in_nc, out_nc, nz, ngf, hw = 2, 1, 8, 8, 128
        sd = O.init_crn(41, in_nc, out_nc, nz, ngf, mode, nlb, True, norm=norm)
        g = RN.define_G(in_nc, out_nc, ngf, "crn", norm, False, n_layers_G=5, noise_nc=nz, upsample_mode=mode,
                        n_layers_CRN_block=nlb, share_label_weights=True, gpu_ids=[])
        assert list(g.state_dict().keys()) == list(sd.keys()), (list(g.state_dict().keys()), list(sd.keys()))
        load_sd(g, sd)
        label = O.np_uniform(401, (1, in_nc, hw, hw)).requires_grad_(True)
        z = O.np_normal(402, (1, nz, hw // 64, hw // 64)).requires_grad_(True)
        r = O.np_normal(403, (1, out_nc, hw, hw))
        y = g.forward(label, z)
        loss = (y * r).sum()
        loss.backward()
        arrs = {"y": y.detach().numpy(), "dlabel": label.grad.numpy(), "dz": z.grad.numpy(), "loss": np.float64(loss.item())}
        for k, p in g.named_parameters():
            arrs["grad/" + k] = p.grad.numpy()
        for k, v in g.state_dict().items():
            if "running" in k or "num_batches" in k:
                arrs["buf/" + k] = v.numpy()
        save(f"crn_small_{tag}.npz", **arrs)


def golden_crn_noise():
    """crn with --add_gaussian_noise (sigma 0.1 on the normalised output of upsample blocks 5..1), ConvTranspose upsampling,
    2-layer blocks; the noise tensors injected from numpy."""
    in_nc, out_nc, nz, ngf, hw, mode, nlb = 2, 1, 8, 8, 128, "convt", 2
    sd = O.init_crn(43, in_nc, out_nc, nz, ngf, mode, nlb, True)
    g = RN.define_G(in_nc, out_nc, ngf, "crn", "instance", False, n_layers_G=5, noise_nc=nz, upsample_mode=mode,
                    n_layers_CRN_block=nlb, share_label_weights=True, add_gaussian_noise=True, gaussian_sigma=0.1, gpu_ids=[])
    load_sd(g, sd)
    label = O.np_uniform(411, (1, in_nc, hw, hw)).requires_grad_(True)
    z = O.np_normal(412, (1, nz, hw // 64, hw // 64)).requires_grad_(True)
    r = O.np_normal(413, (1, out_nc, hw, hw))
    with UnetRandomInjector(0, 80):
        y = g.forward(label, z)
    loss = (y * r).sum()
    loss.backward()
    arrs = {"y": y.detach().numpy(), "dlabel": label.grad.numpy(), "dz": z.grad.numpy(), "loss": np.float64(loss.item())}
    for k, p in g.named_parameters():
        arrs["grad/" + k] = p.grad.numpy()
    save("crn_small_noise.npz", **arrs)


def build_ref_cgan(cfg: "O.CGANConfig", seed: int, tmpdir: str):
    from options.train_options import TrainOptions
    if cfg.variant == "cgan2":
        from models.cgan2_model import CGANModel
    else:
        from models.cgan_model import CGANModel
    netG = {7: "unet_128", 8: "unet_256"}[cfg.num_downs]
    argv = ["x", "--dataroot", "/nonexistent", "--name", "golden", "--model", cfg.variant, "--which_direction", "AtoB",
            "--dataset_mode", "unaligned" if cfg.variant == "cgan2" else "aligned", "--fineSize", str(cfg.fineSize), "--batchSize", "1",
            "--which_model_netG", netG, "--ngf", str(cfg.ngf), "--which_model_netD", "n_layers",
            "--n_layers_D", *map(str, cfg.n_layers_D), "--ndf", str(cfg.ndf), "--scale_factor", *map(str, cfg.scale_factor),
            "--lambda_D", *map(str, cfg.lambda_D), "--lambda_A", str(cfg.lambda_A), "--norm", "instance",
            "--which_channel", "rg_b", "--gpu_ids", "-1", "--display_id", "0", "--checkpoints_dir", tmpdir,
            "--pool_size", str(cfg.pool_size), "--n_layers_G_skip", str(cfg.n_layers_G_skip),
            "--n_update_G", str(cfg.n_update_G)]
    if not cfg.use_dropout:
        argv.append("--no_dropout")
    if cfg.no_lsgan:
        argv.append("--no_lsgan")
    if cfg.add_gaussian_noise:
        argv += ["--add_gaussian_noise", "--gaussian_sigma", str(cfg.gaussian_sigma)]
    if cfg.weights is not None:
        argv += ["--weights", *map(str, cfg.weights)]
    if cfg.train_D_on_fake_fake_pair:
        argv.append("--train_D_on_fake_fake_pair")
    if cfg.train_G_on_fake_fake_pair:
        argv.append("--train_G_on_fake_fake_pair")
    old = sys.argv
    sys.argv = argv
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = old
    opt.scale_factor = [Py2Int(s) if s > 1 else s for s in opt.scale_factor]
    model = CGANModel()
    model.initialize(opt)
    load_sd(model.netG, O.init_unet(seed + 1, cfg.num_downs, cfg.input_nc, cfg.output_nc, cfg.ngf, cfg.n_layers_G_skip))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D, cfg.scale_factor)):
        load_sd(model.netD[i], O.init_nlayer_d(seed + 2 + i, cfg.input_nc + cfg.output_nc, cfg.ndf, nl, sf))
    return to_double(model)


def cgan_batch(cfg, step):
    """The synthetic aligned pair of step `step` (3-channel A and B images; `rg_b` picks A[rg] -> B[b])."""
    return {"A": O.np_uniform(7100 + step, (1, 3, cfg.fineSize, cfg.fineSize)),
            "B": O.np_uniform(7200 + step, (1, 3, cfg.fineSize, cfg.fineSize)), "A_paths": ["synthetic"], "B_paths": ["synthetic"]}


def golden_cgan_step(name, cfg: "O.CGANConfig", seed: int, nsteps: int):
    import random
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        random.seed(1234)           # ImagePool's coin flips (util/image_pool.py:27-37)
        model = build_ref_cgan(cfg, seed, tmp)
        arrs = {}
        losses = []
        with UnetRandomInjector(9000, 9500) as inj:
            inj.count_forwards(model.netG)
            for step in range(nsteps):
                model.set_input(cgan_batch(cfg, step))
                if step > 0:
                    model.optimize_parameters()
                else:
                    model.forward()
                    fake1 = (model.fake_B_from_real_A if cfg.variant == "cgan2" else model.fake_B).detach()
                    arrs["step1/fake_summary"] = np.asarray(O.tensor_summary(fake1))
                    arrs["step1/fake_crop"] = fake1[:, :, :64, :64].numpy().copy()
                    if cfg.variant == "cgan2":
                        fake2 = model.fake_B_from_fake_A.detach()
                        arrs["step1/fake2_summary"] = np.asarray(O.tensor_summary(fake2))
                        arrs["step1/fake2_crop"] = fake2[:, :, :64, :64].numpy().copy()
                    model.optimizer_D.zero_grad()
                    model.backward_D()
                    for i, d in enumerate(model.netD):
                        capture_grads(arrs, f"step1/gradD_{i}", d)
                    arrs["step1/loss_D"] = np.asarray([float(model.loss_D_real), float(model.loss_D_fake)])
                    model.optimizer_D.step()
                    for _ in range(cfg.n_update_G):
                        model.optimizer_G.zero_grad()
                        model.backward_G()
                        model.optimizer_G.step()
                        if cfg.n_update_G > 1:
                            model.sample_noise()
                losses.append([float(model.loss_G), float(model.loss_G_L1), float(model.loss_D_real), float(model.loss_D_fake)])
        arrs["losses"] = np.asarray(losses, dtype=np.float64)
        probe = build_ref_cgan(cfg, seed, tmp)
        probe.set_input(cgan_batch(cfg, 0))
        with UnetRandomInjector(9000, 9500) as pinj:
            pinj.count_forwards(probe.netG)      # cgan2 runs the generator twice per forward(): seeds advance per call
            probe.forward()
            probe.optimizer_G.zero_grad()
            probe.optimizer_D.zero_grad()
            probe.backward_G()
        capture_grads(arrs, "probeG/gradG", probe.netG)
        for i, d in enumerate(probe.netD):
            capture_grads(arrs, f"probeG/gradD_{i}", d)
        arrs["probeG/loss_G"] = np.asarray([float(probe.loss_G), float(probe.loss_G_L1)])
        for label, net in [("G", model.netG)] + [(f"D_{i}", d) for i, d in enumerate(model.netD)]:
            for k, v in net.state_dict().items():
                if v.is_floating_point():
                    arrs[f"summary/{label}/{k}"] = np.asarray(O.tensor_summary(v))
        save(name, **arrs)


def build_ref_segm(cfg: "O.SegmConfig", seed: int, tmpdir: str):
    from options.train_options import TrainOptions
    from models.segm_model import SegmentationModel
    chan = "b_" + "rg"[:cfg.label_nc]
    argv = ["x", "--dataroot", "/nonexistent", "--name", "golden", "--model", "segmentation", "--which_direction", "AtoB",
            "--dataset_mode", "aligned", "--fineSize", str(cfg.fineSize), "--batchSize", "1", "--which_model_netG", {7: "unet_128", 8: "unet_256"}[cfg.num_downs],
            "--ngf", str(cfg.ngf), "--which_model_netD", "n_layers", "--n_layers_D", *map(str, cfg.n_layers_D), "--ndf", str(cfg.ndf),
            "--scale_factor", *map(str, cfg.scale_factor), "--lambda_D", *map(str, cfg.lambda_D), "--norm", "instance", "--which_channel", chan,
            "--gpu_ids", "-1", "--display_id", "0", "--checkpoints_dir", tmpdir, "--pool_size", str(cfg.pool_size), "--no_dropout",
            "--n_update_G", str(cfg.n_update_G), "--manualSeed", "1"]
    if cfg.no_lsgan:
        argv.append("--no_lsgan")
    if cfg.weights is not None:
        argv += ["--weights", *map(str, cfg.weights)]
    if cfg.use_sigmoid_ss:
        argv.append("--use_sigmoid_ss")
    if cfg.add_background_onehot:
        argv.append("--add_background_onehot")
    old = sys.argv
    sys.argv = argv
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = old
    opt.scale_factor = [Py2Int(s) if s > 1 else s for s in opt.scale_factor]
    model = SegmentationModel()
    model.initialize(opt)
    load_sd(model.netG, O.init_unet(seed + 1, cfg.num_downs, cfg.input_nc, cfg.output_nc, cfg.ngf, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D, cfg.scale_factor)):
        load_sd(model.netD[i], O.init_nlayer_d(seed + 2 + i, cfg.input_nc + cfg.output_nc, cfg.ndf, nl, sf))
    return model


def segm_batch(cfg, step):
    """Aligned pair: image channel b of A, label channels r(g) of B -- blocky maps so that argmax labels have structure."""
    n = cfg.fineSize
    lab = O.np_uniform(7400 + step, (1, 3, n // 8, n // 8))
    lab = torch.nn.functional.interpolate(lab, scale_factor=8, mode="nearest")
    return {"A": O.np_uniform(7300 + step, (1, 3, n, n)), "B": lab, "A_paths": ["synthetic"], "B_paths": ["synthetic"]}


def golden_segm_step(name, cfg: "O.SegmConfig", seed: int, nsteps: int):
    """`--model segmentation` (models/segm_model.py): losses of every step, the first step's logits and discriminator gradients,
    parameter summaries after the last step."""
    import random
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        model = build_ref_segm(cfg, seed, tmp)
        random.seed(1234)           # ImagePool's coin flips; after initialize(), which reseeds `random` from --manualSeed (:27)
        arrs, losses = {}, []
        for step in range(nsteps):
            model.set_input(segm_batch(cfg, step))
            if step > 0:
                model.optimize_parameters()
            else:
                model.forward()
                arrs["step1/logit_summary"] = np.asarray(O.tensor_summary(model.logit.detach()))
                arrs["step1/logit_crop"] = model.logit.detach()[:, :, :64, :64].numpy().copy()
                arrs["step1/label_hist"] = np.bincount(model.label.numpy().reshape(-1), minlength=cfg.output_nc)
                model.optimizer_D.zero_grad()
                model.backward_D()
                for i, d in enumerate(model.netD):
                    capture_grads(arrs, f"step1/gradD_{i}", d)
                model.optimizer_D.step()
                for _ in range(cfg.n_update_G):
                    model.optimizer_G.zero_grad()
                    model.backward_G()
                    if _ == 0:
                        capture_grads(arrs, "step1/gradG", model.netG)
                    model.optimizer_G.step()
                    if cfg.n_update_G > 1:
                        model.sample_noise()
            losses.append([float(model.loss_G_CE), float(model.loss_G_GAN), float(model.loss_D_real), float(model.loss_D_fake)])
        arrs["losses"] = np.asarray(losses, dtype=np.float64)
        for label, net in [("G", model.netG)] + [(f"D_{i}", d) for i, d in enumerate(model.netD)]:
            for k, v in net.state_dict().items():
                if v.is_floating_point():
                    arrs[f"summary/{label}/{k}"] = np.asarray(O.tensor_summary(v))
        save(name, **arrs)


def build_ref_segm_cycle(cfg: "O.SegmCycleConfig", seed: int, tmpdir: str):
    from options.train_options import TrainOptions
    from models.segm_cycle_model import SegmentationCycleModel
    L = lambda xs: [str(x) for x in xs]
    unet = {7: "unet_128", 8: "unet_256"}
    argv = ["x", "--dataroot", "/nonexistent", "--name", "golden", "--model", "segmentation_cycle", "--which_direction", "AtoB",
            "--dataset_mode", "aligned", "--fineSize", str(cfg.fineSize), "--batchSize", "1", "--which_channel", "b_" + "rg"[:cfg.label_nc],
            "--which_model_netG1", unet[cfg.num_downs1], "--ngf1", str(cfg.ngf1), "--which_model_netG2", unet[cfg.num_downs2], "--ngf2", str(cfg.ngf2),
            "--which_model_netD2", "n_layers", "--n_layers_D2", *L(cfg.n_layers_D2), "--ndf2", str(cfg.ndf2), "--scale_factor2", *L(cfg.scale_factor2),
            "--lambda_D2", *L(cfg.lambda_D2), "--lambda_A", str(cfg.lambda_A), "--lambda_B", str(cfg.lambda_B), "--lambda_A_cycle", str(cfg.lambda_A_cycle),
            "--lr1", str(cfg.lr1), "--lr2", str(cfg.lr2), "--norm", "instance", "--no_dropout1", "--no_dropout2", "--n_update_G", str(cfg.n_update_G),
            "--pool_size", str(cfg.pool_size), "--gpu_ids", "-1", "--display_id", "0", "--checkpoints_dir", tmpdir, "--manualSeed", "1"]
    if cfg.no_lsgan2:
        argv.append("--no_lsgan2")
    if cfg.weights is not None:
        argv += ["--weights", *L(cfg.weights)]
    if cfg.use_sigmoid_ss:
        argv.append("--use_sigmoid_ss")
    if cfg.add_background_onehot:
        argv.append("--add_background_onehot")
    old = sys.argv
    sys.argv = argv
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = old
    opt.scale_factor2 = [Py2Int(s) if s > 1 else s for s in opt.scale_factor2]
    model = SegmentationCycleModel()
    model.initialize(opt)
    load_sd(model.netG1, O.init_unet(seed + 1, cfg.num_downs1, cfg.input_nc, cfg.num_classes, cfg.ngf1, -1))
    load_sd(model.netG2, O.init_unet(seed + 2, cfg.num_downs2, cfg.num_classes, cfg.input_nc, cfg.ngf2, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D2, cfg.scale_factor2)):
        load_sd(model.netD2[i], O.init_nlayer_d(seed + 3 + i, cfg.input_nc + cfg.num_classes, cfg.ndf2, nl, sf))
    return model


def golden_segm_cycle(name, cfg: "O.SegmCycleConfig", seed: int, nsteps: int):
    """`--model segmentation_cycle`: the six loss terms of every step, first-step logits / G2 outputs, parameter summaries at the end."""
    import random
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        model = build_ref_segm_cycle(cfg, seed, tmp)
        random.seed(1234)
        arrs, losses = {}, []
        for step in range(nsteps):
            model.set_input(segm_batch(cfg, step))
            model.optimize_parameters()
            if step == 0:
                arrs["step1/logit_crop"] = model.logit.detach()[:, :, :64, :64].numpy().copy()
                arrs["step1/fake_A_crop"] = model.fake_A.detach()[:, :, :64, :64].numpy().copy()
                arrs["step1/recon_A_crop"] = model.recon_A.detach()[:, :, :64, :64].numpy().copy()
            losses.append([float(model.loss_G1_CE), float(model.loss_G2_GAN), float(model.loss_G_L1), float(model.loss_G_cycle),
                           float(model.loss_D2_real), float(model.loss_D2_fake)])
        arrs["losses"] = np.asarray(losses, dtype=np.float64)
        for label, net in [("G1", model.netG1), ("G2", model.netG2)] + [(f"D2_{i}", d) for i, d in enumerate(model.netD2)]:
            for k, v in net.state_dict().items():
                if v.is_floating_point():
                    arrs[f"summary/{label}/{k}"] = np.asarray(O.tensor_summary(v))
        save(name, **arrs)


# ---------------------------------------------------------------------------------------
# cgan_cycle
# ---------------------------------------------------------------------------------------
def build_ref_cgan_cycle(cfg: "O.CGANCycleConfig", seed: int, tmpdir: str):
    from options.train_options import TrainOptions
    if cfg.variant == "cgan2_cycle":
        from models.cgan2_cycle_model import CGANCycleModel
    else:
        from models.cgan_cycle_model import CGANCycleModel
    L = lambda xs: [str(x) for x in xs]
    unet = {7: "unet_128", 8: "unet_256"}
    argv = ["x", "--dataroot", "/nonexistent", "--name", "golden", "--model", cfg.variant, "--which_direction", "AtoB",
            "--dataset_mode", "unaligned" if cfg.variant == "cgan2_cycle" else "aligned", "--fineSize", str(cfg.fineSize), "--batchSize", "1", "--which_channel", "rg_b",
            "--lambda_fake_cycle", str(cfg.lambda_fake_cycle),
            "--which_model_netG1", unet[cfg.num_downs1], "--ngf1", str(cfg.ngf1), "--which_model_netG2", unet[cfg.num_downs2],
            "--ngf2", str(cfg.ngf2), "--which_model_netD1", "n_layers", "--n_layers_D1", *L(cfg.n_layers_D1), "--ndf1", str(cfg.ndf1),
            "--scale_factor1", *L(cfg.scale_factor1), "--lambda_D1", *L(cfg.lambda_D1), "--lambda_A", str(cfg.lambda_A),
            "--lambda_B", str(cfg.lambda_B), "--lambda_A_cycle", str(cfg.lambda_A_cycle), "--lr1", str(cfg.lr1), "--lr2", str(cfg.lr2),
            "--norm", "instance", "--no_dropout1", "--no_dropout2", "--n_update_G", str(cfg.n_update_G), "--pool_size", str(cfg.pool_size),
            "--gpu_ids", "-1", "--display_id", "0", "--checkpoints_dir", tmpdir]
    if cfg.no_lsgan1:
        argv.append("--no_lsgan1")
    if cfg.weights is not None:
        argv += ["--weights", *L(cfg.weights)]
    if cfg.train_D_on_fake_fake_pair:
        argv.append("--train_D_on_fake_fake_pair")
    if cfg.train_G_on_fake_fake_pair:
        argv.append("--train_G_on_fake_fake_pair")
    old = sys.argv
    sys.argv = argv
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = old
    opt.scale_factor1 = [Py2Int(s) if s > 1 else s for s in opt.scale_factor1]
    model = CGANCycleModel()
    model.initialize(opt)
    load_sd(model.netG1, O.init_unet(seed + 1, cfg.num_downs1, cfg.input_nc, cfg.output_nc, cfg.ngf1, -1))
    load_sd(model.netG2, O.init_unet(seed + 2, cfg.num_downs2, cfg.output_nc, cfg.input_nc, cfg.ngf2, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D1, cfg.scale_factor1)):
        load_sd(model.netD1[i], O.init_nlayer_d(seed + 3 + i, cfg.input_nc + cfg.output_nc, cfg.ndf1, nl, sf))
    return to_double(model)


def golden_cgan_cycle(name, cfg: "O.CGANCycleConfig", seed: int, nsteps: int):
    import random
    import tempfile
    from collections import namedtuple
    with tempfile.TemporaryDirectory() as tmp:
        arrs = {}
        C = namedtuple("C", "fineSize")(cfg.fineSize)
        probe = build_ref_cgan_cycle(cfg, seed, tmp)
        random.seed(1234)
        probe.set_input(cgan_batch(C, 0))
        probe.forward()
        two = cfg.variant == "cgan2_cycle"
        names = {"fake_B": "fake_B_from_real_A", "fake_A": "fake_A_from_real_B", "recon_A": "recon_real_A", "recon_fake_A": "recon_fake_A"} if two \
            else {"fake_B": "fake_B", "fake_A": "fake_A", "recon_A": "recon_A"}
        for key, attr in names.items():
            t = getattr(probe, attr).detach()
            arrs[f"probe/{key}_summary"] = np.asarray(O.tensor_summary(t))
            arrs[f"probe/{key}_crop"] = t[:, :, :64, :64].numpy().copy()
        probe.optimizer_D1.zero_grad()
        probe.backward_D1()
        for i, d in enumerate(probe.netD1):
            capture_grads(arrs, f"probe/gradD_{i}", d)
        arrs["probe/loss_D"] = np.asarray([float(probe.loss_D_real), float(probe.loss_D_fake)])
        probe.optimizer_D1.zero_grad()
        probe.optimizer_G.zero_grad()
        probe.backward_G()
        capture_grads(arrs, "probe/gradG1", probe.netG1)
        capture_grads(arrs, "probe/gradG2", probe.netG2)
        arrs["probe/loss_G"] = np.asarray([float(probe.loss_G), float(probe.loss_G_GAN), float(probe.loss_G_L1), float(probe.loss_G_CE),
                                           float(probe.loss_G_real_cycle if two else probe.loss_G_cycle)])
        random.seed(1234)
        model = build_ref_cgan_cycle(cfg, seed, tmp)
        losses = []
        for step in range(nsteps):
            model.set_input(cgan_batch(C, step))
            model.optimize_parameters()
            losses.append([float(model.loss_G), float(model.loss_G_real_cycle if two else model.loss_G_cycle), float(model.loss_D)])
        arrs["losses"] = np.asarray(losses, dtype=np.float64)
        save(name, **arrs)


# ---------------------------------------------------------------------------------------
# twostage_cycle (DSGAN)
# ---------------------------------------------------------------------------------------
class TwoNoiseInjector:
    """Latents of G1 / G2 from numpy: the k-th draw of shape1 is np_normal(5000 + k), of shape2 np_normal(6000 + k)."""
    def __init__(self, shape1, shape2):
        self.shapes = {tuple(shape1): [5000, 0], tuple(shape2): [6000, 0]}
        self._orig = torch.Tensor.normal_

    def __enter__(self):
        inj = self

        def patched(t, mean=0.0, std=1.0, *a, **k):
            ent = inj.shapes.get(tuple(t.shape))
            if ent is not None and mean == 0 and std == 1:
                t.copy_(O.np_normal(ent[0] + ent[1], tuple(t.shape)))
                ent[1] += 1
                return t
            return inj._orig(t, mean, std, *a, **k)
        torch.Tensor.normal_ = patched
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self._orig


def build_ref_twostage(cfg: "O.TwoStageConfig", seed: int, tmpdir: str):
    from options.train_options import TrainOptions
    from models.twostage_cycle_model import TwoStageCycleModel
    from models.twostage_model import TwoStageModel
    L = lambda xs: [str(x) for x in xs]
    argv = ["x", "--dataroot", "/nonexistent", "--name", "golden", "--model", "twostage_cycle" if cfg.cycle else "twostage_factd" if cfg.factd else "twostage", "--which_direction", "AtoB",
            "--dataset_mode", "aligned", "--fineSize", str(cfg.fineSize), "--transform_1to2", cfg.transform_1to2, "--batchSize", "1",
            "--which_channel", "rg_b", "--which_model_netG1", "fcgan", "--n_layers_G1", str(cfg.n_layers_G1), "--ngf1", str(cfg.ngf1),
            "--which_model_netD1", "n_layers", "--n_layers_D1", *L(cfg.n_layers_D1), "--ndf1", str(cfg.ndf1),
            "--scale_factor1", *L(cfg.scale_factor1), "--lambda_D1", *L(cfg.lambda_D1), "--which_model_netG2", "crn",
            "--ngf2", str(cfg.ngf2), "--upsample_mode2", cfg.upsample_mode2, "--n_layers_CRN_block2", str(cfg.n_layers_CRN_block2),
            "--which_model_netF2", "unet_128", "--nff2", str(cfg.nff2), "--which_model_netD2", "n_layers",
            "--n_layers_D2", *L(cfg.n_layers_D2), "--ndf2", str(cfg.ndf2), "--scale_factor2", *L(cfg.scale_factor2),
            "--lambda_D2", *L(cfg.lambda_D2), "--lambda_A", str(cfg.lambda_A), "--lambda_B", str(cfg.lambda_B),
            "--lambda_A_cycle", str(cfg.lambda_A_cycle), "--lambda_fake_cycle", str(cfg.lambda_fake_cycle),
            "--noise_nc1", str(cfg.noise_nc1), "--noiseSize1", str(cfg.noiseSize1), "--noise_nc2", str(cfg.noise_nc2),
            "--noiseSize2", str(cfg.noiseSize2), "--norm", "instance", "--no_dropout1", "--no_dropout2", "--n_update_G", "1",
            "--GAN_losses_D2", *cfg.GAN_losses_D2, "--GAN_losses_G2", *cfg.GAN_losses_G2, "--pool_size", str(cfg.pool_size),
            "--gpu_ids", "-1", "--display_id", "0", "--checkpoints_dir", tmpdir]
    if cfg.no_lsgan1:
        argv.append("--no_lsgan1")
    if cfg.no_lsgan2:
        argv.append("--no_lsgan2")
    if cfg.weights is not None:
        argv += ["--weights", *L(cfg.weights)]
    if not cfg.cycle:
        argv += ["--lambda_G1", str(cfg.lambda_G1), "--lambda_G2", str(cfg.lambda_G2)]
    if cfg.use_multi_class_GAN:
        argv.append("--use_multi_class_GAN")
    old = sys.argv
    sys.argv = argv
    try:
        opt = TrainOptions().parse()
    finally:
        sys.argv = old
    opt.scale_factor1 = [Py2Int(s) if s > 1 else s for s in opt.scale_factor1]
    opt.scale_factor2 = [Py2Int(s) if s > 1 else s for s in opt.scale_factor2]
    if cfg.factd:
        from models.twostage_factD_model import TwoStageModel as TwoStageFactDModel
        assert not cfg.cycle
    model = TwoStageCycleModel() if cfg.cycle else TwoStageFactDModel() if cfg.factd else TwoStageModel()
    model.initialize(opt)
    load_sd(model.netG1, O.init_fcgan_g(seed + 1, cfg.noise_nc1, cfg.input_nc, cfg.ngf1, cfg.n_layers_G1))
    load_sd(model.netG2, O.init_crn(seed + 2, cfg.input_nc, cfg.output_nc, cfg.noise_nc2, cfg.ngf2, cfg.upsample_mode2,
                                    cfg.n_layers_CRN_block2, True))
    if cfg.cycle:
        load_sd(model.netF2, O.init_unet(seed + 3, 7, cfg.output_nc, cfg.input_nc, cfg.nff2, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D1, cfg.scale_factor1)):
        load_sd(model.netD1[i], O.init_nlayer_d(seed + 10 + i, cfg.input_nc, cfg.ndf1, nl, sf))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D2, cfg.scale_factor2)):
        load_sd(model.netD2[i], O.init_nlayer_d(seed + 20 + i, cfg.input_nc + cfg.output_nc, cfg.ndf2, nl, sf,
                                                3 if cfg.use_multi_class_GAN else 1))
    return to_double(model)


def golden_twostage(name, cfg: "O.TwoStageConfig", seed: int, nsteps: int):
    import random
    if f64_exists(name):
        return
    import tempfile
    z1 = (1, cfg.noise_nc1, cfg.noiseSize1, cfg.noiseSize1)
    z2 = (1, cfg.noise_nc2, cfg.noiseSize2, cfg.noiseSize2)
    assert z1 != z2
    errs = lambda m: ([float(m.loss_G2_GAN), float(m.loss_G2_real_cycle), float(m.loss_G2_fake_cycle), float(m.loss_D2),
                       float(m.loss_G1_GAN), float(m.loss_D1)] if cfg.cycle else
                      [float(m.loss_G2_GAN), float(m.loss_D2), float(m.loss_G1_GAN), float(m.loss_D1)])
    with tempfile.TemporaryDirectory() as tmp:
        arrs = {}
        # probe: every gradient of the step on the initial weights
        random.seed(1234)
        with TwoNoiseInjector(z1, z2):
            m = build_ref_twostage(cfg, seed, tmp)
            m.set_input(cgan_batch(cfg, 0))
            m.forward()
            outs = [("fake_A", m.fake_A), ("fake_B_from_fake_A", m.fake_B_from_fake_A)] + ([("recon_fake_A", m.recon_fake_A)] if cfg.cycle else [])
            for key, t in outs:
                arrs[f"probe/{key}_summary"] = np.asarray(O.tensor_summary(t))
                arrs[f"probe/{key}_crop"] = t.detach()[:, :, :64, :64].numpy().copy()
            m.optimizer_D1.zero_grad()
            m.backward_D1()
            for i, d in enumerate(m.netD1):
                capture_grads(arrs, f"probe/gradD1_{i}", d)
            m.optimizer_D2.zero_grad()
            m.backward_D2()
            for i, d in enumerate(m.netD2):
                capture_grads(arrs, f"probe/gradD2_{i}", d)
            m.optimizer_G.zero_grad()
            m.backward_G()
            for tag, net in [("G1", m.netG1), ("G2", m.netG2)] + ([("F2", m.netF2)] if cfg.cycle else []):
                for k, p in net.named_parameters():
                    gflat = p.grad.detach().reshape(-1)
                    arrs[f"probe/grad{tag}/summary/{k}"] = np.asarray(O.tensor_summary(gflat))
                    arrs[f"probe/grad{tag}/sample/{k}"] = gflat[torch.from_numpy(grad_sample_idx(gflat.numel()))].numpy()
            arrs["probe/losses"] = np.asarray(errs(m))
            arrs["probe/loss_G"] = np.float64(float(m.loss_G))
        # trajectory
        random.seed(1234)
        with TwoNoiseInjector(z1, z2):
            m = build_ref_twostage(cfg, seed, tmp)
            losses = []
            for step in range(nsteps):
                m.set_input(cgan_batch(cfg, step))
                m.optimize_parameters()
                losses.append(errs(m))
            arrs["losses"] = np.asarray(losses, dtype=np.float64)
        save(name, **arrs)


def main():
    global F64
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if only and only[0] == "f64":      # fp64 arbitration vectors of the full-size (README) cases
        F64 = True
        which = only[1:] or ["fcgan", "cgan", "twostage", "small"]
        if "fcgan" in which:
            golden_step("fcgan_step_full.npz", O.FCGANConfig(), seed=0, nsteps=3, full_params=False)
            golden_step("fcgan_step_full_nug1.npz", O.FCGANConfig(n_update_G=1), seed=0, nsteps=2, full_params=False)
        if "cgan" in which:
            golden_cgan_step("cgan_step_full.npz", O.CGANConfig(**O.CGAN_README), seed=0, nsteps=2)
        if "twostage" in which:
            golden_twostage("twostage_full.npz", O.TwoStageConfig(), seed=0, nsteps=2)
        if "small" in which:      # the small cases whose chains end in 2x2 / 4x4 normalisations (gated through the fp64 run too)
            tw = dict(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8)
            ff = dict(GAN_losses_D2=("real_fake", "fake_fake"), GAN_losses_G2=("real_fake", "fake_fake"))
            golden_twostage("twostage_small.npz", O.TwoStageConfig(**tw, **ff, weights=(2.0, 5.0)), seed=0, nsteps=3)
            golden_twostage("twostage_nocycle_small.npz", O.TwoStageConfig(**tw, **ff, cycle=False, lambda_G1=0.7, lambda_G2=1.3), seed=0, nsteps=3)
            golden_twostage("twostage_multiclass_small.npz",
                            O.TwoStageConfig(**tw, **ff, weights=(2.0, 5.0), use_multi_class_GAN=True, no_lsgan2=True, n_layers_D2=(3, 4),
                                             scale_factor2=(1, 2), lambda_D2=(0.6, 0.4)), seed=0, nsteps=3)
            golden_twostage("twostage_factd_small.npz",
                            O.TwoStageConfig(**tw, **ff, n_layers_D1=(4, 4), n_layers_D2=(3, 3), scale_factor2=(1, 2), lambda_D2=(0.6, 0.4),
                                             no_lsgan2=True, cycle=False, factd=True, lambda_G1=0.7, lambda_G2=1.3), seed=0, nsteps=3)
            golden_cgan_cycle("cgan_cycle_small.npz", O.CGANCycleConfig(), 0, 3)
            golden_cgan_cycle("cgan_cycle_small_d34.npz", O.CGANCycleConfig(scale_factor1=(1, 1), n_layers_D1=(3, 4), weights=None, no_lsgan1=False), 0, 2)
            two = dict(variant="cgan2_cycle", lambda_fake_cycle=0.5)
            golden_cgan_cycle("cgan2_cycle_small.npz", O.CGANCycleConfig(**two), 0, 2)
            golden_cgan_cycle("cgan2_cycle_small_fakefake.npz",
                              O.CGANCycleConfig(**dict(two, train_D_on_fake_fake_pair=True, train_G_on_fake_fake_pair=True, n_update_G=2)), 0, 2)
        return
    if not only or "segmentation_cycle" in only:
        golden_segm_cycle("segm_cycle_small.npz", O.SegmCycleConfig(weights=(1.0, 3.0), lambda_A=2.0, lambda_B=0.5, lambda_A_cycle=1.5, lr2=1e-4), 0, 3)
    if not only or "segmentation" in only:
        small = dict(num_downs=7, ngf=8, ndf=8, fineSize=256, n_layers_D=(3, 3), scale_factor=(1, 2), lambda_D=(0.6, 0.4), no_lsgan=True)
        golden_segm_step("segm_step_small.npz", O.SegmConfig(weights=(1.0, 3.0), n_update_G=2, **small), 0, 3)
        golden_segm_step("segm_step_small_sigmoid_bg.npz",
                         O.SegmConfig(use_sigmoid_ss=True, add_background_onehot=True, weights=(2.0, 1.0, 0.5), **small), 0, 2)
    if not only or "fcgan_star" in only:
        golden_fcgan_star_small()
    if not only or "resnet" in only:
        golden_resnet_small()
    if "g_dropout" in only:
        golden_g_dropout_small()
    if "residual" in only:       # only the --use_residual vectors (added after the others; same generators)
        golden_resnet_small(("6_residual",))
        golden_unet_small(("residual",))
    if "unet_batchnorm" in only:
        golden_unet_small(("batchnorm",))
    if "resnet_batchnorm" in only:
        golden_resnet_small(("6_batchnorm",))
    if "crn_batchnorm" in only:
        golden_crn_small(("bilinear_b2_batchnorm",))
    if not only or "autoencoder" in only:
        golden_autoencoder_small()
        golden_autoencoder_dropout()
    if not only or "sep" in only:
        golden_d_sep()
    if not only or "dcgan" in only:
        golden_dcgan_small()
        golden_g_nofcn_small()
    if not only or "cgan2_cycle" in only:
        two = dict(variant="cgan2_cycle", lambda_fake_cycle=0.5)
        golden_cgan_cycle("cgan2_cycle_small.npz", O.CGANCycleConfig(**two), 0, 2)
        golden_cgan_cycle("cgan2_cycle_small_fakefake.npz",
                          O.CGANCycleConfig(**dict(two, train_D_on_fake_fake_pair=True, train_G_on_fake_fake_pair=True, n_update_G=2)), 0, 2)
    if not only or "cgan_cycle" in only:
        golden_cgan_cycle("cgan_cycle_small.npz", O.CGANCycleConfig(), 0, 3)
        # (n_update_G > 1 is not a case: the reference's sample_noise does not regenerate fake_A, so its second backward_G walks a
        # freed graph and raises)
        golden_cgan_cycle("cgan_cycle_small_d34.npz", O.CGANCycleConfig(scale_factor1=(1, 1), n_layers_D1=(3, 4), weights=None, no_lsgan1=False), 0, 2)
    if not only or "cgan2" in only:
        small = dict(num_downs=7, ngf=8, ndf=8, fineSize=256, weights=(2.0, 5.0), variant="cgan2", n_layers_D=(3, 3), scale_factor=(1, 2),
                     no_lsgan=True, n_update_G=2)
        golden_cgan_step("cgan2_step_small.npz", O.CGANConfig(**small), 0, 2)
        golden_cgan_step("cgan2_step_small_fakefake.npz",
                         O.CGANConfig(**dict(small, train_D_on_fake_fake_pair=True, train_G_on_fake_fake_pair=True)), 0, 2)
    if not only or "crn" in only:
        golden_crn_small()
        golden_crn_noise()
    if not only or "multiclass" in only:
        golden_twostage("twostage_multiclass_small.npz",
                        O.TwoStageConfig(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8,
                                         GAN_losses_D2=("real_fake", "fake_fake"), GAN_losses_G2=("real_fake", "fake_fake"),
                                         weights=(2.0, 5.0), use_multi_class_GAN=True, no_lsgan2=True, n_layers_D2=(3, 4),
                                         scale_factor2=(1, 2), lambda_D2=(0.6, 0.4)), seed=0, nsteps=3)
    if not only or "factd" in only:
        golden_twostage("twostage_factd_small.npz",
                        O.TwoStageConfig(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8, n_layers_D1=(4, 4),
                                         n_layers_D2=(3, 3), scale_factor2=(1, 2), lambda_D2=(0.6, 0.4), no_lsgan2=True,
                                         GAN_losses_D2=("real_fake", "fake_fake"), GAN_losses_G2=("real_fake", "fake_fake"), cycle=False,
                                         factd=True, lambda_G1=0.7, lambda_G2=1.3), seed=0, nsteps=3)
    if not only or "twostage" in only:
        golden_twostage("twostage_small.npz", O.TwoStageConfig(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8,
                                                               ndf2=8, GAN_losses_D2=("real_fake", "fake_fake"),
                                                               GAN_losses_G2=("real_fake", "fake_fake"), weights=(2.0, 5.0)),
                        seed=0, nsteps=3)
        golden_twostage("twostage_full.npz", O.TwoStageConfig(), seed=0, nsteps=2)     # BASELINE configs[4], README.md:18
        golden_twostage("twostage_nocycle_small.npz", O.TwoStageConfig(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8,
                                                                       ndf2=8, GAN_losses_D2=("real_fake", "fake_fake"),
                                                                       GAN_losses_G2=("real_fake", "fake_fake"), cycle=False, lambda_G1=0.7,
                                                                       lambda_G2=1.3), seed=0, nsteps=3)
    if not only or "cgan" in only:
        golden_unet_small()
        golden_cgan_step("cgan_step_small.npz", O.CGANConfig(num_downs=7, ngf=8, ndf=8, fineSize=256, weights=(2.0, 5.0)),
                         seed=0, nsteps=3)
        golden_cgan_step("cgan_step_full.npz", O.CGANConfig(**O.CGAN_README), seed=0, nsteps=2)   # BASELINE configs[2]
    if only and "fcgan" not in only:
        return
    golden_gauss()
    golden_g_small()
    golden_d_small()
    small = O.FCGANConfig(ngf=8, ndf=8, noiseSize=2, n_update_G=2)
    golden_step("fcgan_step_small.npz", small, seed=0, nsteps=3, full_params=True)
    full = O.FCGANConfig()  # README config: 512x512, ngf=ndf=32, n_update_G=2
    golden_step("fcgan_step_full.npz", full, seed=0, nsteps=3, full_params=False)
    full1 = O.FCGANConfig(n_update_G=1)
    golden_step("fcgan_step_full_nug1.npz", full1, seed=0, nsteps=2, full_params=False)


if __name__ == "__main__":
    main()

"""Pin the CPU oracle (oracle/sgan_oracle.py) against vectors produced by the real reference
(oracle/make_golden.py).  CPU-only; no GPU, no /root/reference needed."""
import os

import numpy as np
import pytest
import torch

import sgan_oracle as O

TOL = 1e-3   # north-star tolerance: max|a-b| / max|b|, fp32
TIGHT = 2e-5  # what two fp32 CPU runs of the same math actually achieve


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a):
    return a.detach() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))


def rel(a, b):
    return O.rel_err(_t(a), _t(b))


def test_gauss_filters(golden_dir):
    g = load(golden_dir, "gauss.npz")
    for s in (2, 4):
        for nc in (2, 3):
            w = O.gauss_filter_weight(nc, s).numpy()
            assert w.shape == g[f"s{s}_nc{nc}"].shape
            assert np.abs(w - g[f"s{s}_nc{nc}"]).max() < 1e-7
            assert tuple(g[f"s{s}_nc{nc}_pad"]) == (2 * (s // 2),) * 2
            assert w.shape[-1] == 4 * (s // 2) + 1


def test_fcgan_g_small(golden_dir):
    g = load(golden_dir, "fcgan_g_small.npz")
    sd = O.init_fcgan_g(11, 8, 2, 8, 5)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    z = O.np_normal(101, (1, 8, 2, 2)).requires_grad_(True)
    r = O.np_normal(102, (1, 2, 128, 128))
    taps = {}
    y = O.fcgan_g_forward(sd, z, 5, taps=taps)
    (y * r).sum().backward()
    assert rel(y, g["y"]) < TIGHT
    assert rel(taps["conv0"], g["tap/conv0"]) < TIGHT
    assert rel(taps["conv5"], g["tap/conv5"]) < TIGHT
    assert rel(z.grad, g["dz"]) < TIGHT
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            # conv biases in front of BatchNorm have analytically-zero gradient: compare on the
            # scale of the weight gradient of the same layer instead of their own (noise) scale
            if name.endswith(".bias") and name.replace(".bias", ".weight") in sd and sd[name.replace(".bias", ".weight")].dim() == 4:
                scale = np.abs(g["grad/" + name.replace(".bias", ".weight")]).max()
                assert np.abs(sd[name].grad.numpy() - g[k]).max() < TOL * scale
            else:
                assert rel(sd[name].grad, g[k]) < TIGHT * 10, name
        if k.startswith("buf/"):
            name = k[4:]
            assert rel(sd[name].double(), g[k].astype(np.float64)) < TIGHT, name


def test_fcgan_g_dropout_small(golden_dir):
    """FCGANGenerator with use_dropout (ConvT -> BatchNorm -> Dropout(0.5) -> ReLU above the first block, networks.py:513-521; the
    Sequential indices shift by one per block) restated, against the reference with the same injected masks."""
    g = load(golden_dir, "fcgan_g_dropout_small.npz")
    sd = O.init_fcgan_g(11, 8, 2, 8, 5, use_dropout=True)
    assert {"grad/" + k for k, v in sd.items() if v.is_floating_point() and "running" not in k} == {k for k in g.files if k.startswith("grad/")}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    z = O.np_normal(101, (1, 8, 2, 2)).requires_grad_(True)
    y = O.fcgan_g_forward(sd, z, 5, use_dropout=True, mask_seed=70)
    (y * O.np_normal(102, (1, 2, 128, 128))).sum().backward()
    assert rel(y, g["y"]) < TIGHT and rel(z.grad, g["dz"]) < TIGHT
    for k in g.files:
        if k.startswith("grad/") and (sd[k[5:]].dim() == 4 or "model.1." in k or not k.endswith(".bias") or sd[k[5:].replace(".bias", ".weight")].dim() == 1):
            assert rel(sd[k[5:]].grad, g[k]) < TIGHT * 10, k
        if k.startswith("buf/"):
            assert rel(sd[k[4:]].double(), g[k].astype(np.float64)) < TIGHT, k


@pytest.mark.parametrize("s", [1, 2, 4])
def test_nlayer_d_small(golden_dir, s):
    g = load(golden_dir, f"nlayer_d_small_s{s}.npz")
    sd = O.init_nlayer_d(20 + s, 2, 8, 3, s)
    for v in sd.values():
        v.requires_grad_(True)
    x = O.np_uniform(200 + s, (1, 2, 128, 128)).requires_grad_(True)
    p = O.nlayer_d_forward(sd, x, 3, s, True)
    l_real = O.gan_loss(p, True)
    l_fake = O.gan_loss(O.nlayer_d_forward(sd, x, 3, s, True), False)
    (l_real * 0.7 + l_fake * 0.3).backward()
    assert p.shape == g["p"].shape
    assert rel(p, g["p"]) < TIGHT
    assert abs(float(l_real.detach()) - float(g["l_real"])) < 1e-5
    assert abs(float(l_fake.detach()) - float(g["l_fake"])) < 1e-5
    assert rel(x.grad, g["dx"]) < TIGHT * 10
    for k in g.files:
        if k.startswith("grad/") and k.endswith(".weight"):
            assert rel(sd[k[5:]].grad, g[k]) < TIGHT * 10, k


@pytest.mark.parametrize("s", [1, 2])
def test_nlayer_d_sep_small(golden_dir, s):
    """`--which_model_netD n_layers_sep` (models/networks.py:851-942) against the reference run with its one broken call patched
    (oracle/make_golden.py:golden_d_sep)."""
    g = load(golden_dir, f"nlayer_d_sep_small_s{s}.npz")
    sd = O.init_nlayer_d_sep(40 + s, 8, 3, s)
    for v in sd.values():
        v.requires_grad_(True)
    x = O.np_uniform(240 + s, (1, 3, 128, 128)).requires_grad_(True)
    p = O.nlayer_d_sep_forward(sd, x, 3, s, True)
    loss = O.gan_loss(p, True) * 0.6 + O.gan_loss(O.nlayer_d_sep_forward(sd, x, 3, s, True), False) * 0.4
    loss.backward()
    assert p.shape == g["p"].shape and rel(p, g["p"]) < TIGHT
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    assert rel(x.grad, g["dx"]) < TIGHT * 10
    for k in g.files:
        if k.startswith("grad/") and k.endswith(".weight"):
            assert rel(sd[k[5:]].grad, g[k]) < TIGHT * 10, k


def test_nlayer_d_n4_lsgan(golden_dir):
    g = load(golden_dir, "nlayer_d_small_n4_lsgan.npz")
    sd = O.init_nlayer_d(29, 3, 8, 4, 1)
    for v in sd.values():
        v.requires_grad_(True)
    x = O.np_uniform(209, (1, 3, 128, 128)).requires_grad_(True)
    p = O.nlayer_d_forward(sd, x, 4, 1, False)
    loss = O.gan_loss(p, True, use_lsgan=True)
    loss.backward()
    assert rel(p, g["p"]) < TIGHT
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    assert rel(x.grad, g["dx"]) < TIGHT * 10
    assert rel(sd["model.0.weight"].grad, g["grad/model.0.weight"]) < TIGHT * 10


def make_oracle(cfg, n_init_draws):
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    m = O.FCGANOracle(cfg, seed=0)
    zshape = (1, cfg.noise_nc, cfg.noiseSize, cfg.noiseSize)

    def zs():
        i = n_init_draws
        while True:
            yield O.np_normal(5000 + i, zshape)
            i += 1
    m.noise_iter = zs()
    return m


def real_batch(cfg, step):
    return O.np_uniform(7000 + step, (1, 3, cfg.fineSize, cfg.fineSize))[:, :2].contiguous()


def grad_errors(grad, g, prefix, name, scale_key=None):
    """(max-abs error / scale, relative L2 error) of `grad` against the golden strided sample.  scale =
    max|grad| of the golden tensor, or of `scale_key`'s tensor for analytically-zero gradients."""
    ref_sum = g[f"{prefix}/summary/{name}"]
    ref_smp = g[f"{prefix}/sample/{name}"]
    scale = g[f"{prefix}/summary/{scale_key}"][1] if scale_key else ref_sum[1]
    flat = grad.detach().reshape(-1).cpu()
    smp = flat[torch.from_numpy(O.grad_sample_idx(flat.numel()))].double().numpy()
    e_max = np.abs(smp - ref_smp).max() / (scale + 1e-30)
    e_l2 = np.linalg.norm(smp - ref_smp) / (np.linalg.norm(ref_smp) + 1e-30)
    return float(e_max), float(e_l2)


def load_f64(golden_dir, name):
    """`<name>_f64.npz`: the same reference trainer on the same inputs run in double (oracle/make_golden.py f64), or None."""
    path = os.path.join(golden_dir, name.replace(".npz", "_f64.npz"))
    return np.load(path) if os.path.exists(path) else None


# Which tensors may pass through the "isolated activation flips" clause of check_grads, per golden (round 3: an explicit list, so that
# a NEW tensor needing the clause fails the test instead of passing silently).  They are the discriminators' FIRST layer -- no
# normalisation behind it, so all flips of the 4 M-activation first LeakyReLU land in its bias / weight gradient as whole-channel
# jumps of ~1e-3 .. 5e-3 -- and a few single-row cases seen in the round-2 / round-3 runs (tools: SGAN_TEST_VERBOSE=1).
# Caps: e <= 2e-2 (round 2: 5e-2), relative L2 <= 5e-3, bad elements <= max(8, 10 %); the first-layer tensors, where ONE flipped
# activation moves a whole output channel (48 weights + the bias element), <= 25 %.
_D_FIRST = r"(step1|probeG|probe)/gradD\d?_\d/model\.0\.(weight|bias)"
FLIPS_ALLOWED = {
    "fcgan_step_full.npz": [_D_FIRST],
    "fcgan_step_full_nug1.npz": [_D_FIRST],
    "cgan_step_full.npz": [_D_FIRST, r"(step1|probeG)/gradD_0/model\.(2|8)\.weight"],
    # (model.5: two sampled elements moved when the CRN's one-channel output conv changed kernels (sg_conv_head2_kernel, 1e-7-level
    # differences in fake_B); with SGAN_NO_HEAD2=1 the tensor is back inside the strict bound)
    "twostage_full.npz": [_D_FIRST, r"probe/gradD2_1/model\.(5|11)\.weight"],
    "twostage_small.npz": [_D_FIRST], "twostage_factd_small.npz": [_D_FIRST], "twostage_multiclass_small.npz": [_D_FIRST],
    "twostage_nocycle_small.npz": [_D_FIRST],
    "cgan_cycle_small.npz": [_D_FIRST], "cgan_cycle_small_d34.npz": [_D_FIRST],
    "cgan2_cycle_small.npz": [_D_FIRST], "cgan2_cycle_small_fakefake.npz": [_D_FIRST],
}


def _flips_allowed(g, full_name):
    import re
    zf = getattr(g, "zip", None)
    name = os.path.basename(zf.filename) if zf is not None and zf.filename else ""
    for pat in FLIPS_ALLOWED.get(name, ()):
        if re.fullmatch(pat, full_name):
            return True, bool(re.fullmatch(_D_FIRST, full_name))
    return False, False


def check_grads(grads, g, prefix, undet, tol=TOL, f64=None, tally=None):
    """Every gradient tensor within `tol` (max-abs error / max|g|) of the reference's fp32 golden.

    Where an fp64 run of the same reference code exists (`f64`, see load_f64) it arbitrates instead.  The gradient of these nets
    is a DISCONTINUOUS function of the rounding: a LeakyReLU / ReLU input that lands within an implementation's forward error
    of zero takes the other slope, and because a weight gradient is a sum of ~N random-sign pixel terms, ONE such element moves
    one output channel's row of it by ~1/sqrt(N) of max|g| (measured, tools/diag_bf16x3.py: 1e-3 .. 1e-2 at 512^2; the exact-fp32
    kernels and the reference's own fp32 CPU run show the same jumps against fp64, each on its own elements).  So with
    e = error against fp64 and e_ref = error of the reference's fp32 golden against fp64 (on the golden's strided sample, over
    max|g_fp64|) a tensor passes when
        e <= max(tol, 4 e_ref)                                   -- as accurate as the reference itself is there, or
        the tensor is on the golden's FLIPS_ALLOWED list (above), <= 10 % of its sampled elements (or <= 8 of them; first-layer
        tensors: 25 %) are beyond max(tol, 4 e_ref), e <= 2e-2 and its relative L2 error is <= 5e-3
                                                                  -- isolated rows moved by single activation flips.
    A kernel error would move most elements (and fails the exact adjoint identities of test_hip_ops at the same shapes).
    `tally` collects (prefix, name, e, e_ref, clause)."""
    for k, v in grads.items():
        if f"{prefix}/summary/{k}" not in g.files:
            continue
        scale_key = k.replace(".bias", ".weight") if k in undet else None    # analytically zero: scale of the same layer's weight gradient
        if f64 is not None and f"{prefix}/sample/{k}" in f64.files:
            scale = f64[f"{prefix}/summary/{scale_key or k}"][1] + 1e-30
            flat = v.detach().reshape(-1).cpu()
            smp = flat[torch.from_numpy(O.grad_sample_idx(flat.numel()))].double().numpy()
            truth = f64[f"{prefix}/sample/{k}"]
            err = np.abs(smp - truth) / scale
            e = float(err.max())
            e_ref = float(np.abs(g[f"{prefix}/sample/{k}"].astype(np.float64) - truth).max() / scale)
            bound = max(tol, 4.0 * e_ref)
            clause = "strict" if e <= tol else ("ref" if e <= bound else "flips")
            if tally is not None:
                tally.append((prefix, k, e, e_ref, clause))
            if e > bound:
                nbad = int((err > bound).sum())
                l2 = float(np.linalg.norm(smp - truth) / (np.linalg.norm(truth) + 1e-30)) if scale_key is None else 0.0
                listed, first_layer = _flips_allowed(g, f"{prefix}/{k}")
                cap = max(8, (0.25 if first_layer else 0.10) * err.size)
                ok = listed and nbad <= cap and l2 <= 5e-3 and e <= 2e-2
                if os.environ.get("SGAN_TEST_VERBOSE"):
                    print(f"flips-clause tensor {prefix}/{k}: e {e:.2e} e_ref {e_ref:.2e} nbad {nbad}/{err.size} l2 {l2:.2e}"
                          + ("" if listed else "  <-- NOT on the golden's FLIPS_ALLOWED list") + ("" if ok or not listed else "  <-- beyond the gate"))
                    if os.environ.get("SGAN_TEST_VERBOSE") == "noassert":
                        continue
                assert listed, ("a tensor that is not on FLIPS_ALLOWED needs the flips clause", prefix, k, e, e_ref, nbad, err.size, l2)
                assert ok, (prefix, k, e, e_ref, nbad, err.size, l2)
            continue
        e_max, e_l2 = grad_errors(v, g, prefix, k, scale_key)
        if tally is not None:
            tally.append((prefix, k, e_max, 0.0, "strict"))
        assert e_max <= tol, (prefix, k, e_max, e_l2)


def check_losses(losses, g, f64, tol=TOL):
    """A loss trajectory against the golden one.  With an fp64 run: the trajectory is chaotic (Adam's first steps are
    sign(g)), so the yardstick is the reference's own fp32 deviation from its fp64 trajectory -- every entry within
    max(tol * scale, 4 * max deviation of the fp32 golden).  Without: the 2e-2 band of round 1."""
    losses = np.asarray(losses, dtype=np.float64)
    ref = np.asarray(g["losses"], dtype=np.float64)
    scale = max(1.0, float(np.abs(ref).max()))
    if f64 is not None and "losses" in f64.files:
        truth = np.asarray(f64["losses"], dtype=np.float64)
        dev_ref = float(np.abs(ref - truth).max())
        dev = float(np.abs(losses - truth).max())
        assert dev <= max(tol * scale, 4.0 * dev_ref), (dev, dev_ref, losses, truth)
        return dev, dev_ref
    assert np.abs(losses - ref).max() < 2e-2 * scale, (losses, ref)
    return float(np.abs(losses - ref).max()), None


def check_forward(cap, g, tol=TOL):
    assert rel(cap["fake"][:, :, :64, :64], g["step1/fake_crop"]) < tol
    fs = O.tensor_summary(cap["fake"])
    assert abs(fs[2] - g["step1/fake_summary"][2]) <= tol * g["step1/fake_summary"][2]
    assert np.abs(np.asarray(cap["loss_D"]) - g["step1/loss_D"]).max() < tol


def check_step1(cap, g, cfg, tol=TOL, f64=None, tally=None, check_gradG=True):
    check_forward(cap, g, tol)
    for i, gd in enumerate(cap["gradD"]):
        check_grads(gd, g, f"step1/gradD_{i}", O.norm_cancelled_keys_d(cfg.input_nc, cfg.ndf, cfg.n_layers_D[i]),
                    tol, f64, tally)
    if check_gradG:   # taken after D's first Adam update: only reproducible by bit-identical arithmetic
        assert abs(cap["loss_G"] - float(g["step1/loss_G"])) < tol
        check_grads(cap["gradG"], g, "step1/gradG", O.norm_cancelled_keys_g(cfg.n_layers_G), tol, f64, tally)


def check_probe(pr, g, cfg, tol=TOL, f64=None, tally=None):
    assert abs(pr["loss_G"] - float(g["probeG/loss_G"])) < tol
    check_grads(pr["gradG"], g, "probeG/gradG", O.norm_cancelled_keys_g(cfg.n_layers_G), tol, f64, tally)
    for i, gd in enumerate(pr["gradD"]):
        check_grads(gd, g, f"probeG/gradD_{i}", O.norm_cancelled_keys_d(cfg.input_nc, cfg.ndf, cfg.n_layers_D[i]),
                    tol, f64, tally)


@pytest.mark.parametrize("name,kw", [
    ("fcgan_step_small.npz", dict(ngf=8, ndf=8, noiseSize=2, n_update_G=2)),
    ("fcgan_step_full.npz", dict(n_update_G=2)),          # BASELINE configs[0]: README fcgan @512x512 on CPU
    ("fcgan_step_full_nug1.npz", dict(n_update_G=1)),
])
def test_fcgan_step(golden_dir, name, kw):
    """Step 1 (pre-Adam: fake, losses, every gradient) is the strict gate; the multi-step trajectory is
    only compared through the losses because the reference's own trajectory is chaotic (Adam's first
    steps are sign(g): a 1-thread and an 8-thread CPU run of the reference math already differ by 1e-2
    in `fake` after one update -- see DESIGN.md 'What parity means')."""
    g = load(golden_dir, name)
    cfg = O.FCGANConfig(**kw)
    m = make_oracle(cfg, int(g["n_init_noise_draws"]))
    pr = make_oracle(cfg, int(g["n_init_noise_draws"])).probe_G(real_batch(cfg, 0))
    check_probe(pr, g, cfg, tol=1e-4)
    cap = m.step1_with_captures(real_batch(cfg, 0))
    check_step1(cap, g, cfg, tol=1e-4)
    losses = [list(m.losses().values())]
    for step in range(1, g["losses"].shape[0]):
        m.optimize_parameters(real_batch(cfg, step))
        losses.append(list(m.losses().values()))
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 1e-3, (losses, g["losses"])


# ------------------------------------------------------------------------------------------------
# U-Net generator and the cgan step (BASELINE configs[2])
# ------------------------------------------------------------------------------------------------
UNET_SMALL = {"skipall_dropout": dict(use_dropout=True, num_skips=-1, add_gaussian_noise=False),
              "skip4_noise": dict(use_dropout=False, num_skips=4, add_gaussian_noise=True),
              "residual": dict(use_dropout=False, num_skips=-1, add_gaussian_noise=False, use_residual=True, out_nc=2),
              "batchnorm": dict(use_dropout=True, num_skips=-1, add_gaussian_noise=False, norm="batch")}


def unet_small_inputs(out_nc=1):
    return O.np_uniform(301, (1, 2, 256, 256)), O.np_normal(302, (1, out_nc, 256, 256))


@pytest.mark.parametrize("tag", list(UNET_SMALL))
def test_unet_small(golden_dir, tag):
    g = load(golden_dir, f"unet_small_{tag}.npz")
    kw = UNET_SMALL[tag]
    norm = kw.get("norm", "instance")      # "batchnorm": --norm batch, BatchNorm2d(affine) behind the down / up convs (networks.py:387-389)
    sd = O.init_unet(31, 7, 2, kw.get("out_nc", 1), 8, kw["num_skips"], norm=norm)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    x, r = unet_small_inputs(kw.get("out_nc", 1))
    x.requires_grad_(True)
    y = O.unet_forward(sd, x, 7, 8, kw["num_skips"], kw["use_dropout"], mask_seed=40,
                       add_gaussian_noise=kw["add_gaussian_noise"], gaussian_sigma=0.1, noise_seed=50,
                       use_residual=kw.get("use_residual", False), norm=norm)
    (y * r).sum().backward()
    assert rel(y, g["y"]) < TIGHT * 5
    assert rel(x.grad, g["dx"]) < 1e-4
    undet = O.norm_cancelled_keys_unet(7, 8, kw["num_skips"])
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            assert rel(v.double(), g["buf/" + k].astype(np.float64)) < 1e-5, k
        elif k in undet:
            assert float(v.grad.abs().max()) <= 1e-3 * float(sd[k.replace(".bias", ".weight")].grad.abs().max())
        else:
            assert rel(v.grad, g["grad/" + k]) < 1e-4, k


RESNET_SMALL = {"6": dict(n_blocks=6, use_dropout=False), "9_dropout": dict(n_blocks=9, use_dropout=True),
                "6_residual": dict(n_blocks=6, use_dropout=False, use_residual=True, out_nc=2),
                "6_batchnorm": dict(n_blocks=6, use_dropout=True, norm="batch")}


@pytest.mark.parametrize("tag", list(RESNET_SMALL))
def test_resnet_small(golden_dir, tag):
    """resnet_6blocks / resnet_9blocks (+ dropout) restated (ReflectionPad, k7 / k3 convs, residual blocks, ConvT with output padding)
    against the reference's own nets."""
    g = load(golden_dir, f"resnet_small_{tag}.npz")
    kw = RESNET_SMALL[tag]
    onc = kw.get("out_nc", 1)
    norm = kw.get("norm", "instance")      # "6_batchnorm": --norm batch (BatchNorm2d with affine + running statistics behind every conv but the last)
    sd = O.init_resnet(41, 2, onc, 8, kw["n_blocks"], kw["use_dropout"], norm=norm)
    assert {"grad/" + k for k, v in sd.items() if v.is_floating_point() and "running" not in k} == {k for k in g.files if k.startswith("grad/")}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    x, r = O.np_uniform(311, (1, 2, 64, 64)).requires_grad_(True), O.np_normal(312, (1, onc, 64, 64))
    y = O.resnet_forward(sd, x, kw["n_blocks"], kw["use_dropout"], mask_seed=60, use_residual=kw.get("use_residual", False), norm=norm)
    (y * r).sum().backward()
    assert rel(y, g["y"]) < TIGHT * 5
    assert rel(x.grad, g["dx"]) < 1e-4
    last = f"model.{17 + kw['n_blocks']}.bias"
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            assert rel(v.double(), g["buf/" + k].astype(np.float64)) < 1e-5, k
        elif k.endswith(".bias") and k != last and sd[k.replace(".bias", ".weight")].dim() == 4:       # a conv bias in front of a norm: analytically zero gradient
            assert float(v.grad.abs().max()) <= 1e-3 * float(sd[k.replace(".bias", ".weight")].grad.abs().max())
        else:
            assert rel(v.grad, g["grad/" + k]) < 1e-4, k


def cgan_batch(cfg, step):
    A = O.np_uniform(7100 + step, (1, 3, cfg.fineSize, cfg.fineSize))
    B = O.np_uniform(7200 + step, (1, 3, cfg.fineSize, cfg.fineSize))
    if getattr(cfg, "variant", "cgan") == "cgan2":       # cgan2_model.py:122-124: (label, image) from input['A'], the second label from input['B']
        return A[:, :2].contiguous(), A[:, 2:3].contiguous(), B[:, :2].contiguous()
    return A[:, :2].contiguous(), B[:, 2:3].contiguous()       # --which_channel rg_b


def cgan_undet_D(cfg, i):
    return O.norm_cancelled_keys_d(cfg.input_nc + cfg.output_nc, cfg.ndf, cfg.n_layers_D[i])


def check_cgan_step1(cap, g, cfg, tol=TOL, f64=None, tally=None):
    check_forward(cap, g, tol)
    if "fake2" in cap:
        check_forward({"fake": cap["fake2"], "loss_D": cap["loss_D"]}, {"step1/fake_summary": g["step1/fake2_summary"],
                      "step1/fake_crop": g["step1/fake2_crop"], "step1/loss_D": g["step1/loss_D"]}, tol)
    for i, gd in enumerate(cap["gradD"]):
        check_grads(gd, g, f"step1/gradD_{i}", cgan_undet_D(cfg, i), tol, f64, tally)


def check_cgan_probe(pr, g, cfg, tol=TOL, f64=None, tally=None):
    assert np.abs(np.asarray(pr["loss_G"]) - g["probeG/loss_G"]).max() < tol * max(1.0, float(g["probeG/loss_G"][0]))
    check_grads(pr["gradG"], g, "probeG/gradG", O.norm_cancelled_keys_unet(cfg.num_downs, cfg.ngf, cfg.n_layers_G_skip),
                tol, f64, tally)
    for i, gd in enumerate(pr["gradD"]):
        check_grads(gd, g, f"probeG/gradD_{i}", cgan_undet_D(cfg, i), tol, f64, tally)


CGAN2_SMALL = dict(num_downs=7, ngf=8, ndf=8, fineSize=256, weights=(2.0, 5.0), variant="cgan2", n_layers_D=(3, 3), scale_factor=(1, 2),
                   no_lsgan=True, n_update_G=2)
CGAN_CASES = [("cgan_step_small.npz", dict(num_downs=7, ngf=8, ndf=8, fineSize=256, weights=(2.0, 5.0))),
              ("cgan_step_full.npz", O.CGAN_README),            # BASELINE configs[2]: unet_256 + D 3 4 @512x512
              ("cgan2_step_small.npz", CGAN2_SMALL),            # --model cgan2 (models/cgan2_model.py)
              ("cgan2_step_small_fakefake.npz", dict(CGAN2_SMALL, train_D_on_fake_fake_pair=True, train_G_on_fake_fake_pair=True))]


@pytest.mark.parametrize("name,kw", CGAN_CASES)
def test_cgan_step(golden_dir, name, kw):
    import random
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    g = load(golden_dir, name)
    cfg = O.CGANConfig(**kw)
    pr = O.CGANOracle(cfg, seed=0)
    pr.set_input(*cgan_batch(cfg, 0))
    check_cgan_probe(pr.probe_G(), g, cfg, tol=1e-4)
    random.seed(1234)
    m = O.CGANOracle(cfg, seed=0)
    m.set_input(*cgan_batch(cfg, 0))
    check_cgan_step1(m.step1_with_captures(), g, cfg, tol=1e-4)
    losses = [list(m.losses().values())]
    for step in range(1, g["losses"].shape[0]):
        m.set_input(*cgan_batch(cfg, step))
        m.optimize_parameters()
        losses.append(list(m.losses().values()))
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 2e-3 * max(1.0, np.abs(g["losses"]).max()), (losses, g["losses"])


# ------------------------------------------------------------------------------------------------
# Cascaded refinement network (BASELINE configs[4]'s G2)
# ------------------------------------------------------------------------------------------------
CRN_SMALL = {"convt_b1": ("convt", 1), "bilinear_b2": ("bilinear", 2), "bilinear_b2_batchnorm": ("bilinear", 2)}


@pytest.mark.parametrize("tag", list(CRN_SMALL))
def test_crn_small(golden_dir, tag):
    g = load(golden_dir, f"crn_small_{tag}.npz")
    mode, nlb = CRN_SMALL[tag]
    norm = "batch" if "batchnorm" in tag else "instance"      # --norm batch: BatchNorm2d behind every conv but the last; the shared label block's one runs five times
    sd = O.init_crn(41, 2, 1, 8, 8, mode, nlb, True, norm=norm)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    label = O.np_uniform(401, (1, 2, 128, 128)).requires_grad_(True)
    z = O.np_normal(402, (1, 8, 2, 2)).requires_grad_(True)
    r = O.np_normal(403, (1, 1, 128, 128))
    y = O.crn_forward(sd, label, z, 8, mode, nlb, True, norm=norm)
    (y * r).sum().backward()
    assert rel(y, g["y"]) < TIGHT * 5
    assert rel(label.grad, g["dlabel"]) < 1e-4 and rel(z.grad, g["dz"]) < 1e-4
    undet = O.norm_cancelled_keys_crn(2, 1, 8, 8, mode, nlb, True)
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            assert rel(v.double(), g["buf/" + k].astype(np.float64)) < 1e-5, k
        elif k in undet:
            assert float(v.grad.abs().max()) <= 1e-3 * float(sd[k.replace(".bias", ".weight")].grad.abs().max()), k
        else:
            assert rel(v.grad, g["grad/" + k]) < 1e-4, k


def test_crn_small_noise(golden_dir):
    """--add_gaussian_noise of the CRN (sigma * N(0, 1) on the normalised output of upsample blocks 5..1) against the reference run
    with the same injected noise tensors."""
    g = load(golden_dir, "crn_small_noise.npz")
    sd = O.init_crn(43, 2, 1, 8, 8, "convt", 2, True)
    for v in sd.values():
        v.requires_grad_(True)
    label = O.np_uniform(411, (1, 2, 128, 128)).requires_grad_(True)
    z = O.np_normal(412, (1, 8, 2, 2)).requires_grad_(True)
    r = O.np_normal(413, (1, 1, 128, 128))
    y = O.crn_forward(sd, label, z, 8, "convt", 2, True, gauss_seed=80, gauss_sigma=0.1)
    (y * r).sum().backward()
    assert rel(y, g["y"]) < TIGHT * 5
    assert rel(label.grad, g["dlabel"]) < 1e-4 and rel(z.grad, g["dz"]) < 1e-4
    undet = O.norm_cancelled_keys_crn(2, 1, 8, 8, "convt", 2, True)
    for k, v in sd.items():
        if k not in undet:
            assert rel(v.grad, g["grad/" + k]) < 1e-4, k
    y0 = O.crn_forward(sd, label, z, 8, "convt", 2, True)
    assert rel(y0, g["y"]) > 1e-3        # the noise is really in the golden


# ------------------------------------------------------------------------------------------------
# twostage_cycle (BASELINE configs[4]): G1 fcgan + G2 crn + F2 unet_128 + 2 x D1 + 4 x D2
# ------------------------------------------------------------------------------------------------
TWOSTAGE_CASES = [("twostage_small.npz", dict(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8,
                                             GAN_losses_D2=("real_fake", "fake_fake"), GAN_losses_G2=("real_fake", "fake_fake"),
                                             weights=(2.0, 5.0))),
                  ("twostage_full.npz", dict()),
                  ("twostage_nocycle_small.npz", dict(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8,
                                                      GAN_losses_D2=("real_fake", "fake_fake"), GAN_losses_G2=("real_fake", "fake_fake"),
                                                      cycle=False, lambda_G1=0.7, lambda_G2=1.3)),     # --model twostage
                  ("twostage_multiclass_small.npz", dict(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8,
                                                         GAN_losses_D2=("real_fake", "fake_fake"), GAN_losses_G2=("real_fake", "fake_fake"),
                                                         weights=(2.0, 5.0), use_multi_class_GAN=True, no_lsgan2=True, n_layers_D2=(3, 4),
                                                         scale_factor2=(1, 2), lambda_D2=(0.6, 0.4))),     # --use_multi_class_GAN
                  ("twostage_factd_small.npz", dict(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8,
                                                    n_layers_D1=(4, 4), n_layers_D2=(3, 3), scale_factor2=(1, 2), lambda_D2=(0.6, 0.4), no_lsgan2=True,
                                                    GAN_losses_D2=("real_fake", "fake_fake"), GAN_losses_G2=("real_fake", "fake_fake"),
                                                    cycle=False, factd=True, lambda_G1=0.7, lambda_G2=1.3))]     # --model twostage_factd


def twostage_noise(cfg):
    k = 0
    while True:
        yield (O.np_normal(5000 + k, (1, cfg.noise_nc1, cfg.noiseSize1, cfg.noiseSize1)),
               O.np_normal(6000 + k, (1, cfg.noise_nc2, cfg.noiseSize2, cfg.noiseSize2)))
        k += 1


def twostage_undet(cfg):
    u = {"G1": O.norm_cancelled_keys_g(cfg.n_layers_G1),
         "G2": O.norm_cancelled_keys_crn(cfg.input_nc, cfg.output_nc, cfg.noise_nc2, cfg.ngf2, cfg.upsample_mode2, cfg.n_layers_CRN_block2, True),
         "F2": O.norm_cancelled_keys_unet(7, cfg.nff2, -1)}
    u["D1"] = [O.norm_cancelled_keys_d(cfg.input_nc, cfg.ndf1, nl) for nl in cfg.n_layers_D1]
    u["D2"] = [O.norm_cancelled_keys_d(cfg.input_nc + cfg.output_nc, cfg.ndf2, nl) for nl in cfg.n_layers_D2]
    return u


def check_twostage_probe(cap, g, cfg, tol=TOL, f64=None, tally=None):
    for key in ("fake_A", "fake_B_from_fake_A") + (("recon_fake_A",) if cfg.cycle else ()):
        assert rel(cap[key][:, :, :64, :64], g[f"probe/{key}_crop"]) < tol, key
    assert np.abs(np.asarray(list(cap["losses"].values())) - g["probe/losses"]).max() < tol * max(1.0, np.abs(g["probe/losses"]).max())
    u = twostage_undet(cfg)
    for tag in ("G1", "G2") + (("F2",) if cfg.cycle else ()):
        check_grads(cap["grad" + tag], g, f"probe/grad{tag}", u[tag], tol, f64, tally)
    for tag in ("D1", "D2"):
        for i, gd in enumerate(cap["grad" + tag]):
            check_grads(gd, g, f"probe/grad{tag}_{i}", u[tag][i], tol, f64, tally)


@pytest.mark.parametrize("name,kw", TWOSTAGE_CASES)
def test_twostage_cycle_step(golden_dir, name, kw):
    import random
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    g = load(golden_dir, name)
    cfg = O.TwoStageConfig(**kw)
    random.seed(1234)
    p = O.TwoStageCycleOracle(cfg, seed=0)
    p.noise_iter = twostage_noise(cfg)
    p.set_input(*cgan_batch(cfg, 0))
    check_twostage_probe(p.probe(), g, cfg, tol=1e-4)
    random.seed(1234)
    m = O.TwoStageCycleOracle(cfg, seed=0)
    m.noise_iter = twostage_noise(cfg)
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(*cgan_batch(cfg, step))
        m.optimize_parameters()
        losses.append(list(m.losses().values()))
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 2e-3 * max(1.0, np.abs(g["losses"]).max()), (losses, g["losses"])


def test_autoencoder_small(golden_dir):
    g = load(golden_dir, "autoencoder_small.npz")
    sd = O.init_autoencoder(61, 2, 1, 3, 8)
    for v in sd.values():
        v.requires_grad_(True)
    x = O.np_uniform(601, (1, 2, 128, 128)).requires_grad_(True)
    r = O.np_normal(602, (1, 1, 128, 128))
    y = O.autoencoder_forward(sd, x, 3, 8)
    (y * r).sum().backward()
    assert rel(y, g["y"]) < TIGHT * 5 and rel(x.grad, g["dx"]) < 1e-4
    for k, v in sd.items():
        if k.endswith(".bias"):     # every biased conv of this net feeds an InstanceNorm
            assert float(v.grad.abs().max()) <= 1e-3 * float(sd[k.replace(".bias", ".weight")].grad.abs().max()), k
        else:
            assert rel(v.grad, g["grad/" + k]) < 1e-4, k


# ------------------------------------------------------------------------------------------------
# cgan_cycle ((f) rank 2: G1 label -> image, G2 image -> label, BCE cycle terms)
# ------------------------------------------------------------------------------------------------
CGAN2_CYCLE = dict(variant="cgan2_cycle", lambda_fake_cycle=0.5)
CGAN_CYCLE_CASES = [("cgan_cycle_small.npz", dict()),
                    ("cgan_cycle_small_d34.npz", dict(scale_factor1=(1, 1), n_layers_D1=(3, 4), weights=None, no_lsgan1=False)),
                    ("cgan2_cycle_small.npz", CGAN2_CYCLE),             # --model cgan2_cycle (models/cgan2_cycle_model.py)
                    ("cgan2_cycle_small_fakefake.npz", dict(CGAN2_CYCLE, train_D_on_fake_fake_pair=True, train_G_on_fake_fake_pair=True,
                                                            n_update_G=2))]


def cgan_cycle_batch(cfg, step):
    A = O.np_uniform(7100 + step, (1, 3, cfg.fineSize, cfg.fineSize))
    B = O.np_uniform(7200 + step, (1, 3, cfg.fineSize, cfg.fineSize))
    if cfg.variant == "cgan2_cycle":     # cgan2_cycle_model.py:114-121
        return A[:, :2].contiguous(), A[:, 2:3].contiguous(), B[:, :2].contiguous()
    return A[:, :2].contiguous(), B[:, 2:3].contiguous()


def check_cgan_cycle_probe(pr, g, cfg, tol=TOL, f64=None, tally=None):
    for key in ("fake_B", "fake_A", "recon_A") + (("recon_fake_A",) if "recon_fake_A" in pr else ()):
        assert rel(pr[key][:, :, :64, :64], g[f"probe/{key}_crop"]) < tol, key
        assert abs(O.tensor_summary(pr[key])[2] - g[f"probe/{key}_summary"][2]) <= tol * g[f"probe/{key}_summary"][2], key
    assert np.abs(np.asarray(pr["loss_D"]) - g["probe/loss_D"]).max() < tol
    assert np.abs(np.asarray(pr["loss_G"]) - g["probe/loss_G"]).max() < tol * max(1.0, float(np.abs(g["probe/loss_G"]).max()))
    for i, gd in enumerate(pr["gradD_Dstep"]):
        check_grads(gd, g, f"probe/gradD_{i}", O.norm_cancelled_keys_d(cfg.input_nc + cfg.output_nc, cfg.ndf1, cfg.n_layers_D1[i]),
                    tol, f64, tally)
    check_grads(pr["gradG1"], g, "probe/gradG1", O.norm_cancelled_keys_unet(cfg.num_downs1, cfg.ngf1, -1), tol, f64, tally)
    check_grads(pr["gradG2"], g, "probe/gradG2", O.norm_cancelled_keys_unet(cfg.num_downs2, cfg.ngf2, -1), tol, f64, tally)


@pytest.mark.parametrize("name,kw", CGAN_CYCLE_CASES)
def test_cgan_cycle_step(golden_dir, name, kw):
    import random
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    g = load(golden_dir, name)
    cfg = O.CGANCycleConfig(**kw)
    random.seed(1234)
    pr = O.CGANCycleOracle(cfg, seed=0)
    pr.set_input(*cgan_cycle_batch(cfg, 0))
    check_cgan_cycle_probe(pr.probe(), g, cfg, tol=1e-4)
    random.seed(1234)
    m = O.CGANCycleOracle(cfg, seed=0)
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(*cgan_cycle_batch(cfg, step))
        m.optimize_parameters()
        losses.append(m.losses())
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 2e-3 * max(1.0, np.abs(g["losses"]).max()), (losses, g["losses"])


# ------------------------------------------------------------------------------------------------
# DCGAN generator / discriminator ((f) rank 3 nets)
# ------------------------------------------------------------------------------------------------
def test_dcgan_small(golden_dir):
    g = load(golden_dir, "dcgan_small.npz")
    nz, nc, ngf, ndf = 8, 2, 8, 8
    sd = O.init_dcgan_g(71, nz, nc, ngf)
    for k, v in sd.items():
        if k.endswith((".weight", ".bias")):
            v.requires_grad_(True)
    z = O.np_normal(701, (1, nz, 1, 1)).requires_grad_(True)
    y = O.dcgan_g_forward(sd, z, nz, nc, ngf)
    (y * O.np_normal(702, tuple(y.shape))).sum().backward()
    assert rel(y, g["G/y"]) < 1e-4 and rel(z.grad, g["G/dz"]) < 1e-4
    for k in g.files:
        if k.startswith("G/grad/"):
            assert rel(sd[k[7:]].grad, g[k]) < 1e-4, k
        if k.startswith("G/buf/"):
            assert rel(sd[k[6:]], g[k]) < 1e-5, k
    sdd = O.init_dcgan_d(72, nc, ndf)
    for k, v in sdd.items():
        if k.endswith((".weight", ".bias")):
            v.requires_grad_(True)
    x = O.np_uniform(703, (1, nc, 128, 128)).requires_grad_(True)
    p = O.dcgan_d_forward(sdd, x, nc, ndf)
    loss = torch.nn.functional.binary_cross_entropy(p, torch.ones_like(p))
    loss.backward()
    assert rel(p, g["D/p"]) < 1e-4 and abs(float(loss) - float(g["D/loss"])) < 1e-5 and rel(x.grad, g["D/dx"]) < 1e-4
    for k in g.files:
        if k.startswith("D/grad/"):
            assert rel(sdd[k[7:]].grad, g[k]) < 1e-4, k


def test_fcgan_star_small(golden_dir):
    """FCGANGeneratorStar (models/networks.py:543-640): two deconv chains, chain b fed cat(a, b)."""
    g = load(golden_dir, "fcgan_star_small.npz")
    nz, ngf = 8, 4
    sd = O.init_fcgan_star(81, nz, ngf)
    for k, v in sd.items():
        if k.endswith((".weight", ".bias")):
            v.requires_grad_(True)
    z = O.np_normal(801, (1, nz, 2, 2)).requires_grad_(True)
    y = O.fcgan_star_forward(sd, z, nz)
    assert tuple(y.shape) == (1, 2, 128, 128)
    (y * O.np_normal(802, tuple(y.shape))).sum().backward()
    assert rel(y, g["y"]) < 1e-4 and rel(z.grad, g["dz"]) < 1e-4
    n = 0
    for k in g.files:
        if k.startswith("grad/"):
            assert rel(sd[k[5:]].grad, g[k]) < 1e-4, k
            n += 1
        if k.startswith("buf/"):
            assert rel(sd[k[4:]].double(), g[k]) < 1e-5, k
    assert n == 12 + 2 * 10


def test_fcgan_g_noisesize1(golden_dir):
    """FCGANGenerator with a 1x1 latent (use_fcn False): first ConvT k4 s1 p0."""
    g = load(golden_dir, "fcgan_g_nofcn_small.npz")
    sd = O.init_fcgan_g(12, 8, 2, 8, 5)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    z = O.np_normal(111, (1, 8, 1, 1)).requires_grad_(True)
    y = O.fcgan_g_forward(sd, z, 5, use_fcn=False)
    assert tuple(y.shape) == (1, 2, 128, 128)
    (y * O.np_normal(112, tuple(y.shape))).sum().backward()
    assert rel(y, g["y"]) < TIGHT and rel(z.grad, g["dz"]) < TIGHT
    for k in g.files:
        if k.startswith("grad/") and k.endswith(".weight") and sd[k[5:]].dim() == 4:
            assert rel(sd[k[5:]].grad, g[k]) < 1e-4, k


def test_image_prep_restatement_matches_pillow():
    """oracle/image_prep.py: the index-arithmetic restatement of crop -> flip -> rotate(90 k) -> ToTensor -> Normalize equals the
    Pillow call sequence of the reference (data/base_dataset.py:17-55), bit for bit."""
    import image_prep as IP
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, size=(37, 53, 3), dtype=np.uint8)
    for x0, y0, n in ((0, 0, 37), (16, 3, 32), (21, 5, 17)):
        for flip in (False, True):
            for rot in range(4):
                a, b = IP.prep_pil(img, x0, y0, n, flip, rot), IP.prep_numpy(img, x0, y0, n, flip, rot)
                assert a.dtype == np.float32 and a.shape == (3, n, n)
                assert np.array_equal(a, b), (x0, y0, n, flip, rot)
    assert float(IP.prep_numpy(img, 0, 0, 8, False, 0).min()) >= -1.0 and float(IP.prep_numpy(img, 0, 0, 8, False, 0).max()) <= 1.0


# ------------------------------------------------------------------------------------------------
# SegmentationModel (models/segm_model.py): cgan step with class logits, softmax / sigmoid + cross-entropy
# ------------------------------------------------------------------------------------------------
SEGM_SMALL = dict(num_downs=7, ngf=8, ndf=8, fineSize=256, n_layers_D=(3, 3), scale_factor=(1, 2), lambda_D=(0.6, 0.4), no_lsgan=True)
SEGM_CASES = {"segm_step_small.npz": dict(weights=(1.0, 3.0), n_update_G=2, **SEGM_SMALL),
              "segm_step_small_sigmoid_bg.npz": dict(use_sigmoid_ss=True, add_background_onehot=True, weights=(2.0, 1.0, 0.5), **SEGM_SMALL)}


def segm_batch(cfg, step):
    n = cfg.fineSize
    lab = torch.nn.functional.interpolate(O.np_uniform(7400 + step, (1, 3, n // 8, n // 8)), scale_factor=8, mode="nearest")
    a = O.np_uniform(7300 + step, (1, 3, n, n))
    return a[:, 2:3].contiguous(), lab[:, :cfg.label_nc].contiguous()          # --which_channel b_r / b_rg


@pytest.mark.parametrize("name", list(SEGM_CASES))
def test_segm_step(golden_dir, name):
    import random
    g = load(golden_dir, name)
    cfg = O.SegmConfig(**SEGM_CASES[name])
    random.seed(1234)
    m = O.SegmOracle(cfg, seed=0)
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(*segm_batch(cfg, step))
        if step == 0:
            m.forward()
            assert rel(m.logit[:, :, :64, :64], g["step1/logit_crop"]) < 1e-4
            assert np.array_equal(np.bincount(m.label.numpy().reshape(-1), minlength=cfg.output_nc), g["step1/label_hist"])
            m.opt_D.zero_grad()
            m.backward_D()
            for i, d in enumerate(m.D):
                for k in g.files:
                    if k.startswith(f"step1/gradD_{i}/summary/") and not k.endswith(".bias"):
                        got = O.tensor_summary(d[k.split("/summary/")[1]].grad.reshape(-1))
                        assert np.abs(np.asarray(got) - g[k]).max() < 1e-4 * max(1e-3, np.abs(g[k]).max()), k
            m.opt_D.step()
            m._g_steps()
        else:
            m.optimize_parameters()
        losses.append(list(m.losses().values()))
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 2e-3 * max(1.0, np.abs(g["losses"]).max()), (losses, g["losses"])


SEGM_CYCLE = dict(weights=(1.0, 3.0), lambda_A=2.0, lambda_B=0.5, lambda_A_cycle=1.5, lr2=1e-4)


def test_segm_cycle_step(golden_dir):
    """SegmentationCycleModel (models/segm_cycle_model.py): G1 logits, G2 on the real and on the predicted label, six loss terms."""
    import random
    g = load(golden_dir, "segm_cycle_small.npz")
    cfg = O.SegmCycleConfig(**SEGM_CYCLE)
    random.seed(1234)
    m = O.SegmCycleOracle(cfg, seed=0)
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(*segm_batch(cfg, step))
        m.optimize_parameters()
        if step == 0:
            assert rel(m.logit[:, :, :64, :64], g["step1/logit_crop"]) < 1e-4
            assert rel(m.fake_A[:, :, :64, :64], g["step1/fake_A_crop"]) < 1e-4 and rel(m.recon_A[:, :, :64, :64], g["step1/recon_A_crop"]) < 1e-4
        losses.append(m.losses())
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 2e-3 * max(1.0, np.abs(g["losses"]).max()), (losses, g["losses"])


def test_autoencoder_dropout(golden_dir):
    """The autoencoder with use_dropout against the reference run with injected masks (Dropout(0.2) / Dropout(0.5) between norm
    and ReLU; the Sequential indices move by one per dropout block)."""
    g = load(golden_dir, "autoencoder_dropout.npz")
    sd = O.init_autoencoder(62, 2, 1, 3, 8, True)
    assert [int(k.split(".")[1]) for k in sd if k.endswith(".weight")] == [0, 3, 7, 11, 12, 15, 19, 23]
    for v in sd.values():
        v.requires_grad_(True)
    x = O.np_uniform(611, (1, 2, 128, 128)).requires_grad_(True)
    r = O.np_normal(612, (1, 1, 128, 128))
    y = O.autoencoder_forward(sd, x, 3, 8, True, mask_seed=70)
    (y * r).sum().backward()
    assert rel(y, g["y"]) < TIGHT * 5 and rel(x.grad, g["dx"]) < 1e-4
    for k, v in sd.items():
        if k.endswith(".weight"):
            assert rel(v.grad, g["grad/" + k]) < 1e-4, k


def test_resize_restatement_matches_pillow():
    """oracle/image_prep.py: resize_numpy (precompute_coeffs + normalize_coeffs_8bpc + the two 8-bit passes of Pillow's Resample.c,
    restated) equals Image.resize of the Pillow in this image bit for bit -- it is what tells a reader of sgan_image_resize what
    the kernel must compute; the GPU test compares the kernel with Pillow itself."""
    import image_prep as IP
    rng = np.random.RandomState(4)
    for (h, w, ho, wo) in [(70, 131, 48, 48), (40, 40, 91, 91), (100, 37, 37, 100), (33, 50, 33, 25), (300, 200, 143, 143), (17, 19, 120, 3)]:
        img = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        for f in ("bilinear", "bicubic"):
            assert np.array_equal(IP.resize_pil(img, wo, ho, f), IP.resize_numpy(img, wo, ho, f)), (h, w, ho, wo, f)

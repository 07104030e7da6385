"""Whole-training-step parity on the MI355X for BASELINE configs[1] (fcgan: deconv G + 3 PatchGAN D,
512x512, bs 1) against the reference's golden step and against the CPU oracle.

The strict gate is step 1 BEFORE the optimizer acts (fake, the three losses, every gradient) plus the
Adam kernel on its own (test_hip_ops); the multi-step trajectory is only compared through the losses,
because the reference's trajectory is chaotic (see tests/test_oracle_golden.py::test_fcgan_step)."""
import os

import numpy as np
import pytest
import torch

import sgan_oracle as O
from test_oracle_golden import check_losses, check_probe, check_step1, load_f64, real_batch

pytestmark = pytest.mark.gpu


def build_model(cfg, n_init_draws, extra=()):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd.fcgan_model import FCGANModel
    from supervised_gan_amd.options import TrainOptions
    argv = ["--name", "t", "--model", "fcgan", "--which_direction", "A", "--fineSize", str(cfg.fineSize),
            "--input_nc", str(cfg.input_nc), "--which_model_netG", "deconv", "--n_layers_G", str(cfg.n_layers_G),
            "--ngf", str(cfg.ngf), "--which_model_netD", "n_layers", "--n_layers_D", *map(str, cfg.n_layers_D),
            "--ndf", str(cfg.ndf), "--scale_factor", *map(str, cfg.scale_factor), "--lambda_D", *map(str, cfg.lambda_D),
            "--noise_nc", str(cfg.noise_nc), "--noiseSize", str(cfg.noiseSize), "--norm", "instance", "--no_dropout",
            "--n_update_G", str(cfg.n_update_G), "--no_lsgan", "--which_channel", "rg", "--gpu_ids", "0",
            "--checkpoints_dir", "/tmp/sgan_ckpt", *extra]
    opt = TrainOptions().parse(argv, save=False, verbose=False)
    m = FCGANModel()
    m.initialize(opt)
    m.netG.load_state_dict(O.init_fcgan_g(1, cfg.noise_nc, cfg.input_nc, cfg.ngf, cfg.n_layers_G))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D, cfg.scale_factor)):
        m.netD[i].load_state_dict(O.init_nlayer_d(2 + i, cfg.input_nc, cfg.ndf, nl, sf))
    zshape = (1, cfg.noise_nc, cfg.noiseSize, cfg.noiseSize)
    ctr = {"i": n_init_draws}

    def src():
        z = O.np_normal(5000 + ctr["i"], zshape)
        ctr["i"] += 1
        return z
    m.noise_source = src
    return m


def step1_with_captures(m, real3):
    c = m.opt
    cap = {}
    m.set_input({"A": real3, "A_paths": ["synthetic"]})
    m.forward()
    cap["fake"] = m.fake.detach().cpu().clone()
    m.optimizer_D.zero_grad()
    m.backward_D()
    cap["gradD"] = [{k: p.grad.detach().cpu().clone() for k, p in d.named_parameters() if k.startswith("model.")}
                    for d in m.netD]
    cap["loss_D"] = [float(m.loss_D_real), float(m.loss_D_fake)]
    m.optimizer_D.step()
    for it in range(c.n_update_G):
        m.optimizer_G.zero_grad()
        m.backward_G()
        if it == 0:
            cap["gradG"] = {k: p.grad.detach().cpu().clone() for k, p in m.netG.named_parameters()}
            cap["loss_G"] = float(m.loss_G)
        m.optimizer_G.step()
        if c.n_update_G > 1:
            m.sample_noise()
    return cap


def probe_G(m, real3_):
    """forward() + backward_G() on the initial weights (golden capture point 'probeG')."""
    m.set_input({"A": real3_, "A_paths": ["synthetic"]})
    m.forward()
    m.optimizer_G.zero_grad()
    m.optimizer_D.zero_grad()
    m.backward_G()
    return {"gradG": {k: p.grad.detach().cpu().clone() for k, p in m.netG.named_parameters()},
            "gradD": [{k: p.grad.detach().cpu().clone() for k, p in d.named_parameters() if k.startswith("model.")}
                      for d in m.netD],
            "loss_G": float(m.loss_G)}


def real3(cfg, step):
    return O.np_uniform(7000 + step, (1, 3, cfg.fineSize, cfg.fineSize))


@pytest.mark.parametrize("name,kw,extra,full", [
    ("fcgan_step_small.npz", dict(ngf=8, ndf=8, noiseSize=2, n_update_G=2), (), False),
    ("fcgan_step_full.npz", dict(n_update_G=2), (), True),                           # BASELINE configs[1] shape, fp32
    ("fcgan_step_full_nug1.npz", dict(n_update_G=1), (), True),
    ("fcgan_step_full.npz", dict(n_update_G=2), ("--skip_wasted_D_wgrad",), True),   # same results without the wasted wgrads
])
def test_fcgan_step_vs_reference_golden(golden_dir, name, kw, extra, full):
    g = np.load(os.path.join(golden_dir, name))
    cfg = O.FCGANConfig(**kw)
    tally = []
    # (1) G step through the initial discriminators
    pr = probe_G(build_model(cfg, int(g["n_init_noise_draws"]), extra), real3(cfg, 0))
    if extra:
        assert all(float(v.abs().max()) == 0.0 for gd in pr["gradD"] for v in gd.values())   # really skipped
        pr["gradD"] = []
    f64 = load_f64(golden_dir, name)          # full-size cases: arbitrated by the reference run in double (check_grads)
    assert (f64 is not None) == full
    check_probe(pr, g, cfg, tol=1e-3, f64=f64, tally=tally)
    # (2) step 1 in optimize_parameters' order: fake, D losses, D gradients before the optimizer acts
    m = build_model(cfg, int(g["n_init_noise_draws"]), extra)
    cap = step1_with_captures(m, real3(cfg, 0))
    torch.cuda.synchronize()
    check_step1(cap, g, cfg, tol=1e-3, f64=f64, tally=tally, check_gradG=False)
    worst = max(tally, key=lambda t: t[2])
    n_ref32 = sum(1 for t in tally if t[3] <= 1e-3)
    print(f"{name}: {len(tally)} gradient tensors vs {'the fp64 reference' if f64 is not None else 'the fp32 golden'}: "
          f"{sum(1 for t in tally if t[4] == 'strict')} within 1e-3 (the reference's own fp32: {n_ref32}), "
          f"{sum(1 for t in tally if t[4] == 'ref')} within 4x the reference's own fp32 error, "
          f"{sum(1 for t in tally if t[4] == 'flips')} with isolated activation-flip rows; "
          f"worst {worst[2]:.2e} at {worst[0]}/{worst[1]} (reference fp32 there: {worst[3]:.2e})")
    # (3) trajectory: losses only (Adam's sign-like first steps make the parameters themselves chaotic)
    losses = [list(m.get_current_errors().values())]
    for step in range(1, g["losses"].shape[0]):
        m.set_input({"A": real3(cfg, step), "A_paths": ["synthetic"]})
        m.optimize_parameters()
        losses.append(list(m.get_current_errors().values()))
    print(f"{name}: loss trajectory deviation (this, reference fp32) from fp64 / golden:", check_losses(losses, g, f64))
    assert m.optimizer_D.step_count == g["losses"].shape[0]
    assert m.optimizer_G.step_count == g["losses"].shape[0] * cfg.n_update_G


def test_checkpoint_roundtrip_with_oracle(tmp_path):
    """save() writes the reference's file names / keys / shapes; the CPU oracle (a restatement of the
    reference nets) consumes them directly and reproduces the HIP forward."""
    cfg = O.FCGANConfig(ngf=8, ndf=8, noiseSize=2)
    m = build_model(cfg, 0)
    m.opt.checkpoints_dir = str(tmp_path)
    m.save_dir = str(tmp_path / "t")
    m.save("latest")
    files = sorted(os.listdir(m.save_dir))
    assert files == ["latest_net_D_0.pth", "latest_net_D_1.pth", "latest_net_D_2.pth", "latest_net_G.pth"]
    sdG = torch.load(os.path.join(m.save_dir, "latest_net_G.pth"))
    ref = O.init_fcgan_g(1, 8, 2, 8, 5)
    assert list(sdG.keys()) == list(ref.keys())
    for k in ref:
        assert sdG[k].shape == ref[k].shape and sdG[k].device.type == "cpu" and sdG[k].is_contiguous()
        assert torch.equal(sdG[k], ref[k].detach()), k
    z = O.np_normal(77, (1, 8, 2, 2))
    y_ref = O.fcgan_g_forward({k: v.clone() for k, v in sdG.items()}, z, 5)
    y = m.netG.forward(z.cuda())
    assert O.rel_err(y, y_ref) < 1e-3
    sdD = torch.load(os.path.join(m.save_dir, "latest_net_D_2.pth"))
    assert set(sdD.keys()) == set(O.init_nlayer_d(4, 2, 8, 3, 4).keys())
    x = O.np_uniform(78, (1, 2, 128, 128))
    p_ref = O.nlayer_d_forward(sdD, x, 3, 4, True)
    m.netD[2].fuse_sigmoid_into_loss = False
    assert O.rel_err(m.netD[2].forward(x.cuda()), p_ref) < 1e-3
    # and back: load into a fresh model (old-torch style extras tolerated)
    from supervised_gan_amd.base_model import load_state_dict_compat
    sdD["model.3.running_mean"] = torch.zeros(16)
    sdD["model.3.running_var"] = torch.ones(16)
    load_state_dict_compat(m.netD[2], sdD)


# ------------------------------------------------------------------------------------------------
# cgan: unet G + n_layers D (3, 4) + weighted L1  (BASELINE configs[2])
# ------------------------------------------------------------------------------------------------
from test_hip_nets import inject_unet_random  # noqa: E402
from test_oracle_golden import CGAN_CASES, check_cgan_probe, check_cgan_step1  # noqa: E402


def build_cgan(cfg, extra=()):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    argv = ["--name", "t", "--model", cfg.variant, "--which_direction", "AtoB", "--dataset_mode", "unaligned" if cfg.variant == "cgan2" else "aligned",
            "--fineSize", str(cfg.fineSize), "--which_model_netG", {7: "unet_128", 8: "unet_256"}[cfg.num_downs],
            "--ngf", str(cfg.ngf), "--which_model_netD", "n_layers", "--n_layers_D", *map(str, cfg.n_layers_D),
            "--ndf", str(cfg.ndf), "--scale_factor", *map(str, cfg.scale_factor), "--lambda_D", *map(str, cfg.lambda_D),
            "--lambda_A", str(cfg.lambda_A), "--norm", "instance", "--which_channel", "rg_b", "--gpu_ids", "0",
            "--checkpoints_dir", "/tmp/sgan_ckpt", *extra]
    if cfg.weights is not None:
        argv += ["--weights", *map(str, cfg.weights)]
    if cfg.add_gaussian_noise:
        argv += ["--add_gaussian_noise", "--gaussian_sigma", str(cfg.gaussian_sigma)]
    if cfg.no_lsgan:
        argv.append("--no_lsgan")
    if cfg.train_D_on_fake_fake_pair:
        argv.append("--train_D_on_fake_fake_pair")
    if cfg.train_G_on_fake_fake_pair:
        argv.append("--train_G_on_fake_fake_pair")
    argv += ["--n_update_G", str(cfg.n_update_G)]
    opt = TrainOptions().parse(argv, save=False, verbose=False)
    m = create_model(opt)
    m.netG.load_state_dict(O.init_unet(1, cfg.num_downs, cfg.input_nc, cfg.output_nc, cfg.ngf, cfg.n_layers_G_skip))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D, cfg.scale_factor)):
        m.netD[i].load_state_dict(O.init_nlayer_d(2 + i, cfg.input_nc + cfg.output_nc, cfg.ndf, nl, sf))
    # the k-th generator forward uses the numpy dropout masks / noise seeded 9000 + 100 k / 9500 + 100 k (as the goldens)
    inner, ctr = m.netG.forward, {"k": 0}

    def fwd(*a, **k):
        inject_unet_random(m.netG, cfg.fineSize, 9000 + 100 * ctr["k"], 9500 + 100 * ctr["k"])
        ctr["k"] += 1
        return inner(*a, **k)
    m.netG.forward = fwd
    return m


def cgan_input(cfg, step):
    return {"A": O.np_uniform(7100 + step, (1, 3, cfg.fineSize, cfg.fineSize)),
            "B": O.np_uniform(7200 + step, (1, 3, cfg.fineSize, cfg.fineSize)), "A_paths": ["synthetic"], "B_paths": ["synthetic"]}


def _grads(net, prefix="model."):
    return {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if k.startswith(prefix)}


@pytest.mark.parametrize("name,kw", CGAN_CASES)
def test_cgan_step_vs_reference_golden(golden_dir, name, kw):
    import random
    g = np.load(os.path.join(golden_dir, name))
    cfg = O.CGANConfig(**kw)
    full = cfg.fineSize >= 512
    tally = []
    # (1) G step (GAN + weighted L1) through the initial discriminators
    p = build_cgan(cfg)
    p.set_input(cgan_input(cfg, 0))
    p.forward()
    p.optimizer_G.zero_grad()
    p.optimizer_D.zero_grad()
    p.backward_G()
    torch.cuda.synchronize()
    pr = {"gradG": _grads(p.netG), "gradD": [_grads(d) for d in p.netD], "loss_G": [float(p.loss_G), float(p.loss_G_L1)]}
    f64 = load_f64(golden_dir, name)
    assert (f64 is not None) == full
    check_cgan_probe(pr, g, cfg, tol=1e-3, f64=f64, tally=tally)
    # (2) step 1: fake_B, D losses, D gradients before the optimizer acts
    random.seed(1234)
    m = build_cgan(cfg)
    m.set_input(cgan_input(cfg, 0))
    m.forward()
    cap = {"fake": m.fake_B.detach().cpu().clone()}
    if cfg.variant == "cgan2":
        cap["fake2"] = m.fake_B_from_fake_A.detach().cpu().clone()
    m.optimizer_D.zero_grad()
    m.backward_D()
    cap["gradD"] = [_grads(d) for d in m.netD]
    cap["loss_D"] = [float(m.loss_D_real), float(m.loss_D_fake)]
    check_cgan_step1(cap, g, cfg, tol=1e-3, f64=f64, tally=tally)
    m.optimizer_D.step()
    for _ in range(cfg.n_update_G):
        m.optimizer_G.zero_grad()
        m.backward_G()
        m.optimizer_G.step()
        if cfg.n_update_G > 1:
            m.sample_noise()
    worst = max(tally, key=lambda t: t[2])
    n_ref32 = sum(1 for t in tally if t[3] <= 1e-3)
    print(f"{name}: {len(tally)} gradient tensors vs {'the fp64 reference' if f64 is not None else 'the fp32 golden'}: "
          f"{sum(1 for t in tally if t[4] == 'strict')} within 1e-3 (the reference's own fp32: {n_ref32}), "
          f"{sum(1 for t in tally if t[4] == 'ref')} within 4x the reference's own fp32 error, "
          f"{sum(1 for t in tally if t[4] == 'flips')} with isolated activation-flip rows; "
          f"worst {worst[2]:.2e} at {worst[0]}/{worst[1]} (reference fp32 there: {worst[3]:.2e})")
    # (3) trajectory through the losses
    def errs():     # the golden rows are (loss_G, loss_G_L1, D_real, D_fake); cgan2's get_current_errors has no G_L1 entry
        return [float(m.loss_G), float(m.loss_G_L1), float(m.loss_D_real), float(m.loss_D_fake)]
    losses = [errs()]
    for step in range(1, g["losses"].shape[0]):
        m.set_input(cgan_input(cfg, step))
        m.optimize_parameters()
        losses.append(errs())
    print(f"{name}: loss trajectory deviation:", check_losses(losses, g, f64))


def test_cgan_with_crn_generator():
    """`--model cgan --which_model_netG crn` (the G2 of BASELINE configs[4]) trains through the same trainer: the generator
    output equals the CPU oracle on the same weights / latent, and a few optimizer steps keep every loss finite."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    argv = ["--name", "t", "--model", "cgan", "--which_direction", "AtoB", "--dataset_mode", "aligned", "--fineSize", "128",
            "--which_model_netG", "crn", "--upsample_mode", "bilinear", "--n_layers_CRN_block", "2", "--ngf", "8", "--noise_nc", "8",
            "--noiseSize", "2", "--which_model_netD", "n_layers", "--n_layers_D", "3", "--ndf", "8", "--scale_factor", "1",
            "--lambda_D", "1.0", "--norm", "instance", "--which_channel", "rg_b", "--gpu_ids", "0", "--no_dropout",
            "--checkpoints_dir", "/tmp/sgan_ckpt"]
    m = create_model(TrainOptions().parse(argv, save=False, verbose=False))
    sd = O.init_crn(51, 2, 1, 8, 8, "bilinear", 2, True)
    m.netG.load_state_dict(sd)
    z = O.np_normal(52, (1, 8, 2, 2))
    m.noise_source = lambda: z
    cfg = O.CGANConfig(fineSize=128)
    m.set_input(cgan_input(cfg, 0))
    m.forward()
    A = O.np_uniform(7100, (1, 3, 128, 128))[:, :2].contiguous()
    y_ref = O.crn_forward({k: v.clone() for k, v in sd.items()}, A, z, 8, "bilinear", 2, True)
    assert O.rel_err(m.fake_B, y_ref) < 1e-3
    m.noise_source = None
    for step in range(3):
        m.set_input(cgan_input(cfg, step))
        m.optimize_parameters()
    errs = m.get_current_errors()
    assert all(np.isfinite(v) for v in errs.values()), errs
    assert m.optimizer_G.step_count == 3 and m.optimizer_D.step_count == 3


# ------------------------------------------------------------------------------------------------
# twostage_cycle (BASELINE configs[4])
# ------------------------------------------------------------------------------------------------
from test_oracle_golden import TWOSTAGE_CASES, check_twostage_probe  # noqa: E402


def build_twostage(cfg):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    L = lambda xs: [str(x) for x in xs]
    argv = ["--name", "t", "--model", "twostage_cycle" if cfg.cycle else "twostage_factd" if cfg.factd else "twostage", "--which_direction", "AtoB", "--dataset_mode", "aligned",
            "--fineSize", str(cfg.fineSize), "--transform_1to2", cfg.transform_1to2, "--which_channel", "rg_b",
            "--which_model_netG1", "fcgan", "--n_layers_G1", str(cfg.n_layers_G1), "--ngf1", str(cfg.ngf1),
            "--which_model_netD1", "n_layers", "--n_layers_D1", *L(cfg.n_layers_D1), "--ndf1", str(cfg.ndf1),
            "--scale_factor1", *L(cfg.scale_factor1), "--lambda_D1", *L(cfg.lambda_D1), "--which_model_netG2", "crn",
            "--ngf2", str(cfg.ngf2), "--upsample_mode2", cfg.upsample_mode2, "--n_layers_CRN_block2", str(cfg.n_layers_CRN_block2),
            "--which_model_netF2", "unet_128", "--nff2", str(cfg.nff2), "--which_model_netD2", "n_layers",
            "--n_layers_D2", *L(cfg.n_layers_D2), "--ndf2", str(cfg.ndf2), "--scale_factor2", *L(cfg.scale_factor2),
            "--lambda_D2", *L(cfg.lambda_D2), "--lambda_A", str(cfg.lambda_A), "--lambda_B", str(cfg.lambda_B),
            "--lambda_A_cycle", str(cfg.lambda_A_cycle), "--lambda_fake_cycle", str(cfg.lambda_fake_cycle),
            "--noise_nc1", str(cfg.noise_nc1), "--noiseSize1", str(cfg.noiseSize1), "--noise_nc2", str(cfg.noise_nc2),
            "--noiseSize2", str(cfg.noiseSize2), "--norm", "instance", "--no_dropout1", "--no_dropout2",
            "--GAN_losses_D2", *cfg.GAN_losses_D2, "--GAN_losses_G2", *cfg.GAN_losses_G2, "--gpu_ids", "0",
            "--checkpoints_dir", "/tmp/sgan_ckpt"]
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
    m = create_model(TrainOptions().parse(argv, save=False, verbose=False))
    m.netG1.load_state_dict(O.init_fcgan_g(1, cfg.noise_nc1, cfg.input_nc, cfg.ngf1, cfg.n_layers_G1))
    m.netG2.load_state_dict(O.init_crn(2, cfg.input_nc, cfg.output_nc, cfg.noise_nc2, cfg.ngf2, cfg.upsample_mode2, cfg.n_layers_CRN_block2, True))
    if cfg.cycle:
        m.netF2.load_state_dict(O.init_unet(3, 7, cfg.output_nc, cfg.input_nc, cfg.nff2, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D1, cfg.scale_factor1)):
        m.netD1[i].load_state_dict(O.init_nlayer_d(10 + i, cfg.input_nc, cfg.ndf1, nl, sf))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D2, cfg.scale_factor2)):
        m.netD2[i].load_state_dict(O.init_nlayer_d(20 + i, cfg.input_nc + cfg.output_nc, cfg.ndf2, nl, sf,
                                                   3 if cfg.use_multi_class_GAN else 1))
    ctr = {1: 0, 2: 0}

    def src(which):
        nc, ns = (cfg.noise_nc1, cfg.noiseSize1) if which == 1 else (cfg.noise_nc2, cfg.noiseSize2)
        z = O.np_normal((5000 if which == 1 else 6000) + ctr[which], (1, nc, ns, ns))
        ctr[which] += 1
        return z
    m.noise_source = src
    return m


@pytest.mark.parametrize("name,kw", TWOSTAGE_CASES)
def test_twostage_cycle_vs_reference_golden(golden_dir, name, kw):
    import random
    g = np.load(os.path.join(golden_dir, name))
    cfg = O.TwoStageConfig(**kw)
    full = cfg.fineSize >= 512
    tally = []
    random.seed(1234)
    p = build_twostage(cfg)
    p.set_input(cgan_input(cfg, 0))
    p.forward()
    cap = {k: getattr(p, k).detach().cpu().clone() for k in ("fake_A", "fake_B_from_fake_A") + (("recon_fake_A",) if cfg.cycle else ())}
    p.optimizer_D1.zero_grad()
    p.backward_D1()
    cap["gradD1"] = [_grads(d) for d in p.netD1]
    p.optimizer_D2.zero_grad()
    p.backward_D2()
    cap["gradD2"] = [_grads(d) for d in p.netD2]
    p.optimizer_G.zero_grad()
    p.backward_G()
    torch.cuda.synchronize()
    cap["gradG1"], cap["gradG2"] = _grads(p.netG1), _grads(p.netG2, "")
    cap["gradF2"] = _grads(p.netF2) if cfg.cycle else {}
    cap["losses"] = p.get_current_errors()
    # G1's gradient arrives through seven networks (D1 x2, and via the bilinear transform G2, D2 x4, F2) and ends in a BatchNorm over
    # 4x4 samples, F2's inner blocks normalise 2x2 maps: every case of this trainer is arbitrated by the reference run in double
    f64 = load_f64(golden_dir, name)
    assert f64 is not None
    check_twostage_probe(cap, g, cfg, tol=1e-3, f64=f64, tally=tally)
    worst = max(tally, key=lambda t: t[2])
    n_ref32 = sum(1 for t in tally if t[3] <= 1e-3)
    print(f"{name}: {len(tally)} gradient tensors vs {'the fp64 reference' if f64 is not None else 'the fp32 golden'}: "
          f"{sum(1 for t in tally if t[4] == 'strict')} within 1e-3 (the reference's own fp32: {n_ref32}), "
          f"{sum(1 for t in tally if t[4] == 'ref')} within 4x the reference's own fp32 error, "
          f"{sum(1 for t in tally if t[4] == 'flips')} with isolated activation-flip rows; "
          f"worst {worst[2]:.2e} at {worst[0]}/{worst[1]} (reference fp32 there: {worst[3]:.2e})")
    random.seed(1234)
    m = build_twostage(cfg)
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(cgan_input(cfg, step))
        m.optimize_parameters()
        losses.append(list(m.get_current_errors().values()))
    print(f"{name}: loss trajectory deviation:", check_losses(losses, g, f64))


def test_checkpoints_cgan_and_twostage(tmp_path):
    """save() / load_network() of the U-Net, CRN and two-stage trainers: the reference's file names (base_model.py:44-61,
    cgan_model.py:246-249, twostage_cycle_model.py:466-475), its state_dict keys, and a fresh trainer that continues from
    them reproduces the generator outputs bit for bit."""
    from supervised_gan_amd.base_model import load_state_dict_compat
    cfg = O.CGANConfig(num_downs=7, ngf=8, ndf=8, fineSize=256)
    m = build_cgan(cfg)
    m.save_dir = str(tmp_path / "cgan")
    m.save("latest")
    assert sorted(os.listdir(m.save_dir)) == ["latest_net_D_0.pth", "latest_net_D_1.pth", "latest_net_G.pth"]
    sd = torch.load(os.path.join(m.save_dir, "latest_net_G.pth"))
    ref = O.init_unet(1, 7, 2, 1, 8, -1)
    assert set(sd.keys()) == set(ref.keys()) and all(sd[k].shape == ref[k].shape and sd[k].device.type == "cpu" for k in ref)
    assert all(torch.equal(sd[k], ref[k].detach()) for k in ref)
    m2 = build_cgan(cfg)
    torch.nn.init.normal_(next(m2.netG.parameters()).data)      # perturb, then restore from the checkpoint
    m2.save_dir = m.save_dir
    m2.load_network(m2.netG, "G", "latest")
    for mm in (m, m2):
        mm.set_input(cgan_input(cfg, 0))
        mm.test()
    assert torch.equal(m.fake_B, m2.fake_B)

    tcfg = O.TwoStageConfig(fineSize=256, ngf1=8, noiseSize1=2, ndf1=8, ngf2=8, noiseSize2=4, nff2=8, ndf2=8)
    t = build_twostage(tcfg)
    t.save_dir = str(tmp_path / "two")
    t.save("latest")
    assert sorted(os.listdir(t.save_dir)) == ["latest_net_D1_0.pth", "latest_net_D1_1.pth", "latest_net_D2_0.pth", "latest_net_D2_1.pth",
                                              "latest_net_D2_2.pth", "latest_net_D2_3.pth", "latest_net_F2.pth", "latest_net_G1.pth",
                                              "latest_net_G2.pth"]
    sdG2 = torch.load(os.path.join(t.save_dir, "latest_net_G2.pth"))
    refG2 = O.init_crn(2, 2, 1, 8, 8, "bilinear", 2, True)
    assert list(sdG2.keys()) == list(refG2.keys()) and all(torch.equal(sdG2[k], refG2[k].detach()) for k in refG2)
    # the CPU oracle consumes the files directly and reproduces G2 on the HIP path
    lab = O.np_uniform(91, (1, 2, 256, 256))
    z = O.np_normal(92, (1, 8, 4, 4))
    y_ref = O.crn_forward(sdG2, lab, z, 8, "bilinear", 2, True)
    assert O.rel_err(t.netG2.forward(lab.cuda(), z.cuda()), y_ref) < 1e-3
    # old-torch checkpoints carry InstanceNorm running stats: tolerated
    sdF2 = torch.load(os.path.join(t.save_dir, "latest_net_F2.pth"))
    sdF2["model.1.model.2.running_mean"] = torch.zeros(16)
    sdF2["model.1.model.2.running_var"] = torch.ones(16)
    load_state_dict_compat(t.netF2, sdF2)


# ------------------------------------------------------------------------------------------------
# cgan_cycle
# ------------------------------------------------------------------------------------------------
from test_oracle_golden import CGAN_CYCLE_CASES, check_cgan_cycle_probe  # noqa: E402


def build_cgan_cycle(cfg):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    L = lambda xs: [str(x) for x in xs]
    unet = {7: "unet_128", 8: "unet_256"}
    argv = ["--name", "t", "--model", cfg.variant, "--which_direction", "AtoB", "--dataset_mode", "unaligned" if cfg.variant == "cgan2_cycle" else "aligned",
            "--fineSize", str(cfg.fineSize), "--lambda_fake_cycle", str(cfg.lambda_fake_cycle), "--n_update_G", str(cfg.n_update_G),
            "--which_channel", "rg_b", "--which_model_netG1", unet[cfg.num_downs1], "--ngf1", str(cfg.ngf1),
            "--which_model_netG2", unet[cfg.num_downs2], "--ngf2", str(cfg.ngf2), "--which_model_netD1", "n_layers",
            "--n_layers_D1", *L(cfg.n_layers_D1), "--ndf1", str(cfg.ndf1), "--scale_factor1", *L(cfg.scale_factor1),
            "--lambda_D1", *L(cfg.lambda_D1), "--lambda_A", str(cfg.lambda_A), "--lambda_B", str(cfg.lambda_B),
            "--lambda_A_cycle", str(cfg.lambda_A_cycle), "--lr1", str(cfg.lr1), "--lr2", str(cfg.lr2), "--norm", "instance",
            "--no_dropout1", "--no_dropout2", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/sgan_ckpt"]
    if cfg.no_lsgan1:
        argv.append("--no_lsgan1")
    if cfg.weights is not None:
        argv += ["--weights", *L(cfg.weights)]
    if cfg.train_D_on_fake_fake_pair:
        argv.append("--train_D_on_fake_fake_pair")
    if cfg.train_G_on_fake_fake_pair:
        argv.append("--train_G_on_fake_fake_pair")
    m = create_model(TrainOptions().parse(argv, save=False, verbose=False))
    m.netG1.load_state_dict(O.init_unet(1, cfg.num_downs1, cfg.input_nc, cfg.output_nc, cfg.ngf1, -1))
    m.netG2.load_state_dict(O.init_unet(2, cfg.num_downs2, cfg.output_nc, cfg.input_nc, cfg.ngf2, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D1, cfg.scale_factor1)):
        m.netD1[i].load_state_dict(O.init_nlayer_d(3 + i, cfg.input_nc + cfg.output_nc, cfg.ndf1, nl, sf))
    return m


@pytest.mark.parametrize("name,kw", CGAN_CYCLE_CASES)
def test_cgan_cycle_step_vs_reference_golden(golden_dir, name, kw):
    """`--model cgan_cycle`: forward (three generator calls), D1 step and the joint G1/G2 step before any update, then the
    loss trajectory, against goldens from the reference."""
    import random
    g = np.load(os.path.join(golden_dir, name))
    cfg = O.CGANCycleConfig(**kw)
    random.seed(1234)
    p = build_cgan_cycle(cfg)
    p.set_input(cgan_input(cfg, 0))
    p.forward()
    two = cfg.variant == "cgan2_cycle"
    names = {"fake_B": "fake_B_from_real_A", "fake_A": "fake_A_from_real_B", "recon_A": "recon_real_A", "recon_fake_A": "recon_fake_A"} if two \
        else {"fake_B": "fake_B", "fake_A": "fake_A", "recon_A": "recon_A"}
    pr = {k: getattr(p, a).detach().cpu().clone() for k, a in names.items()}
    p.optimizer_D1.zero_grad()
    p.backward_D1()
    pr["gradD_Dstep"] = [_grads(d) for d in p.netD1]
    pr["loss_D"] = [float(p.loss_D_real), float(p.loss_D_fake)]
    p.optimizer_D1.zero_grad()
    p.optimizer_G.zero_grad()
    p.backward_G()
    torch.cuda.synchronize()
    pr["gradG1"], pr["gradG2"] = _grads(p.netG1), _grads(p.netG2)
    pr["loss_G"] = [float(p.loss_G), float(p.loss_G_GAN), float(p.loss_G_L1), float(p.loss_G_CE),
                    float(p.loss_G_real_cycle if two else p.loss_G_cycle)]
    # the inner U-Net blocks normalise 2x2 - 4x4 maps: arbitrated by the reference run in double
    tally = []
    f64 = load_f64(golden_dir, name)
    assert f64 is not None
    check_cgan_cycle_probe(pr, g, cfg, tol=1e-3, f64=f64, tally=tally)
    worst = max(tally, key=lambda t: t[2])
    n_ref32 = sum(1 for t in tally if t[3] <= 1e-3)
    print(f"{name}: {len(tally)} gradient tensors vs {'the fp64 reference' if f64 is not None else 'the fp32 golden'}: "
          f"{sum(1 for t in tally if t[4] == 'strict')} within 1e-3 (the reference's own fp32: {n_ref32}), "
          f"{sum(1 for t in tally if t[4] == 'ref')} within 4x the reference's own fp32 error, "
          f"{sum(1 for t in tally if t[4] == 'flips')} with isolated activation-flip rows; "
          f"worst {worst[2]:.2e} at {worst[0]}/{worst[1]} (reference fp32 there: {worst[3]:.2e})")
    random.seed(1234)
    m = build_cgan_cycle(cfg)
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(cgan_input(cfg, step))
        m.optimize_parameters()
        losses.append([float(m.loss_G), float(m.loss_G_real_cycle if two else m.loss_G_cycle), float(m.loss_D)])
    print(f"{name}: loss trajectory deviation:", check_losses(losses, g, f64))


# ------------------------------------------------------------------------------------------------
# SegmentationModel (models/segm_model.py): class logits, softmax / sigmoid, cross-entropy + GAN
# ------------------------------------------------------------------------------------------------
SEGM_SMALL = dict(num_downs=7, ngf=8, ndf=8, fineSize=256, n_layers_D=(3, 3), scale_factor=(1, 2), lambda_D=(0.6, 0.4), no_lsgan=True)
SEGM_CASES = {"segm_step_small.npz": dict(weights=(1.0, 3.0), n_update_G=2, **SEGM_SMALL),
              "segm_step_small_sigmoid_bg.npz": dict(use_sigmoid_ss=True, add_background_onehot=True, weights=(2.0, 1.0, 0.5), **SEGM_SMALL)}


def build_segm(cfg, extra=()):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    argv = ["--name", "t", "--model", "segmentation", "--which_direction", "AtoB", "--dataset_mode", "aligned", "--fineSize", str(cfg.fineSize),
            "--which_model_netG", "unet_128", "--ngf", str(cfg.ngf), "--which_model_netD", "n_layers", "--n_layers_D", *map(str, cfg.n_layers_D),
            "--ndf", str(cfg.ndf), "--scale_factor", *map(str, cfg.scale_factor), "--lambda_D", *map(str, cfg.lambda_D), "--norm", "instance",
            "--which_channel", "b_" + "rg"[:cfg.label_nc], "--gpu_ids", "0", "--checkpoints_dir", "/tmp/sgan_ckpt", "--no_dropout", "--no_lsgan",
            "--weights", *map(str, cfg.weights), "--n_update_G", str(cfg.n_update_G), *extra]
    if cfg.use_sigmoid_ss:
        argv.append("--use_sigmoid_ss")
    if cfg.add_background_onehot:
        argv.append("--add_background_onehot")
    m = create_model(TrainOptions().parse(argv, save=False, verbose=False))
    m.netG.load_state_dict(O.init_unet(1, cfg.num_downs, cfg.input_nc, cfg.output_nc, cfg.ngf, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D, cfg.scale_factor)):
        m.netD[i].load_state_dict(O.init_nlayer_d(2 + i, cfg.input_nc + cfg.output_nc, cfg.ndf, nl, sf))
    return m


def segm_input(cfg, step):
    n = cfg.fineSize
    lab = torch.nn.functional.interpolate(O.np_uniform(7400 + step, (1, 3, n // 8, n // 8)), scale_factor=8, mode="nearest")
    return {"A": O.np_uniform(7300 + step, (1, 3, n, n)), "B": lab, "A_paths": ["synthetic"], "B_paths": ["synthetic"]}


@pytest.mark.parametrize("name", list(SEGM_CASES))
def test_segm_steps_match_reference(golden_dir, name):
    """`--model segmentation` on the HIP path against the reference's own steps: first-step logits, label histogram and
    discriminator gradients, then the losses of every step (two generator updates per step in the softmax case)."""
    import random
    g = np.load(os.path.join(golden_dir, name))
    cfg = O.SegmConfig(**SEGM_CASES[name])
    random.seed(1234)
    m = build_segm(cfg)
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(segm_input(cfg, step))
        if step == 0:
            m.forward()
            assert tuple(m.logit.shape) == (1, cfg.output_nc, 256, 256)
            assert O.rel_err(m.logit[:, :, :64, :64].detach().cpu(), torch.from_numpy(g["step1/logit_crop"])) < 1e-3
            assert np.array_equal(np.bincount(m.label.cpu().numpy().reshape(-1), minlength=cfg.output_nc), g["step1/label_hist"])
            m.optimizer_D.zero_grad()
            m.backward_D()
            torch.cuda.synchronize()
            for i, d in enumerate(m.netD):
                for k, gr in _grads(d).items():
                    if k.endswith(".weight"):
                        ref = g[f"step1/gradD_{i}/summary/{k}"]
                        got = np.asarray(O.tensor_summary(gr.reshape(-1)))
                        assert np.abs(got - ref).max() < 2e-3 * max(1e-3, np.abs(ref).max()), (i, k, got, ref)
            m.optimizer_D.step()
            for _ in range(cfg.n_update_G):
                m.optimizer_G.zero_grad()
                m.backward_G()
                m.optimizer_G.step()
                if cfg.n_update_G > 1:
                    m.sample_noise()
        else:
            m.optimize_parameters()
        e = m.get_current_errors()
        losses.append([e["G_CE"], e["G_GAN"], e["D_real"], e["D_fake"]])
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 5e-3 * max(1.0, np.abs(g["losses"]).max()), (losses, g["losses"])
    m.opt.which_metric = ["meanIU"]
    m.accum_accs()
    assert 0.0 <= m.get_current_accs()["meanIU"] <= 1.0


def test_segm_cycle_steps_match_reference(golden_dir):
    """`--model segmentation_cycle` against the reference's own steps: first-step logits / G2 outputs, then all six loss terms."""
    import random
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    from test_oracle_golden import SEGM_CYCLE
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    g = np.load(os.path.join(golden_dir, "segm_cycle_small.npz"))
    cfg = O.SegmCycleConfig(**SEGM_CYCLE)
    L = lambda xs: [str(x) for x in xs]          # noqa: E731
    argv = ["--name", "t", "--model", "segmentation_cycle", "--which_direction", "AtoB", "--dataset_mode", "aligned", "--fineSize", "256",
            "--which_channel", "b_rg", "--which_model_netG1", "unet_128", "--ngf1", "8", "--which_model_netG2", "unet_128", "--ngf2", "8",
            "--which_model_netD2", "n_layers", "--n_layers_D2", *L(cfg.n_layers_D2), "--ndf2", "8", "--scale_factor2", *L(cfg.scale_factor2),
            "--lambda_D2", *L(cfg.lambda_D2), "--lambda_A", "2.0", "--lambda_B", "0.5", "--lambda_A_cycle", "1.5", "--lr1", "2e-4", "--lr2", "1e-4",
            "--norm", "instance", "--no_dropout1", "--no_dropout2", "--no_lsgan2", "--weights", "1.0", "3.0", "--gpu_ids", "0",
            "--checkpoints_dir", "/tmp/sgan_ckpt"]
    random.seed(1234)
    m = create_model(TrainOptions().parse(argv, save=False, verbose=False))
    m.netG1.load_state_dict(O.init_unet(1, 7, 1, 2, 8, -1))
    m.netG2.load_state_dict(O.init_unet(2, 7, 2, 1, 8, -1))
    for i, (nl, sf) in enumerate(zip(cfg.n_layers_D2, cfg.scale_factor2)):
        m.netD2[i].load_state_dict(O.init_nlayer_d(3 + i, 3, 8, nl, sf))
    crop = lambda t: t[:, :, :64, :64].detach().cpu()       # noqa: E731
    losses = []
    for step in range(g["losses"].shape[0]):
        m.set_input(segm_input(cfg, step))
        m.optimize_parameters()
        torch.cuda.synchronize()
        if step == 0:
            for name, t in (("logit", m.logit), ("fake_A", m.fake_A), ("recon_A", m.recon_A)):
                assert O.rel_err(crop(t), torch.from_numpy(g[f"step1/{name}_crop"])) < 1e-3, name
        losses.append(list(m.get_current_errors().values()))
    assert np.abs(np.asarray(losses) - g["losses"]).max() < 5e-3 * max(1.0, np.abs(g["losses"]).max()), (losses, g["losses"])

"""CPU-only tests (`-m "not gpu"`): host-side mirror of the reference interface, checkpoint layout,
C-ABI surface.  No kernel is launched here."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

import sgan_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_declared_symbol(built_lib):
    """libsgan_hip.so loads without a GPU and exports exactly what include/sgan_hip.h declares."""
    from supervised_gan_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "sgan_hip.h")).read()
    declared = set(re.findall(r"\b(sgan_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sgan_norm_desc", "sgan_conv_desc"}
    assert len(declared) >= 18, declared
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in sgan_hip.h but not exported"
    assert declared - {"sgan_version", "sgan_last_error", "sgan_last_kernel"} == set(_lib.SIGNATURES), \
        "ctypes SIGNATURES table out of sync with the header"
    l = _lib.lib()
    assert b"gfx950" in l.sgan_version()
    # argument validation happens before any launch: callable without a GPU
    rc = l.sgan_conv_fwd(None, None, 0, None, None, None, None, 0, 0, None, None, 0, None)
    assert rc < 0 and b"null" in l.sgan_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from supervised_gan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsgan_hip.so")
    with pytest.raises(_lib.SganError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_define_G_D_state_dict_layout_matches_reference():
    """Key names, logical shapes and parameter counts of the reference nets (golden via the oracle's
    init tables, which tests/test_oracle_golden.py pins against the reference)."""
    from supervised_gan_amd import networks as N
    G = N.define_G(2, 0, 32, "deconv", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8)   # README alias
    ref = O.init_fcgan_g(0, 8, 2, 32, 5)
    sd = G.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert sum(p.numel() for p in G.parameters()) == 1772448
    for s, extra in ((1, 0), (2, 100), (4, 324)):
        D = N.define_D(2, 32, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=s)
        ref = O.init_nlayer_d(0, 2, 32, 3, s)
        sd = D.state_dict()
        assert set(sd.keys()) == set(ref.keys())
        for k in ref:
            assert tuple(sd[k].shape) == tuple(ref[k].shape), k
        assert sum(p.numel() for p in D.model.parameters()) == 693729           # what optimizer_D sees
        assert sum(p.numel() for p in D.parameters()) == 693729 + extra
        if s > 1:
            assert torch.allclose(sd["gauss_filter.0.weight"], ref["gauss_filter.0.weight"], atol=1e-7)
    with pytest.raises(NotImplementedError):
        N.define_G(2, 0, 32, "no_such_net")
    with pytest.raises(NotImplementedError):
        N.define_D(2, 32, "no_such_net")


def test_weights_init_distributions_and_load_roundtrip():
    from supervised_gan_amd import networks as N
    torch.manual_seed(0)
    G = N.define_G(2, 0, 32, "fcgan", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8)
    sd = G.state_dict()
    w = sd["model.3.weight"]
    assert abs(float(w.mean())) < 1e-3 and abs(float(w.std()) - 0.02) < 1e-3          # N(0, .02)
    assert abs(float(sd["model.4.weight"].mean()) - 1.0) < 0.01                          # BN gamma N(1, .02)
    assert float(sd["model.4.bias"].abs().max()) == 0.0
    b = sd["model.3.bias"]
    assert float(b.abs().max()) <= 1.0 / np.sqrt(256 * 16) + 1e-7 and float(b.std()) > 0  # torch default bias init
    # padded storage stays zero: the 2-channel output layer is stored with 4 output channels
    L = G.layers[-1]
    master = G._flat[L.w_off: L.w_off + 16 * 4 * 32].view(4, 4, 4, 32)
    assert float(master[:, :, 2:, :].abs().max()) == 0.0
    # load_state_dict from reference-layout tensors, read back identical
    ref = O.init_fcgan_g(5, 8, 2, 32, 5)
    G.load_state_dict(ref)
    for k, v in G.state_dict().items():
        assert torch.equal(v, ref[k].detach()), k
    # parameters are views of ONE flat buffer (single Adam segment / single all-reduce)
    base = G._flat.data_ptr()
    for p in G.parameters():
        assert base <= p.data_ptr() < base + G._flat.numel() * 4
        assert p.grad is not None and p.grad.shape == p.shape


def test_old_torch_checkpoint_compat(tmp_path):
    from supervised_gan_amd import networks as N
    from supervised_gan_amd.base_model import load_state_dict_compat
    D = N.define_D(2, 8, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=1)
    sd = {k: v.clone() for k, v in O.init_nlayer_d(3, 2, 8, 3, 1).items()}
    sd["model.3.running_mean"] = torch.zeros(16)        # torch<=0.3 InstanceNorm buffers
    sd["model.3.running_var"] = torch.ones(16)
    load_state_dict_compat(D, sd)
    assert torch.equal(D.state_dict()["model.2.weight"], sd["model.2.weight"].detach())
    G = N.define_G(2, 0, 8, "fcgan", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8)
    sdg = {k: v.clone() for k, v in O.init_fcgan_g(3, 8, 2, 8, 5).items() if "num_batches_tracked" not in k}
    load_state_dict_compat(G, sdg)                       # torch<0.4.1: no num_batches_tracked
    with pytest.raises(KeyError):
        load_state_dict_compat(G, {"bogus.weight": torch.zeros(1)})


def test_image_pool_policy_matches_reference_restatement():
    from supervised_gan_amd.image_pool import ImagePool
    random.seed(7)
    a = ImagePool(5)
    outs_a = [float(a.query(torch.full((1, 1, 2, 2), float(i)))[0, 0, 0, 0]) for i in range(40)]
    random.seed(7)
    b = O.ImagePool(5)
    outs_b = [float(b.query(torch.full((1, 1, 2, 2), float(i)))[0, 0, 0, 0]) for i in range(40)]
    assert outs_a == outs_b
    assert outs_a[:5] == [0.0, 1.0, 2.0, 3.0, 4.0] and any(o < i for i, o in enumerate(outs_a))
    assert float(ImagePool(0).query(torch.ones(1, 1, 2, 2)).sum()) == 4.0


def test_options_surface_and_trainer_on_cpu_builds_but_refuses_to_run():
    from supervised_gan_amd.fcgan_model import FCGANModel
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    argv = ("--name t --model fcgan --which_direction A --fineSize 128 --input_nc 2 --which_model_netG deconv --n_layers_G 5 "
            "--ngf 8 --which_model_netD n_layers --n_layers_D 3 3 3 --ndf 8 --scale_factor 1 2 4 --lambda_D 0.5 0.4 0.1 "
            "--noise_nc 8 --noiseSize 2 --norm instance --no_dropout --n_update_G 2 --no_lsgan --which_channel rg "
            "--gpu_ids -1 --checkpoints_dir /tmp/sgan_ckpt_cpu").split()
    opt = TrainOptions().parse(argv, save=False, verbose=False)
    assert opt.gpu_ids == [] and opt.scale_factor == [1, 2, 4] and opt.lambda_D == [0.5, 0.4, 0.1] and opt.isTrain
    assert opt.pool_size == 50 and opt.beta1 == 0.5 and opt.lr == 2e-4 and opt.n_update_D == 1
    m = create_model(opt)
    assert isinstance(m, FCGANModel) and opt.input_nc == 2 and len(m.netD) == 3
    from supervised_gan_amd._lib import SganError
    with pytest.raises(SganError, match="MI355X"):
        m.netG.forward(torch.zeros(1, 8, 2, 2))
    opt2 = TrainOptions().parse(argv[:2] + ["--model", "no_such_model"] + argv[4:], save=False, verbose=False)
    with pytest.raises(ValueError):
        create_model(opt2)
    # LR schedule (fcgan_model.py:228-236): linear decay by lr/niter_decay per call
    m.optimizer_D.sync_lr = lambda: None
    m.optimizer_G.sync_lr = lambda: None
    m.update_learning_rate()
    assert abs(m.optimizer_G.param_groups[0]["lr"] - (2e-4 - 2e-4 / 100)) < 1e-12


def test_phase_tables_cover_every_output_exactly_once():
    """Geometry of the implicit GEMM (restated in numpy from sgan_common.h): for ConvT k4 s2 p1 the 4
    phases x 4 taps tile the output, and each (ky,kx) weight tap belongs to exactly one phase."""
    k, s, p, Hin = 4, 2, 1, 5
    Hout = (Hin - 1) * s - 2 * p + k
    cover = np.zeros((Hout, Hout), int)
    taps_seen = set()
    for a in range(s):
        for b in range(s):
            Hp, Wp = -(-(Hout - a) // s), -(-(Hout - b) // s)
            for py in range(Hp):
                for px in range(Wp):
                    cover[py * s + a, px * s + b] += 1
            for ky in range(k):
                if (a + p - ky) % s:
                    continue
                for kx in range(k):
                    if (b + p - kx) % s:
                        continue
                    assert (ky, kx) not in taps_seen
                    taps_seen.add((ky, kx))
    assert (cover == 1).all() and len(taps_seen) == k * k


def test_rand_f_score_vs_restatement():
    """compute_Rand_F_scores (scipy labelling + one bincount) == the plain-loop restatement of util/util.py:86-128, on hand-built
    boundary maps (a perfect prediction scores 1, a merged and a split region less) and on random ones."""
    import rand_score as R
    from supervised_gan_amd.util import compute_Rand_F_scores
    t = np.zeros((24, 24))
    t[:, 8] = t[:, 16] = t[12, :] = 1                       # six regions
    assert abs(compute_Rand_F_scores(t, t)[0] - 1.0) < 1e-12
    merged = t.copy()
    merged[:, 8] = 0
    merged[12, :] = 1
    split = t.copy()
    split[6, :] = 1
    diag = np.zeros((24, 24))
    diag[np.arange(24), np.arange(24)] = 1                  # 8-connectivity: a diagonal line does NOT separate two regions
    for s in (merged, split, diag):
        got = compute_Rand_F_scores(s, t)[0]
        assert abs(got - R.rand_f_score(s, t)) < 1e-12 and 0 < got < 1
    rng = np.random.default_rng(3)
    S = (rng.random((3, 1, 20, 28)) < 0.3).astype(np.float32) * 0.9
    T = (rng.random((3, 1, 20, 28)) < 0.25).astype(np.float32)
    got = compute_Rand_F_scores(S, T)
    for k in range(3):
        assert abs(got[k] - R.rand_f_score(S[k, 0], T[k, 0])) < 1e-12
    with pytest.raises(NotImplementedError):
        compute_Rand_F_scores(S, T, do_thin=True)


def test_visualizer_and_html_page(tmp_path):
    """util/visualizer.py + util/html.py: loss_log.txt lines, images/epoch%.3d_<label>.png, index.html rebuilt with every epoch so far
    (newest first) with the reference's table structure; test.py's result page through save_images()."""
    from types import SimpleNamespace
    from supervised_gan_amd import html
    from supervised_gan_amd.visualizer import Visualizer
    opt = SimpleNamespace(isTrain=True, no_html=False, display_winsize=128, name="exp", checkpoints_dir=str(tmp_path), display_id=0)
    vis = Visualizer(opt)
    img = (np.arange(8 * 8 * 3) % 255).astype(np.uint8).reshape(8, 8, 3)
    t = torch.zeros(1, 2, 8, 8)                                   # trainers hand tensors in [-1, 1]: two channels -> zero blue plane
    for epoch in (1, 2):
        vis.display_current_results({"fake": img, "real": t}, epoch)
    vis.print_current_errors(2, 64, {"G_GAN": 0.5, "D_real": 1.25}, 0.0031)
    web = tmp_path / "exp" / "web"
    assert sorted(os.listdir(web / "images")) == ["epoch001_fake.png", "epoch001_real.png", "epoch002_fake.png", "epoch002_real.png"]
    page = (web / "index.html").read_text()
    assert page.index("epoch [2]") < page.index("epoch [1]")
    assert '<table border="1" style="table-layout: fixed;">' in page and 'href="images/epoch002_fake.png"' in page
    assert '<img style="width:128px" src="images/epoch001_real.png">' in page and "<title>Experiment name = exp</title>" in page
    log = (tmp_path / "exp" / "loss_log.txt").read_text().splitlines()
    assert log[0].startswith("================ Training Loss") and log[-1] == "(epoch: 2, iters: 64, time: 0.003) G_GAN: 0.500 D_real: 1.250 "
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(web / "images" / "epoch002_fake.png")), img)
    # test.py's page
    wp = html.HTML(str(tmp_path / "res"), "Experiment = exp, Phase = test, Epoch = latest")
    written = vis.save_images(wp, {"fake_B": img}, ["/data/testA/0007_x.png"])
    wp.save()
    assert written == [str(tmp_path / "res" / "images" / "0007_x_fake_B.png")] and os.path.exists(written[0])
    assert "<h3>0007_x</h3>" in (tmp_path / "res" / "index.html").read_text()

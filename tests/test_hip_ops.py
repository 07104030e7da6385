"""Per-kernel parity of libsgan_hip.so (through the C ABI) against plain PyTorch fp32 on CPU.
Tolerance: max|a-b| / max|b| <= 1e-3 (north-star tolerance; observed values are ~1e-6)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3
EXPECT = 2e-5     # what exact-fp32 MFMA accumulation should actually deliver


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd import ops
    from supervised_gan_amd import _lib
    _lib.lib()   # raises if libsgan_hip.so is missing -- no fallback
    return ops


@pytest.fixture(params=["bf16x3", "f32"])
def math_mode(request, hip):
    """Both arithmetic modes of the MFMA conv kernels (sgan_conv_desc.math): split-bf16 (default) and exact fp32."""
    prev = hip.get_math()
    hip.set_math(request.param)
    yield request.param
    hip.set_math(prev)


def _norm_act(x, norm, gamma, beta, act, slope):
    if norm == "in":
        x = F.instance_norm(x, eps=1e-5)
    elif norm == "bn":
        x = F.batch_norm(x, None, None, gamma, beta, training=True, eps=1e-5)
    if act == 1:
        x = F.relu(x)
    elif act == 2:
        x = F.leaky_relu(x, slope)
    return x


CASES = [
    # kind, k, s, p, cin, cout, H, W, norm(in-side), act
    ("conv", 4, 2, 2, 2, 32, 37, 41, None, 0),      # D first layer on an odd-sized 2-channel image
    ("conv", 4, 2, 2, 32, 64, 33, 33, None, 2),     # D second layer (input = LReLU(conv0), no norm)
    ("conv", 4, 2, 2, 64, 128, 17, 19, "in", 2),
    ("conv", 4, 1, 2, 128, 256, 9, 11, "in", 2),    # stride-1 pad-2 (H+1 outputs)
    ("conv", 4, 1, 2, 256, 1, 10, 12, "in", 2),     # logits head: Cout = 1
    ("convT", 4, 2, 1, 8, 64, 4, 4, None, 0),       # G first layer (latent input)
    ("convT", 4, 2, 1, 64, 32, 9, 7, "bn", 1),      # BN(gamma,beta)+ReLU on load
    ("convT", 4, 2, 1, 32, 2, 16, 16, "bn", 1),     # G last layer: Cout = 2
    ("conv", 3, 1, 1, 10, 64, 8, 8, None, 0),       # CRN-style k3 with 10 (padded to 12) channels
    ("conv", 4, 2, 1, 64, 128, 16, 16, "in", 2),    # unet down
    ("conv", 4, 2, 2, 3, 64, 64, 64, None, 0),      # cgan D first layer (3 channels)
    ("conv", 4, 2, 2, 64, 128, 129, 129, "in", 2),  # > 1 split, odd
    ("conv", 3, 1, 1, 8, 1, 20, 22, "in", 1),       # CRN output conv: k3, Cout = 1 (K = 9 taps x 4: a partial k-tile in dgrad)
    ("conv", 3, 1, 1, 64, 1, 21, 19, "bn", 1),      # one-channel head, k3, BN + ReLU on load, ragged 8 x 8 tiles (sg_conv_head_kernel)
    ("conv", 3, 1, 1, 16, 8, 12, 12, "in", 0),      # CRN stage conv: norm without activation on load
    ("conv", 3, 1, 1, 2, 8, 16, 16, None, 0),       # CRN label conv
    ("convT", 4, 2, 1, 16, 8, 6, 6, "in", 0),       # CRN ConvT upsampling, norm without activation on load
    ("conv", 4, 2, 1, 512, 512, 8, 8, "in", 2),     # U-Net inner down conv: 8 tiles x 256 k-tiles -> split-K 32 AND two wave groups
    ("convT", 4, 2, 1, 512, 256, 4, 4, "in", 1),    # U-Net inner up conv (4 phases, deep reduction on a 4x4 map)
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}_k{c[1]}s{c[2]}p{c[3]}_{c[4]}to{c[5]}_{c[6]}x{c[7]}_{c[8]}" for c in CASES])
def test_conv_layer_fwd_bwd(hip, case, math_mode):
    from hip_utils import from_buf, from_master, master_weight, pad_vec, rel, stats_of, to_buf
    from supervised_gan_amd.ops import pad4
    ops = hip
    kind, k, s, p, cin, cout, H, W, norm, act = case
    tr = kind == "convT"
    g = torch.Generator().manual_seed(1000 + CASES.index(case))
    x = (torch.randn(1, cin, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = (torch.randn(*wshape, generator=g) * 0.05).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(cin, generator=g)).requires_grad_(True) if norm == "bn" else None
    beta = (0.1 * torch.randn(cin, generator=g)).requires_grad_(True) if norm == "bn" else None
    slope = 0.2

    a = _norm_act(x, norm, gamma, beta, act, slope)
    out = F.conv_transpose2d(a, w, b, stride=s, padding=p) if tr else F.conv2d(a, w, b, stride=s, padding=p)
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    Ho, Wo = out.shape[2:]

    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, W, pad4(cin), Ho, Wo, pad4(cout), cin, cout)   # with the logical-channel hint
    desc_plain = ops.conv_desc(1 if tr else 0, k, s, p, H, W, pad4(cin), Ho, Wo, pad4(cout))
    xb = to_buf(x.detach())
    wm = master_weight(w.detach(), tr)
    bb = pad_vec(b.detach())
    st_in = stats_of(x.detach()) if norm else None
    gam = pad_vec(gamma.detach()) if gamma is not None else None
    bet = pad_vec(beta.detach()) if beta is not None else None
    in_norm = ops.norm_desc(st_in, gam, bet, H * W, 1e-5, act, slope)

    # ---- forward (+ output statistics) ----
    ob = torch.full((Ho, Wo, pad4(cout)), float("nan"), device="cuda")
    ost = torch.zeros(2 * pad4(cout), dtype=torch.float64, device="cuda")
    ops.conv_fwd(desc, xb, in_norm, wm, bb, ob, 0, ost)
    torch.cuda.synchronize()
    assert torch.isfinite(ob).all()
    e = rel(from_buf(ob, cout), out)
    assert e < TOL, e
    assert e < EXPECT * 10, e      # split-bf16: ~1e-5 of the output scale, exact fp32: ~1e-6
    ref_st = stats_of(out.detach(), "cpu")
    assert rel(ost, ref_st) < 1e-4
    if pad4(cout) != cout:
        assert float(ob[..., cout:].abs().max()) == 0.0     # padded channels stay exactly zero
        ob2 = torch.full_like(ob, float("nan"))
        ops.conv_fwd(desc_plain, xb, in_norm, wm, bb, ob2, 0, None)      # same result without the hint
        assert rel(ob2, ob) < 1e-5

    # the same forward without statistics (how a layer with no normalisation behind it is called): the first conv on the image
    # (4 stored channels, > 16 result channels) then runs on sg_conv_c4_kernel -- one MFMA per tap, weights in registers
    ob3 = torch.full_like(ob, float("nan"))
    ops.conv_fwd(desc, xb, in_norm, wm, bb, ob3, 0, None)
    torch.cuda.synchronize()
    from supervised_gan_amd import _lib as _L
    if pad4(cin) == 4 and cout > 16 and not norm and act == 0:
        assert _L.lib().sgan_last_kernel().decode() == "sg_conv_c4_kernel"
    assert rel(ob3, ob) < 2e-6

    # ---- backward data (+ act', norm sums) then norm backward ----
    Rb = to_buf(R)
    din = torch.full((H, W, pad4(cin)), float("nan"), device="cuda")
    sums = torch.zeros(2 * pad4(cin), dtype=torch.float64, device="cuda") if norm else None
    ops.conv_dgrad(desc, Rb, wm, din, xb, in_norm, sums)
    dgam = torch.zeros(pad4(cin), device="cuda") if norm == "bn" else None
    dbet = torch.zeros(pad4(cin), device="cuda") if norm == "bn" else None
    if norm:
        ops.norm_bwd_apply(din, xb, in_norm, sums, dgam, dbet)
    torch.cuda.synchronize()
    e = rel(from_buf(din, cin), x.grad)
    assert e < TOL, e
    if norm == "bn":
        assert rel(dgam[:cin], gamma.grad) < TOL
        assert rel(dbet[:cin], beta.grad) < TOL
    # same through the transposed weight copy [tap][Cin][Cout] (what the networks use: k-contiguous staging)
    wt = wm._sgan_wt        # made by sgan_pack_weights together with the split-bf16 copies (hip_utils.derived_copies)
    assert torch.equal(wt.view(k * k, pad4(cin), pad4(cout)), wm.view(k * k, pad4(cout), pad4(cin)).transpose(1, 2))
    wt_old = torch.empty_like(wm)
    ops.transpose_weights(wm, wt_old, [(0, k * k, pad4(cout), pad4(cin))])
    assert torch.equal(wt_old, wt)
    din2 = torch.full((H, W, pad4(cin)), float("nan"), device="cuda")
    sums2 = torch.zeros(2 * pad4(cin), dtype=torch.float64, device="cuda") if norm else None
    ops.conv_dgrad(desc, Rb, wt, din2, xb, in_norm, sums2, w_transposed=True)
    if norm:
        ops.norm_bwd_apply(din2, xb, in_norm, sums2)
    torch.cuda.synchronize()
    assert rel(from_buf(din2, cin), x.grad) < TOL
    if pad4(cin) != cin and not norm:
        din3 = torch.full((H, W, pad4(cin)), float("nan"), device="cuda")
        ops.conv_dgrad(desc_plain, Rb, wm, din3, xb, in_norm, None)
        torch.cuda.synchronize()
        assert rel(din3, din) < 1e-6 and float(din[..., cin:].abs().max()) == 0.0

    # ---- backward weight / bias (accumulating) ----
    dw = torch.zeros_like(wm)
    db = torch.zeros_like(bb)
    ops.conv_wgrad(desc, xb, in_norm, Rb, dw, db)
    torch.cuda.synchronize()
    e = rel(from_master(dw, k, cin, cout, tr), w.grad)
    assert e < TOL, e
    assert rel(db[:cout], b.grad) < TOL
    # accumulate semantics: a second call doubles
    ops.conv_wgrad(desc, xb, in_norm, Rb, dw, db)
    torch.cuda.synchronize()
    assert rel(from_master(dw, k, cin, cout, tr), 2 * w.grad) < TOL


def test_conv_plain_dgrad_and_tanh(hip):
    """dgrad without a forward tensor (image gradient) and the tanh epilogue / tanh backward."""
    from hip_utils import from_buf, master_weight, rel, to_buf
    from supervised_gan_amd.ops import pad4
    ops = hip
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 2, 21, 23, generator=g, requires_grad=True)
    w = (torch.randn(16, 2, 4, 4, generator=g) * 0.1)
    out = torch.tanh(F.conv2d(x, w, None, stride=2, padding=2))
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    Ho, Wo = out.shape[2:]
    desc = ops.conv_desc(0, 4, 2, 2, 21, 23, 4, Ho, Wo, 16)
    xb, wm = to_buf(x.detach()), master_weight(w, False)
    ob = torch.empty(Ho, Wo, 16, device="cuda")
    ops.conv_fwd(desc, xb, None, wm, None, ob, 3, None)
    assert rel(from_buf(ob, 16), out) < 1e-5
    d = torch.empty_like(ob)
    ops.tanh_bwd(to_buf(R), ob, d)
    dx = torch.empty(21, 23, 4, device="cuda")
    ops.conv_dgrad(desc, d, wm, dx, None, None, None)
    torch.cuda.synchronize()
    assert rel(from_buf(dx, 2), x.grad) < 1e-4
    assert float(dx[..., 2:].abs().max()) == 0.0


def test_grouped_deep_reduction(hip, math_mode):
    """Two deep 512 -> 512 problems of different size (@33x33, @33x31) in one grouped call -- 256 k-tiles per workgroup, two wave
    groups, no split along K (each alone would be split).  Forward (+ statistics) and backward-data against torch."""
    from hip_utils import from_buf, master_weight, pad_vec, rel, stats_of, to_buf
    ops = hip
    g = torch.Generator().manual_seed(77)
    sizes = [(33, 33), (33, 31)]
    cin = cout = 512
    w = torch.randn(cout, cin, 4, 4, generator=g) * 0.02
    b = torch.randn(cout, generator=g) * 0.1
    wm, bb = master_weight(w, False), pad_vec(b)
    fjobs, djobs, refs, keep = [], [], [], []
    for H, W in sizes:
        x = (torch.randn(1, cin, H, W, generator=g) * 1.5 + 0.3).requires_grad_(True)
        a = F.leaky_relu(F.instance_norm(x, eps=1e-5), 0.2)
        out = F.conv2d(a, w, b, stride=1, padding=1)
        R = torch.randn(out.shape, generator=g)
        (out * R).sum().backward()
        Ho, Wo = out.shape[2:]
        desc = ops.conv_desc(0, 4, 1, 1, H, W, cin, Ho, Wo, cout)
        xb = to_buf(x.detach())
        nd = ops.norm_desc(stats_of(x.detach()), None, None, H * W, 1e-5, 2, 0.2)
        ob = torch.full((Ho, Wo, cout), float("nan"), device="cuda")
        ost = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
        din = torch.full((H, W, cin), float("nan"), device="cuda")
        sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
        fjobs.append((desc, xb, nd, wm, bb, ob, ost))
        djobs.append((desc, to_buf(R), wm._sgan_wt, din, xb, nd, sums, 0, False, True))
        refs.append((out.detach(), x.grad, ob, ost, din, sums, nd, xb, cin))
        keep.append((desc, nd))
    ops.conv_fwd_grouped(fjobs)
    ops.conv_dgrad_grouped(djobs)
    for out, xg, ob, ost, din, sums, nd, xb, c in refs:
        ops.norm_bwd_apply(din, xb, nd, sums, None, None)
    torch.cuda.synchronize()
    for out, xg, ob, ost, din, sums, nd, xb, c in refs:
        assert rel(from_buf(ob, cout), out) < 10 * EXPECT
        assert rel(ost, stats_of(out, "cpu")) < 1e-4
        assert rel(from_buf(din, c), xg) < TOL


@pytest.mark.parametrize("s,nc", [(2, 2), (4, 2), (2, 3)])
def test_gauss_down(hip, s, nc):
    import sgan_oracle as O
    from hip_utils import from_buf, rel, to_buf
    ops = hip
    H = W = 64 + s
    x = torch.randn(1, nc, H, W, requires_grad=True)
    wg = O.gauss_filter_weight(nc, s)
    y = O.gauss_down(x, wg, s)
    R = torch.randn_like(y)
    (y * R).sum().backward()
    kg, padg = 4 * (s // 2) + 1, 2 * (s // 2)
    Ho, Wo = y.shape[2:]
    xb = to_buf(x.detach())
    out = torch.empty(Ho, Wo, 4, device="cuda")
    wgd = wg.cuda()
    ops.gauss_down_fwd(xb, nc, wgd, (nc + 1) * kg * kg, kg, padg, s, out)
    din = torch.empty(H, W, 4, device="cuda")
    ops.gauss_down_bwd(to_buf(R), nc, wgd, (nc + 1) * kg * kg, kg, padg, s, din)
    din2 = din.clone()
    ops.gauss_down_bwd(to_buf(R), nc, wgd, (nc + 1) * kg * kg, kg, padg, s, din2, accumulate=True)   # din2 += ...
    torch.cuda.synchronize()
    assert rel(din2, 2 * din) < 1e-6
    assert rel(from_buf(out, nc), y) < 1e-5
    assert rel(from_buf(din, nc), x.grad) < 1e-5
    assert float(out[..., nc:].abs().max()) == 0.0


def test_gauss_down_multi(hip):
    """Scale-2 and scale-4 pre-filters of one image in one launch each way == the single-job entry points."""
    import sgan_oracle as O
    ops = hip
    nc, H, W = 2, 37, 41
    g = torch.Generator().manual_seed(77)
    xb = torch.randn(H, W, 4, generator=g).cuda()
    xb[..., nc:] = 0
    jobs, outs, refs, douts = [], [], [], []
    for s in (2, 4):
        kg, padg = 4 * (s // 2) + 1, 2 * (s // 2)
        wg = O.gauss_filter_weight(nc, s).cuda()
        Ho, Wo = (H + 2 * padg - kg) // s + 1, (W + 2 * padg - kg) // s + 1
        out, ref = torch.empty(Ho, Wo, 4, device="cuda"), torch.empty(Ho, Wo, 4, device="cuda")
        ops.gauss_down_fwd(xb, nc, wg, (nc + 1) * kg * kg, kg, padg, s, ref)
        jobs.append((xb, out, wg, (nc + 1) * kg * kg, kg, padg, s))
        outs.append(out)
        refs.append(ref)
        douts.append(torch.randn(Ho, Wo, 4, generator=g).cuda())
    ops.gauss_down_multi_fwd(jobs, nc)
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)
    din_ref = torch.zeros(H, W, 4, device="cuda")
    for (x_, o_, wg, gcs, kg, padg, s), d in zip(jobs, douts):
        ops.gauss_down_bwd(d, nc, wg, gcs, kg, padg, s, din_ref, accumulate=True)
    base = torch.randn(H, W, 4, generator=g).cuda()
    din = base.clone()
    ops.gauss_down_multi_bwd([(din, d, wg, gcs, kg, padg, s) for (x_, o_, wg, gcs, kg, padg, s), d in zip(jobs, douts)], nc, accumulate=True)
    din2 = torch.full((H, W, 4), float("nan"), device="cuda")
    ops.gauss_down_multi_bwd([(din2, d, wg, gcs, kg, padg, s) for (x_, o_, wg, gcs, kg, padg, s), d in zip(jobs, douts)], nc)
    torch.cuda.synchronize()
    assert float((din2 - din_ref).abs().max()) <= 1e-6 * float(din_ref.abs().max())
    assert float((din - base - din_ref).abs().max()) <= 1e-5 * float(din_ref.abs().max())


@pytest.mark.parametrize("mode,target", [(0, 1.0), (0, 0.0), (1, 1.0), (1, 0.0)])
def test_gan_loss(hip, mode, target):
    from hip_utils import from_buf, rel, to_buf
    ops = hip
    x = (torch.randn(1, 1, 67, 67) * 3).requires_grad_(True)
    x.data[0, 0, 0, :4] = torch.tensor([-120.0, 120.0, -30.0, 30.0])    # exercise the -100 clamp / saturation
    t = torch.full_like(x, target)
    loss = F.binary_cross_entropy(torch.sigmoid(x), t) if mode == 0 else F.mse_loss(x, t)
    (loss * 0.37).backward()
    xb = to_buf(x.detach())
    lo = torch.zeros((), device="cuda")
    ops.gan_loss_fwd(xb, target, mode, lo)
    d = torch.empty_like(xb)
    ops.gan_loss_bwd(xb, target, mode, torch.tensor(0.37, device="cuda"), d)
    torch.cuda.synchronize()
    assert abs(float(lo) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
    assert rel(from_buf(d, 1), x.grad) < 1e-5
    assert float(d[..., 1:].abs().max()) == 0.0


def test_sigmoid_and_layout(hip):
    from hip_utils import from_buf, rel, to_buf
    ops = hip
    x = torch.randn(1, 1, 19, 19, requires_grad=True)
    p = torch.sigmoid(x)
    R = torch.randn_like(p)
    (p * R).sum().backward()
    xb = to_buf(x.detach())
    pb = torch.empty_like(xb)
    ops.sigmoid_fwd(xb, pb)
    dx = torch.empty_like(xb)
    ops.sigmoid_bwd(to_buf(R), pb, dx)
    assert rel(from_buf(pb, 1), p) < 1e-6 and rel(from_buf(dx, 1), x.grad) < 1e-5
    # layout boundary: arbitrary strides -> padded NHWC; registered views come back zero-copy
    t = torch.randn(1, 3, 9, 11, device="cuda")
    nb = ops.as_nhwc(t)
    assert nb.shape == (9, 11, 4) and torch.equal(nb[..., :3], t[0].permute(1, 2, 0)) and float(nb[..., 3].abs().max()) == 0
    v = ops.logical_view(nb, 3)
    assert torch.equal(v, t)
    assert ops.as_nhwc(v).data_ptr() == nb.data_ptr()
    assert ops.as_nhwc(v.detach()).data_ptr() == nb.data_ptr()
    assert ops.as_nhwc(t.permute(0, 1, 3, 2)).shape == (11, 9, 4)


def test_adam_matches_reference_form(hip):
    import sgan_oracle as O
    ops = hip
    n = 10007 * 4
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = O.Adam([ref_p], lr=2e-4, beta1=0.5)
    p = p0.clone().cuda()
    gr = torch.zeros(n, device="cuda")
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    state = torch.zeros(4, dtype=torch.int32, device="cuda")
    lr = torch.full((1,), 2e-4, device="cuda")
    for step in range(5):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        ref_p.grad = grad.clone()
        opt.step()
        gr.copy_(grad)
        ops.adam_multi([(p, gr, m, v, n)], lr, 0.5, 0.999, 1e-8, state)
    torch.cuda.synchronize()
    assert int(state[0]) == 5
    assert float((p.cpu() - ref_p.detach()).abs().max()) < 1e-6
    assert float((m.cpu() - opt.m[0]).abs().max()) < 1e-6 * float(opt.m[0].abs().max()) + 1e-12


@pytest.mark.parametrize("momentum", [0.0, 0.9])
def test_sgd_matches_torch(hip, momentum):
    """sgan_sgd_multi against torch.optim.SGD (CPU) over five steps, plain and with momentum; odd length (tail elements)."""
    ops = hip
    n = 10007
    g = torch.Generator().manual_seed(6)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref_p], lr=0.05, momentum=momentum)
    p, gr = p0.clone().cuda(), torch.zeros(n, device="cuda")
    buf = torch.zeros(n, device="cuda") if momentum else None
    lr = torch.full((1,), 0.05, device="cuda")
    for step in range(5):
        grad = torch.randn(n, generator=g)
        ref_p.grad = grad.clone()
        opt.step()
        gr.copy_(grad)
        ops.sgd_multi([(p, gr, buf, n)], lr, momentum)
    torch.cuda.synchronize()
    assert float((p.cpu() - ref_p.detach()).abs().max()) < 1e-5


def test_bn_running_update(hip):
    from hip_utils import stats_of
    ops = hip
    x = torch.randn(1, 8, 16, 16) * 2 + 1
    rm, rv = torch.zeros(8), torch.ones(8)
    F.batch_norm(x, rm, rv, None, None, training=True, momentum=0.1)
    st = stats_of(x)
    rmd, rvd = torch.zeros(8, device="cuda"), torch.ones(8, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    ops.bn_running_update([(st, rmd, rvd, nbt, 8, 256)], 0.1)
    torch.cuda.synchronize()
    assert float((rmd.cpu() - rm).abs().max()) < 1e-6 and float((rvd.cpu() - rv).abs().max()) < 1e-5
    assert int(nbt) == 1


def test_normal_fill_moments_and_counter(hip):
    ops = hip
    n = 1 << 20
    a = torch.empty(n, device="cuda")
    b = torch.empty(n, device="cuda")
    off = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.normal_fill(a, 123, off)
    ops.normal_fill(b, 123, off)
    torch.cuda.synchronize()
    assert int(off) == 2 * (n // 4)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a * b).mean())) < 5e-3          # the two draws are independent
    k = float(((a - a.mean()) ** 4).mean() / a.var() ** 2)
    assert abs(k - 3.0) < 0.05
    c = torch.empty(n, device="cuda")
    ops.normal_fill(c, 123, None)
    assert torch.equal(a, c)                           # counter-based: same (seed, offset) -> same numbers


def test_cat_pair_forward_backward(hip):
    """networks.cat_pair == torch.cat((a, b), 1) (one kernel, padded NHWC result registered as a zero-copy view) and its gradient
    slices; channel counts that fill, underfill and exceed one 4-channel group."""
    from supervised_gan_amd import networks as N
    g = torch.Generator().manual_seed(8)
    for Ca, Cb in ((2, 1), (1, 1), (3, 3), (4, 2)):
        a = torch.randn(1, Ca, 13, 9, generator=g).cuda().requires_grad_(True)
        b = torch.randn(1, Cb, 13, 9, generator=g).cuda().requires_grad_(True)
        r = torch.randn(1, Ca + Cb, 13, 9, generator=g).cuda()
        out = N.cat_pair(a, b)
        assert torch.equal(out, torch.cat((a, b), 1))
        buf = hip.as_nhwc(out)
        assert buf.data_ptr() == out.data_ptr() and float(buf[..., Ca + Cb:].abs().sum()) == 0.0      # zero-copy, zero padding
        (out * r).sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(a.grad, r[:, :Ca]) and torch.equal(b.grad, r[:, Ca:])
    a = torch.randn(1, 2, 8, 8).cuda()
    b = torch.randn(1, 1, 8, 8).cuda().requires_grad_(True)
    out = N.cat_pair(a, b)
    out.sum().backward()
    assert a.grad is None and torch.equal(b.grad, torch.ones_like(b))


def test_cat_pair_output_with_a_gradient_stays_zero_copy(hip):
    """The pair a conditional discriminator is fed with in the G step (the image half needs a gradient): autograd returns an alias
    of the view made in forward(); it must still map back to the concat buffer -- no sg_to_nhwc launch in front of the discriminator."""
    from supervised_gan_amd import _lib, networks as N
    a, b = torch.randn(1, 2, 16, 12).cuda(), torch.randn(1, 1, 16, 12).cuda().requires_grad_(True)
    pair = N.cat_pair(a, b)
    hip.tanh_bwd(torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda"))      # moves sgan_last_kernel on
    before = _lib.lib().sgan_last_kernel()
    buf = hip.as_nhwc(pair)
    assert buf.data_ptr() == pair.data_ptr() and buf.shape == (16, 12, 4) and _lib.lib().sgan_last_kernel() == before
    assert torch.equal(buf[..., :3].permute(2, 0, 1).unsqueeze(0), torch.cat((a, b), 1)) and float(buf[..., 3].abs().sum()) == 0.0
    # ... and the pair neither member of which needs one, handed on detached (the D step; ImagePool.query detaches as well)
    import gc
    pair = N.cat_pair(a, b.detach()).detach()
    gc.collect()
    before = _lib.lib().sgan_last_kernel()
    buf = hip.as_nhwc(pair)
    assert buf.data_ptr() == pair.data_ptr() and buf.shape == (16, 12, 4) and _lib.lib().sgan_last_kernel() == before
    assert torch.equal(buf[..., :3].permute(2, 0, 1).unsqueeze(0), torch.cat((a, b), 1)) and float(buf[..., 3].abs().sum()) == 0.0


def test_explicit_device_of_the_boundary(hip):
    """sgan_stream_device / sgan_set_device: what a host thread that never chose a device calls before its first entry point."""
    import ctypes as C
    from supervised_gan_amd import _lib
    lib = _lib.lib()
    dev = C.c_int32(-1)
    assert lib.sgan_stream_device(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(dev)) == 0 and dev.value == torch.cuda.current_device()
    assert lib.sgan_stream_device(None, C.byref(dev)) == 0 and dev.value == torch.cuda.current_device()      # the null stream: the current device
    assert lib.sgan_set_device(dev.value) == 0
    assert lib.sgan_set_device(4096) < 0 and b"hipSetDevice" in lib.sgan_last_error()
    assert lib.sgan_stream_device(None, None) < 0
    # the failed calls leave no sticky HIP error behind: the next launch of this thread is checked clean
    hip.tanh_bwd(torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda"))


def test_image_resize_bit_exact_vs_pillow(hip):
    """sgan_image_resize against Image.resize of the Pillow in this image: bilinear and bicubic, up- and down-scaling (the filter
    support grows with the down-scale factor), one axis unchanged, a 1-pixel-wide result, the aligned dataset's 2:1 shape."""
    import image_prep as IP
    ops = hip
    rng = np.random.RandomState(16)
    for (h, w, ho, wo) in [(70, 131, 48, 48), (64, 64, 143, 143), (100, 37, 37, 100), (150, 300, 140, 280), (33, 50, 33, 25),
                           (600, 400, 286, 286), (17, 19, 200, 3), (256, 256, 256, 256), (1024, 1024, 286, 572)]:
        img = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
        dev = torch.from_numpy(img).cuda()
        for f in ("bilinear", "bicubic"):
            out = ops.image_resize(dev, wo, ho, f)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), IP.resize_pil(img, wo, ho, f)), (h, w, ho, wo, f)
    from supervised_gan_amd._lib import SganError
    with pytest.raises(SganError, match="filter"):
        ops.image_resize(dev, 10, 10, 1)


def test_normal_fill_nhwc_is_the_flat_fill_in_place(hip):
    """The latent drawn straight into the generator's padded NHWC buffer holds the values normal_fill gives the contiguous
    [C, H, W] tensor, at (h, w, c); the padding channels are left alone; the offset moves the same way."""
    ops = hip
    Cr, H, W = 6, 8, 5
    flat = torch.empty(Cr * H * W, device="cuda")
    buf = torch.full((H, W, 8), 7.0, device="cuda")
    o1 = torch.tensor([11], dtype=torch.int64, device="cuda")
    o2 = o1.clone()
    ops.normal_fill(flat, 99, o1)
    ops.normal_fill_nhwc(buf, Cr, 99, o2)
    torch.cuda.synchronize()
    assert torch.equal(buf[..., :Cr].permute(2, 0, 1).contiguous().view(-1), flat)
    assert float(buf[..., Cr:].min()) == 7.0 and float(buf[..., Cr:].max()) == 7.0
    assert int(o1) == int(o2) == 11 + (Cr * H * W + 3) // 4
    # two latents + a cleared arena in one launch == two calls (bit for bit, same advance of the stream)
    a1, b1 = torch.full((H, W, 8), 7.0, device="cuda"), torch.full((H, W, 8), 7.0, device="cuda")
    a2, b2 = a1.clone(), b1.clone()
    z = torch.full((1000,), 3.0, dtype=torch.float64, device="cuda")
    o3, o4 = o1.clone(), o1.clone()
    ops.normal_fill_nhwc(a1, Cr, 99, o3)
    ops.normal_fill_nhwc(b1, Cr, 99, o3)
    ops.normal_fill_nhwc_pair(a2, b2, Cr, 99, o4, z[:998])
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(b1, b2) and not torch.equal(a2[..., :Cr], b2[..., :Cr]) and int(o3) == int(o4)
    assert float(z[:998].abs().max()) == 0.0 and float(z[998:].min()) == 3.0


def test_fused_gan_loss_one_kernel(hip):
    """_GanLossMultiFn: the forward kernel finishes the terms itself (ticket counter, left at zero) and writes the unit-gradient
    d total / d logits; backward() returns them for a registered unit gradient and rescales them for any other upstream gradient.
    Both against torch autograd on the same formula, BCE-with-sigmoid and LSGAN."""
    ops = hip
    from supervised_gan_amd import networks as N
    g = torch.Generator().manual_seed(12)
    sizes, targets, weights = [(67, 67), (35, 35), (19, 21)], [1.0, 0.0, 1.0], [0.5, 0.25, 2.0]
    for mode in (0, 1):
        raw = [(torch.randn(1, 1, h, w, generator=g) * 3).cuda() for h, w in sizes]
        for gscale in (None, 0.37):
            xs = [r.clone().requires_grad_(True) for r in raw]
            ref_x = [r.clone().double().requires_grad_(True) for r in raw]
            terms = []
            for x, t in zip(ref_x, targets):
                tt = torch.full_like(x, t)
                terms.append(torch.nn.functional.binary_cross_entropy_with_logits(x, tt) if mode == 0 else ((x - tt) ** 2).mean())
            ref_total = sum(w * l for w, l in zip(weights, terms))
            total, each = N._GanLossMultiFn.apply(targets, weights, mode, *xs)
            if gscale is None:
                one = torch.ones_like(total)
                ops.register_unit_grad(one)
                total.backward(one)
                ref_total.backward()
            else:
                (total * gscale).backward()
                (ref_total * gscale).backward()
            torch.cuda.synchronize()
            assert abs(float(total) - float(ref_total)) < 1e-5 * max(1.0, abs(float(ref_total)))
            assert torch.allclose(each.double().cpu(), torch.stack([l.detach() for l in terms]).cpu(), rtol=1e-5, atol=1e-6)
            for x, rx in zip(xs, ref_x):
                assert torch.allclose(x.grad.double(), rx.grad, rtol=2e-5, atol=1e-9), (mode, gscale)
    ws = ops._gan_loss_workspace(torch.device("cuda", 0))
    assert int(ws.view(torch.int32)[-4]) == 0 and int(ws.view(torch.int64)[8 * 16]) == 0      # the ticket counter is back at zero


# BASELINE configs[1] layer shapes (fcgan: deconv G ngf 32 on an 8x8x8 latent, PatchGAN D ndf 32 on 2x512x512)
FULL_LAYERS = [
    ("convT", 4, 2, 1, 8, 256, 8), ("convT", 4, 2, 1, 256, 256, 16), ("convT", 4, 2, 1, 256, 128, 32),
    ("convT", 4, 2, 1, 128, 64, 64), ("convT", 4, 2, 1, 64, 32, 128), ("convT", 4, 2, 1, 32, 2, 256),
    ("conv", 4, 2, 2, 2, 32, 512), ("conv", 4, 2, 2, 32, 64, 257), ("conv", 4, 2, 2, 64, 128, 129),
    ("conv", 4, 1, 2, 128, 256, 65), ("conv", 4, 1, 2, 256, 1, 66),
]


@pytest.mark.parametrize("layer", FULL_LAYERS, ids=[f"{l[0]}_{l[4]}to{l[5]}_{l[6]}" for l in FULL_LAYERS])
def test_full_size_adjoint_identities(hip, layer, math_mode):
    """Size-independent property at the full BASELINE shapes: the three kernels are exact transposes of
    one another,  <conv(x; w), r>  ==  <x, dgrad(r; w)>  ==  <w, wgrad(x, r)>  (bilinear, no activation
    kinks involved), and the bias gradient is the pixel sum of r."""
    from supervised_gan_amd.ops import pad4
    ops = hip
    kind, k, s, p, cin, cout, H = layer
    tr = kind == "convT"
    Ho = (H - 1) * s - 2 * p + k if tr else (H + 2 * p - k) // s + 1
    cs_i, cs_o = pad4(cin), pad4(cout)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.zeros(H, H, cs_i, device="cuda")
    x[..., :cin] = torch.randn(H, H, cin, device="cuda", generator=g)
    w = torch.zeros(k * k, cs_o, cs_i, device="cuda")
    w[:, :cout, :cin] = torch.randn(k * k, cout, cin, device="cuda", generator=g) * 0.05
    r = torch.zeros(Ho, Ho, cs_o, device="cuda")
    r[..., :cout] = torch.randn(Ho, Ho, cout, device="cuda", generator=g)
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, H, cs_i, Ho, Ho, cs_o)
    y = torch.empty(Ho, Ho, cs_o, device="cuda")
    from hip_utils import derived_copies
    wm, wt = derived_copies(w.view(-1), k, cs_o, cs_i)
    ops.conv_fwd(desc, x, None, wm, None, y, 0, None)
    dx = torch.empty(H, H, cs_i, device="cuda")
    ops.conv_dgrad(desc, r, wt, dx, None, None, None, w_transposed=True)
    dw = torch.zeros_like(w)
    db = torch.zeros(cs_o, device="cuda")
    ops.conv_wgrad(desc, x, None, r, dw.view(-1), db)
    torch.cuda.synchronize()
    a = float((y.double() * r.double()).sum())
    b = float((x.double() * dx.double()).sum())
    c = float((w.double() * dw.double()).sum())
    scale = float(y.double().norm() * r.double().norm())
    tol = 1e-5 if math_mode == "f32" else 3e-5      # split-bf16 drops a_lo * b_lo: ~2^-16 per product, averaged over the sums
    assert abs(a - b) <= tol * scale and abs(a - c) <= tol * scale, (a, b, c, scale)
    assert float((db.double() - r.double().sum((0, 1))).abs().max()) <= 1e-4 * float(r.double().abs().sum((0, 1)).max())


# ------------------------------------------------------------------------------------------------
# U-Net / cgan elementwise kernels
# ------------------------------------------------------------------------------------------------
def test_norm_apply_fwd_bwd(hip):
    """y = IN(u) * mask + sigma * noise into a channel slice; backward sums then norm_bwd_apply == autograd."""
    from hip_utils import from_buf, master_weight, pad_vec, rel, stats_of, to_buf
    import sgan_oracle as O
    ops = hip
    torch.manual_seed(0)
    C, H = 16, 12
    u = torch.randn(1, C, H, H) * 2 + 0.5
    mask = (torch.rand(1, C, H, H) >= 0.5).float() * 2
    noise = torch.randn(1, C, H, H)
    r = torch.randn(1, C, H, H)
    ur = u.clone().requires_grad_(True)
    y = torch.nn.functional.instance_norm(ur, eps=1e-5) * mask + 0.1 * noise
    (y * r).sum().backward()
    ub = to_buf(u)
    wide = torch.zeros(H, H, 2 * C, device="cuda")
    st = stats_of(u)
    un = ops.norm_desc(st, None, None, H * H, 1e-5, 0, 0.0)
    hw = lambda t: t[0].permute(1, 2, 0).contiguous().cuda()
    ops.norm_apply_fwd(ub, un, wide[:, :, :C], hw(mask), hw(noise), 0.1)
    assert rel(from_buf(wide[:, :, :C].contiguous(), C), y) < 1e-5
    assert float(wide[:, :, C:].abs().max()) == 0.0
    dwide = torch.zeros(H, H, 2 * C, device="cuda")
    dwide[:, :, :C] = hw(r)
    sums = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    ops.norm_apply_bwd_sums(dwide[:, :, :C], ub, un, sums, hw(mask))
    ops.norm_bwd_apply(dwide[:, :, :C], ub, un, sums)
    assert rel(from_buf(dwide[:, :, :C].contiguous(), C), ur.grad) < 1e-4


def test_stat_slices_and_accumulate(hip, math_mode):
    """A conv writing into the right half of a wider buffer with its statistics in a slice (sq_stride), and a dgrad
    accumulating into a slice: equal to the plain calls."""
    from hip_utils import from_buf, master_weight, pad_vec, rel, stats_of, to_buf
    import sgan_oracle as O
    ops = hip
    torch.manual_seed(1)
    cin, cout, H = 16, 32, 20      # both convs run the split-bf16 kernels in that mode
    x = torch.randn(1, cin, H, H)
    w = torch.randn(cout, cin, 4, 4) * 0.1
    b = torch.randn(cout)
    desc = ops.conv_desc(0, 4, 2, 1, H, H, cin, H // 2, H // 2, cout)
    xb, wm, bb = to_buf(x), master_weight(w, False), pad_vec(b)
    plain = torch.empty(H // 2, H // 2, cout, device="cuda")
    st = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
    ops.conv_fwd(desc, xb, None, wm, bb, plain, 0, st)
    wide = torch.zeros(H // 2, H // 2, 3 * cout, device="cuda")
    stw = torch.zeros(2 * 3 * cout, dtype=torch.float64, device="cuda")
    ops.conv_fwd(desc, xb, None, wm, bb, wide[:, :, cout:2 * cout], 0, stw[cout:], 3 * cout)
    torch.cuda.synchronize()
    assert torch.equal(wide[:, :, cout:2 * cout], plain)
    assert float(wide[:, :, :cout].abs().max()) == 0 and float(wide[:, :, 2 * cout:].abs().max()) == 0
    assert rel(stw[cout:2 * cout], st[:cout]) < 1e-6 and rel(stw[4 * cout:5 * cout], st[cout:]) < 1e-6   # fp32 partial sums, atomic order
    # consumer reads the slice with the sliced statistics: same result as reading the plain tensor
    desc2 = ops.conv_desc(0, 4, 2, 1, H // 2, H // 2, cout, H // 4, H // 4, cout)
    w2 = master_weight(torch.randn(cout, cout, 4, 4) * 0.1, False)
    n_plain = ops.norm_desc(st, None, None, (H // 2) ** 2, 1e-5, 2, 0.2)
    n_slice = ops.norm_desc(stw[cout:], None, None, (H // 2) ** 2, 1e-5, 2, 0.2, 3 * cout)
    o1 = torch.empty(H // 4, H // 4, cout, device="cuda")
    o2 = torch.empty_like(o1)
    ops.conv_fwd(desc2, plain, n_plain, w2, None, o1, 0, None)
    ops.conv_fwd(desc2, wide[:, :, cout:2 * cout], n_slice, w2, None, o2, 0, None)
    assert rel(o2, o1) < 1e-5
    # accumulate
    dy = torch.randn(H // 4, H // 4, cout, device="cuda")
    d1 = torch.empty_like(plain)
    s1 = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
    ops.conv_dgrad(desc2, dy, w2._sgan_wt, d1, plain, n_plain, s1, w_transposed=True)
    base = torch.randn(H // 2, H // 2, 3 * cout, device="cuda")
    dw = base.clone()
    sw = torch.zeros(2 * 3 * cout, dtype=torch.float64, device="cuda")
    ops.conv_dgrad(desc2, dy, w2._sgan_wt, dw[:, :, cout:2 * cout], wide[:, :, cout:2 * cout], n_slice, sw[cout:], 3 * cout, accumulate=True,
                   w_transposed=True)
    torch.cuda.synchronize()
    assert rel(dw[:, :, cout:2 * cout], base[:, :, cout:2 * cout] + d1) < 1e-5
    assert torch.equal(dw[:, :, :cout], base[:, :, :cout]) and torch.equal(dw[:, :, 2 * cout:], base[:, :, 2 * cout:])
    assert rel(sw[cout:2 * cout], s1[:cout]) < 1e-5 and rel(sw[4 * cout:5 * cout], s1[cout:]) < 1e-5
    # norm_bwd_apply on the slice with sliced sums == on the plain tensor
    d1c = d1.clone()
    ops.norm_bwd_apply(d1c, plain, n_plain, s1)
    dz = torch.zeros(H // 2, H // 2, 3 * cout, device="cuda")
    dz[:, :, cout:2 * cout] = d1
    ops.norm_bwd_apply(dz[:, :, cout:2 * cout], wide[:, :, cout:2 * cout], n_slice, sw[cout:], None, None, 3 * cout)
    assert rel(dz[:, :, cout:2 * cout], d1c) < 1e-5


@pytest.mark.parametrize("weighted", [False, True])
def test_weighted_l1(hip, weighted):
    from hip_utils import from_buf, master_weight, pad_vec, rel, stats_of, to_buf
    import sgan_oracle as O
    ops = hip
    torch.manual_seed(2)
    H = 64
    x = torch.randn(1, 1, H, H, requires_grad=True)
    y = torch.randn(1, 1, H, H)
    A = torch.rand(1, 2, H, H) * 2 - 1
    wts = [2.0, 5.0]
    w = None
    if weighted:
        w = torch.ones(1, 1, H, H)
        for i, wv in enumerate(wts):
            w = w + (A.narrow(1, i, 1) + 1) / 2 * (wv - 1.0)
    loss = O.weighted_l1(x, y, w) * 10.0
    (loss * 0.7).backward()
    xb, yb, ab = to_buf(x.detach()), to_buf(y), to_buf(A)
    out = torch.empty((), device="cuda")
    g = torch.empty_like(xb)
    wd = torch.tensor(wts, device="cuda")
    ops.l1w_fwd(xb, yb, 1, ab if weighted else None, wd if weighted else None, 2 if weighted else 0, 10.0, out, g)
    dx = torch.empty_like(xb)
    ops.scale(torch.tensor(0.7, device="cuda"), g, dx)
    assert abs(float(out) - float(loss.detach())) < 1e-5 * abs(float(loss.detach()))
    assert rel(from_buf(dx, 1), x.grad) < 1e-5
    assert float(dx[:, :, 1:].abs().max()) == 0


def test_bilinear_up2_and_pyramid(hip):
    from hip_utils import from_buf, rel, to_buf
    ops = hip
    torch.manual_seed(3)
    C, H, W = 8, 10, 14
    x = torch.randn(1, C, H, W, requires_grad=True)
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    r = torch.randn_like(y)
    (y * r).sum().backward()
    out = torch.empty(2 * H, 2 * W, C, device="cuda")
    st = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    ops.bilinear_up2_fwd(to_buf(x.detach()), out, st)
    din = torch.empty(H, W, C, device="cuda")
    ops.bilinear_up2_bwd(to_buf(r), din)
    torch.cuda.synchronize()
    assert rel(from_buf(out, C), y) < 1e-6
    assert rel(from_buf(din, C), x.grad) < 1e-5
    yd = y.detach().double()
    assert rel(st[:C], yd.sum((0, 2, 3))) < 1e-5 and rel(st[C:], (yd * yd).sum((0, 2, 3))) < 1e-5
    # label pyramid: AvgPool2d(2^(s+1)), s = 0..5, and its adjoint
    lab = torch.rand(1, 2, 128, 192, requires_grad=True)
    lvl = [F.avg_pool2d(lab, 2 ** (s + 1), 2 ** (s + 1)) for s in range(6)]
    rs = [torch.randn_like(t) for t in lvl]
    sum((a * b).sum() for a, b in zip(lvl, rs)).backward()
    bufs = [torch.empty(128 >> (s + 1), 192 >> (s + 1), 4, device="cuda") for s in range(6)]
    ops.avgpool_pyramid_fwd(to_buf(lab.detach()), bufs)
    dl = torch.empty(128, 192, 4, device="cuda")
    ops.avgpool_pyramid_bwd([to_buf(t) for t in rs], dl)
    torch.cuda.synchronize()
    for s in range(6):
        assert rel(from_buf(bufs[s], 2), lvl[s]) < 5e-6, s    # hierarchical vs flat summation order
    assert rel(from_buf(dl, 2), lab.grad) < 1e-6


def test_c_abi_error_paths(hip):
    """Bad arguments come back as a negative status with a message (no exception across the ABI, nothing launched): geometry that
    does not match the conv arithmetic, channel counts that are not multiples of 4, tensors beyond the 31-bit offsets, null tensors,
    a grouped call mixing layer types, a LeakyReLU slope above 1, a net input smaller than its receptive field."""
    from supervised_gan_amd import networks as N
    from supervised_gan_amd._lib import SganError
    ops = hip
    x = torch.zeros(16, 16, 8, device="cuda")
    w = torch.zeros(16 * 8 * 8, device="cuda")
    y = torch.zeros(9, 9, 8, device="cuda")
    good = ops.conv_desc(0, 4, 2, 2, 16, 16, 8, 9, 9, 8)
    ops.conv_fwd(good, x, None, w, None, y)                                           # the baseline call is fine
    with pytest.raises(SganError, match="geometry"):
        ops.conv_fwd(ops.conv_desc(0, 4, 2, 2, 16, 16, 8, 8, 8, 8), x, None, w, None, y)
    with pytest.raises(SganError, match="multiples of 4"):
        ops.conv_fwd(ops.conv_desc(0, 4, 2, 2, 16, 16, 6, 9, 9, 8), x, None, w, None, y)
    with pytest.raises(SganError, match="too large"):
        ops.conv_fwd(ops.conv_desc(0, 4, 1, 2, 16384, 16384, 4, 16385, 16385, 4), x, None, w, None, y)
    with pytest.raises(SganError, match="kernel size"):
        ops.conv_fwd(ops.conv_desc(0, 9, 1, 4, 16, 16, 8, 16, 16, 8), x, None, w, None, y)
    with pytest.raises(SganError, match="bad conv kind"):
        ops.conv_fwd(ops.conv_desc(5, 4, 2, 2, 16, 16, 8, 9, 9, 8), x, None, w, None, y)
    with pytest.raises(SganError, match="same layer type"):
        other = ops.conv_desc(0, 4, 1, 2, 16, 16, 8, 17, 17, 8)
        ops.conv_fwd_grouped([(good, x, None, w, None, y, None), (other, x, None, w, None, torch.zeros(17, 17, 8, device="cuda"), None)])
    with pytest.raises(SganError, match="slope"):
        ops.conv_fwd(good, x, ops.norm_desc(None, None, None, 256, 0.0, 2, 1.5), w, None, y)
    rc = L_raw().sgan_conv_fwd(None, None, 0, None, None, None, None, 0, 0, None, None, 0, None)
    assert rc < 0 and L_raw().sgan_last_error()
    # Round 2's memory access fault: this very test asked for k = 7 as its "unsupported kernel size" in a build whose
    # tap table had just grown to 49 entries.  The descriptor was legal, so the library ran a k7 8 -> 8 layer on a 16 x 16 map over the
    # buffers above: 3136 weights read from a 1024-element tensor, 2048 results stored into a 648-element one.  The C ABI cannot
    # see allocation sizes; the wrapper that owns the tensors now refuses before anything is launched.
    with pytest.raises(SganError, match="does not match the descriptor"):
        ops.conv_fwd(ops.conv_desc(0, 7, 1, 3, 16, 16, 8, 16, 16, 8), x, None, w, None, y)
    y16 = torch.zeros(16, 16, 8, device="cuda")
    with pytest.raises(SganError, match="weight tensor holds 1024"):
        ops.conv_fwd(ops.conv_desc(0, 7, 1, 3, 16, 16, 8, 16, 16, 8), x, None, w, None, y16)
    with pytest.raises(SganError, match="does not match the descriptor"):
        ops.conv_dgrad(ops.conv_desc(0, 7, 1, 3, 16, 16, 8, 16, 16, 8), y, torch.zeros(49 * 64, device="cuda"), x)
    with pytest.raises(SganError, match="does not match the descriptor"):
        ops.conv_wgrad(ops.conv_desc(0, 7, 1, 3, 16, 16, 8, 16, 16, 8), x, None, y, torch.zeros(49 * 64, device="cuda"), None)
    assert float(y.abs().max()) == 0.0 and float(y16.abs().max()) == 0.0           # nothing ran
    D = N.define_D(2, 8, "dcgan", gpu_ids=[0])             # five k4 s2 p1 convs, then k4 s1 p0 on what must be a 4x4 map
    with pytest.raises(SganError, match="too small"):
        D.forward(torch.rand(1, 2, 32, 32, device="cuda"))
    torch.cuda.synchronize()


def L_raw():
    from supervised_gan_amd import _lib
    return _lib.lib()


@pytest.mark.parametrize("case", [(256, 4, 2, 10, 12, "in", 2, 0), (64, 3, 1, 21, 19, "bn", 1, 3), (512, 4, 2, 9, 9, "in", 2, 0), (128, 4, 2, 33, 17, None, 2, 0),
                                  (96, 3, 1, 18, 35, "in", 0, 3), (256, 4, 1, 40, 23, None, 0, 0)],
                         ids=lambda c: f"C{c[0]}_k{c[1]}p{c[2]}_{c[3]}x{c[4]}_{c[5]}")
def test_head_forward_channel_per_thread(hip, case):
    """sg_conv_head2_kernel: the one-channel stride-1 head forward (thread = channel, input-stationary), two problems of different
    size in one launch, norm / activation on load, bias, optional tanh -- against torch."""
    from hip_utils import from_buf, master_weight, pad_vec, rel, stats_of, to_buf
    from supervised_gan_amd import _lib
    ops = hip
    C, k, p, H, W, norm, act, out_act = case
    g = torch.Generator().manual_seed(7 * C + H)
    w = torch.randn(1, C, k, k, generator=g) * 0.05
    b = torch.randn(1, generator=g) * 0.1
    gamma = (1 + 0.2 * torch.randn(C, generator=g)) if norm == "bn" else None
    beta = (0.1 * torch.randn(C, generator=g)) if norm == "bn" else None
    wm, bb = master_weight(w, False), pad_vec(b)
    jobs, refs = [], []
    for (h, w_) in ((H, W), (H + 7, W + 3)):
        x = torch.randn(1, C, h, w_, generator=g) * 1.5 + 0.3
        ref = F.conv2d(_norm_act(x, norm, gamma, beta, act, 0.2), w, b, stride=1, padding=p)
        if out_act == 3:
            ref = torch.tanh(ref)
        ho, wo = ref.shape[2:]
        desc = ops.conv_desc(0, k, 1, p, h, w_, C, ho, wo, 4, C, 1)
        nd = ops.norm_desc(stats_of(x) if norm else None, pad_vec(gamma) if gamma is not None else None, pad_vec(beta) if beta is not None else None,
                           h * w_, 1e-5, act, 0.2)
        out = torch.full((ho, wo, 4), float("nan"), device="cuda")
        jobs.append((desc, to_buf(x), nd, wm, bb, out, None, 0, 0))
        refs.append((out, ref))
    ops.conv_fwd_grouped(jobs, out_act)
    torch.cuda.synchronize()
    assert _lib.lib().sgan_last_kernel().decode() == "sg_conv_head2_kernel"
    for out, ref in refs:
        assert rel(from_buf(out, 1), ref) < 2e-5
        assert float(out[..., 1:].abs().max()) == 0.0


@pytest.mark.parametrize("case", [(256, 4, 2, 10, 12, "in", 2), (64, 3, 1, 21, 19, "bn", 1), (512, 4, 2, 9, 9, "in", 2), (128, 4, 2, 33, 17, None, 2)],
                         ids=lambda c: f"C{c[0]}_k{c[1]}p{c[2]}_{c[3]}x{c[4]}_{c[5]}")
def test_head_backward_one_launch(hip, case):
    """sgan_conv_head_bwd (sg_head_bwd_kernel): backward-data (+ activation derivative, norm-backward sums) and backward-weight / bias of
    the one-channel stride-1 head in ONE launch, two problems grouped, against torch autograd through norm + activation + conv."""
    from hip_utils import from_buf, from_master, master_weight, pad_vec, rel, stats_of, to_buf
    from supervised_gan_amd import _lib
    ops = hip
    C, k, p, H, W, norm, act = case
    g = torch.Generator().manual_seed(C + H)
    w = (torch.randn(1, C, k, k, generator=g) * 0.05).requires_grad_(True)
    b = (torch.randn(1, generator=g) * 0.1).requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).requires_grad_(True) if norm == "bn" else None
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True) if norm == "bn" else None
    wm = master_weight(w.detach(), False)
    dw, db = torch.zeros_like(wm), torch.zeros(4, device="cuda")
    djobs, wjobs, refs = [], [], []
    for (h, w_) in ((H, W), (H + 5, W + 2)):
        x = (torch.randn(1, C, h, w_, generator=g) * 1.5 + 0.3).requires_grad_(True)
        a = _norm_act(x, norm, gamma, beta, act, 0.2)
        out = F.conv2d(a, w, b, stride=1, padding=p)
        R = torch.randn(out.shape, generator=g)
        (out * R).sum().backward()
        ho, wo = out.shape[2:]
        desc = ops.conv_desc(0, k, 1, p, h, w_, C, ho, wo, 4, C, 1)
        xb, Rb = to_buf(x.detach()), to_buf(R)
        nd = ops.norm_desc(stats_of(x.detach()) if norm else None, pad_vec(gamma.detach()) if gamma is not None else None,
                           pad_vec(beta.detach()) if beta is not None else None, h * w_, 1e-5, act, 0.2)
        din = torch.full((h, w_, C), float("nan"), device="cuda")
        sums = torch.zeros(2 * C, dtype=torch.float64, device="cuda") if norm else None
        djobs.append((desc, Rb, wm._sgan_wt, din, xb, nd, sums, 0, False, True, 0))
        wjobs.append((desc, xb, nd, Rb, dw, db))
        refs.append((x, din, xb, nd, sums))
    assert ops.conv_bwd_grouped(djobs, wjobs) is True
    assert _lib.lib().sgan_last_kernel().decode() == "sg_head_bwd_kernel"
    dgam = torch.zeros(C, device="cuda") if norm == "bn" else None
    dbet = torch.zeros(C, device="cuda") if norm == "bn" else None
    for x, din, xb, nd, sums in refs:
        if norm:
            ops.norm_bwd_apply(din, xb, nd, sums, dgam, dbet)
    torch.cuda.synchronize()
    for x, din, xb, nd, sums in refs:
        assert rel(from_buf(din, C), x.grad) < 2e-5
    assert rel(from_master(dw, k, C, 1, False), w.grad) < 2e-5
    assert abs(float(db[0]) - float(b.grad)) < 2e-5 * max(1.0, abs(float(b.grad))) and float(db[1:].abs().max()) == 0.0
    if norm == "bn":
        assert rel(dgam, gamma.grad) < 1e-4 and rel(dbet, beta.grad) < 1e-4
    # the untransposed weight layout and "input gradient only" (the generator step with --skip_wasted_D_wgrad)
    din2 = torch.full_like(refs[0][1], float("nan"))
    d0 = djobs[0]
    from supervised_gan_amd import _lib as LL
    arr = ops._dgrad_array([(d0[0], d0[1], wm, din2, d0[4], d0[5], None, 0, False, False, 0)])
    assert LL.lib().sgan_conv_head_bwd(arr, None, 1, ops._stream()) == 0
    torch.cuda.synchronize()
    sums0 = torch.zeros(2 * C, dtype=torch.float64, device="cuda") if norm else None
    din3 = torch.full_like(din2, float("nan"))
    ops.conv_dgrad(d0[0], d0[1], wm, din3, d0[4], d0[5], sums0)      # the generic kernel, same raw result (before the norm backward)
    torch.cuda.synchronize()
    assert rel(din2, din3) < 2e-6


@pytest.mark.parametrize("case", [("convT", 4, 2, 1, 32, 2, [(64, 64)], "bn", 1), ("convT", 4, 2, 1, 32, 2, [(37, 50)], "bn", 1),
                                  ("conv", 4, 2, 2, 2, 32, [(130, 130), (66, 66), (34, 34)], None, 0), ("conv", 4, 2, 2, 2, 32, [(257, 131)], None, 0)],
                         ids=lambda c: f"{c[0]}_{c[4]}to{c[5]}_n{len(c[6])}_{c[6][0][0]}x{c[6][0][1]}")
def test_thin_pair_backward(hip, case, math_mode):
    """sgan_conv_bwd_thin_pair (sg_bwd_thin_pair_kernel): backward-data and backward-weight of a layer with a 4-channel side in one grid --
    the generator's output ConvTranspose2d (32 -> 2) and the first PatchGAN conv (2 -> 32) -- against the two grouped calls."""
    from hip_utils import master_weight, pad_vec, rel, stats_of, to_buf
    from supervised_gan_amd import _lib
    from supervised_gan_amd.ops import pad4
    ops = hip
    kind, k, s_, p, cin, cout, sizes, norm, act = case
    tr = kind == "convT"
    g = torch.Generator().manual_seed(41 + cin)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    wm = master_weight(torch.randn(*wshape, generator=g) * 0.05, tr)
    gam = pad_vec(1 + 0.2 * torch.randn(cin, generator=g)) if norm == "bn" else None
    bet = pad_vec(0.1 * torch.randn(cin, generator=g)) if norm == "bn" else None
    probs = []
    for (H, W) in sizes:
        x = torch.randn(1, cin, H, W, generator=g) * 1.5 + 0.3
        ho, wo = ((H - 1) * s_ - 2 * p + k, (W - 1) * s_ - 2 * p + k) if tr else ((H + 2 * p - k) // s_ + 1, (W + 2 * p - k) // s_ + 1)
        desc = ops.conv_desc(1 if tr else 0, k, s_, p, H, W, pad4(cin), ho, wo, pad4(cout), cin, cout)
        nd = ops.norm_desc(stats_of(x), gam, bet, H * W, 1e-5, act, 0.2) if norm else None
        probs.append((desc, to_buf(x), nd, to_buf(torch.randn(1, cout, ho, wo, generator=g)), H, W))
    res = {}
    for mode in ("apart", "pair"):
        dw, db = torch.zeros_like(wm), torch.zeros(pad4(cout), device="cuda")
        dj, wj, keep = [], [], []
        for desc, xb, nd, rb, H, W in probs:
            din = torch.full((H, W, pad4(cin)), float("nan"), device="cuda")
            sums = torch.zeros(2 * pad4(cin), dtype=torch.float64, device="cuda") if norm else None
            dj.append((desc, rb, wm._sgan_wt, din, xb if norm else None, nd, sums, 0, False, True, 0))
            wj.append((desc, xb, nd, rb, dw, db))
            keep.append((din, sums))
        if mode == "apart":
            ops.conv_wgrad_grouped(wj)
            ops.conv_dgrad_grouped(dj)
        else:
            assert ops.conv_bwd_grouped(dj, wj) is True
            assert _lib.lib().sgan_last_kernel().decode() == "sg_bwd_thin_pair_kernel"
        torch.cuda.synchronize()
        res[mode] = (keep, dw, db)
    for (da, sa), (dp, sp) in zip(res["apart"][0], res["pair"][0]):
        assert torch.isfinite(dp).all() and rel(dp, da) < 2e-6
        if sa is not None:
            assert rel(sp, sa) < 1e-6
    assert rel(res["pair"][1], res["apart"][1]) < 3e-6 and rel(res["pair"][2], res["apart"][2]) < 3e-6


def test_head_backward_declines_full_size_maps(hip):
    """Past ~128 tiles per weight tensor the same-address atomics of sg_head_bwd_kernel cost more than the two generic launches
    (CRN output conv on 512 x 512: 300 us against 70): sgan_conv_head_bwd answers 1 and conv_bwd_grouped runs the generic pair."""
    from hip_utils import master_weight
    from supervised_gan_amd import _lib
    ops = hip
    C, k, p, H, W = 64, 3, 1, 160, 160
    wm = master_weight(torch.randn(1, C, k, k) * 0.05, False)
    x, R = torch.randn(H, W, C, device="cuda"), torch.randn(H, W, 4, device="cuda")
    desc = ops.conv_desc(0, k, 1, p, H, W, C, H, W, 4, C, 1)
    nd = ops.norm_desc(None, None, None, H * W, 1e-5, 1, 0.0)
    din, dw, db = torch.empty(H, W, C, device="cuda"), torch.zeros_like(wm), torch.zeros(4, device="cuda")
    dj, wj = [(desc, R, wm._sgan_wt, din, x, nd, None, 0, False, True, 0)], [(desc, x, nd, R, dw, db)]
    assert _lib.lib().sgan_conv_head_bwd(ops._dgrad_array(dj), ops._wgrad_array(wj), 1, ops._stream()) == 1
    ops.conv_bwd_grouped(dj, wj)
    torch.cuda.synchronize()
    assert _lib.lib().sgan_last_kernel().decode() != "sg_head_bwd_kernel" and float(dw.abs().max()) > 0


@pytest.mark.parametrize("C,H,W,weighted", [(3, 67, 67, False), (3, 35, 19, True), (5, 64, 48, True), (12, 9, 7, False)])
def test_cross_entropy_and_softmax_kernels(hip, C, H, W, weighted):
    """sgan_ce_fwd / sgan_ce_bwd against F.cross_entropy (class weights, an int64 label map with ignored pixels, and the one-class
    form GANLossMultiClass uses), sgan_softmax_fwd / _bwd against F.softmax, through their autograd wrappers (losses.py)."""
    from supervised_gan_amd.losses import cross_entropy_logits, softmax_channels
    g = torch.Generator().manual_seed(C * 100 + H)
    z = (torch.randn(1, C, H, W, generator=g) * 3).requires_grad_(True)
    lab = torch.randint(0, C, (1, H, W), generator=g)
    lab[0, 0, :3] = -100                                    # torch's ignore_index
    cw = (torch.rand(C, generator=g) + 0.5) if weighted else None
    ref = F.cross_entropy(z, lab, weight=cw)
    ref.backward()
    zc = z.detach().cuda().requires_grad_(True)
    loss = cross_entropy_logits(zc, lab.cuda(), 0, cw.cuda() if cw is not None else None)
    (loss * 1.7).backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    assert float((zc.grad.cpu() - 1.7 * z.grad).abs().max()) < 1e-6 * float(z.grad.abs().max()) + 1e-9
    for k in (0, C - 1):                                    # every pixel the same class (the multi-class GAN objective)
        z2 = z.detach().clone().requires_grad_(True)
        r2 = F.cross_entropy(z2.permute(0, 2, 3, 1).reshape(-1, C), torch.full((H * W,), k, dtype=torch.long))
        r2.backward()
        z2c = z.detach().cuda().requires_grad_(True)
        l2 = cross_entropy_logits(z2c, None, k)
        l2.backward()
        assert abs(float(l2) - float(r2)) < 2e-6 * max(1.0, abs(float(r2)))
        assert float((z2c.grad.cpu() - z2.grad).abs().max()) < 1e-6 * float(z2.grad.abs().max()) + 1e-9
    z3 = z.detach().clone().requires_grad_(True)
    R = torch.randn(1, C, H, W, generator=g)
    (F.softmax(z3, dim=1) * R).sum().backward()
    z3c = z.detach().cuda().requires_grad_(True)
    p = softmax_channels(z3c)
    (p * R.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert float((p.detach().cpu() - F.softmax(z.detach(), dim=1)).abs().max()) < 1e-6
    assert float((z3c.grad.cpu() - z3.grad).abs().max()) < 2e-6 * float(z3.grad.abs().max()) + 1e-9


GUARD = 4096      # fp32 elements of sentinel behind every operand (16 KB: more than any vector over-read)


def _guarded(t, fill=float("nan")):
    """A copy of `t` whose storage ends in GUARD sentinel elements (NaN: an over-READ that is used turns the result NaN; an
    over-WRITE changes the sentinel's bit pattern).  Returns (view shaped like t, the guard)."""
    flat = torch.full((t.numel() + GUARD,), fill, dtype=t.dtype, device="cuda")
    flat[: t.numel()] = t.reshape(-1)
    return flat[: t.numel()].view(t.shape), flat[t.numel():]


def _guard_intact(g):
    return bool(torch.isnan(g).all())


@pytest.mark.parametrize("case", [
    ("conv", 7, 1, 3, 8, 8, 16, 16),        # the layer round 2's fault ran on undersized buffers (k7, 49 taps)
    ("conv", 7, 1, 0, 8, 64, 22, 22),       # resnet head after the reflection pad (49 taps, p0)
    ("conv", 7, 1, 0, 64, 2, 22, 22),       # resnet output layer: 49 taps into a 4-channel stored result
    ("convT", 3, 2, 1, 32, 16, 9, 7),       # resnet up layer with output_padding = 1 (Hout = 2 Hin)
    ("conv", 4, 2, 2, 2, 32, 37, 41),       # first PatchGAN conv (4 stored channels: c4 / thin kernels)
    ("conv", 4, 1, 2, 256, 1, 10, 12),      # logits head
], ids=lambda c: f"{c[0]}_k{c[1]}s{c[2]}p{c[3]}_{c[4]}to{c[5]}_{c[6]}x{c[7]}")
def test_conv_operands_with_guarded_tails(hip, case, math_mode):
    """Every operand of forward / backward-data / backward-weight is placed so that its storage ends in a NaN guard: results match
    torch (an over-read that reaches a product would make them NaN) and every guard keeps its bit pattern (no store past an end).
    Regression shapes for the round-2 fault: the 49-tap layers, ConvT output padding, the 4-channel kernels."""
    from hip_utils import from_buf, from_master, master_weight, pad_vec, rel, to_buf
    from supervised_gan_amd.ops import pad4
    ops = hip
    kind, k, s, p, cin, cout, H, W = case
    tr = kind == "convT"
    op = 1 if (tr and k == 3) else 0
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(1, cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), generator=g) * 0.05).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).requires_grad_(True)
    out = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op) if tr else F.conv2d(x, w, b, stride=s, padding=p)
    R = torch.randn(out.shape, generator=g)
    (out * R).sum().backward()
    Ho, Wo = out.shape[2:]
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, W, pad4(cin), Ho, Wo, pad4(cout), cin, cout)
    guards = []

    def G(t):
        v, gd = _guarded(t)
        guards.append(gd)
        return v
    xb, Rb, bb = G(to_buf(x.detach())), G(to_buf(R)), G(pad_vec(b.detach()))
    wm0 = master_weight(w.detach(), tr)
    wm, wt = G(wm0), G(wm0._sgan_wt)
    pf, pb, pbh = G(wm0._sgan_pk), G(wm0._sgan_wt._sgan_pk), G(wm0._sgan_wt._sgan_pk16)
    ops.with_packed(wm, pf); ops.with_packed(wt, pb, pbh)
    ob, din = G(torch.zeros(Ho, Wo, pad4(cout), device="cuda")), G(torch.zeros(H, W, pad4(cin), device="cuda"))
    dw, db = G(torch.zeros_like(wm0)), G(torch.zeros(pad4(cout), device="cuda"))
    ops.conv_fwd(desc, xb, None, wm, bb, ob)
    ops.conv_dgrad(desc, Rb, wt, din, None, None, None, w_transposed=True)
    ops.conv_wgrad(desc, xb, None, Rb, dw, db)
    torch.cuda.synchronize()
    assert rel(from_buf(ob, cout), out) < 1e-4
    assert rel(from_buf(din, cin), x.grad) < TOL
    assert rel(from_master(dw, k, cin, cout, tr), w.grad) < TOL and rel(db[:cout], b.grad) < TOL
    assert all(_guard_intact(gd) for gd in guards), [i for i, gd in enumerate(guards) if not _guard_intact(gd)]


def test_image_prep_bit_exact(hip):
    """sgan_image_prep against the Pillow sequence of the reference's transforms: every flip / rotation, windows touching the image
    border, a non-square source; integer gather and the two fp32 divisions are bit-exact."""
    import image_prep as IP
    from hip_utils import from_buf
    ops = hip
    rng = np.random.RandomState(6)
    img = rng.randint(0, 256, size=(70, 131, 3), dtype=np.uint8)
    dev = torch.from_numpy(img).cuda()
    for x0, y0, n in ((0, 0, 70), (61, 0, 70), (17, 9, 48), (130, 69, 1)):
        for flip in (False, True):
            for rot in range(4):
                buf = ops.image_prep(dev, x0, y0, n, flip, rot)
                torch.cuda.synchronize()
                assert tuple(buf.shape) == (n, n, 4) and float(buf[..., 3].abs().max()) == 0.0
                assert np.array_equal(from_buf(buf, 3)[0].numpy(), IP.prep_pil(img, x0, y0, n, flip, rot)), (x0, y0, n, flip, rot)
    from supervised_gan_amd._lib import SganError
    with pytest.raises(SganError, match="outside"):
        ops.image_prep(dev, 100, 0, 70, False, 0)

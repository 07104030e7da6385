"""The split-bf16 ("bf16x3") arithmetic mode of the conv kernels: the packed weight copies bit for bit, every tile shape of
sg_igemm3_kernel against the exact-fp32 kernel and against an fp64 CPU result, ragged sizes, grouped launches and split-K."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    from supervised_gan_amd import _lib, ops
    _lib.lib()
    prev = ops.get_math()
    yield ops
    ops.set_math(prev)
    os.environ.pop("SGAN_TILE3", None)
    os.environ.pop("SGAN_IGEMM3P", None)


def _bf16_rne_bits(x):
    """uint16 bits of bf16(x), round to nearest even -- numpy restatement of what v_cvt_pk_bf16_f32 does on finite values."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def _split(x):
    hi = _bf16_rne_bits(x)
    hif = (hi.astype(np.uint32) << 16).view(np.float32)
    lo = _bf16_rne_bits(x.astype(np.float32) - hif)
    return hi, lo


def _split_f16(x, shift=10):
    """fp16 planes of x * 2^shift (forward copy): numpy's float32 -> float16 conversion rounds to nearest even like v_cvt_pk_f16_f32."""
    xs = (x.astype(np.float32) * np.float32(2.0 ** shift)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16)


def test_pack_weights_bit_exact(ops):
    """flat_t / packed_fwd (fp16 planes of w * 2^10) / packed_bwd (bf16 planes) of two segments (one with Cin % 8 != 0: its forward
    copy is left untouched)."""
    rng = np.random.default_rng(5)
    segs = [(0, 16, 40, 24), (16 * 40 * 24 + 8, 9, 16, 12)]      # (off, taps, cout, cin); 8 floats of bias between them
    total = segs[1][0] + 9 * 16 * 12
    flat = (rng.standard_normal(total) * np.exp(rng.uniform(-12, 2, total))).astype(np.float32)      # |w| from 1e-6 to ~30
    f = torch.from_numpy(flat).cuda()
    ft, pf, pb, pbh = torch.zeros_like(f), torch.zeros_like(f), torch.zeros_like(f), torch.zeros_like(f)
    ops.pack_weights(f, ft, pf, pb, segs, pbh)
    torch.cuda.synchronize()
    ft, pf, pb, pbh = ft.cpu().numpy(), pf.cpu().numpy().view(np.uint16), pb.cpu().numpy().view(np.uint16), pbh.cpu().numpy().view(np.uint16)
    for off, taps, co, ci in segs:
        w = flat[off: off + taps * co * ci].reshape(taps, co, ci)
        assert np.array_equal(ft[off: off + taps * co * ci].reshape(taps, ci, co), w.transpose(0, 2, 1))
        for name, pk, rows, cols, m in (("fwd", pf, co, ci, w), ("bwd", pb, ci, co, w.transpose(0, 2, 1)), ("bwd16", pbh, ci, co, w.transpose(0, 2, 1))):
            got = pk[2 * off: 2 * (off + taps * co * ci)].reshape(taps, rows, cols // 8 if cols % 8 == 0 else 1, -1)
            if cols % 8:
                assert not got.any(), name       # untouched (zeros from the allocation)
                continue
            hi, lo = (_split if name == "bwd" else _split_f16)(np.ascontiguousarray(m))
            want = np.stack([hi.reshape(taps, rows, cols // 8, 8), lo.reshape(taps, rows, cols // 8, 8)], axis=3)
            assert np.array_equal(got.reshape(taps, rows, cols // 8, 2, 8), want), name


SHAPES = [
    # kind, k, s, p, cin, cout, H, W, norm, act
    ("conv", 4, 2, 2, 32, 64, 67, 45, None, 2),
    ("conv", 4, 1, 2, 128, 256, 33, 29, "in", 2),
    ("conv", 3, 1, 1, 64, 64, 40, 56, "in", 1),
    ("convT", 4, 2, 1, 256, 128, 16, 12, "bn", 1),
    ("convT", 4, 2, 1, 64, 32, 31, 33, "bn", 1),
    ("conv", 4, 2, 1, 24, 40, 30, 30, "in", 2),        # channel counts that are multiples of 8 but not of 32
    ("conv", 7, 1, 3, 32, 64, 30, 41, "in", 1),        # 49 taps (resnet stem shape): a 14 x 14 patch
    ("conv", 4, 2, 2, 64, 128, 65, 67, "in", 2),       # backward-data = 4 phases of 2 x 2 taps, 64 result channels (patch kernel)
    ("convT", 4, 2, 1, 128, 64, 21, 19, "bn", 1),      # forward = 4 phases of 2 x 2 taps on odd sizes
]
TILES = ["auto", "64x64", "128x64", "128x128", "patch"]
_FUSED_SEEN = {}


def _select_tile(tile):
    """SGAN_TILE3 forces a tile of sg_igemm3_kernel; "patch" forces the patch-stationary kernel wherever it is eligible."""
    os.environ.pop("SGAN_TILE3", None)
    os.environ.pop("SGAN_IGEMM3P", None)
    if tile == "patch":
        os.environ["SGAN_IGEMM3P"] = "1"
    elif tile != "auto":
        os.environ["SGAN_TILE3"] = tile
        os.environ["SGAN_IGEMM3P"] = "0"


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("shape", SHAPES, ids=[f"{c[0]}_k{c[1]}s{c[2]}_{c[4]}to{c[5]}_{c[6]}x{c[7]}" for c in SHAPES])
def test_igemm3_vs_fp32_kernel_and_fp64(ops, shape, tile):
    from hip_utils import from_buf, master_weight, pad_vec, rel, stats_of, to_buf
    from supervised_gan_amd import _lib
    kind, k, s, p, cin, cout, H, W, norm, act = shape
    tr = kind == "convT"
    _select_tile(tile)
    g = torch.Generator().manual_seed(99)
    x = (torch.randn(1, cin, H, W, generator=g) * 1.5 + 0.3).double().requires_grad_(True)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = (torch.randn(*wshape, generator=g) * 0.05).double().requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).double().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(cin, generator=g)).double() if norm == "bn" else None
    beta = (0.1 * torch.randn(cin, generator=g)).double() if norm == "bn" else None
    a = x
    if norm == "in":
        a = F.instance_norm(a, eps=1e-5)
    elif norm == "bn":
        a = F.batch_norm(a, None, None, gamma, beta, training=True, eps=1e-5)
    a = F.relu(a) if act == 1 else (F.leaky_relu(a, 0.2) if act == 2 else a)
    out = F.conv_transpose2d(a, w, b, stride=s, padding=p) if tr else F.conv2d(a, w, b, stride=s, padding=p)
    R = torch.randn(out.shape, generator=g).double()
    (out * R).sum().backward()
    Ho, Wo = out.shape[2:]
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, W, cin, Ho, Wo, cout, cin, cout)
    xb, wm, bb, Rb = to_buf(x.detach().float()), master_weight(w.detach().float(), tr), pad_vec(b.detach().float()), to_buf(R.float())
    st_in = stats_of(x.detach()) if norm else None
    in_norm = ops.norm_desc(st_in, pad_vec(gamma.float()) if gamma is not None else None, pad_vec(beta.float()) if beta is not None else None,
                            H * W, 1e-5, act, 0.2)
    res = {}
    for mode in ("f32", "bf16x3"):
        ops.set_math(mode)
        ob = torch.full((Ho, Wo, cout), float("nan"), device="cuda")
        ost = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
        ops.conv_fwd(desc, xb, in_norm, wm, bb, ob, 0, ost)
        kf = _lib.lib().sgan_last_kernel().decode()
        din = torch.full((H, W, cin), float("nan"), device="cuda")
        sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda") if norm else None
        ops.conv_dgrad(desc, Rb, wm._sgan_wt, din, xb, in_norm, sums, w_transposed=True)
        kd = _lib.lib().sgan_last_kernel().decode()
        if mode == "bf16x3" and tile == "patch" and cin % 32 == 0 and cout % 32 == 0 and H * W >= 256:
            # unit-stride gathers (forward of a stride-1 Conv2d / of any ConvTranspose2d, backward-data of any Conv2d) and, since
            # round 3, stride-2 gathers through the parity-plane patch (forward of a stride-2 Conv2d, backward-data of a stride-2
            # ConvTranspose2d); result channels > 32: narrower layers keep the 128 x 32 tile of sg_igemm3_kernel
            assert ("igemm3p" in kf) == (cout > 32) and ("igemm3p" in kd) == (cin > 32), (kf, kd)
            assert ("s2" in kf) == (not tr and s == 2 and cout > 32) and ("s2" in kd) == (tr and s == 2 and cin > 32), (kf, kd)
        if norm:
            ops.norm_bwd_apply(din, xb, in_norm, sums)
        dw, db = torch.zeros_like(wm), torch.zeros_like(bb)
        ops.conv_wgrad(desc, xb, in_norm, Rb, dw, db)
        torch.cuda.synchronize()
        res[mode] = (ob, ost, din, dw, db)
        assert torch.isfinite(ob).all() and torch.isfinite(din).all() and torch.isfinite(dw).all()
    ops.set_math("bf16x3")
    e32 = rel(from_buf(res["f32"][0], cout), out)
    e3 = rel(from_buf(res["bf16x3"][0], cout), out)
    d32 = rel(from_buf(res["f32"][2], cin), x.grad)
    d3 = rel(from_buf(res["bf16x3"][2], cin), x.grad)
    from hip_utils import from_master
    w32 = rel(from_master(res["f32"][3], k, cin, cout, tr), w.grad)
    w3 = rel(from_master(res["bf16x3"][3], k, cin, cout, tr), w.grad)
    b3 = rel(res["bf16x3"][4][:cout], b.grad)
    print(f"fwd err vs fp64: f32 {e32:.2e} bf16x3 {e3:.2e}; dgrad: f32 {d32:.2e} bf16x3 {d3:.2e}; wgrad: f32 {w32:.2e} bf16x3 {w3:.2e}")
    # forward: fp16 planes, fp32-equivalent (the same 1e-6 as the exact-fp32 kernel); backward: bf16 planes, ~5e-6
    assert e3 < 3e-6 and d3 < 1e-4 and w3 < 3e-5 and b3 < 1e-5, (e3, d3, w3, b3)
    assert rel(res["bf16x3"][1], stats_of(out.detach(), "cpu")) < 1e-4
    assert rel(res["bf16x3"][0], res["f32"][0]) < 3e-6
    # round 3: the same backward calls with the gradient's maximum published (what sgan_norm_bwd_apply_multi leaves behind):
    # fp16 planes of dOut * 2^s -- fp32-equivalent like the forward pass.  Tiny gradients (1e-7 of scale) as they occur in training.
    for gscale in (1.0, 3e-7):
        Rs = Rb * gscale
        Rs._sgan_amax = Rs.abs().max().reshape(1).float()
        din = torch.full((H, W, cin), float("nan"), device="cuda")
        sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda") if norm else None
        ops.conv_dgrad(desc, Rs, wm._sgan_wt, din, xb, in_norm, sums, w_transposed=True)
        if norm:
            ops.norm_bwd_apply(din, xb, in_norm, sums)
        dw, db = torch.zeros_like(wm), torch.zeros_like(bb)
        ops.conv_wgrad(desc, xb, in_norm, Rs, dw, db)
        torch.cuda.synchronize()
        d16 = rel(from_buf(din, cin), x.grad * gscale)
        w16 = rel(from_master(dw, k, cin, cout, tr), w.grad * gscale)
        b16 = rel(db[:cout], b.grad * gscale)
        print(f"fp16-plane backward (gradient scale {gscale:g}): dgrad {d16:.2e} wgrad {w16:.2e}")
        assert d16 < 3e-6 and w16 < 3e-6 and b16 < 1e-5, (gscale, d16, d32, w16, w32, b16)      # the forward pass's bound: fp32-equivalent


@pytest.mark.parametrize("tile", ["auto", "patch"])
def test_igemm3_grouped_and_splitk(ops, tile):
    """Three problems of one layer type and different sizes in one grouped launch, and a deep reduction on a tiny map (split-K
    through the slab epilogue), both in split-bf16 against the exact-fp32 kernels."""
    from hip_utils import master_weight, pad_vec, rel, stats_of, to_buf
    _select_tile(tile)
    g = torch.Generator().manual_seed(3)
    cin, cout = 128, 256
    w = torch.randn(cout, cin, 4, 4, generator=g) * 0.03
    wm, bb = master_weight(w, False), pad_vec(torch.randn(cout, generator=g) * 0.1)
    sizes = [(65, 65), (33, 33), (17, 19)]
    xs = [torch.randn(1, cin, H, W, generator=g) for H, W in sizes]
    out = {}
    for mode in ("f32", "bf16x3"):
        ops.set_math(mode)
        jobs, keep = [], []
        for (H, W), x in zip(sizes, xs):
            desc = ops.conv_desc(0, 4, 1, 2, H, W, cin, H + 1, W + 1, cout)
            nd = ops.norm_desc(stats_of(x), None, None, H * W, 1e-5, 2, 0.2)
            ob = torch.full((H + 1, W + 1, cout), float("nan"), device="cuda")
            ost = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
            jobs.append((desc, to_buf(x), nd, wm, bb, ob, ost))
            keep.append((ob, ost))
        ops.conv_fwd_grouped(jobs)
        torch.cuda.synchronize()
        out[mode] = keep
    for (o3, s3), (o1, s1) in zip(out["bf16x3"], out["f32"]):
        assert rel(o3, o1) < 3e-5 and rel(s3, s1) < 1e-5
    # split-K: 512 -> 512 on an 8x8 map (U-Net inner block)
    cin = cout = 512
    w = torch.randn(cout, cin, 4, 4, generator=g) * 0.02
    wm = master_weight(w, False)
    x = torch.randn(1, cin, 8, 8, generator=g)
    desc = ops.conv_desc(0, 4, 2, 1, 8, 8, cin, 4, 4, cout)
    nd = ops.norm_desc(stats_of(x), None, None, 64, 1e-5, 2, 0.2)
    got = {}
    for mode in ("f32", "bf16x3"):
        ops.set_math(mode)
        ob = torch.full((4, 4, cout), float("nan"), device="cuda")
        ost = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
        ops.conv_fwd(desc, to_buf(x), nd, wm, None, ob, 0, ost)
        torch.cuda.synchronize()
        got[mode] = (ob, ost)
    ops.set_math("bf16x3")
    assert rel(got["bf16x3"][0], got["f32"][0]) < 3e-5 and rel(got["bf16x3"][1], got["f32"][1]) < 1e-5


@pytest.mark.parametrize("shape", [("convT", 4, 2, 1, 256, 128, 32, 32, "bn", 1), ("convT", 4, 2, 1, 256, 256, 16, 16, "bn", 1),
                                   ("conv", 4, 2, 1, 512, 512, 32, 32, "in", 2), ("convT", 4, 2, 1, 512, 256, 16, 16, "in", 1)],
                         ids=lambda c: f"{c[0]}_{c[4]}to{c[5]}_{c[6]}x{c[7]}")
def test_fused_backward_with_a_split_k_half(ops, shape):
    """Deep reductions on small maps (one problem): the backward-data half keeps its split-K INSIDE the fused launch
    (sgan_conv_bwd_fused_ws: slabs in the caller's workspace, sg_splitk_epilogue_kernel afterwards) -- same results as the two
    grouped calls."""
    from hip_utils import master_weight, pad_vec, rel, stats_of, to_buf
    from supervised_gan_amd import _lib
    kind, k, s, p, cin, cout, H, W, norm, act = shape
    tr = kind == "convT"
    _select_tile("auto")
    ops.set_math("bf16x3")
    g = torch.Generator().manual_seed(23)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    wm = master_weight(torch.randn(*wshape, generator=g) * 0.05, tr)
    x = torch.randn(1, cin, H, W, generator=g) * 1.5 + 0.3
    ho, wo = ((H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k) if tr else ((H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, W, cin, ho, wo, cout, cin, cout)
    gam = pad_vec(1 + 0.2 * torch.randn(cin, generator=g)) if norm == "bn" else None
    bet = pad_vec(0.1 * torch.randn(cin, generator=g)) if norm == "bn" else None
    nd = ops.norm_desc(stats_of(x), gam, bet, H * W, 1e-5, act, 0.2)
    xb, rb = to_buf(x), to_buf(torch.randn(1, cout, ho, wo, generator=g))
    res = {}
    for mode in ("apart", "fused"):
        din = torch.full((H, W, cin), float("nan"), device="cuda")
        sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda")
        dw, db = torch.zeros_like(wm), torch.zeros(cout, device="cuda")
        dj, wj = [(desc, rb, wm._sgan_wt, din, xb, nd, sums, 0, False, True, 0)], [(desc, xb, nd, rb, dw, db)]
        if mode == "apart":
            ops.conv_wgrad_grouped(wj)
            ops.conv_dgrad_grouped(dj)
            apart_kernel = _lib.lib().sgan_last_kernel().decode()      # sg_igemm3_kernel<64,64,2,2> (split-K) on the ConvTranspose cases
        else:
            assert ops.conv_bwd_grouped(dj, wj) is True
            assert _lib.lib().sgan_last_kernel().decode() == "sg_bwd_fused_kernel"
        torch.cuda.synchronize()
        res[mode] = (din, sums, dw, db)
    a, f = res["apart"], res["fused"]
    assert torch.isfinite(f[0]).all()
    assert torch.equal(a[0], f[0]) or rel(f[0], a[0]) < 4e-6
    assert rel(f[1], a[1]) < 1e-6 and rel(f[2], a[2]) < 2e-6 and rel(f[3], a[3]) < 2e-6


@pytest.mark.parametrize("dmath", [None, "f16"])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"{c[0]}_k{c[1]}s{c[2]}_{c[4]}to{c[5]}_{c[6]}x{c[7]}" for c in SHAPES])
def test_fused_backward_equals_the_two_launches(ops, shape, dmath):
    """sgan_conv_bwd_fused (one grid for a layer's backward-data and backward-weight) against sgan_conv_dgrad_grouped +
    sgan_conv_wgrad_grouped on the same two-problem job lists: the input gradients bit for bit (same body, same tile order), the
    weight / bias gradients and the norm-backward sums up to the order of their atomic adds.  dmath "f16": the gradient tensors carry
    their published maximum, so backward-data runs on fp16 planes (round 3: what replaced the exact-fp32 variant for backward-data
    into a layer without a normalisation); the backward-weight half reads an untagged alias and stays on bf16 planes in both runs."""
    from hip_utils import master_weight, pad_vec, rel, stats_of, to_buf
    from supervised_gan_amd import _lib
    kind, k, s, p, cin, cout, H, W, norm, act = shape
    tr = kind == "convT"
    if dmath == "f16" and cin > 32:
        pytest.skip("the fused launch carries fp16 planes only for its 128 x 32 backward-data variant (<= 32 result channels)")
    _select_tile("auto")
    ops.set_math("bf16x3")
    g = torch.Generator().manual_seed(17)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    wm = master_weight(torch.randn(*wshape, generator=g) * 0.05, tr)
    sizes = [(H, W), (H + 3, W + 2)]
    probs = []
    for h, w_ in sizes:
        x = torch.randn(1, cin, h, w_, generator=g) * 1.5 + 0.3
        ho, wo = ((h - 1) * s - 2 * p + k, (w_ - 1) * s - 2 * p + k) if tr else ((h + 2 * p - k) // s + 1, (w_ + 2 * p - k) // s + 1)
        desc = ops.conv_desc(1 if tr else 0, k, s, p, h, w_, cin, ho, wo, cout, cin, cout)
        nd = ops.norm_desc(stats_of(x), None, None, h * w_, 1e-5, act, 0.2) if norm else None
        probs.append((desc, to_buf(x), nd, to_buf(torch.randn(1, cout, ho, wo, generator=g)), h, w_))
    res = {}
    for mode in ("apart", "fused"):
        dw, db = torch.zeros_like(wm), torch.zeros(pad_vec(torch.zeros(cout)).numel(), device="cuda")
        djobs, wjobs, keep = [], [], []
        for desc, xb, nd, dy, h, w_ in probs:
            din = torch.full((h, w_, cin), float("nan"), device="cuda")
            sums = torch.zeros(2 * cin, dtype=torch.float64, device="cuda") if norm else None
            if dmath == "f16":
                dy._sgan_amax = dy.abs().max().reshape(1).float()
            djobs.append((desc, dy, wm._sgan_wt, din, xb, nd, sums, 0, False, True, 0))
            wjobs.append((desc, xb, nd, dy.view_as(dy), dw, db))
            keep.append((din, sums))
        if mode == "apart":
            ops.conv_wgrad_grouped(wjobs)
            ops.conv_dgrad_grouped(djobs)
        else:
            fused = ops.conv_bwd_grouped(djobs, wjobs, None)
            if fused:
                assert _lib.lib().sgan_last_kernel().decode().startswith("sg_bwd_fused_kernel")
        torch.cuda.synchronize()
        res[mode] = (keep, dw, db)
    print("fused launch:", fused)
    _FUSED_SEEN[(shape, dmath)] = fused
    for (da, sa), (df, sf) in zip(res["apart"][0], res["fused"][0]):
        # the same arithmetic; bit-equal unless the stand-alone launch is small enough for two wave groups per workgroup
        # (sg_igemm3p_kernel<..., KW = 2>: the two halves of the channel blocks are added at the end, another rounding order)
        assert torch.equal(da, df) or rel(df, da) < 4e-6
        if sa is not None:
            assert rel(sf, sa) < (1e-12 if torch.equal(da, df) else 4e-6)
    assert rel(res["fused"][1], res["apart"][1]) < 2e-6 and rel(res["fused"][2], res["apart"][2]) < 2e-6


def test_fused_backward_is_taken(ops):
    """The comparison above is vacuous where the fused entry point declines (backward-data on a tile shape the fused kernel does not
    carry, split-K): of the listed shapes, the stride-1 convs with >= 64 channels on both sides must have gone through
    sg_bwd_fused_kernel, and at least one more."""
    want = [c for c in SHAPES if c[0] == "conv" and c[2] == 1 and c[4] >= 64 and c[5] >= 64 and c[6] * c[7] >= 256]
    assert sum(bool(v) for (c, m), v in _FUSED_SEEN.items() if m is None) > len(want)
    assert want and all(_FUSED_SEEN.get((c, None)) for c in want), {c: _FUSED_SEEN.get((c, None)) for c in want}
    # backward-data with <= 32 result channels (the 128 x 32 tile) on fp16 planes: the second PatchGAN layer's launch
    assert _FUSED_SEEN.get((SHAPES[0], "f16")) and _FUSED_SEEN.get((SHAPES[0], None)), _FUSED_SEEN


@pytest.mark.parametrize("shape", [("conv", 4, 1, 2, 128, 256, 33, 29), ("convT", 4, 2, 1, 256, 256, 16, 16), ("conv", 3, 1, 1, 160, 64, 24, 24),
                                   ("conv", 4, 1, 2, 224, 128, 20, 20), ("conv", 4, 2, 2, 64, 128, 65, 63), ("conv", 4, 2, 1, 96, 64, 34, 40),
                                   ("conv", 3, 2, 1, 64, 64, 41, 37)],
                         ids=["4blocks", "8blocks_convT", "5blocks", "7blocks", "stride2_k4p2", "stride2_k4p1_3blocks", "stride2_k3"])
def test_patch_kernel_small_launches_repeat(ops, shape):
    """Forward launches of the patch kernel with fewer workgroups than CUs and 4 / 8 / 5 / 7 channel blocks: against an fp64 result,
    and twenty repeats bit for bit (a synchronisation slip inside the kernel would show as a run that differs)."""
    from hip_utils import master_weight, pad_vec, stats_of, to_buf
    from supervised_gan_amd import _lib
    kind, k, s, p, cin, cout, H, W = shape
    tr = kind == "convT"
    s2 = not tr and s == 2       # stride-2 gather: the parity-plane patch (forced: these grids are under its automatic threshold)
    _select_tile("patch" if s2 else "auto")
    ops.set_math("bf16x3")
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, cin, H, W, generator=g)
    w = torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), generator=g) * 0.05
    b = torch.randn(cout, generator=g) * 0.1
    a = F.relu(F.instance_norm(x.double(), eps=1e-5))
    ref = F.conv_transpose2d(a, w.double(), b.double(), stride=s, padding=p) if tr else F.conv2d(a, w.double(), b.double(), stride=s, padding=p)
    Ho, Wo = ref.shape[2:]
    desc = ops.conv_desc(1 if tr else 0, k, s, p, H, W, cin, Ho, Wo, cout)
    nd = ops.norm_desc(stats_of(x), None, None, H * W, 1e-5, 1, 0.0)
    xb, wm, bb = to_buf(x), master_weight(w, tr), pad_vec(b)
    first = None
    for _ in range(20):
        ob = torch.full((Ho, Wo, cout), float("nan"), device="cuda")
        ost = torch.zeros(2 * cout, dtype=torch.float64, device="cuda")
        ops.conv_fwd(desc, xb, nd, wm, bb, ob, 0, ost)
        assert _lib.lib().sgan_last_kernel().decode() == ("sg_igemm3p_kernel<64,s2>" if s2 else "sg_igemm3p_kernel<64>")
        if first is None:
            first = ob.clone()
        else:
            assert torch.equal(ob, first)
    got = first.permute(2, 0, 1).unsqueeze(0).double().cpu()
    assert float((got - ref).abs().max() / ref.abs().max()) < 3e-6
    assert float((ost.cpu() - torch.cat([ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))])).abs().max() / (ref * ref).sum((0, 2, 3)).max()) < 1e-5
    _select_tile("auto")


def test_bf16x3_needs_packed_weights(ops):
    """No silent change of arithmetic: a layer the split kernels cover, asked for in bf16x3 without the packed copy, raises."""
    from supervised_gan_amd._lib import SganError
    ops.set_math("bf16x3")
    x = torch.zeros(40, 40, 32, device="cuda")
    w = torch.zeros(16 * 32 * 32, device="cuda")        # not tagged with a packed copy
    y = torch.zeros(21, 21, 32, device="cuda")
    with pytest.raises(SganError, match="w_packed"):
        ops.conv_fwd(ops.conv_desc(0, 4, 2, 2, 40, 40, 32, 21, 21, 32), x, None, w, None, y)
    ops.set_math("f32")
    ops.conv_fwd(ops.conv_desc(0, 4, 2, 2, 40, 40, 32, 21, 21, 32), x, None, w, None, y)
    ops.set_math("bf16x3")
    # maps under SGAN_BF16X3_MIN_PIXELS stay on the exact-fp32 kernels in either mode: no packed copy needed
    ops.conv_fwd(ops.conv_desc(0, 4, 2, 2, 16, 16, 32, 9, 9, 32), torch.zeros(16, 16, 32, device="cuda"), None, w, None, torch.zeros(9, 9, 32, device="cuda"))

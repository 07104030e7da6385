"""Generators of the reference's `define_G` on the MI355X path (models/networks.py:221-794,1015-1072): fcgan / deconv, fcgan_star,
dcgan, autoencoder, resnet_6blocks / resnet_9blocks, unet_128 / unet_256, crn -- each a layer program over `chain.ChainNet`."""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, CONV, CONVT, SganError
from .ops import pad4
from .chain import BN_EPS, BN_MOMENTUM, IN_EPS, ChainNet, LayerSpec, _BwdArena, _ChainFn, _ParamBox      # noqa: F401


class FCGANGenerator(ChainNet):
    """FCGANGenerator (models/networks.py:493-540): ConvT(k4,s2,p1) -> BatchNorm -> ReLU x n_layers,
    ConvT -> Tanh.  `norm_layer` is hard-wired to BatchNorm by define_G (models/networks.py:87) and
    the net never leaves train mode."""
    final_act = ACT_TANH

    def __init__(self, noise_nc, input_nc, ngf=64, n_layers=3, use_dropout=False, use_fcn=False, gpu_ids=[]):
        layers = []
        nf = min(2 ** (n_layers - 1), 8)
        # --noiseSize 1 (use_fcn False): the first ConvT is k4 s1 p0 and turns the 1x1 latent into a 4x4 map (:503-504)
        layers.append(LayerSpec("0", CONVT, 4, 2 if use_fcn else 1, 1 if use_fcn else 0, noise_nc, ngf * nf, False, "bn", ACT_RELU))
        idx = 3
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(2 ** (n_layers - n - 1), 8)
            # with use_dropout every block above the first is ConvT -> BatchNorm -> Dropout(0.5) -> ReLU (:513-521): four modules
            layers.append(LayerSpec(str(idx), CONVT, 4, 2, 1, ngf * nf_prev, ngf * nf, True, "bn", ACT_RELU, drop=0.5 if use_dropout else 0.0))
            idx += 4 if use_dropout else 3
        layers.append(LayerSpec(str(idx), CONVT, 4, 2, 1, ngf, input_nc, False, None, ACT_NONE))
        super().__init__(layers)
        self.gpu_ids = gpu_ids

    def _prepare_input(self, x, memo=None):
        return {"chain_in": ops.as_nhwc(x)}

    def _finish_input_grad(self, xb, dchain):
        return ops.logical_view(dchain, self.layers[0].cin)

    def forward(self, x, activation=None):
        params = list(self.model.parameters())
        return self._apply_with_activation(activation, lambda: _ChainFn.apply(self, x, *params))

    def _wrap_output(self, y):
        return y


class FCGANGeneratorStar(ChainNet):
    """FCGANGeneratorStar (models/networks.py:543-640): two bias-free ConvT(k4,s2,p1) -> BatchNorm -> ReLU chains of six layers.
    Chain a runs on the second half of the latent; every layer of chain b above the first reads cat([ha, hb]) of the level below;
    the image is tanh(cat([ha, hb])).

    Layout: both chains of a level write their raw outputs side by side into ONE [H, W, 2C] buffer (chain a the first C channels)
    with one [sum(2C) | sumsq(2C)] statistics array, and the level's BN affine parameters sit side by side in the flat storage --
    so the concatenation is never materialised: chain a's next layer reads the first half through a leading dimension, chain b's
    the whole buffer.  Backward-data of the b layer writes all 2C channels, the a layer's accumulates into the first C."""

    def __init__(self, noise_nc, input_nc, ngf=64, n_layers=3, use_dropout=False, use_fcn=False, gpu_ids=[]):
        assert n_layers == 5 and use_fcn is True and input_nc == 2        # models/networks.py:550-552
        half = int(noise_nc / 2)
        ch = [ngf * 8, ngf * 8, ngf * 4, ngf * 2, ngf]
        if ngf % 4:
            raise NotImplementedError("FCGANGeneratorStar on the MI355X path needs ngf % 4 == 0 (channel slices are read in 16-byte chunks)")
        self.ch, self.la, self.lb = ch, [], []
        for i in range(6):
            cout = ch[i] if i < 5 else 1
            nrm, act = ("bn", ACT_RELU) if i < 5 else (None, ACT_NONE)
            self.la.append(LayerSpec(f"conv{i}a.0", CONVT, 4, 2, 1, half if i == 0 else ch[i - 1], cout, False, nrm, act))
            self.lb.append(LayerSpec(f"conv{i}b.0", CONVT, 4, 2, 1, half if i == 0 else 2 * ch[i - 1], cout, False, nrm, act))
        super().__init__(self.la + self.lb)        # the reference's module order: chain a, then chain b
        del self.model                              # layers are direct attributes there: no `model.` prefix in state_dict keys
        self.noise_nc = half
        self.gpu_ids = gpu_ids

    def _param_root(self):
        return self

    def _assign_offsets(self, layers):
        off = 0
        for A, B in zip(self.la, self.lb):
            for L in (A, B):
                L.w_off = off
                off += L.k * L.k * L.cout_s * L.cin_s
            if A.norm == "bn":          # [gamma_a | gamma_b][beta_a | beta_b]: the affine of the concatenated tensor, contiguous
                A.g_off, B.g_off = off, off + A.cout_s
                off += 2 * A.cout_s
                A.be_off, B.be_off = off, off + A.cout_s
                off += 2 * A.cout_s
        return off

    def _desc(self, L, h, w):
        key = ("star", L.key, h, w)
        if key not in self._geom_cache:
            ho, wo = L.out_hw(h, w)
            self._geom_cache[key] = ops.conv_desc(L.kind, L.k, L.stride, L.pad, h, w, L.cin_s, ho, wo, L.cout_s, L.cin, L.cout)
        return self._geom_cache[key]

    def _level_norms(self, i, stats, count):
        """How the next layers read level i: (chain a's half, the whole concatenation)."""
        A, C = self.la[i], self.ch[i]
        f = self._flat
        na = ops.norm_desc(stats[i], f[A.g_off: A.g_off + C], f[A.be_off: A.be_off + C], count, BN_EPS, ACT_RELU, 0.0, sq_stride=2 * C)
        nb = ops.norm_desc(stats[i], f[A.g_off: A.g_off + 2 * C], f[A.be_off: A.be_off + 2 * C], count, BN_EPS, ACT_RELU, 0.0)
        return na, nb

    def run_forward(self, x, update_running=True):
        xa, xb = x["a"], x["b"]
        ops.require_gpu(xa, type(self).__name__)
        if self._flat.device != xa.device:
            raise SganError(f"module parameters are on {self._flat.device}, input on {xa.device}")
        dev, ch = xa.device, self.ch
        n_stats = sum(4 * c for c in ch)
        arena = torch.zeros(2 * n_stats, dtype=torch.float64, device=dev)      # forward statistics | backward sums
        stats, o = [], 0
        for c in ch:
            stats.append(arena[o: o + 4 * c])
            o += 4 * c
        h, w = xa.shape[0], xa.shape[1]
        cats, rl = [], []
        src_a, src_b, na, nb = xa, xb, None, None
        for i in range(5):
            A, B, C = self.la[i], self.lb[i], ch[i]
            ho, wo = A.out_hw(h, w)
            cat = torch.empty((ho, wo, 2 * C), dtype=torch.float32, device=dev)
            ops.conv_fwd(self._desc(B, h, w), src_b, nb, self._wb(B)[0], None, cat[..., C:], ACT_NONE, stats[i][C:], 2 * C)
            ops.conv_fwd(self._desc(A, h, w), src_a, na, self._wb(A)[0], None, cat[..., :C], ACT_NONE, stats[i], 2 * C)
            for L, st in ((A, stats[i]), (B, stats[i][C:])):
                nbx = self._bn_boxes[L.key]
                rl.append((st, nbx.running_mean, nbx.running_var, nbx.num_batches_tracked, C, ho * wo, 2 * C))
            cats.append(cat)
            h, w = ho, wo
            na, nb = self._level_norms(i, stats, h * w)
            src_a, src_b = cat[..., :C], cat
        A, B = self.la[5], self.lb[5]
        ho, wo = A.out_hw(h, w)
        out_a = torch.empty((ho, wo, A.cout_s), dtype=torch.float32, device=dev)
        out_b = torch.empty((ho, wo, B.cout_s), dtype=torch.float32, device=dev)
        ops.conv_fwd(self._desc(B, h, w), src_b, nb, self._wb(B)[0], None, out_b, ACT_NONE, None)
        ops.conv_fwd(self._desc(A, h, w), src_a, na, self._wb(A)[0], None, out_a, ACT_NONE, None)
        if update_running:
            ops.bn_running_update(rl, BN_MOMENTUM)
        return (out_a, out_b), {"x": x, "cats": cats, "stats": stats, "bwd": _BwdArena(arena[n_stats:])}

    def run_backward(self, x, outs, saved, douts, need_dx, want_wgrad):
        """douts: gradients of the two raw last-layer outputs ([H, W, 4] each).  Returns (dxa, dxb) or (None, None)."""
        xa, xb = saved["x"]["a"], saved["x"]["b"]
        cats, stats, ch = saved["cats"], saved["stats"], self.ch
        dev = xa.device
        if want_wgrad:
            self._ensure_grads()
        n_stats = sum(4 * c for c in ch)
        arena = saved["bwd"].take(n_stats)
        sums, o = [], 0
        for c in ch:
            sums.append(arena[o: o + 4 * c])
            o += 4 * c
        d_a, d_b = douts
        for i in range(5, 0, -1):               # layer i of both chains reads level i - 1
            A, B, C, cat = self.la[i], self.lb[i], ch[i - 1], cats[i - 1]
            h, w = cat.shape[0], cat.shape[1]
            na, nb = self._level_norms(i - 1, stats, h * w)
            da, db = self._desc(A, h, w), self._desc(B, h, w)
            if want_wgrad:
                ops.conv_wgrad(db, cat, nb, d_b, self._gwb(B)[0], None)
                ops.conv_wgrad(da, cat[..., :C], na, d_a, self._gwb(A)[0], None)
            dcat = torch.empty_like(cat)
            ops.conv_dgrad(db, d_b, self._wt(B), dcat, cat, nb, sums[i - 1], w_transposed=True)
            ops.conv_dgrad(da, d_a, self._wt(A), dcat[..., :C], cat[..., :C], na, sums[i - 1], sums_sq=2 * C, accumulate=True,
                           w_transposed=True)
            P = self.la[i - 1]
            dg = self._gflat[P.g_off: P.g_off + 2 * C] if want_wgrad else None
            dbe = self._gflat[P.be_off: P.be_off + 2 * C] if want_wgrad else None
            ops.norm_bwd_apply(dcat, cat, nb, sums[i - 1], dg, dbe)
            d_a, d_b = dcat[..., :C], dcat[..., C:]
        A, B = self.la[0], self.lb[0]
        h, w = xa.shape[0], xa.shape[1]
        if want_wgrad:
            ops.conv_wgrad(self._desc(B, h, w), xb, None, d_b, self._gwb(B)[0], None)
            ops.conv_wgrad(self._desc(A, h, w), xa, None, d_a, self._gwb(A)[0], None)
        if not need_dx:
            return None, None
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
        ops.conv_dgrad(self._desc(A, h, w), d_a, self._wt(A), dxa, None, None, None, w_transposed=True)
        ops.conv_dgrad(self._desc(B, h, w), d_b, self._wt(B), dxb, None, None, None, w_transposed=True)
        return dxa, dxb

    def forward(self, noise, activation=None):
        ha, hb = _StarFn.apply(self, noise, *list(self.parameters()))
        y = torch.cat([ha, hb], 1)
        return torch.tanh(y) if activation is None else activation(y)

    def _wrap_output(self, y):
        return y


class _StarFn(torch.autograd.Function):
    """One autograd node for both chains of FCGANGeneratorStar; returns the two raw single-channel images."""

    @staticmethod
    def forward(ctx, net, noise, *params):
        half = net.noise_nc
        x = {"b": ops.as_nhwc(noise.narrow(1, 0, half)), "a": ops.as_nhwc(noise.narrow(1, half, half))}   # :626-629
        (out_a, out_b), saved = net.run_forward(x)
        ctx.net, ctx.saved = net, saved
        ctx.need_dx = ctx.needs_input_grad[1]
        ctx.want_wgrad = net.compute_param_grads and any(ctx.needs_input_grad[2:])
        return ops.logical_view(out_a, 1), ops.logical_view(out_b, 1)

    @staticmethod
    def backward(ctx, ga, gb):
        net = ctx.net
        dxa, dxb = net.run_backward(None, None, ctx.saved, (ops.as_nhwc(ga.contiguous()), ops.as_nhwc(gb.contiguous())), ctx.need_dx,
                                    ctx.want_wgrad)
        dz = None
        if ctx.need_dx:
            half = net.noise_nc
            dz = torch.cat([ops.logical_view(dxb, half), ops.logical_view(dxa, half)], 1)
        return (None, dz) + (None,) * (len(ctx.needs_input_grad) - 2)


class DCGANGenerator(ChainNet):
    """DCGANGenerator (models/networks.py:1015-1071): ConvT(nz -> 8 ngf, k4, s1, p0) on a 1x1 latent, four ConvT(k4,s2,p1)
    halving the channels down to ngf/2, each followed by BatchNorm + ReLU, then ConvT(ngf/2 -> nc) -> Tanh (128x128 output);
    no biases.  The Tanh is part of `model` in the reference; here it is the last conv's epilogue."""
    final_act = ACT_TANH

    def __init__(self, gpu_ids=[], nz=100, nc=3, ngf=64):
        chans = [ngf * 8, ngf * 4, ngf * 2, ngf, int(ngf / 2)]
        layers = [LayerSpec("0", CONVT, 4, 1, 0, nz, chans[0], False, "bn", ACT_RELU)]
        for i in range(1, 5):
            layers.append(LayerSpec(str(3 * i), CONVT, 4, 2, 1, chans[i - 1], chans[i], False, "bn", ACT_RELU))
        layers.append(LayerSpec("15", CONVT, 4, 2, 1, chans[4], nc, False, None, ACT_NONE))
        super().__init__(layers)
        self.gpu_ids = gpu_ids

    def _prepare_input(self, x, memo=None):
        return {"chain_in": ops.as_nhwc(x)}

    def _finish_input_grad(self, xb, dchain):
        return ops.logical_view(dchain, self.layers[0].cin)

    def forward(self, input):
        return _ChainFn.apply(self, input, *list(self.model.parameters()))

    def _wrap_output(self, y):
        return y


class AutoEncoder(ChainNet):
    """AutoEncoder (models/networks.py:421-490): Conv(k4,s2,p1)+norm+ReLU x n_layers, a bias-free latent Conv with nothing after
    it, then ConvT(k4,s2,p1)+norm+ReLU x n_layers and a bias-free ConvT -> Tanh; with `use_dropout` every block but the first of
    each half has nn.Dropout(0.2) (encoder) / nn.Dropout(0.5) (decoder) between its norm and its ReLU (ChainNet's `drop`)."""
    final_act = ACT_TANH

    def __init__(self, input_nc, output_nc, n_layers=3, ngf=64, norm="batch", use_dropout=False, gpu_ids=[]):
        nrm = {"instance": "in", "batch": "bn"}[norm]
        step = 4 if use_dropout else 3      # modules per block in the reference's nn.Sequential
        layers, idx = [], 0
        nf = 1
        layers.append(LayerSpec(str(idx), CONV, 4, 2, 1, input_nc, ngf, True, nrm, ACT_RELU))
        idx += 3
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(2 ** n, 8)
            layers.append(LayerSpec(str(idx), CONV, 4, 2, 1, nf_prev * ngf, ngf * nf, True, nrm, ACT_RELU, drop=0.2 if use_dropout else 0.0))
            idx += step
        latent_nc = min(2 ** n_layers, 8)
        layers.append(LayerSpec(str(idx), CONV, 4, 2, 1, nf * ngf, latent_nc, False, None, ACT_NONE))
        idx += 1
        nf = min(2 ** (n_layers - 1), 8)
        layers.append(LayerSpec(str(idx), CONVT, 4, 2, 1, latent_nc, ngf * nf, False, nrm, ACT_RELU))
        idx += 3
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(2 ** (n_layers - n - 1), 8)
            layers.append(LayerSpec(str(idx), CONVT, 4, 2, 1, ngf * nf_prev, ngf * nf, True, nrm, ACT_RELU, drop=0.5 if use_dropout else 0.0))
            idx += step
        layers.append(LayerSpec(str(idx), CONVT, 4, 2, 1, ngf, output_nc, False, None, ACT_NONE))
        super().__init__(layers)
        self.gpu_ids = gpu_ids
        self.input_nc = input_nc

    def _prepare_input(self, x, memo=None):
        return {"chain_in": ops.as_nhwc(x)}

    def _finish_input_grad(self, xb, dchain):
        return ops.logical_view(dchain, self.input_nc)

    def forward(self, x, noise=None, activation=None):
        return self._apply_with_activation(activation, lambda: _ChainFn.apply(self, x, *list(self.model.parameters())))

    def _wrap_output(self, y):
        return y


class _ResidualActFn(torch.autograd.Function):
    """act(x + y) of two logical images in one pass (`nn.Tanh()(x + y)`, models/networks.py:268, :367); one gradient serves both."""

    @staticmethod
    def forward(ctx, x, y, act):
        a, b = ops.as_nhwc(x).contiguous(), ops.as_nhwc(y).contiguous()
        out = torch.empty_like(a)
        ops.add_act_fwd(a, b, out, act)
        ctx.act, ctx.out, ctx.nc = act, out, x.shape[1]
        return ops.logical_view(out, x.shape[1])

    @staticmethod
    def backward(ctx, g):
        if ctx.act == ACT_TANH:
            d = torch.empty_like(ctx.out)
            ops.tanh_bwd(ops.as_nhwc(g).contiguous(), ctx.out, d)
            g = ops.logical_view(d, ctx.nc)
        return g, g, None


def _residual_forward(net, x, activation, run):
    """`activation(x + net.model(x))` for a generator built with use_residual: the chain ends raw, the sum and a Tanh are one kernel."""
    if x.shape[1] != net.output_nc:
        raise ValueError(f"use_residual adds the input to the output: input_nc {x.shape[1]} != output_nc {net.output_nc}")
    net._call_act = ACT_NONE
    try:
        y = run()
    finally:
        net._call_act = None
    custom = activation is not None and not isinstance(activation, nn.Tanh)
    s = _ResidualActFn.apply(x, y, ACT_NONE if custom else ACT_TANH)
    return activation(s) if custom else s


class ResnetGenerator(ChainNet):
    """ResnetGenerator + ResnetBlock (models/networks.py:221-311; `resnet_6blocks` / `resnet_9blocks`, padding_type 'reflect'):
        ReflectionPad(3) Conv(k7) IN ReLU -> 2 x [Conv(k3,s2,p1) IN ReLU] -> n x ResnetBlock -> 2 x [ConvT(k3,s2,p1,op1) IN ReLU]
        -> ReflectionPad(3) Conv(k7) -> Tanh,     ResnetBlock: x + [ReflectionPad(1) Conv(k3) IN ReLU (Dropout) ReflectionPad(1) Conv(k3) IN](x)

    On the MI355X path the reflection-padded tensors are materialised by one gather pass each (`sgan_pad_reflect_fwd`, which also
    applies the InstanceNorm + ReLU (+ dropout mask) the reference runs before the padding); the convs behind them run with pad 0
    and no prologue.  The stride-2 convs and the second ConvT normalise on load like every other net.  A block's output
    x + IN(conv) is one `norm_apply_fwd` pass (the residual rides in its additive input).  49-tap k7 layers: SGAN_MAX_TAPS."""
    final_act = ACT_TANH

    def __init__(self, input_nc, output_nc, ngf=64, norm="instance", use_dropout=False, n_blocks=6, use_residual=False, gpu_ids=[]):
        nrm = {"instance": "in", "batch": "bn"}[norm]      # get_norm_layer (networks.py:43-50)
        self.use_residual = bool(use_residual)      # the Sequential then ends without its nn.Tanh (:258-259); forward() adds the input
        self.n_blocks, self.use_dropout, self.input_nc, self.output_nc = int(n_blocks), bool(use_dropout), input_nc, output_nc
        C = 4 * ngf
        self.c0 = LayerSpec("1", CONV, 7, 1, 0, input_nc, ngf, True, nrm, ACT_RELU)
        self.d1 = LayerSpec("4", CONV, 3, 2, 1, ngf, 2 * ngf, True, nrm, ACT_RELU)
        self.d2 = LayerSpec("7", CONV, 3, 2, 1, 2 * ngf, C, True, nrm, ACT_RELU)
        second = 6 if use_dropout else 5
        self.blocks = [(LayerSpec("%d.conv_block.1" % (10 + i), CONV, 3, 1, 0, C, C, True, nrm, ACT_RELU),
                        LayerSpec("%d.conv_block.%d" % (10 + i, second), CONV, 3, 1, 0, C, C, True, nrm, ACT_NONE)) for i in range(n_blocks)]
        nb = 10 + n_blocks
        self.u1 = LayerSpec(str(nb), CONVT, 3, 2, 1, C, 2 * ngf, True, nrm, ACT_RELU)
        self.u2 = LayerSpec(str(nb + 3), CONVT, 3, 2, 1, 2 * ngf, ngf, True, nrm, ACT_RELU)
        self.cl = LayerSpec(str(nb + 7), CONV, 7, 1, 0, ngf, output_nc, True, None, ACT_NONE)
        super().__init__([self.c0, self.d1, self.d2] + [l for ab in self.blocks for l in ab] + [self.u1, self.u2, self.cl])
        self.gpu_ids = gpu_ids
        self._rng_seed, self._rng_offset = 0, None

    # ---- module API ----------------------------------------------------------------------------
    def _prepare_input(self, x, memo=None):
        return {"chain_in": ops.as_nhwc(x)}

    def _finish_input_grad(self, xb, dchain):
        return ops.logical_view(dchain, self.input_nc)

    def forward(self, x, noise=None, activation=None):
        # the reference's forward() applies nn.Tanh() to the output of self.model, which (without --use_residual) already ends in
        # nn.Tanh() (models/networks.py:261-262,268): tanh(tanh(conv)).  The first is the conv epilogue, the second one elementwise op
        # on the output image.
        if self.use_residual:
            return _residual_forward(self, x, None, lambda: _ChainFn.apply(self, x, *list(self.model.parameters())))
        return torch.tanh(_ChainFn.apply(self, x, *list(self.model.parameters())))

    def _wrap_output(self, y):
        return y

    # ---- geometry ------------------------------------------------------------------------------
    def _desc(self, L, hin, win, hout, wout):
        key = (L.key, hin, win)
        if key not in self._geom_cache:
            self._geom_cache[key] = ops.conv_desc(L.kind, L.k, L.stride, L.pad, hin, win, L.cin_s, hout, wout, L.cout_s, L.cin, L.cout)
        return self._geom_cache[key]

    def _in(self, L, st, count, act):
        """How a consumer reads layer L's raw output: its norm (InstanceNorm, or BatchNorm with the layer's affine) + activation."""
        if L.norm == "bn":
            return ops.norm_desc(st, self._flat[L.g_off: L.g_off + L.cout_s], self._flat[L.be_off: L.be_off + L.cout_s], count, BN_EPS, act, 0.0)
        return ops.norm_desc(st, None, None, count, IN_EPS, act, 0.0)

    # ---- programs ------------------------------------------------------------------------------
    def run_forward(self, x, update_running=True):
        ops.require_gpu(x, type(self).__name__)
        if self._flat.device != x.device:
            raise SganError(f"module parameters are on {self._flat.device}, input on {x.device}")
        H, W, Cs = x.shape
        assert Cs == self.c0.cin_s and H % 4 == 0 and W % 4 == 0, (x.shape, "resnet generators need H, W divisible by 4")
        dev = x.device
        final_act = self._take_call_act()
        E = lambda h, w, c: torch.empty((h, w, c), dtype=torch.float32, device=dev)      # noqa: E731
        normed = [self.c0, self.d1, self.d2] + [l for ab in self.blocks for l in ab] + [self.u1, self.u2]
        n_stats = sum(2 * L.cout_s for L in normed)
        arena = torch.zeros(2 * n_stats, dtype=torch.float64, device=dev)      # forward statistics | backward sums
        st, o = {}, 0
        for L in normed:
            st[L.key] = arena[o: o + 2 * L.cout_s]
            o += 2 * L.cout_s
        h2, w2, h4, w4 = H // 2, W // 2, H // 4, W // 4
        ngf, C = self.c0.cout_s, self.d2.cout_s
        S = dict(final_act=final_act, x=x, st=st, bwd=_BwdArena(arena[n_stats:]), n_stats=n_stats)
        xp = E(H + 6, W + 6, Cs)
        ops.pad_reflect_fwd(x, None, 3, xp)
        c0 = E(H, W, ngf)
        ops.conv_fwd(self._desc(self.c0, H + 6, W + 6, H, W), xp, None, *self._wb(self.c0), c0, ACT_NONE, st[self.c0.key])
        d1 = E(h2, w2, self.d1.cout_s)
        ops.conv_fwd(self._desc(self.d1, H, W, h2, w2), c0, self._in(self.c0, st[self.c0.key], H * W, ACT_RELU), *self._wb(self.d1), d1, ACT_NONE, st[self.d1.key])
        d2 = E(h4, w4, C)
        ops.conv_fwd(self._desc(self.d2, h2, w2, h4, w4), d1, self._in(self.d1, st[self.d1.key], h2 * w2, ACT_RELU), *self._wb(self.d2), d2, ACT_NONE, st[self.d2.key])
        b = E(h4, w4, C)
        ops.pad_reflect_fwd(d2, self._in(self.d2, st[self.d2.key], h4 * w4, ACT_RELU), 0, b)
        if self.use_dropout and (self._rng_offset is None or self._rng_offset.device != dev):
            self._rng_offset = torch.zeros(1, dtype=torch.int64, device=dev)
        d3 = self._desc(self.blocks[0][0], h4 + 2, w4 + 2, h4, w4) if self.blocks else None
        blk = []
        for i, (A, B) in enumerate(self.blocks):
            p1 = E(h4 + 2, w4 + 2, C)
            ops.pad_reflect_fwd(b, None, 1, p1)
            a = E(h4, w4, C)
            ops.conv_fwd(d3, p1, None, *self._wb(A), a, ACT_NONE, st[A.key])
            mask = None
            if self.use_dropout:
                mask = E(h4, w4, C)
                src = getattr(self, "mask_source", None)        # tests inject the reference's masks
                if src is not None:
                    mask.copy_(src(i, (h4, w4, C)))
                else:
                    ops.dropout_mask(mask, 0.5, self._rng_seed + i, self._rng_offset, advance=False)
            p2 = E(h4 + 2, w4 + 2, C)
            ops.pad_reflect_fwd(a, self._in(A, st[A.key], h4 * w4, ACT_RELU), 1, p2, mask)
            c = E(h4, w4, C)
            ops.conv_fwd(d3, p2, None, *self._wb(B), c, ACT_NONE, st[B.key])
            bn = E(h4, w4, C)
            ops.norm_apply_fwd(c, self._in(B, st[B.key], h4 * w4, ACT_NONE), bn, None, b, 1.0)      # x + IN(conv)
            blk.append((p1, a, mask, p2, c))
            b = bn
        if self.use_dropout and getattr(self, "mask_source", None) is None and self.blocks:
            ops.rng_advance(self._rng_offset, (h4 * w4 * C + 3) // 4)
        u1 = E(h2, w2, self.u1.cout_s)
        ops.conv_fwd(self._desc(self.u1, h4, w4, h2, w2), b, None, *self._wb(self.u1), u1, ACT_NONE, st[self.u1.key])
        u2 = E(H, W, ngf)
        ops.conv_fwd(self._desc(self.u2, h2, w2, H, W), u1, self._in(self.u1, st[self.u1.key], h2 * w2, ACT_RELU), *self._wb(self.u2), u2, ACT_NONE, st[self.u2.key])
        pl = E(H + 6, W + 6, ngf)
        ops.pad_reflect_fwd(u2, self._in(self.u2, st[self.u2.key], H * W, ACT_RELU), 3, pl)
        y = E(H, W, self.cl.cout_s)
        ops.conv_fwd(self._desc(self.cl, H + 6, W + 6, H, W), pl, None, *self._wb(self.cl), y, final_act, None)
        if self._bn_boxes and update_running and self.training:      # --norm batch: running statistics of every BatchNorm
            cnt = {self.c0.key: H * W, self.d1.key: h2 * w2, self.d2.key: h4 * w4, self.u1.key: h2 * w2, self.u2.key: H * W}
            rl = []
            for L in normed:
                nb = self._bn_boxes[L.key]
                rl.append((st[L.key], nb.running_mean, nb.running_var, nb.num_batches_tracked, L.cout, cnt.get(L.key, h4 * w4), L.cout_s))
            for i0 in range(0, len(rl), 8):
                ops.bn_running_update(rl[i0: i0 + 8], BN_MOMENTUM)
        S.update(xp=xp, c0=c0, d1=d1, d2=d2, blk=blk, b_last=b, u1=u1, u2=u2, pl=pl, y=y)
        return [y], S

    def run_backward(self, x, outs, S, dout, need_dx, want_wgrad):
        H, W, Cs = x.shape
        dev = x.device
        h2, w2, h4, w4 = H // 2, W // 2, H // 4, W // 4
        ngf, C = self.c0.cout_s, self.d2.cout_s
        st = S["st"]
        E = lambda h, w, c: torch.empty((h, w, c), dtype=torch.float32, device=dev)      # noqa: E731
        if want_wgrad:
            self._ensure_grads()
        arena = S["bwd"].take(S["n_stats"])
        sums, o = {}, 0
        for key, t in st.items():
            sums[key] = arena[o: o + t.numel()]
            o += t.numel()
        if S["final_act"] == ACT_TANH:
            dy = torch.empty_like(S["y"])
            ops.tanh_bwd(dout.contiguous(), S["y"], dy)
        else:
            dy = dout.contiguous()

        def wgrad(L, desc, src, nrm, d):
            if want_wgrad:
                ops.conv_wgrad(desc, src, nrm, d, *self._gwb(L))

        def affine_grads(L):
            if L.norm != "bn" or not want_wgrad:
                return None, None
            return self._gflat[L.g_off: L.g_off + L.cout_s], self._gflat[L.be_off: L.be_off + L.cout_s]

        def norm_bwd(d, xraw, L, count, act):
            ops.norm_bwd_apply(d, xraw, self._in(L, st[L.key], count, act), sums[L.key], *affine_grads(L))

        # last conv (k7 over the padded, activated u2)
        dcl = self._desc(self.cl, H + 6, W + 6, H, W)
        wgrad(self.cl, dcl, S["pl"], None, dy)
        dpl = E(H + 6, W + 6, ngf)
        ops.conv_dgrad(dcl, dy, self._wt(self.cl), dpl, None, None, None, w_transposed=True)
        du2 = E(H, W, ngf)
        ops.pad_reflect_bwd(dpl, 3, du2, S["u2"], self._in(self.u2, st[self.u2.key], H * W, ACT_RELU), None, sums[self.u2.key])
        norm_bwd(du2, S["u2"], self.u2, H * W, ACT_RELU)
        # the two transposed convs
        n_u1 = self._in(self.u1, st[self.u1.key], h2 * w2, ACT_RELU)
        du = self._desc(self.u2, h2, w2, H, W)
        wgrad(self.u2, du, S["u1"], n_u1, du2)
        du1 = E(h2, w2, self.u1.cout_s)
        ops.conv_dgrad(du, du2, self._wt(self.u2), du1, S["u1"], n_u1, sums[self.u1.key], w_transposed=True)
        norm_bwd(du1, S["u1"], self.u1, h2 * w2, ACT_RELU)
        du = self._desc(self.u1, h4, w4, h2, w2)
        wgrad(self.u1, du, S["b_last"], None, du1)
        db = E(h4, w4, C)
        ops.conv_dgrad(du, du1, self._wt(self.u1), db, None, None, None, w_transposed=True)
        # residual blocks, last to first: b_out = b_in + IN(conv_b(pad(mask * relu(IN(conv_a(pad(b_in)))))))
        d3 = self._desc(self.blocks[0][0], h4 + 2, w4 + 2, h4, w4) if self.blocks else None
        for (A, B), (p1, a, mask, p2, c) in zip(reversed(self.blocks), reversed(S["blk"])):
            dc = db.clone()
            n_c = self._in(B, st[B.key], h4 * w4, ACT_NONE)
            ops.norm_apply_bwd_sums(dc, c, n_c, sums[B.key])
            ops.norm_bwd_apply(dc, c, n_c, sums[B.key], *affine_grads(B))
            wgrad(B, d3, p2, None, dc)
            dp2 = E(h4 + 2, w4 + 2, C)
            ops.conv_dgrad(d3, dc, self._wt(B), dp2, None, None, None, w_transposed=True)
            da = E(h4, w4, C)
            ops.pad_reflect_bwd(dp2, 1, da, a, self._in(A, st[A.key], h4 * w4, ACT_RELU), mask, sums[A.key])
            norm_bwd(da, a, A, h4 * w4, ACT_RELU)
            wgrad(A, d3, p1, None, da)
            dp1 = E(h4 + 2, w4 + 2, C)
            ops.conv_dgrad(d3, da, self._wt(A), dp1, None, None, None, w_transposed=True)
            dbi = E(h4, w4, C)
            ops.pad_reflect_bwd(dp1, 1, dbi)
            db.add_(dbi)
        # block input = relu(IN(d2)), materialised with pad 0
        dd2 = E(h4, w4, C)
        ops.pad_reflect_bwd(db, 0, dd2, S["d2"], self._in(self.d2, st[self.d2.key], h4 * w4, ACT_RELU), None, sums[self.d2.key])
        norm_bwd(dd2, S["d2"], self.d2, h4 * w4, ACT_RELU)
        n_d1 = self._in(self.d1, st[self.d1.key], h2 * w2, ACT_RELU)
        dd = self._desc(self.d2, h2, w2, h4, w4)
        wgrad(self.d2, dd, S["d1"], n_d1, dd2)
        dd1 = E(h2, w2, self.d1.cout_s)
        ops.conv_dgrad(dd, dd2, self._wt(self.d2), dd1, S["d1"], n_d1, sums[self.d1.key], w_transposed=True)
        norm_bwd(dd1, S["d1"], self.d1, h2 * w2, ACT_RELU)
        n_c0 = self._in(self.c0, st[self.c0.key], H * W, ACT_RELU)
        dd = self._desc(self.d1, H, W, h2, w2)
        wgrad(self.d1, dd, S["c0"], n_c0, dd1)
        dc0 = E(H, W, ngf)
        ops.conv_dgrad(dd, dd1, self._wt(self.d1), dc0, S["c0"], n_c0, sums[self.c0.key], w_transposed=True)
        norm_bwd(dc0, S["c0"], self.c0, H * W, ACT_RELU)
        d0 = self._desc(self.c0, H + 6, W + 6, H, W)
        wgrad(self.c0, d0, S["xp"], None, dc0)
        if not need_dx:
            return None
        dxp = E(H + 6, W + 6, Cs)
        ops.conv_dgrad(d0, dc0, self._wt(self.c0), dxp, None, None, None, w_transposed=True)
        dx = E(H, W, Cs)
        ops.pad_reflect_bwd(dxp, 3, dx)
        return dx


class UnetGenerator(ChainNet):
    """UnetGenerator + UnetSkipConnectionBlock (models/networks.py:318-419) as a layer program over a DAG.

    Level l = 0..n-1: `down[l]` Conv(k4,s2,p1) produces x_l (c_l channels at H/2^(l+1)); `up[l]` ConvT(k4,s2,p1) is the
    transposed conv of the block wrapping x_l.  Block l (1..n-1) computes
        y_l = Dropout?(IN(up[l](ReLU(sub)))) [+ sigma * noise],   returns cat([y_l, x_{l-1}]) if skip_l else y_l
    with sub = IN(down[l](LeakyReLU(x_{l-1}))) fed to block l+1 (innermost: no IN, no sub-block).

    MI355X layout: cat([y_l, x_{l-1}]) is never assembled -- `down[l-1]` writes its raw output straight into the
    right half of the concat buffer (pixel stride 2c) with its InstanceNorm statistics in a slice of the buffer's
    statistics, and one pass (`norm_apply_fwd`) writes y_l into the left half.  Consumers normalise on load with
    per-channel statistics (left half: identity entries), so the skip tensors exist once and IN/LeakyReLU/ReLU
    never run as passes.  Backward: the two consumers of x_{l-1} (ReLU via the concat, LeakyReLU via down[l])
    accumulate into one gradient buffer (dgrad `accumulate`), then one `norm_bwd_apply`."""
    final_act = ACT_TANH

    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm="instance", use_dropout=False, use_residual=False,
                 add_gaussian_noise=False, gaussian_sigma=0.1, num_skips=-1, gpu_ids=[]):
        nrm = {"instance": "in", "batch": "bn"}[norm]      # get_norm_layer (networks.py:43-50): InstanceNorm2d(affine=False) / BatchNorm2d(affine=True)
        self.bn = nrm == "bn"
        self.use_residual = bool(use_residual)
        if num_downs < 5:
            raise ValueError("UnetGenerator needs num_downs >= 5")
        n = num_downs
        if num_skips < 0:
            num_skips = n
        self.n = n
        self.c = [ngf * min(2 ** l, 8) for l in range(n)]
        self.skip = [False] + [num_skips >= n - l for l in range(1, n)]
        self.use_dropout = bool(use_dropout)
        self.drop = [bool(use_dropout and 4 <= l <= n - 2) for l in range(n)]
        self.add_gauss, self.gauss_sigma = bool(add_gaussian_noise), float(gaussian_sigma)
        self.input_nc, self.output_nc = input_nc, output_nc
        c, skip = self.c, self.skip
        self.down, self.up = [], []
        for l in range(n):
            inner = l == n - 1
            if l == 0:
                dk, uk = "0", "3"
                d = LayerSpec(dk, CONV, 4, 2, 1, input_nc, c[0], True, None, ACT_NONE)
                u = LayerSpec(uk, CONVT, 4, 2, 1, c[0] * (2 if skip[1] else 1), output_nc, True, None, ACT_NONE)
            else:
                prefix = "1" + ".model.3" * (l - 1)
                dk, uk = prefix + ".model.1", prefix + (".model.3" if inner else ".model.5")
                d = LayerSpec(dk, CONV, 4, 2, 1, c[l - 1], c[l], True, None if inner else nrm, ACT_NONE)
                u = LayerSpec(uk, CONVT, 4, 2, 1, c[l] if inner else c[l] * (2 if skip[l + 1] else 1), c[l - 1], True, nrm, ACT_NONE)
            self.down.append(d)
            self.up.append(u)
        # parameter order = the reference's nn.Sequential traversal: down[0], (down[1], (down[2] ... up[2]), up[1]), up[0]
        super().__init__(self.down + self.up[::-1])
        self.gpu_ids = gpu_ids
        self._rng_seed = 0
        self._rng_offset = None
        self.mask_override = None     # tests: {level: [h, w, c] keep-mask (0 / 2)}
        self.noise_override = None    # tests: {level: [h, w, c] N(0,1) tensor}

    # ---- geometry / buffers ---------------------------------------------------------------------
    def _unet_geometry(self, H, W):
        key = ("unet", H, W)
        if key not in self._geom_cache:
            n = self.n
            if H % (1 << n) or W % (1 << n):
                raise SganError(f"UnetGenerator with {n} downsamplings needs H, W divisible by {1 << n}, got {H}x{W}")
            hw = [(H >> (l + 1), W >> (l + 1)) for l in range(n)]
            dn, upd = [], []
            for l in range(n):
                hi, wi = (H, W) if l == 0 else hw[l - 1]
                ho, wo = hw[l]
                d, u = self.down[l], self.up[l]
                dn.append(ops.conv_desc(CONV, 4, 2, 1, hi, wi, d.cin_s, ho, wo, d.cout_s, d.cin, d.cout))
                upd.append(ops.conv_desc(CONVT, 4, 2, 1, ho, wo, u.cin_s, hi, wi, u.cout_s, u.cin, u.cout))
            self._geom_cache[key] = (hw, dn, upd)
        return self._geom_cache[key]

    def _stat_layout(self, hw):
        """Offsets inside one float64 arena: per concat buffer [2 * width], per up-conv output [2 * c]; and the
        template holding the identity entries (sum 0, sumsq count * (1 - eps) => mean 0, rstd 1)."""
        n, c, skip = self.n, self.c, self.skip
        off, lay = 0, {}
        for l in range(1, n):
            wdt = c[l - 1] * (2 if skip[l] else 1)
            lay[("cat", l)] = (off, wdt)
            off += 2 * wdt
            lay[("u", l)] = (off, c[l - 1])
            off += 2 * c[l - 1]
        for l in range(1, n - 1):
            if not skip[l + 1]:       # normalised x_l that is not part of a concat buffer
                lay[("x", l)] = (off, c[l])
                off += 2 * c[l]
        return lay, off

    def _stat_template(self, hw, dev):
        key = ("tmpl", hw[0], str(dev))
        if key not in self._geom_cache:
            lay, total = self._stat_layout(hw)
            t = torch.zeros(2 * total, dtype=torch.float64)      # forward statistics | backward sums (zeros)
            one_minus_eps = 1.0 - float(np.float32(IN_EPS))
            for l in range(1, self.n):
                o, wdt = lay[("cat", l)]
                cnt = hw[l - 1][0] * hw[l - 1][1]
                cy = self.c[l - 1]
                t[o + wdt: o + wdt + cy] = cnt * one_minus_eps              # y half: already normalised
                if self.skip[l] and l - 1 == 0:
                    t[o + wdt + cy: o + 2 * wdt] = cnt * one_minus_eps      # x_0 has no norm
            self._geom_cache[key] = (lay, total, t.to(dev))
        return self._geom_cache[key]

    def _wb(self, L):
        return super()._wb(L)

    def _affine(self, L, grad=False):
        """(gamma, beta) of layer L's BatchNorm in the flat parameter (or gradient) storage; (None, None) for InstanceNorm."""
        if L.norm != "bn":
            return None, None
        flat = self._gflat if grad else self._flat
        return flat[L.g_off: L.g_off + L.cout_s], flat[L.be_off: L.be_off + L.cout_s]

    def _x_norm(self, l, hw, xstat, act, slope=0.0):
        """How a consumer reads x_l from its raw conv output."""
        if l == 0 or l == self.n - 1:
            return ops.norm_desc(None, None, None, 1, 0.0, act, slope)
        st, sq = xstat[l]
        g, b = self._affine(self.down[l])
        return ops.norm_desc(st, g, b, hw[l][0] * hw[l][1], IN_EPS, act, slope, sq)

    def _cat_affine(self):
        """--norm batch: per concat buffer the affine its consumer applies on load -- (1, 0) for the up half (materialised with its
        own affine already), (gamma, beta) of down[l-1]'s BatchNorm for the skip half.  Two torch.cat launches per forward."""
        n, c, skip = self.n, self.c, self.skip
        dev = self._flat.device
        key = ("catconst", str(dev))
        if key not in self._geom_cache:
            self._geom_cache[key] = {l: (torch.ones(c[l - 1], device=dev), torch.zeros(c[l - 1], device=dev)) for l in range(2, n)}
        const = self._geom_cache[key]
        gs, bs, where, o = [], [], {}, 0
        for l in range(2, n):
            if not skip[l]:
                continue
            g, b = self._affine(self.down[l - 1])
            gs += [const[l][0], g]
            bs += [const[l][1], b]
            where[l] = (o, 2 * c[l - 1])
            o += 2 * c[l - 1]
        if not gs:
            return {}
        G, B = torch.cat(gs), torch.cat(bs)
        return {l: (G[o: o + w], B[o: o + w]) for l, (o, w) in where.items()}

    def _cat_norm(self, l, hw, catstat, cataff=None):
        """ReLU(cat_l) as read by up[l-1]: identity for y_l (and x_0), the norm of x_{l-1} (its statistics, and with --norm batch
        its affine: `cataff` from _cat_affine) for the skip half."""
        if l == 1 or not self.skip[l]:
            return ops.norm_desc(None, None, None, 1, 0.0, ACT_RELU, 0.0)
        g, b = cataff[l] if (self.bn and cataff) else (None, None)
        return ops.norm_desc(catstat[l], g, b, hw[l - 1][0] * hw[l - 1][1], IN_EPS, ACT_RELU, 0.0, 0)

    def _random(self, l, shape, dev):
        """Dropout mask / Gaussian noise of level l.  Every (level, kind) is its own Philox stream (seed), all read the same
        offset; run_forward moves the offset once per pass (one launch instead of one per tensor)."""
        mask = noise = None
        if self.drop[l]:
            if self.mask_override is not None:
                mask = self.mask_override[l]
            else:
                mask = torch.empty(shape, dtype=torch.float32, device=dev)
                ops.dropout_mask(mask, 0.5, self._rng_seed + 2 * l, self._rng_offset, advance=False)
                self._rng_drawn = max(self._rng_drawn, (mask.numel() + 3) // 4)
        if self.add_gauss:
            if self.noise_override is not None:
                noise = self.noise_override[l]
            else:
                noise = torch.empty(shape, dtype=torch.float32, device=dev)
                ops.normal_fill(noise, self._rng_seed + 2 * l + 1, self._rng_offset, advance=False)
                self._rng_drawn = max(self._rng_drawn, (noise.numel() + 3) // 4)
        return mask, noise

    # ---- programs -------------------------------------------------------------------------------
    def run_forward(self, x, update_running=True):
        with ops.math_scope(os.environ.get("SGAN_UNET_MATH")):      # diagnostics: force an arithmetic mode for the U-Nets only
            return self._run_forward(x, update_running)

    def run_backward(self, x, outs, S, dout, need_dx, want_wgrad):
        with ops.math_scope(os.environ.get("SGAN_UNET_MATH")):
            return self._run_backward(x, outs, S, dout, need_dx, want_wgrad)

    def _run_forward(self, x, update_running=True):
        ops.require_gpu(x, type(self).__name__)
        if self._flat.device != x.device:
            raise SganError(f"module parameters are on {self._flat.device}, input on {x.device}")
        H, W, Cs = x.shape
        assert Cs == self.down[0].cin_s, (Cs, self.down[0].cin_s)
        n, c, skip = self.n, self.c, self.skip
        dev = x.device
        hw, dn, upd = self._unet_geometry(H, W)
        if self._rng_offset is None or self._rng_offset.device != dev:
            self._rng_offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self._rng_drawn = 0     # longest stream drawn in this pass (in Philox blocks of 4 values)
        lay, total, tmpl = self._stat_template(hw, dev)
        arena = tmpl.clone()
        cataff = self._cat_affine() if self.bn else None
        catw = [0] * (n + 1)
        cat, catstat, ustat = [None] * (n + 1), [None] * (n + 1), [None] * (n + 1)
        for l in range(1, n):
            o, wdt = lay[("cat", l)]
            catw[l] = wdt
            cat[l] = torch.empty(hw[l - 1] + (wdt,), dtype=torch.float32, device=dev)
            catstat[l] = arena[o: o + 2 * wdt]
            o, cu = lay[("u", l)]
            ustat[l] = arena[o: o + 2 * cu]
        xr, xstat = [None] * n, [None] * n
        for l in range(n):
            if l + 1 <= n - 1 and skip[l + 1]:
                xr[l] = cat[l + 1][:, :, c[l]:]
                xstat[l] = (catstat[l + 1][c[l]:], catw[l + 1])
            else:
                xr[l] = torch.empty(hw[l] + (c[l],), dtype=torch.float32, device=dev)
                if ("x", l) in lay:
                    o, cx = lay[("x", l)]
                    xstat[l] = (arena[o: o + 2 * cx], 0)
        # encoder
        for l in range(n):
            L = self.down[l]
            wt, b = self._wb(L)
            src = x if l == 0 else xr[l - 1]
            in_norm = None if l == 0 else self._x_norm(l - 1, hw, xstat, ACT_LRELU, 0.2)
            if 1 <= l <= n - 2:
                st, sq = xstat[l]
                ops.conv_fwd(dn[l], src, in_norm, wt, b, xr[l], ACT_NONE, st, sq)
            else:
                ops.conv_fwd(dn[l], src, in_norm, wt, b, xr[l], ACT_NONE, None)
        # decoder
        u, masks = [None] * n, [None] * n
        for l in range(n - 1, 0, -1):
            L = self.up[l]
            wt, b = self._wb(L)
            if l == n - 1:
                src, in_norm = xr[l], ops.norm_desc(None, None, None, 1, 0.0, ACT_RELU, 0.0)
            else:
                src, in_norm = cat[l + 1], self._cat_norm(l + 1, hw, catstat, cataff)
            u[l] = torch.empty(hw[l - 1] + (c[l - 1],), dtype=torch.float32, device=dev)
            ops.conv_fwd(upd[l], src, in_norm, wt, b, u[l], ACT_NONE, ustat[l])
            mask, noise = self._random(l, u[l].shape, dev)
            masks[l] = mask
            ug, ub = self._affine(L)
            un = ops.norm_desc(ustat[l], ug, ub, hw[l - 1][0] * hw[l - 1][1], IN_EPS, ACT_NONE, 0.0)
            ops.norm_apply_fwd(u[l], un, cat[l][:, :, :c[l - 1]], mask, noise, self.gauss_sigma if noise is not None else 0.0)
        L = self.up[0]
        wt, b = self._wb(L)
        out = torch.empty((H, W, L.cout_s), dtype=torch.float32, device=dev)
        final_act = self._take_call_act()
        ops.conv_fwd(upd[0], cat[1], self._cat_norm(1, hw, catstat, cataff), wt, b, out, final_act, None)
        if self._rng_drawn:
            ops.rng_advance(self._rng_offset, self._rng_drawn)
        if self.bn and update_running and self.training:      # running statistics of every BatchNorm (train-mode side effect of the forward)
            rl = []
            for l in range(1, n - 1):
                nb = self._bn_boxes[self.down[l].key]
                st, sq = xstat[l]
                rl.append((st, nb.running_mean, nb.running_var, nb.num_batches_tracked, c[l], hw[l][0] * hw[l][1], sq if sq else c[l]))
            for l in range(1, n):
                nb = self._bn_boxes[self.up[l].key]
                rl.append((ustat[l], nb.running_mean, nb.running_var, nb.num_batches_tracked, c[l - 1], hw[l - 1][0] * hw[l - 1][1], c[l - 1]))
            for i0 in range(0, len(rl), 8):
                ops.bn_running_update(rl[i0: i0 + 8], BN_MOMENTUM)
        saved = dict(final_act=final_act, x=x, hw=hw, cat=cat, catw=catw, catstat=catstat, ustat=ustat, xr=xr, xstat=xstat, u=u, masks=masks,
                     out=out, lay=lay, total=total, bwd=_BwdArena(arena[total:]), cataff=cataff)
        return [out], saved

    def _run_backward(self, x, outs, S, dout, need_dx, want_wgrad):
        n, c, skip = self.n, self.c, self.skip
        dev = x.device
        hw, dn, upd = self._unet_geometry(x.shape[0], x.shape[1])
        cat, catw, catstat, ustat, xr, xstat, u, masks = (S[k] for k in ("cat", "catw", "catstat", "ustat", "xr", "xstat", "u", "masks"))
        if want_wgrad:
            self._ensure_grads()
        if S["final_act"] == ACT_TANH:
            d0 = torch.empty_like(S["out"])
            ops.tanh_bwd(dout.contiguous(), S["out"], d0)
        else:
            d0 = dout.contiguous()
        lay = S["lay"]
        arena = S["bwd"].take(S["total"])
        csum, usum, xsum = [None] * (n + 1), [None] * (n + 1), [None] * n
        for l in range(1, n):
            o, wdt = lay[("cat", l)]
            csum[l] = arena[o: o + 2 * wdt]
            o, cu = lay[("u", l)]
            usum[l] = arena[o: o + 2 * cu]
        for l in range(1, n - 1):
            if skip[l + 1]:
                xsum[l] = (csum[l + 1][c[l]:], catw[l + 1])
            else:
                o, cx = lay[("x", l)]
                xsum[l] = (arena[o: o + 2 * cx], 0)
        dcat = [None] * (n + 1)
        for l in range(1, n):
            dcat[l] = torch.empty_like(cat[l])

        def wgrad(L, desc, src, nrm, dy):
            if want_wgrad:
                gw, gb = self._gwb(L)
                ops.conv_wgrad(desc, src, nrm, dy, gw, gb)

        def bwd(L, desc, src, nrm, dy, din, sums, sq=0, accumulate=False):
            """Backward-weight and backward-data of one conv: one fused launch where sgan_conv_bwd_fused covers the layer."""
            djob = [(desc, dy, self._wt(L), din, src, nrm, sums, sq, accumulate, True, 0)]
            if want_wgrad:
                ops.conv_bwd_grouped(djob, [(desc, src, nrm, dy) + self._gwb(L)])
            else:
                ops.conv_dgrad_grouped(djob)

        cataff = S.get("cataff")
        # final transposed conv: gradient of ReLU(cat_1)
        nrm = self._cat_norm(1, hw, catstat, cataff)
        bwd(self.up[0], upd[0], cat[1], nrm, d0, dcat[1], None)
        # decoder, outermost block first
        d_inner = None
        for l in range(1, n):
            dy = dcat[l][:, :, :c[l - 1]]
            ug, ub = self._affine(self.up[l])
            un = ops.norm_desc(ustat[l], ug, ub, hw[l - 1][0] * hw[l - 1][1], IN_EPS, ACT_NONE, 0.0)
            ops.norm_apply_bwd_sums(dy, u[l], un, usum[l], masks[l])
            dug, dub = self._affine(self.up[l], grad=True) if want_wgrad else (None, None)
            ops.norm_bwd_apply(dy, u[l], un, usum[l], dug, dub)             # dy is now d(up[l] output)
            if l == n - 1:
                src, nrm = xr[l], ops.norm_desc(None, None, None, 1, 0.0, ACT_RELU, 0.0)
                d_inner = torch.empty(hw[l] + (c[l],), dtype=torch.float32, device=dev)
                din, sums = d_inner, None
            else:
                src, nrm = cat[l + 1], self._cat_norm(l + 1, hw, catstat, cataff)
                din = dcat[l + 1]
                sums = csum[l + 1] if (skip[l + 1] and l + 1 > 1) else None
            bwd(self.up[l], upd[l], src, nrm, dy, din, sums)
        # encoder, innermost first: dr = gradient w.r.t. the raw output of down[l]
        dr = d_inner
        for l in range(n - 1, 0, -1):
            src = xr[l - 1]
            nrm = self._x_norm(l - 1, hw, xstat, ACT_LRELU, 0.2)
            normed = 1 <= l - 1 <= n - 2
            sums, sq = xsum[l - 1] if normed else (None, 0)
            if skip[l]:
                din = dcat[l][:, :, c[l - 1]:]
                bwd(self.down[l], dn[l], src, nrm, dr, din, sums, sq, accumulate=True)
            else:
                din = torch.empty(hw[l - 1] + (c[l - 1],), dtype=torch.float32, device=dev)
                bwd(self.down[l], dn[l], src, nrm, dr, din, sums, sq)
            if normed:
                dg, db = self._affine(self.down[l - 1], grad=True) if want_wgrad else (None, None)
                ops.norm_bwd_apply(din, src, nrm, sums, dg, db, sq)
            dr = din
        wgrad(self.down[0], dn[0], x, None, dr)
        dx = None
        if need_dx:
            dx = torch.empty_like(x)
            ops.conv_dgrad(dn[0], dr, self._wt(self.down[0]), dx, None, None, None, w_transposed=True)
        return dx

    # ---- module protocol ---------------------------------------------------------------------------
    def _prepare_input(self, x, memo=None):
        return {"chain_in": ops.as_nhwc(x)}

    def _finish_input_grad(self, xb, dchain):
        return ops.logical_view(dchain, self.input_nc)

    def forward(self, x, noise=None, activation=None):
        """`noise` is accepted and ignored like in the reference (models/networks.py:362)."""
        params = list(self.model.parameters())
        if self.use_residual:
            return _residual_forward(self, x, activation, lambda: _ChainFn.apply(self, x, *params))
        return self._apply_with_activation(activation, lambda: _ChainFn.apply(self, x, *params))

    def _wrap_output(self, y):
        return y


class CascadedRefinementNetwork(ChainNet):
    """CascadedRefinementNetwork + CrnUpsampleBlock + CrnInterBlock (models/networks.py:642-794), n_layers = 5:
    six stages from H/64 to H.  Stage s reads cat([label branch l_s, h_{s+1}]) (stage 5: cat([AvgPool64(label), noise])),
    upsamples by 2 (ConvT k4 s2 p1 + IN, or Conv3x3 + bilinear + IN) and applies n_layers_block x (ReLU, Conv3x3, IN);
    the last stage ends in Conv3x3 -> Tanh.  l_s = IN(Conv3x3(AvgPool_{2^(s+1)}(label))) with one shared conv.

    MI355X layout: as in the U-Net, cat([l_s, h]) is never assembled -- the label conv and the previous stage's last
    conv write their raw outputs into the two halves of one buffer, their InstanceNorm statistics into the two
    halves of one statistics array, and the stage's first conv normalises on load.  The six label maps come from one
    pyramid kernel; the bilinear kernel accumulates the statistics of its own output."""
    final_act = ACT_TANH

    def __init__(self, input_nc, output_nc, noise_nc, ngf=64, n_layers=5, norm="instance", upsample_mode='convt',
                 add_gaussian_noise=False, gaussian_sigma=0.1, share_label_weights=True, n_layers_block=1, gpu_ids=[]):
        assert n_layers == 5
        nrm = {"instance": "in", "batch": "bn"}[norm]      # get_norm_layer (networks.py:43-50)
        self.bn = nrm == "bn"
        if upsample_mode not in ('convt', 'bilinear'):
            raise NotImplementedError('UpsampleBlock mode [%s] is not recognized' % upsample_mode)
        if input_nc > 4:
            raise NotImplementedError("label images with more than 4 channels are not on the MI355X path")
        self.input_nc, self.output_nc, self.noise_nc, self.ngf = input_nc, output_nc, noise_nc, ngf
        self.mode, self.nlb, self.share = upsample_mode, n_layers_block, share_label_weights
        # --add_gaussian_noise: sigma * N(0, 1) on the normalised output of every upsample block but the last (networks.py:655-680,757-760)
        self.add_gauss, self.gauss_sigma = bool(add_gaussian_noise), float(gaussian_sigma)
        self.noise_override = None      # tests: {stage: [2h, 2w, ngf] NHWC tensor}
        self._rng_seed, self._rng_offset = 0, None
        self.up, self.inter, self.lab = {}, {}, {}
        layers = []
        for s in range(5, -1, -1):
            cin = noise_nc + input_nc if s == 5 else 2 * ngf
            if upsample_mode == 'convt':
                u = LayerSpec(f"blockh{s}.0.model.0", CONVT, 4, 2, 1, cin, ngf, False, nrm, ACT_NONE)
            else:      # [Conv2d, Upsample, norm]: the norm is child 2 (networks.py:749-753)
                u = LayerSpec(f"blockh{s}.0.model.0", CONV, 3, 1, 1, cin, ngf, True, nrm, ACT_NONE, norm_key=f"blockh{s}.0.model.2")
            self.up[s] = u
            layers.append(u)
            self.inter[s] = []
            for i in range(n_layers_block):
                last = s == 0 and i == n_layers_block - 1
                L = LayerSpec(f"blockh{s}.1.model.{3 * i + 1}", CONV, 3, 1, 1, ngf, output_nc if last else ngf, True,
                              None if last else nrm, ACT_NONE)
                self.inter[s].append(L)
                layers.append(L)
        if share_label_weights:
            L = LayerSpec("blockl.0", CONV, 3, 1, 1, input_nc, ngf, True, nrm, ACT_NONE)
            layers.append(L)
            for s in range(5):
                self.lab[s] = L
        else:
            for s in range(4, -1, -1):
                self.lab[s] = LayerSpec(f"blockl{s}.0", CONV, 3, 1, 1, input_nc, ngf, True, nrm, ACT_NONE)
                layers.append(self.lab[s])
        super().__init__(layers)
        del self.model          # the reference keeps its blocks as direct attributes: no `model.` prefix in state_dict keys
        self.gpu_ids = gpu_ids

    def _param_root(self):
        return self

    # ---- programs -------------------------------------------------------------------------------
    def _desc(self, L, h, w):
        key = ("crn", L.key, h, w)
        if key not in self._geom_cache:
            ho, wo = L.out_hw(h, w)
            self._geom_cache[key] = ops.conv_desc(L.kind, L.k, L.stride, L.pad, h, w, L.cin_s, ho, wo, L.cout_s, L.cin, L.cout)
        return self._geom_cache[key]

    def _aff(self, L, grad=False):
        """(gamma, beta) of layer L's BatchNorm in the flat parameter (or gradient) storage; (None, None) for InstanceNorm."""
        if L is None or L.norm != "bn":
            return None, None
        flat = self._gflat if grad else self._flat
        return flat[L.g_off: L.g_off + L.cout_s], flat[L.be_off: L.be_off + L.cout_s]

    def _cat_affine(self):
        """--norm batch: per concat buffer cat([l_s, h_{s+1}]) the affine of its two halves side by side (the label conv's BatchNorm,
        the BatchNorm of stage s + 1's last inter conv).  Two torch.cat launches per forward."""
        gs, bs = [], []
        for s in range(5):
            for L in (self.lab[s], self.inter[s + 1][-1]):
                g, b = self._aff(L)
                gs.append(g)
                bs.append(b)
        G, B = torch.cat(gs), torch.cat(bs)
        C2 = 2 * pad4(self.ngf)
        return {s: (G[s * C2: (s + 1) * C2], B[s * C2: (s + 1) * C2]) for s in range(5)}

    def run_forward(self, x, update_running=True):
        """x: dict(label=[H, W, 4] buffer, first=[H/64, W/64, pad4(input_nc + noise_nc)] buffer = cat([AvgPool64(label), noise]))
        -- the caller (forward) builds `first` because its channel order interleaves two tensors."""
        label = x["label"]
        ops.require_gpu(label, type(self).__name__)
        if self._flat.device != label.device:
            raise SganError(f"module parameters are on {self._flat.device}, input on {label.device}")
        H, W, _ = label.shape
        dev = label.device
        ngf, nlb = self.ngf, self.nlb
        C2 = 2 * ngf
        # statistics arena: per stage s <= 4 the concat statistics [2 * C2]; per stage the upsampled tensor [2 * ngf] and
        # the inner inter-block convs [2 * ngf] each; doubled for the backward sums
        lay, off = {}, 0
        for s in range(5, -1, -1):
            if s <= 4:
                lay[("cat", s)] = off
                off += 2 * C2
            lay[("u", s)] = off
            off += 2 * ngf
            for i in range(nlb - 1):
                lay[("t", s, i)] = off
                off += 2 * ngf
        arena = torch.zeros(2 * off, dtype=torch.float64, device=dev)
        st = lambda k, n: arena[lay[k]: lay[k] + n]
        res = {s: (H >> (s + 1), W >> (s + 1)) for s in range(6)}
        cat = {s: torch.empty(res[s] + (C2,), dtype=torch.float32, device=dev) for s in range(5)}
        # label branch: pyramid, then the (shared) label conv into the left halves
        lv = [torch.empty(res[s] + (4,), dtype=torch.float32, device=dev) for s in range(5)] + [x["pool64"]]
        ops.avgpool_pyramid_fwd(label, lv)
        if x.get("first") is None:
            x["first"] = x["first_fn"]()
        for s in range(5):
            L = self.lab[s]
            wt, b = self._wb(L)
            ops.conv_fwd(self._desc(L, *res[s]), lv[s], None, wt, b, cat[s][:, :, :ngf], ACT_NONE, st(("cat", s), 2 * C2), C2)
        final_act = self._take_call_act()
        cataff = self._cat_affine() if self.bn else {s: (None, None) for s in range(5)}
        saved = dict(final_act=final_act, label=label, first=x["first"], lv=lv, cat=cat, c={}, u={}, un={}, t={}, arena=arena, lay=lay, off=off, res=res,
                     cataff=cataff)
        running = []      # --norm batch: (layer, statistics, count, sq stride) in the reference's module order, for the running-statistics updates
        out = None
        drawn = 0
        for s in range(5, -1, -1):
            h, w = res[s]
            U = self.up[s]
            wt, b = self._wb(U)
            src = x["first"] if s == 5 else cat[s]
            nrm = None if s == 5 else ops.norm_desc(st(("cat", s), 2 * C2), *cataff[s], h * w, IN_EPS, ACT_NONE, 0.0)
            if s <= 4:
                running.append((self.lab[s], st(("cat", s), 2 * C2), h * w, C2))
            u = torch.empty((2 * h, 2 * w, ngf), dtype=torch.float32, device=dev)
            ustat = st(("u", s), 2 * ngf)
            running.append((U, ustat, 4 * h * w, 0))
            if self.mode == 'convt':
                ops.conv_fwd(self._desc(U, h, w), src, nrm, wt, b, u, ACT_NONE, ustat)
            else:
                c = torch.empty((h, w, ngf), dtype=torch.float32, device=dev)
                ops.conv_fwd(self._desc(U, h, w), src, nrm, wt, b, c, ACT_NONE, None)
                ops.bilinear_up2_fwd(c, u, ustat)
                saved["c"][s] = c
            saved["u"][s] = u
            cur, cur_stat = u, ustat
            if self.add_gauss and s > 0:      # t = norm(u) + sigma * noise, materialised; the inter block reads ReLU(t) with no norm
                if self.noise_override is not None:
                    nz = self.noise_override[s]
                else:
                    if self._rng_offset is None or self._rng_offset.device != dev:
                        self._rng_offset = torch.zeros(1, dtype=torch.int64, device=dev)
                    nz = torch.empty_like(u)
                    ops.normal_fill(nz, self._rng_seed + s, self._rng_offset, advance=False)
                    drawn = max(drawn, (nz.numel() + 3) // 4)
                tn = torch.empty_like(u)
                ops.norm_apply_fwd(u, ops.norm_desc(ustat, *self._aff(U), 4 * h * w, IN_EPS, ACT_NONE, 0.0), tn, None, nz, self.gauss_sigma)
                saved["un"][s] = tn
                cur, cur_stat = tn, None
            cur_L = U if cur_stat is not None else None      # whose norm the next conv applies on load
            for i, L in enumerate(self.inter[s]):
                wt, b = self._wb(L)
                nrm = ops.norm_desc(cur_stat, *self._aff(cur_L), 4 * h * w, IN_EPS, ACT_RELU, 0.0)
                last_i = i == nlb - 1
                if last_i and s == 0:
                    out = torch.empty((2 * h, 2 * w, L.cout_s), dtype=torch.float32, device=dev)
                    ops.conv_fwd(self._desc(L, 2 * h, 2 * w), cur, nrm, wt, b, out, final_act, None)
                elif last_i:   # feeds the next stage: right half of its concat buffer, statistics into the matching slice
                    dst = cat[s - 1][:, :, ngf:]
                    ops.conv_fwd(self._desc(L, 2 * h, 2 * w), cur, nrm, wt, b, dst, ACT_NONE, st(("cat", s - 1), 2 * C2)[ngf:], C2)
                    running.append((L, st(("cat", s - 1), 2 * C2)[ngf:], 4 * h * w, C2))
                else:
                    t = torch.empty((2 * h, 2 * w, ngf), dtype=torch.float32, device=dev)
                    tstat = st(("t", s, i), 2 * ngf)
                    ops.conv_fwd(self._desc(L, 2 * h, 2 * w), cur, nrm, wt, b, t, ACT_NONE, tstat)
                    saved["t"][(s, i)] = t
                    cur, cur_stat, cur_L = t, tstat, L
                    running.append((L, tstat, 4 * h * w, 0))
        if self.bn and update_running and self.training:
            # one launch per entry: the shared label block's BatchNorm is updated five times per forward, in order
            for L, stt, cnt, sq in running:
                nb = self._bn_boxes[L.key]
                ops.bn_running_update([(stt, nb.running_mean, nb.running_var, nb.num_batches_tracked, L.cout, cnt, sq if sq else L.cout_s)], BN_MOMENTUM)
        if drawn:
            ops.rng_advance(self._rng_offset, drawn)      # every stage read the same offset with its own seed
        saved["out"] = out
        return [out], saved

    def run_backward(self, x, outs, S, dout, need_dx, want_wgrad):
        """Returns (dlabel buffer or None, dfirst buffer or None)."""
        dev = dout.device
        ngf, nlb = self.ngf, self.nlb
        C2 = 2 * ngf
        res, cat, lay, off, arena = S["res"], S["cat"], S["lay"], S["off"], S["arena"]
        st = lambda k, n: arena[lay[k]: lay[k] + n]
        sm = lambda k, n: arena[off + lay[k]: off + lay[k] + n]          # backward sums live in the arena's second half
        if want_wgrad:
            self._ensure_grads()

        def wgrad(L, desc, src, nrm, dy):
            if want_wgrad:
                gw, gb = self._gwb(L)
                ops.conv_wgrad(desc, src, nrm, dy, gw, gb)

        def bwd(L, desc, src, nrm, dy, din, sums):
            """Backward-weight and backward-data of one conv: one fused launch where sgan_conv_bwd_fused covers the layer."""
            djob = [(desc, dy, self._wt(L), din, src, nrm, sums, 0, False, True, 0)]
            if want_wgrad:
                ops.conv_bwd_grouped(djob, [(desc, src, nrm, dy) + self._gwb(L)])
            else:
                ops.conv_dgrad_grouped(djob)

        if S["final_act"] == ACT_TANH:
            d = torch.empty_like(S["out"])
            ops.tanh_bwd(dout.contiguous(), S["out"], d)
        else:
            d = dout.contiguous()
        dcat_next = None        # gradient w.r.t. cat[s - 1] produced while walking stage s - 1; consumed by stage s
        dlv = [None] * 6
        dfirst = None
        for s in range(0, 6):
            h, w = res[s]
            u, ustat = S["u"][s], st(("u", s), 2 * ngf)
            # inter block, last conv first: `d` is the gradient w.r.t. the raw output of inter[s][-1]
            for i in range(nlb - 1, -1, -1):
                L = self.inter[s][i]
                noisy = i == 0 and s in S["un"]
                src = (S["un"][s] if noisy else u) if i == 0 else S["t"][(s, i - 1)]
                sstat = (None if noisy else ustat) if i == 0 else st(("t", s, i - 1), 2 * ngf)
                ssum = sm(("u", s), 2 * ngf) if i == 0 else sm(("t", s, i - 1), 2 * ngf)
                src_L = (None if noisy else self.up[s]) if i == 0 else self.inter[s][i - 1]      # whose norm `src` carries
                nrm = ops.norm_desc(sstat, *self._aff(src_L), 4 * h * w, IN_EPS, ACT_RELU, 0.0)
                desc = self._desc(L, 2 * h, 2 * w)
                din = torch.empty((2 * h, 2 * w, ngf), dtype=torch.float32, device=dev)
                bwd(L, desc, src, nrm, d, din, None if noisy else ssum)
                if noisy:      # din = d t (the noise has no gradient): sums of the norm backward, then the norm backward itself
                    unrm = ops.norm_desc(ustat, *self._aff(self.up[s]), 4 * h * w, IN_EPS, ACT_NONE, 0.0)
                    ops.norm_apply_bwd_sums(din, u, unrm, ssum, None)
                    ops.norm_bwd_apply(din, u, unrm, ssum, *(self._aff(self.up[s], grad=True) if want_wgrad else (None, None)))
                else:
                    ops.norm_bwd_apply(din, src, nrm, ssum, *(self._aff(src_L, grad=True) if want_wgrad else (None, None)))
                d = din
            # d = gradient w.r.t. u_s (raw, before its InstanceNorm)
            U = self.up[s]
            desc = self._desc(U, h, w)
            if self.mode == 'bilinear':
                dc = torch.empty((h, w, ngf), dtype=torch.float32, device=dev)
                ops.bilinear_up2_bwd(d, dc)
                d = dc
            src = S["first"] if s == 5 else cat[s]
            nrm = None if s == 5 else ops.norm_desc(st(("cat", s), 2 * C2), *S["cataff"][s], h * w, IN_EPS, ACT_NONE, 0.0)
            if s == 5:
                wgrad(U, desc, src, nrm, d)
                if need_dx:
                    dfirst = torch.empty_like(S["first"])
                    ops.conv_dgrad(desc, d, self._wt(U), dfirst, None, None, None, w_transposed=True)
                break
            dc_ = torch.empty_like(cat[s])
            csum = sm(("cat", s), 2 * C2)
            bwd(U, desc, src, nrm, d, dc_, csum)
            if self.bn and want_wgrad:      # the two halves' affine gradients come out side by side: add them where each layer keeps its own
                dgb = torch.zeros(2 * C2, dtype=torch.float32, device=dev)
                ops.norm_bwd_apply(dc_, cat[s], nrm, csum, dgb[:C2], dgb[C2:])
                for L_, lo in ((self.lab[s], 0), (self.inter[s + 1][-1], ngf)):
                    dg, db = self._aff(L_, grad=True)
                    dg.add_(dgb[lo: lo + ngf])
                    db.add_(dgb[C2 + lo: C2 + lo + ngf])
            else:
                ops.norm_bwd_apply(dc_, cat[s], nrm, csum)          # both halves at once: raw gradients of l_s and of h_{s+1}
            # label branch of this stage
            Ll = self.lab[s]
            ldesc = self._desc(Ll, h, w)
            wgrad(Ll, ldesc, S["lv"][s], None, dc_[:, :, :ngf])
            if need_dx:
                dlv[s] = torch.empty(res[s] + (4,), dtype=torch.float32, device=dev)
                ops.conv_dgrad(ldesc, dc_[:, :, :ngf], self._wt(Ll), dlv[s], None, None, None, w_transposed=True)
            d = dc_[:, :, ngf:]      # gradient w.r.t. the raw output of stage s + 1's last conv
        dlabel = None
        if need_dx:
            dlabel = torch.empty_like(S["label"])
            ops.avgpool_pyramid_bwd(dlv, dlabel, accumulate=False)     # level 5 travels with `dfirst`
        return dlabel, dfirst

    # ---- module protocol ---------------------------------------------------------------------------
    def forward(self, label, noise, activation=None):
        params = list(self.parameters())
        return self._apply_with_activation(activation, lambda: _CrnFn.apply(self, label, noise, *params))

    def _wrap_output(self, y):
        return y


class _CrnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, label, noise, *params):
        lb = ops.as_nhwc(label)
        H, W, _ = lb.shape
        if H % 64 or W % 64:
            raise SganError(f"CascadedRefinementNetwork needs H, W divisible by 64, got {H}x{W}")
        if tuple(noise.shape[2:]) != (H // 64, W // 64):
            raise SganError(f"noise must be {H // 64}x{W // 64} for a {H}x{W} label, got {tuple(noise.shape[2:])}")
        pool64 = torch.empty((H // 64, W // 64, 4), dtype=torch.float32, device=lb.device)
        x = {"label": lb, "pool64": pool64, "first": None}
        # cat([AvgPool64(label), noise], 1) interleaves two tensors channel-wise: assembled by torch on the 8x8 map.
        # The pyramid kernel has to run first, so the generator's first buffer is filled right after it.
        net_first = lambda: ops.as_nhwc(torch.cat([ops.logical_view(pool64, net.input_nc), noise], 1))
        x["first_fn"] = net_first
        outs, saved = net.run_forward(x)
        ctx.net, ctx.saved = net, saved
        ctx.need_dlabel, ctx.need_dnoise = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        ctx.want_wgrad = net.compute_param_grads and any(ctx.needs_input_grad[3:])
        return ops.logical_view(outs[-1], net.output_nc)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        need_dx = ctx.need_dlabel or ctx.need_dnoise
        dlabel, dfirst = net.run_backward(None, None, ctx.saved, ops.as_nhwc(gout), need_dx, ctx.want_wgrad)
        gl = gn = None
        if need_dx:
            dfl = ops.logical_view(dfirst, net.input_nc + net.noise_nc)
            if ctx.need_dnoise:
                gn = dfl[:, net.input_nc:]
            if ctx.need_dlabel:
                # level 5 of the pyramid: its gradient is the first input_nc channels of dfirst
                d5 = ops.as_nhwc(dfl[:, :net.input_nc])
                ops.avgpool_pyramid_bwd([None] * 5 + [d5], dlabel, accumulate=True)
                gl = ops.logical_view(dlabel, net.input_nc)
        return (None, gl, gn) + (None,) * (len(ctx.needs_input_grad) - 3)

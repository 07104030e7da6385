"""Discriminators of the reference's `define_D` on the MI355X path (models/networks.py:798-837,1074-1129): the multi-scale PatchGAN
(`n_layers` / `basic`, Gaussian pre-filter, logits fused into the loss) and the dcgan discriminator."""
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


# ------------------------------------------------------------------------------------------------
# helpers restated from the reference
# ------------------------------------------------------------------------------------------------
def matlab_style_gauss2D(shape=(3, 3), sigma=0.5):
    """fspecial('gaussian') (models/networks.py:22-33)."""
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x + y * y) / (2. * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    sumh = h.sum()
    if sumh != 0:
        h /= sumh
    return h


def init_gauss_filters(nf, kw, sigma):
    """models/networks.py:36-40."""
    filters = np.zeros((nf, nf, kw, kw))
    for i in range(nf):
        filters[i, i, :, :] = matlab_style_gauss2D((kw, kw), sigma)
    return filters


class DCGANDiscriminator(ChainNet):
    """DCGANDiscriminator (models/networks.py:1074-1129) for 128x128 inputs: Conv(nc -> ndf/2, k4,s2,p1) + LeakyReLU(0.2),
    four Conv(k4,s2,p1) + BatchNorm + LeakyReLU doubling the channels to 8 ndf, Conv(8 ndf -> 1, k4, s1, p0) -> Sigmoid,
    output flattened to [N]; no biases."""

    def __init__(self, gpu_ids=[], nc=3, ndf=64):
        chans = [int(ndf / 2), ndf, ndf * 2, ndf * 4, ndf * 8]
        layers = [LayerSpec("0", CONV, 4, 2, 1, nc, chans[0], False, None, ACT_LRELU, 0.2)]
        for i in range(1, 5):
            layers.append(LayerSpec(str(3 * i - 1), CONV, 4, 2, 1, chans[i - 1], chans[i], False, "bn", ACT_LRELU, 0.2))
        layers.append(LayerSpec("14", CONV, 4, 1, 0, chans[4], 1, False, None, ACT_NONE))
        super().__init__(layers)
        self.gpu_ids = gpu_ids
        self.input_nc = nc
        self.use_sigmoid = True
        self.gauss_filter = None
        self.fuse_sigmoid_into_loss = False     # trainers feeding GANLoss set it: forward then returns the tagged logits

    def _prepare_input(self, x, memo=None):
        key = (x.data_ptr(), tuple(x.shape), x.stride())
        img = memo.get(key) if memo is not None else None
        if img is None:
            img = ops.as_nhwc(x)
            if memo is not None:
                memo[key] = img
        return {"img": img, "chain_in": img}

    def _finish_input_grad(self, xb, dchain, into=None):
        return ops.logical_view(dchain, self.input_nc)

    def forward(self, input):
        return self._wrap_output(_ChainFn.apply(self, input, *list(self.model.parameters())))

    def _wrap_output(self, logits):
        if self.fuse_sigmoid_into_loss:
            logits._sgan_pending_sigmoid = True
            return logits
        p = _SigmoidFn.apply(logits)
        p._sgan_logits = logits
        return p.view(-1, 1).squeeze(1)


class NLayerDiscriminator(ChainNet):
    """NLayerDiscriminator (models/networks.py:798-847): [gauss prefilter + stride pick] ->
    Conv(k4,s2,p2)+LReLU -> (Conv s2 + norm + LReLU) x (n-1) -> Conv s1 + norm + LReLU -> Conv s1 [-> Sigmoid]."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm="instance", use_sigmoid=False, scale_factor=1,
                 num_classes=2, gpu_ids=[]):
        logit_nc = 1 if num_classes == 2 else int(num_classes)      # models/networks.py:806
        nrm = {"instance": "in", "batch": "bn"}[norm]
        kw, padw = 4, int(np.ceil((4 - 1) / 2))
        layers = [LayerSpec("0", CONV, kw, 2, padw, input_nc, ndf, True, None, ACT_LRELU, 0.2)]
        nf, idx = 1, 2
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(2 ** n, 8)
            layers.append(LayerSpec(str(idx), CONV, kw, 2, padw, ndf * nf_prev, ndf * nf, True, nrm, ACT_LRELU, 0.2))
            idx += 3
        nf_prev, nf = nf, min(2 ** n_layers, 8)
        layers.append(LayerSpec(str(idx), CONV, kw, 1, padw, ndf * nf_prev, ndf * nf, True, nrm, ACT_LRELU, 0.2))
        idx += 3
        layers.append(LayerSpec(str(idx), CONV, kw, 1, padw, ndf * nf, logit_nc, True, None, ACT_NONE))
        super().__init__(layers)
        self.gpu_ids = gpu_ids
        self.logit_nc = logit_nc
        self.use_sigmoid = use_sigmoid
        self.scale_factor = int(scale_factor)
        self.input_nc = input_nc
        self.gauss_filter = None
        # trainers that feed the output straight into GANLoss set this: forward then returns the logits
        # tagged for the fused sigmoid+BCE kernel instead of launching a separate sigmoid
        self.fuse_sigmoid_into_loss = False
        if self.scale_factor > 1:
            sigma = self.scale_factor // 2        # Python-2 integer division in the reference (:808)
            kg = 4 * sigma + 1
            box = _ParamBox("conv")
            box.weight = nn.Parameter(torch.zeros(input_nc, input_nc, kg, kg))
            self.gauss_filter = nn.Module()
            self.gauss_filter.add_module("0", box)
            self._gauss = (kg, 2 * sigma)

    def _extra_parameters(self):
        return [self.gauss_filter._modules["0"].weight] if self.gauss_filter is not None else []

    def _gauss_args(self):
        wg = self.gauss_filter._modules["0"].weight
        kg, padg = self._gauss
        return wg, (self.input_nc + 1) * kg * kg, kg, padg

    def _prepare_input(self, x, memo=None, defer=None):
        """`defer`: a list that collects the pre-filter jobs instead of launching them (the caller flushes the list with
        ops.gauss_down_multi_fwd: one launch for the scale-2 and scale-4 discriminators of a multi-scale set)."""
        key = (x.data_ptr(), tuple(x.shape), x.stride())
        img = memo.get(key) if memo is not None else None
        if img is None:
            img = ops.as_nhwc(x)
            if memo is not None:
                memo[key] = img
        xb = {"img": img}
        if self.scale_factor > 1:
            wg, gcs, kg, padg = self._gauss_args()
            H, W, Cs = xb["img"].shape
            s = self.scale_factor
            Ho, Wo = (H + 2 * padg - kg) // 1 + 1, (W + 2 * padg - kg) // 1 + 1       # conv output
            Ho, Wo = (Ho - 1) // s + 1, (Wo - 1) // s + 1                              # AvgPool2d(1, stride s)
            out = torch.empty((Ho, Wo, Cs), dtype=torch.float32, device=x.device)
            # conv(pad) then pick every s-th pixel == strided conv with the same pad
            if defer is not None:
                defer.append((self.input_nc, (xb["img"], out, wg, gcs, kg, padg, s)))
            else:
                ops.gauss_down_fwd(xb["img"], self.input_nc, wg, gcs, kg, padg, s, out)
            xb["chain_in"] = out
        else:
            xb["chain_in"] = xb["img"]
        return xb

    def _finish_input_grad(self, xb, dchain, into=None):
        """Gradient w.r.t. the image.  `into`: an NHWC image-gradient buffer another discriminator fed with the same image
        already produced -- this one's contribution is added to it and None is returned."""
        if self.scale_factor > 1:
            wg, gcs, kg, padg = self._gauss_args()
            dimg = into if into is not None else torch.empty_like(xb["img"])
            ops.gauss_down_bwd(dchain, self.input_nc, wg, gcs, kg, padg, self.scale_factor, dimg, accumulate=into is not None)
            dchain = dimg
        elif into is not None:
            into.add_(dchain)
        return None if into is not None else ops.logical_view(dchain, self.input_nc)

    def forward(self, x):
        params = list(self.model.parameters())
        return self._wrap_output(_ChainFn.apply(self, x, *params))

    def _wrap_output(self, logits):
        if not self.use_sigmoid:
            return logits
        if self.logit_nc > 1:      # class scores (--use_multi_class_GAN): a 3 x 67 x 67 map, plain elementwise sigmoid
            return torch.sigmoid(logits)
        if self.fuse_sigmoid_into_loss:
            logits._sgan_pending_sigmoid = True
            return logits
        p = _SigmoidFn.apply(logits)
        p._sgan_logits = logits
        return p


class NLayerDiscriminatorSep(NLayerDiscriminator):
    """NLayerDiscriminatorSep (models/networks.py:851-942): the 3-channel pair is split into its 2 label channels and its image
    channel, each goes through its own two-conv stem (netA, netB: Conv(k4,s2,p2)+LReLU -> Conv s2 + norm + LReLU), the two
    feature maps are concatenated and `model` (the rest of the PatchGAN) runs on them.

    On this path the two stems are ONE chain with block-diagonal weights: conv(2 -> ndf) (+) conv(1 -> ndf) is a conv 3 -> 2 ndf whose
    master weight holds netA's taps in the (outputs 0 .. ndf-1, inputs 0 .. 1) block and netB's in (ndf .. 2 ndf-1, input 2), zeros
    elsewhere; the second stem layer likewise (2 ndf -> 4 ndf).  Normalisation and LeakyReLU are per channel, and the channel
    order of the result IS torch.cat([y_A, y_B], 1) -- so the whole discriminator is one layer program on the existing kernels
    (the zero blocks cost 2x the MACs of two small layers).  The Parameters `netA.*` / `netB.*` are strided views of the diagonal
    blocks under the reference's state_dict keys; the off-diagonal blocks are never written by init or load_state_dict and their
    gradients are zeroed after every backward pass.

    The reference's forward applies netA to BOTH halves on the CPU branch (`y_B = self.netA(x_B)`, :940) and therefore raises on
    any input; its data_parallel branch uses netB, which is the evident intent and what this class computes (the golden comes
    from the reference with that one call patched, oracle/make_golden.py)."""

    no_group = True       # chain._same_architecture: never part of a grouped launch (the gradient masks live in run_backward)

    def __init__(self, input_nc, ndf=64, n_layers=3, norm="instance", use_sigmoid=False, scale_factor=1, num_classes=2, gpu_ids=[]):
        assert input_nc == 3, "n_layers_sep splits a 3-channel pair (models/networks.py:862)"
        assert n_layers >= 2 and pad4(ndf) == ndf, "n_layers_sep: n_layers >= 2, ndf a multiple of 4"
        logit_nc = 1 if num_classes == 2 else int(num_classes)
        nrm = {"instance": "in", "batch": "bn"}[norm]
        kw, padw = 4, 2
        layers = [LayerSpec("stem.0", CONV, kw, 2, padw, 3, 2 * ndf, True, None, ACT_LRELU, 0.2),
                  LayerSpec("stem.2", CONV, kw, 2, padw, 2 * ndf, 4 * ndf, True, nrm, ACT_LRELU, 0.2)]
        nf, idx = 4, 0      # models/networks.py:903: nf_mult = 2 * nf_mult after the stems
        for n in range(2, n_layers):
            nf_prev, nf = nf, min(2 ** n, 8)
            layers.append(LayerSpec(f"model.{idx}", CONV, kw, 2, padw, ndf * nf_prev, ndf * nf, True, nrm, ACT_LRELU, 0.2))
            idx += 3
        nf_prev, nf = nf, min(2 ** n_layers, 8)
        layers.append(LayerSpec(f"model.{idx}", CONV, kw, 1, padw, ndf * nf_prev, ndf * nf, True, nrm, ACT_LRELU, 0.2))
        idx += 3
        layers.append(LayerSpec(f"model.{idx}", CONV, kw, 1, padw, ndf * nf, logit_nc, True, None, ACT_NONE))
        self._ndf = ndf
        self._stem_boxes = {}
        ChainNet.__init__(self, layers)
        self.gpu_ids, self.logit_nc, self.use_sigmoid = gpu_ids, logit_nc, use_sigmoid
        self.scale_factor, self.input_nc = int(scale_factor), input_nc
        self.gauss_filter = None
        self.fuse_sigmoid_into_loss = False
        if self.scale_factor > 1:
            sigma = self.scale_factor // 2
            kg = 4 * sigma + 1
            box = _ParamBox("conv")
            box.weight = nn.Parameter(torch.zeros(input_nc, input_nc, kg, kg))
            self.gauss_filter = nn.Module()
            self.gauss_filter.add_module("0", box)
            self._gauss = (kg, 2 * sigma)

    # ---- module tree: `stem.*` boxes hold the full block-diagonal tensors and stay OUT of the tree; netA / netB boxes are registered ----
    def _param_root(self):
        return self

    def _add_box(self, key, box):
        if key.startswith("stem."):
            self._stem_boxes[key] = box
            if getattr(box, "_sgan_kind", None) == "conv":
                for net in ("netA", "netB"):
                    half = _ParamBox("conv")
                    half.weight = nn.Parameter(torch.empty(0))
                    half.bias = nn.Parameter(torch.empty(0))
                    ChainNet._add_box(self, f"{net}.{key[5:]}", half)
            else:      # BatchNorm behind the second stem conv: halves of gamma / beta / running statistics
                for net in ("netA", "netB"):
                    half = _ParamBox("bn")
                    half.weight, half.bias = nn.Parameter(torch.empty(0)), nn.Parameter(torch.empty(0))
                    n = box.running_mean.numel() // 2
                    half.register_buffer("running_mean", torch.zeros(n))
                    half.register_buffer("running_var", torch.ones(n))
                    half.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
                    ChainNet._add_box(self, f"{net}.{key[5:]}", half)
            return
        ChainNet._add_box(self, key, box)

    def _box(self, L):
        return self._stem_boxes[L.key] if L.key.startswith("stem.") else ChainNet._box(self, L)

    def _half_views(self, flat, L):
        """((wA, bA), (wB, bB)) diagonal blocks of stem layer L in `flat`, logical [Cout/2, Cin_half, kh, kw] / [Cout/2]."""
        m = flat[L.w_off: L.w_off + L.k * L.k * L.cout_s * L.cin_s].view(L.k, L.k, L.cout_s, L.cin_s).permute(2, 3, 0, 1)
        h = L.cout // 2
        ca = 2 if L.key == "stem.0" else L.cin // 2          # netA reads the 2 label channels, netB the image channel
        cb0, cb1 = (2, 3) if L.key == "stem.0" else (L.cin // 2, L.cin)
        b = flat[L.b_off: L.b_off + L.cout]
        return (m[:h, :ca], b[:h]), (m[h:L.cout, cb0:cb1], b[h:])

    def _rebind(self):
        ChainNet._rebind(self)
        for L in self.layers:
            if not L.key.startswith("stem."):
                continue
            n = L.k * L.k * L.cout_s * L.cin_s
            for net, (w, b), (gw, gb), wseg, boff in zip(("netA", "netB"), self._half_views(self._flat, L), self._half_views(self._gflat, L),
                                                         ((L.w_off, n), (L.w_off + n, 0)), (0, L.cout // 2)):
                box = ChainNet._box(self, LayerSpec(f"{net}.{L.key[5:]}", CONV, 4, 2, 2, 1, 1, True, None, ACT_NONE))
                box.weight.data, box.weight.grad = w, gw
                box.bias.data, box.bias.grad = b, gb
                box.weight._sgan_seg = (self, wseg[0], wseg[1])      # the optimizer's flat range: the whole slab rides with netA's half
                box.bias._sgan_seg = (self, L.b_off + boff, L.cout // 2)
            if L.norm == "bn":
                full = self._bn_boxes[L.key]
                g, be = self._flat[L.g_off: L.g_off + L.cout], self._flat[L.be_off: L.be_off + L.cout]
                gg, gbe = self._gflat[L.g_off: L.g_off + L.cout], self._gflat[L.be_off: L.be_off + L.cout]
                h = L.cout // 2
                for net, sl in (("netA", slice(0, h)), ("netB", slice(h, L.cout))):
                    nb = ChainNet._box(self, LayerSpec(f"{net}.{int(L.key[5:]) + 1}", CONV, 4, 2, 2, 1, 1, True, None, ACT_NONE))
                    nb.weight.data, nb.weight.grad = g[sl], gg[sl]
                    nb.bias.data, nb.bias.grad = be[sl], gbe[sl]
                    nb.weight._sgan_seg = (self, L.g_off + sl.start, h)
                    nb.bias._sgan_seg = (self, L.be_off + sl.start, h)
                    nb._buffers["running_mean"] = full.running_mean[sl]
                    nb._buffers["running_var"] = full.running_var[sl]

    def _ensure_grads(self):
        ChainNet._ensure_grads(self)
        self._rebind()

    def _apply(self, fn, recurse=True):
        for box in self._stem_boxes.values():      # not in the module tree: their BatchNorm buffers move with the net all the same
            for k, buf in box._buffers.items():
                if buf is not None:
                    box._buffers[k] = fn(buf)
        return ChainNet._apply(self, fn, recurse)

    def _default_bias_init(self):
        ChainNet._default_bias_init(self)
        for L in self.layers:      # each stem half keeps torch's bound for ITS fan-in (2 / 1 input channels; ndf each)
            if L.key.startswith("stem."):
                (_, ba), (_, bb) = self._half_views(self._flat, L)
                fa, fb = ((2, 1) if L.key == "stem.0" else (L.cin // 2, L.cin // 2))
                ba.uniform_(-1.0 / math.sqrt(fa * 16), 1.0 / math.sqrt(fa * 16))
                bb.uniform_(-1.0 / math.sqrt(fb * 16), 1.0 / math.sqrt(fb * 16))

    def _mask_offdiag(self, flat):
        for L in self.layers:
            if L.key.startswith("stem."):
                m = flat[L.w_off: L.w_off + L.k * L.k * L.cout_s * L.cin_s].view(L.k * L.k, L.cout_s, L.cin_s)
                h = L.cout // 2
                ca = 2 if L.key == "stem.0" else L.cin // 2
                m[:, :h, ca:].zero_()
                m[:, h:, :ca].zero_()

    def run_backward(self, x, outs, stats, dout, need_dx, want_wgrad):
        dx = ChainNet.run_backward(self, x, outs, stats, dout, need_dx, want_wgrad)
        if want_wgrad:
            self._mask_offdiag(self._gflat)      # the dense backward-weight kernels also fill the cross blocks: not parameters
        return dx

    def forward(self, x):
        return self._wrap_output(_ChainFn.apply(self, x, *[p for p in self.parameters()]))


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        lb = ops.as_nhwc(logits)
        pb = torch.empty_like(lb)
        ops.sigmoid_fwd(lb, pb)
        ctx.pb = pb
        return ops.logical_view(pb, 1)

    @staticmethod
    def backward(ctx, g):
        gb = ops.as_nhwc(g)
        dx = torch.empty_like(ctx.pb)
        ops.sigmoid_bwd(gb, ctx.pb, dx)
        return ops.logical_view(dx, 1)

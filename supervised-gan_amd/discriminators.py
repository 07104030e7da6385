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

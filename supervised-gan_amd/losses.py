"""Losses and small differentiable ops of the trainers on the MI355X path (models/networks.py:152-214 GANLoss / GANLossMultiClass /
WeightedL1Loss; the (label, image) pair concat, BCE on rescaled tanh outputs, bilinear x2 upsampling): one kernel forward, one backward."""
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


class _GanLossFn(torch.autograd.Function):
    """Sigmoid + BCELoss(mean) (or MSELoss) against a constant target, on the logits map."""

    @staticmethod
    def forward(ctx, logits, target, mode):
        lb = ops.as_nhwc(logits)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        ops.gan_loss_fwd(lb, target, mode, loss)
        ctx.lb, ctx.target, ctx.mode = lb, target, mode
        return loss

    @staticmethod
    def backward(ctx, gout):
        d = torch.empty_like(ctx.lb)
        ops.gan_loss_bwd(ctx.lb, ctx.target, ctx.mode, gout.contiguous(), d)
        return ops.logical_view(d, 1), None, None


class _CatPairFn(torch.autograd.Function):
    """torch.cat((a, b), 1) of two logical [1, C, H, W] tensors as ONE kernel that writes the padded NHWC buffer the discriminators
    read (no CatArrayBatchedCopy + layout pass), and one slice kernel per member that needs a gradient in backward."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.Ca, ctx.Cb = a.shape[1], b.shape[1]
        out = ops.concat_nhwc(ops.as_nhwc(a), ctx.Ca, ops.as_nhwc(b), ctx.Cb)
        # autograd hands the caller a detached alias of a differentiable view output; the buffer object must outlive this call
        # for that alias to map back to it without a copy (ops.as_nhwc looks the buffer up by address)
        ctx.out = out
        return ops.logical_view(out, ctx.Ca + ctx.Cb)

    @staticmethod
    def backward(ctx, g):
        gb = ops.as_nhwc(g)
        ga = ops.logical_view(ops.slice_nhwc(gb, 0, ctx.Ca), ctx.Ca) if ctx.needs_input_grad[0] else None
        gbb = ops.logical_view(ops.slice_nhwc(gb, ctx.Ca, ctx.Cb), ctx.Cb) if ctx.needs_input_grad[1] else None
        return ga, gbb


def cat_pair(a, b):
    """The conditional discriminators' input cat((label, image), 1) (models/cgan_model.py:162,172,187) on the HIP path; anything
    that is not a batch-1 fp32 device pair goes to torch.cat."""
    if a.is_cuda and b.is_cuda and a.dim() == 4 and a.shape[0] == 1 and a.shape[2:] == b.shape[2:] and a.dtype == b.dtype == torch.float32:
        return _CatPairFn.apply(a, b)
    return torch.cat((a, b), 1)


class _GanLossMultiFn(torch.autograd.Function):
    """total = sum_i w_i * GANLoss(pred_i, target_i) -- ONE kernel for all terms, their finish and (when a gradient will be asked
    for) d total / d pred_i for an upstream gradient of 1.  backward() hands those out as they are when the upstream gradient is
    the trainers' cached unit gradient (ops.register_unit_grad), and rescales them with one more kernel per term otherwise."""

    @staticmethod
    def forward(ctx, targets, weights, mode, *logits):
        lbs = [ops.as_nhwc(l) for l in logits]
        dev = logits[0].device
        each = torch.empty(len(lbs), dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        ds = [torch.empty_like(lb) for lb in lbs] if any(ctx.needs_input_grad[3:]) else None
        ops.gan_loss_multi_fwd(lbs, targets, weights, mode, each, total, ds)
        ctx.ds = ds
        ctx.mark_non_differentiable(each)
        ctx.set_materialize_grads(False)      # or autograd zero-fills a gradient for `each` on every backward (one more launch)
        return total, each

    @staticmethod
    def backward(ctx, gtotal, _geach):
        ds = ctx.ds
        if gtotal is None:
            return (None,) * (3 + len(ds))
        if not ops.is_unit_grad(gtotal):
            scaled = [torch.empty_like(d) for d in ds]
            g = gtotal.contiguous()
            for d, o in zip(ds, scaled):
                ops.scale(g, d, o)
            ds = scaled
        return (None, None, None) + tuple(ops.logical_view(d, 1) for d in ds)


class GANLoss(nn.Module):
    """GANLoss (models/networks.py:152-185).  With `use_lsgan=False` the reference applies BCELoss to
    the discriminator's Sigmoid output; here the loss kernel consumes the logits behind that output
    (numerically the same function, torch's -100 log clamp included)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor):
        super().__init__()
        self.real_label = target_real_label
        self.fake_label = target_fake_label
        self.use_lsgan = use_lsgan
        self.Tensor = tensor

    def _logits_of(self, input):
        if self.use_lsgan:
            return input
        logits = getattr(input, "_sgan_logits", None)
        if logits is None and getattr(input, "_sgan_pending_sigmoid", False):
            logits = input
        if logits is None:
            raise SganError("GANLoss(use_lsgan=False) needs the output of a supervised_gan_amd discriminator built with "
                            "use_sigmoid=True (it carries its logits); got a plain tensor")
        return logits

    def __call__(self, input, target_is_real):
        t = self.real_label if target_is_real else self.fake_label
        return _GanLossFn.apply(self._logits_of(input), t, 1 if self.use_lsgan else 0)

    def weighted_sum(self, inputs, targets_are_real, weights):
        """sum_i weights[i] * self(inputs[i], targets_are_real[i]) as ONE autograd node (<= 8 terms): returns
        (total, each) where `each` holds the unweighted terms for logging."""
        ts = [self.real_label if r else self.fake_label for r in targets_are_real]
        return _GanLossMultiFn.apply(ts, [float(w) for w in weights], 1 if self.use_lsgan else 0,
                                     *[self._logits_of(i) for i in inputs])


class _CEFn(torch.autograd.Function):
    """Class-weighted cross-entropy of a [1, C, H, W] logits map against an int64 label map [1, H, W] (or one class for every
    pixel): sgan_ce_fwd / sgan_ce_bwd.  The forward keeps nothing but the two fp64 sums; the backward recomputes the softmax."""

    @staticmethod
    def forward(ctx, logits, label, const_label, class_w):
        zb = ops.as_nhwc(logits)
        acc = ops.stat_arena(3, logits.device)      # [sum w nll, sum w, ticket]; zeroed
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        lab = label.reshape(-1).contiguous() if label is not None else None
        ops.ce_fwd(zb, logits.shape[1], lab, const_label, class_w, acc, loss)
        ctx.zb, ctx.lab, ctx.cl, ctx.cw, ctx.acc, ctx.C = zb, lab, const_label, class_w, acc, logits.shape[1]
        return loss

    @staticmethod
    def backward(ctx, gout):
        dz = torch.empty_like(ctx.zb)
        ops.ce_bwd(ctx.zb, ctx.C, ctx.lab, ctx.cl, ctx.cw, ctx.acc, gout.contiguous().float(), dz)
        return ops.logical_view(dz, ctx.C), None, None, None


def cross_entropy_logits(logits, label=None, const_label=0, class_weights=None):
    """nn.CrossEntropyLoss(weight=class_weights)(logits, label) for a [1, C, H, W] map on the HIP kernel (C <= 16, batch 1); `label`
    None: every pixel has class `const_label`."""
    assert logits.dim() == 4 and logits.shape[0] == 1 and logits.shape[1] <= 16, logits.shape
    cw = class_weights.detach().float().contiguous() if class_weights is not None else None
    return _CEFn.apply(logits, label, int(const_label), cw)


class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits):
        zb = ops.as_nhwc(logits)
        pb = torch.empty_like(zb)
        ops.softmax_fwd(zb, logits.shape[1], pb)
        ctx.pb, ctx.C = pb, logits.shape[1]
        return ops.logical_view(pb, logits.shape[1])

    @staticmethod
    def backward(ctx, g):
        dz = torch.empty_like(ctx.pb)
        ops.softmax_bwd(ops.as_nhwc(g), ctx.pb, ctx.C, dz)
        return ops.logical_view(dz, ctx.C)


def softmax_channels(logits):
    """F.softmax(logits, dim=1) of a [1, C, H, W] map on the HIP kernel (the result is NHWC-backed: the discriminators read it with
    no layout copy)."""
    assert logits.dim() == 4 and logits.shape[0] == 1 and logits.shape[1] <= 16, logits.shape
    return _SoftmaxFn.apply(logits)


class GANLossMultiClass(nn.Module):
    """GANLossMultiClass (models/networks.py:188-202): CrossEntropyLoss over the class channel of every pixel of a
    discriminator map against one class: one forward and one backward launch (sgan_ce_fwd / sgan_ce_bwd)."""

    def __init__(self, use_lsgan=False, num_classes=3, use_gpu=False):
        super().__init__()
        assert use_lsgan is False
        self.num_classes = num_classes

    def __call__(self, input, target_label):
        assert input.shape[1] == self.num_classes
        return cross_entropy_logits(input, None, int(target_label))


class _L1Fn(torch.autograd.Function):
    """lambda * mean(|x - y| * w) with w = 1 + sum_i (A_i + 1) / 2 * (weights_i - 1), or w a per-pixel map, or 1."""

    @staticmethod
    def forward(ctx, x, y, a, wts, nw, lam):
        xb = ops.as_nhwc(x)
        yb = ops.as_nhwc(y)
        ab = None
        if a is not None:
            ab = ops.as_nhwc(a)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        g = torch.empty_like(xb)
        ops.l1w_fwd(xb, yb, x.shape[1], ab, wts, nw, lam, loss, g)
        ctx.g, ctx.C = g, x.shape[1]
        return loss

    @staticmethod
    def backward(ctx, gout):
        dx = torch.empty_like(ctx.g)
        ops.scale(gout.contiguous(), ctx.g, dx)
        return ops.logical_view(dx, ctx.C), None, None, None, None, None


class _Bce01Fn(torch.autograd.Function):
    """BCELoss((x + 1) / 2, (t + 1) / 2), gradient w.r.t. x only."""

    @staticmethod
    def forward(ctx, x, t):
        xb, tb = ops.as_nhwc(x), ops.as_nhwc(t)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        g = torch.empty_like(xb)
        ops.bce01_fwd(xb, tb, x.shape[1], loss, g)
        ctx.g, ctx.C = g, x.shape[1]
        return loss

    @staticmethod
    def backward(ctx, gout):
        dx = torch.empty_like(ctx.g)
        ops.scale(gout.contiguous(), ctx.g, dx)
        return ops.logical_view(dx, ctx.C), None


def bce_on_rescaled(x, t):
    """torch.nn.BCELoss()((x + 1) / 2, (t + 1) / 2) of the two-stage trainers (twostage_cycle_model.py:398-403) as one
    forward and one backward kernel; `t` is treated as a constant."""
    return _Bce01Fn.apply(x, t.detach())


class _Bilinear2xFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='bilinear') on a logical [1, C, H, W] tensor (`--transform_1to2 bilinear_2`)."""

    @staticmethod
    def forward(ctx, x):
        xb = ops.as_nhwc(x)
        H, W, Cs = xb.shape
        out = torch.empty((2 * H, 2 * W, Cs), dtype=torch.float32, device=x.device)
        ops.bilinear_up2_fwd(xb, out, None)
        ctx.shape, ctx.C = (H, W, Cs), x.shape[1]
        return ops.logical_view(out, x.shape[1])

    @staticmethod
    def backward(ctx, g):
        din = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        ops.bilinear_up2_bwd(ops.as_nhwc(g), din)
        return ops.logical_view(din, ctx.C)


def bilinear_upsample2x(x):
    return _Bilinear2xFn.apply(x)


class WeightedL1Loss(nn.Module):
    """WeightedL1Loss (models/networks.py:205-214): mean(|x - y| * w).  One forward kernel (which also writes the
    gradient for a unit upstream) and one scaling kernel in backward."""

    def __call__(self, x, y, w=None):
        return _L1Fn.apply(x, y, w, None, 0, 1.0)

    def from_labels(self, x, y, real_A, weights, lam=1.0):
        """lam * self(x, y, w) with the weight map of CGANModel.backward_G (models/cgan_model.py:198-207),
        w = 1 + sum_i (real_A[:, i] + 1) / 2 * (weights[i] - 1), evaluated inside the kernel."""
        if weights is None:
            return _L1Fn.apply(x, y, None, None, 0, float(lam))
        wts = getattr(self, "_wts", None)
        if wts is None or wts.device != x.device or wts.numel() != len(weights):
            wts = self._wts = torch.tensor([float(v) for v in weights], dtype=torch.float32, device=x.device)
        return _L1Fn.apply(x, y, real_A, wts, len(weights), float(lam))

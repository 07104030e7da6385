"""Drop-in counterpart of the reference's network factory (models/networks.py) on MI355X.

Same entry points -- `define_G`, `define_D`, `GANLoss`, `WeightedL1Loss`, `print_network`,
`weights_init` -- same argument names, same returned-object protocol (`.forward`, `.parameters()`,
`netD.model.parameters()`, `.state_dict()` with the reference's key names and logical shapes,
`netD.gauss_filter`), but a network is a *layer program* over NHWC buffers executed by
hand-written gfx950 kernels (include/sgan_hip.h); normalisation and activations never exist as
separate passes (they are applied while the consumer conv stages its input) and the whole net is
one autograd node.

Implemented: which_model_netG in {fcgan, deconv (README alias), unet_128, unet_256, crn, autoencoder, dcgan}, which_model_netD in
{n_layers, basic}.  Other names raise NotImplementedError like the reference does for unknown
names (models/networks.py:95,123)."""
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

BN_EPS = 1e-5       # nn.BatchNorm2d default
IN_EPS = 1e-5       # nn.InstanceNorm2d default
BN_MOMENTUM = 0.1


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


def weights_init(m):
    """models/networks.py:13-19, applied to our parameter containers (class-name free)."""
    kind = getattr(m, "_sgan_kind", None)
    if kind == "conv":
        m.weight.data.normal_(0.0, 0.02)
        seg = getattr(m.weight, "_sgan_seg", None)
        if seg is not None:      # a write through `.data` moves no version counter: tell the net its derived copies are stale
            seg[0].invalidate_derived()
    elif kind == "bn":
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def print_network(net):
    """models/networks.py:135-140."""
    num_params = sum(p.numel() for p in net.parameters())
    print(net)
    print('Total number of parameters: %d' % num_params)


# ------------------------------------------------------------------------------------------------
# layer program
# ------------------------------------------------------------------------------------------------
@dataclass
class LayerSpec:
    key: str                 # index of the conv inside the reference's nn.Sequential ("0", "3", ...)
    kind: int                # CONV / CONVT
    k: int
    stride: int
    pad: int
    cin: int                 # logical channels
    cout: int
    bias: bool
    norm: Optional[str]      # None | "in" | "bn" : normalisation of THIS layer's output
    act: int                 # activation after the norm (applied by the consumer on load)
    slope: float = 0.0
    drop: float = 0.0        # nn.Dropout(p) between this layer's norm and its activation (training mode only)
    # filled by the net
    w_off: int = 0
    b_off: int = -1
    g_off: int = -1          # BN gamma / beta offsets
    be_off: int = -1

    @property
    def cin_s(self):
        return pad4(self.cin)

    @property
    def cout_s(self):
        return pad4(self.cout)

    def out_hw(self, h, w):
        if self.kind == CONV:
            return (h + 2 * self.pad - self.k) // self.stride + 1, (w + 2 * self.pad - self.k) // self.stride + 1
        return (h - 1) * self.stride - 2 * self.pad + self.k, (w - 1) * self.stride - 2 * self.pad + self.k


class _ParamBox(nn.Module):
    """Stand-in for one numbered child of the reference's nn.Sequential: owns `weight` / `bias`
    Parameters that are strided views into the net's flat storage."""

    def __init__(self, kind):
        super().__init__()
        self._sgan_kind = kind

    def extra_repr(self):
        return ", ".join(f"{n}={tuple(p.shape)}" for n, p in self._parameters.items() if p is not None)


class _BwdArena:
    """Zeroed fp64 scratch for the backward sums, carved from the same fill as the forward statistics.  A second
    backward through the same forward (retain_graph) gets a fresh zeroed buffer."""

    def __init__(self, buf):
        self.buf, self.dev = buf, buf.device

    def take(self, n):
        buf, self.buf = self.buf, None
        if buf is None or buf.numel() < n:
            return torch.zeros(n, dtype=torch.float64, device=self.dev)
        return buf[:n]


class ChainNet(nn.Module):
    """A sequential conv net as a layer program over flat fp32 storage.

    Master layouts (include/sgan_hip.h): conv weight [kh*kw][Cout_s][Cin_s]; exposed to
    state_dict()/optimizers as strided views with the reference's logical shapes, so checkpoints
    interchange with the reference without any conversion pass."""

    final_act = ACT_NONE

    def __init__(self, layers: List[LayerSpec]):
        super().__init__()
        self.layers = layers
        off = self._assign_offsets(layers)
        self._nflat = off
        self._flat = torch.zeros(off, dtype=torch.float32)
        self._gflat = torch.zeros(off, dtype=torch.float32)
        self._arena = (self._flat, self._gflat, 0)   # (param arena, grad arena, this net's offset); see pack_flat()
        self.model = nn.Module()
        self._bn_boxes = {}
        for L in layers:
            box = _ParamBox("conv")
            box.weight = nn.Parameter(torch.empty(0))
            box.bias = nn.Parameter(torch.empty(0)) if L.bias else None
            self._add_box(L.key, box)
            if L.norm == "bn":
                nb = _ParamBox("bn")
                nb.weight = nn.Parameter(torch.empty(0))
                nb.bias = nn.Parameter(torch.empty(0))
                nb.register_buffer("running_mean", torch.zeros(L.cout))
                nb.register_buffer("running_var", torch.ones(L.cout))
                nb.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
                parts = L.key.split(".")      # the norm is the next numbered child of the same nn.Sequential
                self._add_box(".".join(parts[:-1] + [str(int(parts[-1]) + 1)]), nb)
                self._bn_boxes[L.key] = nb
        self._rebind()
        self._default_bias_init()
        self.compute_param_grads = True   # trainers may clear this while only dX is wanted (G step)
        self._geom_cache = {}

    def _take_call_act(self):
        """Output activation of THIS call: `final_act`, unless forward() was handed its own callable -- then the chain ends raw
        (ACT_NONE) and the callable runs on the logical output (the reference's `activation=` argument, models/networks.py:535-540)."""
        act = getattr(self, "_call_act", None)
        self._call_act = None
        return self.final_act if act is None else act

    def _apply_with_activation(self, activation, run):
        """run() -> output of the autograd node; a non-Tanh `activation` switches the fused tanh off for this call."""
        custom = activation is not None and not isinstance(activation, nn.Tanh)
        self._call_act = ACT_NONE if custom else None
        try:
            y = run()
        finally:
            self._call_act = None
        return activation(y) if custom else y

    def _assign_offsets(self, layers):
        """Place every layer's weight / bias / BN affine in the flat storage; returns the total length."""
        off = 0
        for L in layers:
            L.w_off = off
            off += L.k * L.k * L.cout_s * L.cin_s
            if L.bias:
                L.b_off = off
                off += L.cout_s
            if L.norm == "bn":
                L.g_off = off
                off += L.cout_s
                L.be_off = off
                off += L.cout_s
        return off

    # ---- module tree ---------------------------------------------------------------------------
    def _add_box(self, key, box):
        """Register `box` under self.model at a dotted path ("1.model.3.model.1"), creating plain containers on
        the way, so state_dict() keys equal the reference's nested nn.Sequential names."""
        node = self._param_root()
        parts = key.split(".")
        for part in parts[:-1]:
            if part not in node._modules:
                node.add_module(part, nn.Module())
            node = node._modules[part]
        node.add_module(parts[-1], box)

    def _param_root(self):
        """Module under which the parameter boxes live: `self.model` mirrors the reference nets that keep their layers
        in `self.model`; nets whose blocks are direct attributes (CRN) return self."""
        return self.model

    def _box(self, L: LayerSpec):
        node = self._param_root()
        for part in L.key.split("."):
            node = node._modules[part]
        return node

    # ---- storage <-> Parameter views -------------------------------------------------------
    def _views(self, flat, L: LayerSpec):
        m = flat[L.w_off: L.w_off + L.k * L.k * L.cout_s * L.cin_s].view(L.k, L.k, L.cout_s, L.cin_s)
        if L.kind == CONVT:
            w = m.permute(3, 2, 0, 1)[:L.cin, :L.cout]       # logical [Cin, Cout, kh, kw]
        else:
            w = m.permute(2, 3, 0, 1)[:L.cout, :L.cin]       # logical [Cout, Cin, kh, kw]
        b = flat[L.b_off: L.b_off + L.cout] if L.bias else None
        g = flat[L.g_off: L.g_off + L.cout] if L.norm == "bn" else None
        be = flat[L.be_off: L.be_off + L.cout] if L.norm == "bn" else None
        return w, b, g, be

    def _rebind(self):
        for L in self.layers:
            box = self._box(L)
            w, b, g, be = self._views(self._flat, L)
            gw, gb, gg, gbe = self._views(self._gflat, L)
            box.weight.data = w
            box.weight.grad = gw
            box.weight._sgan_seg = (self, L.w_off, L.k * L.k * L.cout_s * L.cin_s)
            if L.bias:
                box.bias.data = b
                box.bias.grad = gb
                box.bias._sgan_seg = (self, L.b_off, L.cout_s)
            if L.norm == "bn":
                nb = self._bn_boxes[L.key]
                nb.weight.data, nb.weight.grad = g, gg
                nb.bias.data, nb.bias.grad = be, gbe
                nb.weight._sgan_seg = (self, L.g_off, L.cout_s)
                nb.bias._sgan_seg = (self, L.be_off, L.cout_s)

    def _default_bias_init(self):
        """torch's default conv bias init U(+-1/sqrt(fan_in)); weights_init leaves it in place in the
        reference (models/networks.py:13-19 touches only .weight of convs)."""
        for L in self.layers:
            if L.bias:
                fan_in = (L.cin if L.kind == CONV else L.cout) * L.k * L.k
                bound = 1.0 / math.sqrt(fan_in)
                self._box(L).bias.data.uniform_(-bound, bound)

    def _ensure_grads(self):
        """Re-attach .grad views if someone set them to None (torch's zero_grad(set_to_none=True))."""
        for L in self.layers:
            box = self._box(L)
            gw, gb, gg, gbe = self._views(self._gflat, L)
            pairs = [(box.weight, gw)]
            if L.bias:
                pairs.append((box.bias, gb))
            if L.norm == "bn":
                nb = self._bn_boxes[L.key]
                pairs += [(nb.weight, gg), (nb.bias, gbe)]
            for p, gv in pairs:
                if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                    gv.zero_()
                    p.grad = gv

    def _apply(self, fn, recurse=True):
        self._flat = fn(self._flat).clone() if self._arena[0] is not self._flat else fn(self._flat)
        self._gflat = fn(self._gflat).clone() if self._arena[1] is not self._gflat else fn(self._gflat)
        self._arena = (self._flat, self._gflat, 0)
        for mod in self.modules():
            for k, buf in mod._buffers.items():
                if buf is not None:
                    mod._buffers[k] = fn(buf)
        for p in self._extra_parameters():
            p.data = fn(p.data)
            if p.grad is not None:
                p.grad = fn(p.grad)
        self._rebind()
        self._geom_cache = {}
        return self

    def _extra_parameters(self):
        return []

    def flat_segment(self):
        """(params, grads, numel) of the contiguous storage behind `self.model.parameters()`."""
        return self._flat, self._gflat, self._nflat

    def zero_grad_flat(self):
        self._gflat.zero_()

    # ---- geometry ---------------------------------------------------------------------------
    def _geometry(self, H, W):
        key = (H, W)
        if key not in self._geom_cache:
            geo = []
            h, w = H, W
            for L in self.layers:
                ho, wo = L.out_hw(h, w)
                if ho <= 0 or wo <= 0:
                    raise SganError(f"input {H}x{W} too small for layer {L.key}")
                geo.append((ops.conv_desc(L.kind, L.k, L.stride, L.pad, h, w, L.cin_s, ho, wo, L.cout_s, L.cin, L.cout), h, w, ho, wo))
                h, w = ho, wo
            self._geom_cache[key] = geo
        return self._geom_cache[key]

    def _wb(self, L: LayerSpec):
        """(weight, bias) of layer L for the forward pass; the weight slice carries the matching slice of the split-bf16
        forward copy for the bf16x3 kernels."""
        n = L.k * L.k * L.cout_s * L.cin_s
        w = self._flat[L.w_off: L.w_off + n]
        if ops.get_math() == "bf16x3":
            self._refresh_derived()
            ops.with_packed(w, self._pk_f[L.w_off: L.w_off + n])
        b = self._flat[L.b_off: L.b_off + L.cout_s] if L.bias else None
        return w, b

    def _wt(self, L: LayerSpec):
        """Weights of layer L from the transposed copy [tap][Cin][Cout] that backward-data reads (+ its split-bf16 twin)."""
        self._refresh_derived()
        n = L.k * L.k * L.cout_s * L.cin_s
        return ops.with_packed(self._flat_t[L.w_off: L.w_off + n], self._pk_b[L.w_off: L.w_off + n])

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_derived()
        return out

    def invalidate_derived(self):
        """Call after writing the parameters behind torch's back (e.g. through `.data` of a foreign alias): the derived weight
        copies are re-made before the next kernel that reads them."""
        self._wt_epoch = getattr(self, "_wt_epoch", 0) + 1

    def _derived_key(self):
        # Every way the flat storage changes must move this key: FusedAdam.step() / load_state_dict / _apply / weights_init bump
        # `_wt_epoch`; torch optimizers, `param.copy_` and the like bump the Parameters' own version counters (set_data gave each
        # Parameter a counter of its own, so `_flat._version` alone misses them); in-place ops on `_flat` itself bump its counter.
        return (self._flat.data_ptr(), self._flat._version, getattr(self, "_wt_epoch", 0),
                tuple(p._version for p in self._conv_weight_params()))

    def _conv_weight_params(self):
        ps = getattr(self, "_cw_params", None)
        if ps is None:
            ps = self._cw_params = [self._box(L).weight for L in self.layers]
        return ps

    def _refresh_derived(self):
        """The three derived weight copies (fp32 transposed, split-bf16 forward / backward: sgan_pack_weights), refreshed lazily
        by ONE launch whenever the parameters changed since they were made."""
        key = self._derived_key()
        if getattr(self, "_wt_key", None) == key:
            return
        mates = getattr(self, "_arena_mates", None)
        if mates is not None and self._refresh_arena(mates):
            return
        if getattr(self, "_flat_t", None) is None or self._flat_t.shape != self._flat.shape or self._flat_t.device != self._flat.device:
            self._flat_t = torch.zeros_like(self._flat)
            self._pk_f = torch.zeros_like(self._flat)
            self._pk_b = torch.zeros_like(self._flat)
        ops.pack_weights(self._flat, self._flat_t, self._pk_f, self._pk_b, self._conv_segments())
        self._wt_key = key

    def _conv_segments(self, base=0):
        segs, seen = [], set()
        for Lx in self.layers:
            if Lx.w_off not in seen:
                seen.add(Lx.w_off)
                segs.append((base + Lx.w_off, Lx.k * Lx.k, Lx.cout_s, Lx.cin_s))
        return segs

    def _refresh_arena(self, mates) -> bool:
        """Networks that share one parameter arena (pack_flat) and one optimizer go stale together: refresh the derived copies of
        all of them that are stale in ONE launch over the arena.  False: the arena was re-homed since; take the per-net path."""
        arena_p = self._arena[0]
        if any(getattr(n, "_arena", (None,))[0] is not arena_p or n._flat.data_ptr() != arena_p.data_ptr() + 4 * n._arena[2] for n in mates):
            return False
        der = getattr(arena_p, "_sgan_derived", None)
        if der is None or der[0].shape != arena_p.shape or der[0].device != arena_p.device:
            der = arena_p._sgan_derived = tuple(torch.zeros_like(arena_p) for _ in range(3))
        segs, stale = [], []
        for n in mates:
            off = n._arena[2]
            if getattr(n, "_flat_t", None) is None or n._flat_t.data_ptr() != der[0].data_ptr() + 4 * off:
                n._flat_t, n._pk_f, n._pk_b = (d[off: off + n._nflat] for d in der)
                n._wt_key = None
            key = n._derived_key()
            if n._wt_key != key:
                segs += n._conv_segments(off)
                stale.append((n, key))
        ops.pack_weights(arena_p, der[0], der[1], der[2], segs)
        for n, key in stale:
            n._wt_key = key
        return True

    def _gwb(self, L: LayerSpec):
        w = self._gflat[L.w_off: L.w_off + L.k * L.k * L.cout_s * L.cin_s]
        b = self._gflat[L.b_off: L.b_off + L.cout_s] if L.bias else None
        return w, b

    def _norm_of(self, li, stats, count):
        """How a consumer reads layer li's raw output: its norm (from `stats`) + activation."""
        L = self.layers[li]
        if L.norm is None:
            return ops.norm_desc(None, None, None, count, 0.0, L.act, L.slope)
        st = stats[li]
        if L.norm == "bn":
            g = self._flat[L.g_off: L.g_off + L.cout_s]
            be = self._flat[L.be_off: L.be_off + L.cout_s]
            return ops.norm_desc(st, g, be, count, BN_EPS, L.act, L.slope)
        return ops.norm_desc(st, None, None, count, IN_EPS, L.act, L.slope)

    def _norm_in(self, li, stats, count, drop):
        """_norm_of for a consumer that may read the materialised dropout tensor of layer li: only the activation is left to apply."""
        if li in drop:
            L = self.layers[li]
            return ops.norm_desc(None, None, None, count, 0.0, L.act, L.slope)
        return self._norm_of(li, stats, count)

    # ---- forward / backward programs ----------------------------------------------------------
    def run_forward(self, x: torch.Tensor, update_running=True):
        """x: [H, W, Cs] NHWC buffer.  Returns (outs, stats): raw conv outputs and per-layer stats."""
        ops.require_gpu(x, type(self).__name__)
        if self._flat.device != x.device:
            raise SganError(f"module parameters are on {self._flat.device}, input on {x.device}")
        H, W, Cs = x.shape
        assert Cs == self.layers[0].cin_s, (Cs, self.layers[0].cin_s)
        geo = self._geometry(H, W)
        final_act = self._take_call_act()
        n_stats = sum(2 * L.cout_s for L in self.layers if L.norm)
        # one zero-fill serves the forward statistics and the backward sums (second half, consumed by run_backward)
        arena = torch.zeros(max(2 * n_stats, 1), dtype=torch.float64, device=x.device)
        stats, o = [], 0
        for L in self.layers:
            if L.norm:
                stats.append(arena[o: o + 2 * L.cout_s])
                o += 2 * L.cout_s
            else:
                stats.append(None)
        stats.append(_BwdArena(arena[n_stats:]))
        stats[-1].final_act = final_act
        outs = []
        cur = x
        # Dropout layers (norm -> Dropout(p) -> ReLU, the AutoEncoder's): the mask commutes with the ReLU, so the masked normalised
        # tensor t = norm(y) * mask is materialised by one pass (sgan_norm_apply_fwd) and the consumer reads ReLU(t) with no norm.
        drop = {}
        if self.training and any(L.drop > 0 for L in self.layers):
            if getattr(self, "_rng_offset", None) is None or self._rng_offset.device != x.device:
                self._rng_offset = torch.zeros(1, dtype=torch.int64, device=x.device)
            drawn = 0
        for li, L in enumerate(self.layers):
            desc, h, w, ho, wo = geo[li]
            out = torch.empty((ho, wo, L.cout_s), dtype=torch.float32, device=x.device)
            in_norm = self._norm_in(li - 1, stats, h * w, drop) if li > 0 else None
            wt, b = self._wb(L)
            last = li == len(self.layers) - 1
            ops.conv_fwd(desc, cur, in_norm, wt, b, out, final_act if last else ACT_NONE, stats[li])
            outs.append(out)
            cur = out
            if self.training and L.drop > 0:
                mask = torch.empty((ho, wo, L.cout_s), dtype=torch.float32, device=x.device)
                src = getattr(self, "mask_source", None)        # tests inject the reference's masks
                if src is not None:
                    mask.copy_(src(li, (ho, wo, L.cout_s)))
                else:
                    ops.dropout_mask(mask, L.drop, getattr(self, "_rng_seed", 0) + li, self._rng_offset, advance=False)
                    drawn = max(drawn, (mask.numel() + 3) // 4)
                t = torch.empty_like(out)
                ops.norm_apply_fwd(out, ops.norm_desc(stats[li], None, None, ho * wo, IN_EPS, ACT_NONE, 0.0), t, mask)
                drop[li] = (t, mask)
                cur = t
        if drop and getattr(self, "mask_source", None) is None:
            ops.rng_advance(self._rng_offset, drawn)
        stats[-1].drop = drop
        if update_running and self._bn_boxes:
            rl = []
            for li, L in enumerate(self.layers):
                if L.norm == "bn":
                    nb = self._bn_boxes[L.key]
                    _, _, _, ho, wo = geo[li]
                    rl.append((stats[li], nb.running_mean, nb.running_var, nb.num_batches_tracked, L.cout, ho * wo, L.cout_s))
            ops.bn_running_update(rl, BN_MOMENTUM)
        return outs, stats

    def run_backward(self, x, outs, stats, dout, need_dx: bool, want_wgrad: bool):
        """dout: gradient w.r.t. the net output (after final_act), [Ho, Wo, Cs].  Returns dx or None."""
        geo = self._geometry(x.shape[0], x.shape[1])
        nL = len(self.layers)
        dev = x.device
        if want_wgrad:
            self._ensure_grads()
        dcur = dout
        if getattr(stats[-1], "final_act", self.final_act) == ACT_TANH:
            d2 = torch.empty_like(outs[-1])
            ops.tanh_bwd(dcur.contiguous(), outs[-1], d2)
            dcur = d2
        n_sums = sum(2 * L.cout_s for L in self.layers if L.norm)
        arena = stats[-1].take(max(n_sums, 1))
        sums, o = [], 0
        for L in self.layers:
            if L.norm:
                sums.append(arena[o: o + 2 * L.cout_s])
                o += 2 * L.cout_s
            else:
                sums.append(None)
        dx = None
        drop = getattr(stats[-1], "drop", {})
        for li in range(nL - 1, -1, -1):
            L = self.layers[li]
            desc, h, w, ho, wo = geo[li]
            dropped = (li - 1) in drop
            src = (drop[li - 1][0] if dropped else outs[li - 1]) if li > 0 else x
            in_norm = self._norm_in(li - 1, stats, h * w, drop) if li > 0 else None
            wt, _ = self._wb(L)
            if want_wgrad:
                gw, gb = self._gwb(L)
                ops.conv_wgrad(desc, src, in_norm, dcur, gw, gb)
            if li > 0:
                P = self.layers[li - 1]
                din = torch.empty((h, w, P.cout_s), dtype=torch.float32, device=dev)
                with ops.math_scope(_dgrad_math(P)):
                    ops.conv_dgrad(desc, dcur, self._wt(L), din, src, in_norm, None if dropped else sums[li - 1], w_transposed=True)
                if dropped:      # din = d t * ReLU'(t); through the mask, with the two norm-backward sums of the masked gradient
                    raw_norm = self._norm_of(li - 1, stats, h * w)
                    ops.norm_apply_bwd_sums(din, outs[li - 1], raw_norm, sums[li - 1], drop[li - 1][1])
                    ops.norm_bwd_apply(din, outs[li - 1], raw_norm, sums[li - 1])
                elif P.norm:
                    dg = self._gflat[P.g_off: P.g_off + P.cout_s] if (P.norm == "bn" and want_wgrad) else None
                    db = self._gflat[P.be_off: P.be_off + P.cout_s] if (P.norm == "bn" and want_wgrad) else None
                    ops.norm_bwd_apply(din, src, in_norm, sums[li - 1], dg, db)
                dcur = din
            elif need_dx:
                dx = torch.empty((h, w, L.cin_s), dtype=torch.float32, device=dev)
                ops.conv_dgrad(desc, dcur, self._wt(L), dx, None, None, None, w_transposed=True)
        return dx


def _dgrad_math(P):
    """Arithmetic of the backward-data launch whose result is the gradient of layer P's output.  Behind a normalisation the
    result goes through the norm backward, which re-centres it with sums taken from the very same values: the 5e-6 element errors
    of the split products stay 5e-6.  Without one (the first PatchGAN layer) the result is used as is, and the layer's bias
    gradient sums it over every pixel -- terms that cancel to a small residual (the gradient that reaches it left a normalisation
    as a zero-sum field) while unbiased element errors do not: measured 7e-3 of the bias gradient against the fp64 reference,
    where the reference's own fp32 is at 7e-6.  Those launches run on the exact-fp32 kernel (one per discriminator pass)."""
    return None if P.norm else "f32"


def pack_flat(nets):
    """Re-home the flat parameter / gradient storage of several networks in ONE contiguous arena, so an
    optimizer over all of them is a single Adam segment and a single gradient all-reduce (the three
    fcgan discriminators: 3 x 693,729 parameters -> one 8.3 MB buffer)."""
    nets = list(nets)
    dev = nets[0]._flat.device
    total = sum(n._nflat for n in nets)
    arena_p = torch.empty(total, dtype=torch.float32, device=dev)
    arena_g = torch.zeros(total, dtype=torch.float32, device=dev)
    off = 0
    for n in nets:
        arena_p[off: off + n._nflat].copy_(n._flat)
        n._flat = arena_p[off: off + n._nflat]
        n._gflat = arena_g[off: off + n._nflat]
        n._arena = (arena_p, arena_g, off)
        n._rebind()
        off += n._nflat
    for n in nets:
        if isinstance(n, ChainNet) and all(isinstance(m, ChainNet) for m in nets):
            n._arena_mates = nets
            n._flat_t = None      # derived copies move into arena-wide buffers on the next refresh
    return arena_p, arena_g


class _ChainFn(torch.autograd.Function):
    """One autograd node per network call."""

    @staticmethod
    def forward(ctx, net: "ChainNet", x_logical, *params):
        xb = net._prepare_input(x_logical)
        outs, stats = net.run_forward(xb["chain_in"])
        ctx.net, ctx.xb, ctx.outs, ctx.stats = net, xb, outs, stats
        ctx.want_wgrad = net.compute_param_grads and any(ctx.needs_input_grad[2:])
        ctx.need_dx = ctx.needs_input_grad[1]
        return ops.logical_view(outs[-1], net.layers[-1].cout)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        g = ops.as_nhwc(gout)
        dchain = net.run_backward(ctx.xb["chain_in"], ctx.outs, ctx.stats, g, ctx.need_dx, ctx.want_wgrad)
        dx = net._finish_input_grad(ctx.xb, dchain) if ctx.need_dx else None
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


# ------------------------------------------------------------------------------------------------
# grouped execution: several chains of the same architecture, one kernel launch per layer
# ------------------------------------------------------------------------------------------------
def _same_architecture(a: "ChainNet", b: "ChainNet") -> bool:
    if len(a.layers) != len(b.layers) or a.final_act != b.final_act:
        return False
    key = lambda L: (L.kind, L.k, L.stride, L.pad, L.cin, L.cout, L.bias, L.norm, L.act, L.slope)
    return all(key(x) == key(y) for x, y in zip(a.layers, b.layers))


def can_group(nets) -> bool:
    nets = list(nets)
    return 1 < len(nets) <= 8 and all(_same_architecture(nets[0], n) for n in nets[1:])


def _grouped_forward(nets, xs):
    """nets[j] applied to xs[j] ([H,W,Cs] buffers); per layer ONE grouped launch.  Returns per-job (outs, stats)."""
    dev = xs[0].device
    J = len(nets)
    geos = [n._geometry(x.shape[0], x.shape[1]) for n, x in zip(nets, xs)]
    per_job = sum(2 * L.cout_s for L in nets[0].layers if L.norm)
    arena = torch.zeros(max(2 * per_job * J, 1), dtype=torch.float64, device=dev)   # forward statistics | backward sums
    bwd = _BwdArena(arena[per_job * J:])
    stats = []
    for j in range(J):
        st, o = [], j * per_job
        for L in nets[j].layers:
            if L.norm:
                st.append(arena[o: o + 2 * L.cout_s])
                o += 2 * L.cout_s
            else:
                st.append(None)
        st.append(bwd)
        stats.append(st)
    outs = [[] for _ in range(J)]
    cur = list(xs)
    nL = len(nets[0].layers)
    for li in range(nL):
        jobs = []
        for j, net in enumerate(nets):
            L = net.layers[li]
            desc, h, w, ho, wo = geos[j][li]
            out = torch.empty((ho, wo, L.cout_s), dtype=torch.float32, device=dev)
            in_norm = net._norm_of(li - 1, stats[j], h * w) if li > 0 else None
            wt, b = net._wb(L)
            jobs.append((desc, cur[j], in_norm, wt, b, out, stats[j][li]))
            outs[j].append(out)
            cur[j] = out
        ops.conv_fwd_grouped(jobs, nets[0].final_act if li == nL - 1 else ACT_NONE)
    for j, net in enumerate(nets):
        if net._bn_boxes:
            rl = []
            for li, L in enumerate(net.layers):
                if L.norm == "bn":
                    nb = net._bn_boxes[L.key]
                    _, _, _, ho, wo = geos[j][li]
                    rl.append((stats[j][li], nb.running_mean, nb.running_var, nb.num_batches_tracked, L.cout, ho * wo, L.cout_s))
            ops.bn_running_update(rl, BN_MOMENTUM)
    return outs, stats


def _grouped_backward(nets, xs, outs, stats, douts, need_dx, want_wgrad):
    dev = xs[0].device
    J = len(nets)
    geos = [n._geometry(x.shape[0], x.shape[1]) for n, x in zip(nets, xs)]
    nL = len(nets[0].layers)
    for j, net in enumerate(nets):
        if want_wgrad[j]:
            net._ensure_grads()
    dcur = list(douts)
    if nets[0].final_act == ACT_TANH:
        for j in range(J):
            d2 = torch.empty_like(outs[j][-1])
            ops.tanh_bwd(dcur[j].contiguous(), outs[j][-1], d2)
            dcur[j] = d2
    per_job = sum(2 * L.cout_s for L in nets[0].layers if L.norm)
    arena = stats[0][-1].take(max(per_job * J, 1))
    sums = []
    for j in range(J):
        sm, o = [], j * per_job
        for L in nets[j].layers:
            if L.norm:
                sm.append(arena[o: o + 2 * L.cout_s])
                o += 2 * L.cout_s
            else:
                sm.append(None)
        sums.append(sm)
    dxs = [None] * J
    for li in range(nL - 1, -1, -1):
        srcs = [outs[j][li - 1] if li > 0 else xs[j] for j in range(J)]
        norms = [nets[j]._norm_of(li - 1, stats[j], geos[j][li][1] * geos[j][li][2]) if li > 0 else None for j in range(J)]
        wj = [j for j in range(J) if want_wgrad[j]]
        if wj:
            ops.conv_wgrad_grouped([(geos[j][li][0], srcs[j], norms[j], dcur[j]) + nets[j]._gwb(nets[j].layers[li]) for j in wj])
        if li > 0:
            jobs, dins = [], []
            for j, net in enumerate(nets):
                Pv = net.layers[li - 1]
                desc, h, w, ho, wo = geos[j][li]
                din = torch.empty((h, w, Pv.cout_s), dtype=torch.float32, device=dev)
                dins.append(din)
                jobs.append((desc, dcur[j], net._wt(net.layers[li]), din, srcs[j], norms[j], sums[j][li - 1], 0, False, True))
            with ops.math_scope(_dgrad_math(nets[0].layers[li - 1])):
                ops.conv_dgrad_grouped(jobs)
            nb = []
            for j, net in enumerate(nets):
                Pv = net.layers[li - 1]
                if Pv.norm:
                    bn = Pv.norm == "bn" and want_wgrad[j]
                    dg = net._gflat[Pv.g_off: Pv.g_off + Pv.cout_s] if bn else None
                    db = net._gflat[Pv.be_off: Pv.be_off + Pv.cout_s] if bn else None
                    nb.append((dins[j], srcs[j], norms[j], sums[j][li - 1], dg, db))
                dcur[j] = dins[j]
            if nb:
                ops.norm_bwd_apply_multi(nb)
        else:
            dj = [j for j in range(J) if need_dx[j]]
            if dj:
                jobs = []
                for j in dj:
                    L = nets[j].layers[0]
                    desc, h, w, ho, wo = geos[j][0]
                    dxs[j] = torch.empty((h, w, L.cin_s), dtype=torch.float32, device=dev)
                    jobs.append((desc, dcur[j], nets[j]._wt(L), dxs[j], None, None, None, 0, False, True))
                ops.conv_dgrad_grouped(jobs)
    return dxs


class _MultiChainFn(torch.autograd.Function):
    """One autograd node for several network calls that share an architecture (the three discriminators on
    the fake and the real batch): each layer of all of them is ONE kernel launch."""

    @staticmethod
    def forward(ctx, nets, *tensors):
        J = len(nets)
        xlog = tensors[:J]
        memo = {}       # the same image goes to several discriminators: one layout conversion ...
        gauss = []      # ... and one launch for all their Gaussian pre-filters
        xbs = [net._prepare_input(x, memo, gauss) if hasattr(net, "scale_factor") else net._prepare_input(x, memo)
               for net, x in zip(nets, xlog)]
        for creal in sorted({c for c, _ in gauss}):
            ops.gauss_down_multi_fwd([job for c, job in gauss if c == creal], creal)
        outs, stats = _grouped_forward(nets, [xb["chain_in"] for xb in xbs])
        ctx.nets, ctx.xbs, ctx.outs, ctx.stats = nets, xbs, outs, stats
        ctx.in_keys = [(x.data_ptr(), tuple(x.shape), tuple(x.stride())) for x in xlog]
        ctx.need_dx = [bool(ctx.needs_input_grad[1 + j]) for j in range(J)]
        any_param = any(ctx.needs_input_grad[1 + J:])
        ctx.want_wgrad = [net.compute_param_grads and any_param for net in nets]
        return tuple(ops.logical_view(outs[j][-1], nets[j].layers[-1].cout) for j in range(J))

    @staticmethod
    def backward(ctx, *gouts):
        nets = ctx.nets
        J = len(nets)
        douts = []
        for j in range(J):
            g = gouts[j]
            if g is None:
                g = torch.zeros_like(ops.logical_view(ctx.outs[j][-1], nets[j].layers[-1].cout))
            douts.append(ops.as_nhwc(g))
        dch = _grouped_backward(nets, [xb["chain_in"] for xb in ctx.xbs], ctx.outs, ctx.stats, douts, ctx.need_dx, ctx.want_wgrad)
        # discriminators fed with the same image (the multi-scale set on `fake`) share one image-gradient buffer: the
        # scale-1 chain's backward-data wrote it, the pre-filter backward of all the others adds into it in one launch,
        # and autograd is handed one gradient and Nones -- no gradient-accumulation kernels afterwards
        dxs, groups = [None] * J, {}
        for j in range(J):
            if not ctx.need_dx[j]:
                continue
            if hasattr(nets[j], "scale_factor"):
                groups.setdefault(ctx.in_keys[j], []).append(j)
            else:
                dxs[j] = nets[j]._finish_input_grad(ctx.xbs[j], dch[j])
        for js in groups.values():
            ones = [j for j in js if nets[j].scale_factor == 1]
            downs = [j for j in js if nets[j].scale_factor > 1]
            nc = nets[js[0]].input_nc
            if ones:
                buf = dch[ones[0]]
                for j in ones[1:]:
                    buf.add_(dch[j])
            else:
                buf = torch.empty_like(ctx.xbs[js[0]]["img"])
            if downs:
                jobs = []
                for j in downs:
                    wg, gcs, kg, padg = nets[j]._gauss_args()
                    jobs.append((buf, dch[j], wg, gcs, kg, padg, nets[j].scale_factor))
                ops.gauss_down_multi_bwd(jobs, nc, accumulate=bool(ones))
            dxs[js[0]] = ops.logical_view(buf, nc)
        return (None,) + tuple(dxs) + (None,) * (len(ctx.needs_input_grad) - 1 - J)


def multi_forward(jobs):
    """[(net, x)] -> [net.forward(x)].  Jobs are partitioned into sets of same-architecture nets (<= 8 each) and every
    set runs with one kernel launch per layer; a net alone in its set is called on its own."""
    jobs = list(jobs)
    groups = []
    for idx, (n, _) in enumerate(jobs):
        for grp in groups:
            if len(grp) < 8 and _same_architecture(jobs[grp[0]][0], n):
                grp.append(idx)
                break
        else:
            groups.append([idx])
    results = [None] * len(jobs)
    for grp in groups:
        if len(grp) == 1:
            n, x = jobs[grp[0]]
            results[grp[0]] = n.forward(x)
            continue
        nets = [jobs[i][0] for i in grp]
        params, seen = [], set()
        for n in nets:
            if id(n) not in seen:
                seen.add(id(n))
                params += list(n.model.parameters())
        outs = _MultiChainFn.apply(nets, *[jobs[i][1] for i in grp], *params)
        for i, n, o in zip(grp, nets, outs):
            results[i] = n._wrap_output(o)
    return results


class FCGANGenerator(ChainNet):
    """FCGANGenerator (models/networks.py:493-540): ConvT(k4,s2,p1) -> BatchNorm -> ReLU x n_layers,
    ConvT -> Tanh.  `norm_layer` is hard-wired to BatchNorm by define_G (models/networks.py:87) and
    the net never leaves train mode."""
    final_act = ACT_TANH

    def __init__(self, noise_nc, input_nc, ngf=64, n_layers=3, use_dropout=False, use_fcn=False, gpu_ids=[]):
        if use_dropout:
            raise NotImplementedError("FCGANGenerator dropout is not on the MI355X path (README uses --no_dropout)")
        layers = []
        nf = min(2 ** (n_layers - 1), 8)
        # --noiseSize 1 (use_fcn False): the first ConvT is k4 s1 p0 and turns the 1x1 latent into a 4x4 map (:503-504)
        layers.append(LayerSpec("0", CONVT, 4, 2 if use_fcn else 1, 1 if use_fcn else 0, noise_nc, ngf * nf, False, "bn", ACT_RELU))
        idx = 3
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(2 ** (n_layers - n - 1), 8)
            layers.append(LayerSpec(str(idx), CONVT, 4, 2, 1, ngf * nf_prev, ngf * nf, True, "bn", ACT_RELU))
            idx += 3
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


class AutoEncoder(ChainNet):
    """AutoEncoder (models/networks.py:421-490): Conv(k4,s2,p1)+norm+ReLU x n_layers, a bias-free latent Conv with nothing after
    it, then ConvT(k4,s2,p1)+norm+ReLU x n_layers and a bias-free ConvT -> Tanh; with `use_dropout` every block but the first of
    each half has nn.Dropout(0.2) (encoder) / nn.Dropout(0.5) (decoder) between its norm and its ReLU (ChainNet's `drop`)."""
    final_act = ACT_TANH

    def __init__(self, input_nc, output_nc, n_layers=3, ngf=64, norm="batch", use_dropout=False, gpu_ids=[]):
        nrm = {"instance": "in", "batch": "bn"}[norm]
        if use_dropout and nrm != "in":
            raise NotImplementedError("AutoEncoder dropout on the MI355X path implements --norm instance (the masked tensor is "
                                      "materialised without an affine)")
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
        if norm != "instance":
            raise NotImplementedError("ResnetGenerator on the MI355X path implements --norm instance")
        if use_residual:
            raise NotImplementedError("ResnetGenerator --use_residual (tanh(x + y)) is not on the MI355X path")
        self.n_blocks, self.use_dropout, self.input_nc, self.output_nc = int(n_blocks), bool(use_dropout), input_nc, output_nc
        C = 4 * ngf
        self.c0 = LayerSpec("1", CONV, 7, 1, 0, input_nc, ngf, True, "in", ACT_RELU)
        self.d1 = LayerSpec("4", CONV, 3, 2, 1, ngf, 2 * ngf, True, "in", ACT_RELU)
        self.d2 = LayerSpec("7", CONV, 3, 2, 1, 2 * ngf, C, True, "in", ACT_RELU)
        second = 6 if use_dropout else 5
        self.blocks = [(LayerSpec("%d.conv_block.1" % (10 + i), CONV, 3, 1, 0, C, C, True, "in", ACT_RELU),
                        LayerSpec("%d.conv_block.%d" % (10 + i, second), CONV, 3, 1, 0, C, C, True, "in", ACT_NONE)) for i in range(n_blocks)]
        nb = 10 + n_blocks
        self.u1 = LayerSpec(str(nb), CONVT, 3, 2, 1, C, 2 * ngf, True, "in", ACT_RELU)
        self.u2 = LayerSpec(str(nb + 3), CONVT, 3, 2, 1, 2 * ngf, ngf, True, "in", ACT_RELU)
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
        return torch.tanh(_ChainFn.apply(self, x, *list(self.model.parameters())))

    def _wrap_output(self, y):
        return y

    # ---- geometry ------------------------------------------------------------------------------
    def _desc(self, L, hin, win, hout, wout):
        key = (L.key, hin, win)
        if key not in self._geom_cache:
            self._geom_cache[key] = ops.conv_desc(L.kind, L.k, L.stride, L.pad, hin, win, L.cin_s, hout, wout, L.cout_s, L.cin, L.cout)
        return self._geom_cache[key]

    def _in(self, st, count, act):
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
        ops.conv_fwd(self._desc(self.d1, H, W, h2, w2), c0, self._in(st[self.c0.key], H * W, ACT_RELU), *self._wb(self.d1), d1, ACT_NONE, st[self.d1.key])
        d2 = E(h4, w4, C)
        ops.conv_fwd(self._desc(self.d2, h2, w2, h4, w4), d1, self._in(st[self.d1.key], h2 * w2, ACT_RELU), *self._wb(self.d2), d2, ACT_NONE, st[self.d2.key])
        b = E(h4, w4, C)
        ops.pad_reflect_fwd(d2, self._in(st[self.d2.key], h4 * w4, ACT_RELU), 0, b)
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
            ops.pad_reflect_fwd(a, self._in(st[A.key], h4 * w4, ACT_RELU), 1, p2, mask)
            c = E(h4, w4, C)
            ops.conv_fwd(d3, p2, None, *self._wb(B), c, ACT_NONE, st[B.key])
            bn = E(h4, w4, C)
            ops.norm_apply_fwd(c, self._in(st[B.key], h4 * w4, ACT_NONE), bn, None, b, 1.0)      # x + IN(conv)
            blk.append((p1, a, mask, p2, c))
            b = bn
        if self.use_dropout and getattr(self, "mask_source", None) is None and self.blocks:
            ops.rng_advance(self._rng_offset, (h4 * w4 * C + 3) // 4)
        u1 = E(h2, w2, self.u1.cout_s)
        ops.conv_fwd(self._desc(self.u1, h4, w4, h2, w2), b, None, *self._wb(self.u1), u1, ACT_NONE, st[self.u1.key])
        u2 = E(H, W, ngf)
        ops.conv_fwd(self._desc(self.u2, h2, w2, H, W), u1, self._in(st[self.u1.key], h2 * w2, ACT_RELU), *self._wb(self.u2), u2, ACT_NONE, st[self.u2.key])
        pl = E(H + 6, W + 6, ngf)
        ops.pad_reflect_fwd(u2, self._in(st[self.u2.key], H * W, ACT_RELU), 3, pl)
        y = E(H, W, self.cl.cout_s)
        ops.conv_fwd(self._desc(self.cl, H + 6, W + 6, H, W), pl, None, *self._wb(self.cl), y, final_act, None)
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

        def norm_bwd(d, xraw, L, count, act):
            ops.norm_bwd_apply(d, xraw, self._in(st[L.key], count, act), sums[L.key])

        # last conv (k7 over the padded, activated u2)
        dcl = self._desc(self.cl, H + 6, W + 6, H, W)
        wgrad(self.cl, dcl, S["pl"], None, dy)
        dpl = E(H + 6, W + 6, ngf)
        ops.conv_dgrad(dcl, dy, self._wt(self.cl), dpl, None, None, None, w_transposed=True)
        du2 = E(H, W, ngf)
        ops.pad_reflect_bwd(dpl, 3, du2, S["u2"], self._in(st[self.u2.key], H * W, ACT_RELU), None, sums[self.u2.key])
        norm_bwd(du2, S["u2"], self.u2, H * W, ACT_RELU)
        # the two transposed convs
        n_u1 = self._in(st[self.u1.key], h2 * w2, ACT_RELU)
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
            n_c = self._in(st[B.key], h4 * w4, ACT_NONE)
            ops.norm_apply_bwd_sums(dc, c, n_c, sums[B.key])
            ops.norm_bwd_apply(dc, c, n_c, sums[B.key])
            wgrad(B, d3, p2, None, dc)
            dp2 = E(h4 + 2, w4 + 2, C)
            ops.conv_dgrad(d3, dc, self._wt(B), dp2, None, None, None, w_transposed=True)
            da = E(h4, w4, C)
            ops.pad_reflect_bwd(dp2, 1, da, a, self._in(st[A.key], h4 * w4, ACT_RELU), mask, sums[A.key])
            norm_bwd(da, a, A, h4 * w4, ACT_RELU)
            wgrad(A, d3, p1, None, da)
            dp1 = E(h4 + 2, w4 + 2, C)
            ops.conv_dgrad(d3, da, self._wt(A), dp1, None, None, None, w_transposed=True)
            dbi = E(h4, w4, C)
            ops.pad_reflect_bwd(dp1, 1, dbi)
            db.add_(dbi)
        # block input = relu(IN(d2)), materialised with pad 0
        dd2 = E(h4, w4, C)
        ops.pad_reflect_bwd(db, 0, dd2, S["d2"], self._in(st[self.d2.key], h4 * w4, ACT_RELU), None, sums[self.d2.key])
        norm_bwd(dd2, S["d2"], self.d2, h4 * w4, ACT_RELU)
        n_d1 = self._in(st[self.d1.key], h2 * w2, ACT_RELU)
        dd = self._desc(self.d2, h2, w2, h4, w4)
        wgrad(self.d2, dd, S["d1"], n_d1, dd2)
        dd1 = E(h2, w2, self.d1.cout_s)
        ops.conv_dgrad(dd, dd2, self._wt(self.d2), dd1, S["d1"], n_d1, sums[self.d1.key], w_transposed=True)
        norm_bwd(dd1, S["d1"], self.d1, h2 * w2, ACT_RELU)
        n_c0 = self._in(st[self.c0.key], H * W, ACT_RELU)
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
        if norm != "instance":
            raise NotImplementedError("UnetGenerator on the MI355X path implements --norm instance (the reference default)")
        if use_residual:
            raise NotImplementedError("UnetGenerator --use_residual is not on the MI355X path")
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
                d = LayerSpec(dk, CONV, 4, 2, 1, c[l - 1], c[l], True, None if inner else "in", ACT_NONE)
                u = LayerSpec(uk, CONVT, 4, 2, 1, c[l] if inner else c[l] * (2 if skip[l + 1] else 1), c[l - 1], True, "in", ACT_NONE)
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

    def _x_norm(self, l, hw, xstat, act, slope=0.0):
        """How a consumer reads x_l from its raw conv output."""
        if l == 0 or l == self.n - 1:
            return ops.norm_desc(None, None, None, 1, 0.0, act, slope)
        st, sq = xstat[l]
        return ops.norm_desc(st, None, None, hw[l][0] * hw[l][1], IN_EPS, act, slope, sq)

    def _cat_norm(self, l, hw, catstat):
        """ReLU(cat_l) as read by up[l-1]: identity for y_l (and x_0), InstanceNorm statistics for x_{l-1}."""
        if l == 1 or not self.skip[l]:
            return ops.norm_desc(None, None, None, 1, 0.0, ACT_RELU, 0.0)
        return ops.norm_desc(catstat[l], None, None, hw[l - 1][0] * hw[l - 1][1], IN_EPS, ACT_RELU, 0.0, 0)

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
                src, in_norm = cat[l + 1], self._cat_norm(l + 1, hw, catstat)
            u[l] = torch.empty(hw[l - 1] + (c[l - 1],), dtype=torch.float32, device=dev)
            ops.conv_fwd(upd[l], src, in_norm, wt, b, u[l], ACT_NONE, ustat[l])
            mask, noise = self._random(l, u[l].shape, dev)
            masks[l] = mask
            un = ops.norm_desc(ustat[l], None, None, hw[l - 1][0] * hw[l - 1][1], IN_EPS, ACT_NONE, 0.0)
            ops.norm_apply_fwd(u[l], un, cat[l][:, :, :c[l - 1]], mask, noise, self.gauss_sigma if noise is not None else 0.0)
        L = self.up[0]
        wt, b = self._wb(L)
        out = torch.empty((H, W, L.cout_s), dtype=torch.float32, device=dev)
        final_act = self._take_call_act()
        ops.conv_fwd(upd[0], cat[1], self._cat_norm(1, hw, catstat), wt, b, out, final_act, None)
        if self._rng_drawn:
            ops.rng_advance(self._rng_offset, self._rng_drawn)
        saved = dict(final_act=final_act, x=x, hw=hw, cat=cat, catw=catw, catstat=catstat, ustat=ustat, xr=xr, xstat=xstat, u=u, masks=masks,
                     out=out, lay=lay, total=total, bwd=_BwdArena(arena[total:]))
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

        # final transposed conv: gradient of ReLU(cat_1)
        nrm = self._cat_norm(1, hw, catstat)
        wgrad(self.up[0], upd[0], cat[1], nrm, d0)
        ops.conv_dgrad(upd[0], d0, self._wt(self.up[0]), dcat[1], cat[1], nrm, None, w_transposed=True)
        # decoder, outermost block first
        d_inner = None
        for l in range(1, n):
            dy = dcat[l][:, :, :c[l - 1]]
            un = ops.norm_desc(ustat[l], None, None, hw[l - 1][0] * hw[l - 1][1], IN_EPS, ACT_NONE, 0.0)
            ops.norm_apply_bwd_sums(dy, u[l], un, usum[l], masks[l])
            ops.norm_bwd_apply(dy, u[l], un, usum[l])                       # dy is now d(up[l] output)
            if l == n - 1:
                src, nrm = xr[l], ops.norm_desc(None, None, None, 1, 0.0, ACT_RELU, 0.0)
                d_inner = torch.empty(hw[l] + (c[l],), dtype=torch.float32, device=dev)
                din, sums = d_inner, None
            else:
                src, nrm = cat[l + 1], self._cat_norm(l + 1, hw, catstat)
                din = dcat[l + 1]
                sums = csum[l + 1] if (skip[l + 1] and l + 1 > 1) else None
            wgrad(self.up[l], upd[l], src, nrm, dy)
            ops.conv_dgrad(upd[l], dy, self._wt(self.up[l]), din, src, nrm, sums, w_transposed=True)
        # encoder, innermost first: dr = gradient w.r.t. the raw output of down[l]
        dr = d_inner
        for l in range(n - 1, 0, -1):
            src = xr[l - 1]
            nrm = self._x_norm(l - 1, hw, xstat, ACT_LRELU, 0.2)
            wgrad(self.down[l], dn[l], src, nrm, dr)
            normed = 1 <= l - 1 <= n - 2
            sums, sq = xsum[l - 1] if normed else (None, 0)
            if skip[l]:
                din = dcat[l][:, :, c[l - 1]:]
                ops.conv_dgrad(dn[l], dr, self._wt(self.down[l]), din, src, nrm, sums, sq, accumulate=True, w_transposed=True)
            else:
                din = torch.empty(hw[l - 1] + (c[l - 1],), dtype=torch.float32, device=dev)
                ops.conv_dgrad(dn[l], dr, self._wt(self.down[l]), din, src, nrm, sums, sq, w_transposed=True)
            if normed:
                ops.norm_bwd_apply(din, src, nrm, sums, None, None, sq)
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
        if norm != "instance":
            raise NotImplementedError("CascadedRefinementNetwork on the MI355X path implements --norm instance")
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
                u = LayerSpec(f"blockh{s}.0.model.0", CONVT, 4, 2, 1, cin, ngf, False, "in", ACT_NONE)
            else:
                u = LayerSpec(f"blockh{s}.0.model.0", CONV, 3, 1, 1, cin, ngf, True, "in", ACT_NONE)
            self.up[s] = u
            layers.append(u)
            self.inter[s] = []
            for i in range(n_layers_block):
                last = s == 0 and i == n_layers_block - 1
                L = LayerSpec(f"blockh{s}.1.model.{3 * i + 1}", CONV, 3, 1, 1, ngf, output_nc if last else ngf, True,
                              None if last else "in", ACT_NONE)
                self.inter[s].append(L)
                layers.append(L)
        if share_label_weights:
            L = LayerSpec("blockl.0", CONV, 3, 1, 1, input_nc, ngf, True, "in", ACT_NONE)
            layers.append(L)
            for s in range(5):
                self.lab[s] = L
        else:
            for s in range(4, -1, -1):
                self.lab[s] = LayerSpec(f"blockl{s}.0", CONV, 3, 1, 1, input_nc, ngf, True, "in", ACT_NONE)
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
        saved = dict(final_act=final_act, label=label, first=x["first"], lv=lv, cat=cat, c={}, u={}, un={}, t={}, arena=arena, lay=lay, off=off, res=res)
        out = None
        drawn = 0
        for s in range(5, -1, -1):
            h, w = res[s]
            U = self.up[s]
            wt, b = self._wb(U)
            src = x["first"] if s == 5 else cat[s]
            nrm = None if s == 5 else ops.norm_desc(st(("cat", s), 2 * C2), None, None, h * w, IN_EPS, ACT_NONE, 0.0)
            u = torch.empty((2 * h, 2 * w, ngf), dtype=torch.float32, device=dev)
            ustat = st(("u", s), 2 * ngf)
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
                ops.norm_apply_fwd(u, ops.norm_desc(ustat, None, None, 4 * h * w, IN_EPS, ACT_NONE, 0.0), tn, None, nz, self.gauss_sigma)
                saved["un"][s] = tn
                cur, cur_stat = tn, None
            for i, L in enumerate(self.inter[s]):
                wt, b = self._wb(L)
                nrm = ops.norm_desc(cur_stat, None, None, 4 * h * w, IN_EPS, ACT_RELU, 0.0)
                last_i = i == nlb - 1
                if last_i and s == 0:
                    out = torch.empty((2 * h, 2 * w, L.cout_s), dtype=torch.float32, device=dev)
                    ops.conv_fwd(self._desc(L, 2 * h, 2 * w), cur, nrm, wt, b, out, final_act, None)
                elif last_i:   # feeds the next stage: right half of its concat buffer, statistics into the matching slice
                    dst = cat[s - 1][:, :, ngf:]
                    ops.conv_fwd(self._desc(L, 2 * h, 2 * w), cur, nrm, wt, b, dst, ACT_NONE, st(("cat", s - 1), 2 * C2)[ngf:], C2)
                else:
                    t = torch.empty((2 * h, 2 * w, ngf), dtype=torch.float32, device=dev)
                    tstat = st(("t", s, i), 2 * ngf)
                    ops.conv_fwd(self._desc(L, 2 * h, 2 * w), cur, nrm, wt, b, t, ACT_NONE, tstat)
                    saved["t"][(s, i)] = t
                    cur, cur_stat = t, tstat
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
                nrm = ops.norm_desc(sstat, None, None, 4 * h * w, IN_EPS, ACT_RELU, 0.0)
                desc = self._desc(L, 2 * h, 2 * w)
                wgrad(L, desc, src, nrm, d)
                din = torch.empty((2 * h, 2 * w, ngf), dtype=torch.float32, device=dev)
                ops.conv_dgrad(desc, d, self._wt(L), din, src, nrm, None if noisy else ssum, w_transposed=True)
                if noisy:      # din = d t (the noise has no gradient): sums of the norm backward, then the norm backward itself
                    unrm = ops.norm_desc(ustat, None, None, 4 * h * w, IN_EPS, ACT_NONE, 0.0)
                    ops.norm_apply_bwd_sums(din, u, unrm, ssum, None)
                    ops.norm_bwd_apply(din, u, unrm, ssum)
                else:
                    ops.norm_bwd_apply(din, src, nrm, ssum)
                d = din
            # d = gradient w.r.t. u_s (raw, before its InstanceNorm)
            U = self.up[s]
            desc = self._desc(U, h, w)
            if self.mode == 'bilinear':
                dc = torch.empty((h, w, ngf), dtype=torch.float32, device=dev)
                ops.bilinear_up2_bwd(d, dc)
                d = dc
            src = S["first"] if s == 5 else cat[s]
            nrm = None if s == 5 else ops.norm_desc(st(("cat", s), 2 * C2), None, None, h * w, IN_EPS, ACT_NONE, 0.0)
            wgrad(U, desc, src, nrm, d)
            if s == 5:
                if need_dx:
                    dfirst = torch.empty_like(S["first"])
                    ops.conv_dgrad(desc, d, self._wt(U), dfirst, None, None, None, w_transposed=True)
                break
            dc_ = torch.empty_like(cat[s])
            csum = sm(("cat", s), 2 * C2)
            ops.conv_dgrad(desc, d, self._wt(U), dc_, cat[s], nrm, csum, w_transposed=True)
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
        return total, each

    @staticmethod
    def backward(ctx, gtotal, _geach):
        ds = ctx.ds
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


class GANLossMultiClass(nn.Module):
    """GANLossMultiClass (models/networks.py:188-202): CrossEntropyLoss over the class channel of every pixel of a
    discriminator map.  The maps are 3 x 67 x 67: the loss runs on PyTorch's own kernels."""

    def __init__(self, use_lsgan=False, num_classes=3, use_gpu=False):
        super().__init__()
        assert use_lsgan is False
        self.num_classes = num_classes

    def __call__(self, input, target_label):
        flat = input.permute(0, 2, 3, 1).reshape(-1, self.num_classes)
        tgt = getattr(self, "_tgt", None)
        if tgt is None or tgt.device != flat.device or tgt.shape[1] != flat.shape[0]:
            tgt = self._tgt = torch.arange(self.num_classes, device=flat.device).view(-1, 1).expand(-1, flat.shape[0]).contiguous()
        return F.cross_entropy(flat, tgt[int(target_label)])


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


# ------------------------------------------------------------------------------------------------
# factories (models/networks.py:53-132)
# ------------------------------------------------------------------------------------------------
def define_G(input_nc, output_nc, ngf, which_model_netG, norm='batch', use_dropout=False, n_layers_G=5,
             use_residual=False, use_fcn=False, noise_nc=0, add_gaussian_noise=False, gaussian_sigma=0.1,
             n_layers_G_skip=-1, upsample_mode='convt', share_label_weights=True, n_layers_CRN_block=1, gpu_ids=[]):
    if which_model_netG in ('fcgan', 'deconv'):   # README spells it `deconv` (README.md:33)
        netG = FCGANGenerator(noise_nc, input_nc, ngf, n_layers=n_layers_G, use_dropout=use_dropout, use_fcn=use_fcn,
                              gpu_ids=gpu_ids)
    elif which_model_netG in ('unet_128', 'unet_256'):
        netG = UnetGenerator(input_nc, output_nc, 7 if which_model_netG == 'unet_128' else 8, ngf, norm=norm,
                             use_dropout=use_dropout, use_residual=use_residual, add_gaussian_noise=add_gaussian_noise,
                             gaussian_sigma=gaussian_sigma, num_skips=n_layers_G_skip, gpu_ids=gpu_ids)
    elif which_model_netG == 'crn':
        netG = CascadedRefinementNetwork(input_nc, output_nc, noise_nc, ngf=ngf, n_layers=n_layers_G, norm=norm,
                                         upsample_mode=upsample_mode, add_gaussian_noise=add_gaussian_noise,
                                         gaussian_sigma=gaussian_sigma, share_label_weights=share_label_weights,
                                         n_layers_block=n_layers_CRN_block, gpu_ids=gpu_ids)
    elif which_model_netG == 'autoencoder':
        netG = AutoEncoder(input_nc, output_nc, n_layers_G, ngf, norm=norm, use_dropout=use_dropout, gpu_ids=gpu_ids)
    elif which_model_netG == 'dcgan':
        netG = DCGANGenerator(gpu_ids=gpu_ids, nz=noise_nc, nc=input_nc, ngf=ngf)
    elif which_model_netG == 'fcgan_star':
        netG = FCGANGeneratorStar(noise_nc, input_nc, ngf, n_layers=n_layers_G, use_dropout=use_dropout, use_fcn=use_fcn,
                                  gpu_ids=gpu_ids)
    elif which_model_netG in ('resnet_9blocks', 'resnet_6blocks'):
        netG = ResnetGenerator(input_nc, output_nc, ngf, norm=norm, use_dropout=use_dropout, n_blocks=9 if which_model_netG == 'resnet_9blocks' else 6,
                               use_residual=use_residual, gpu_ids=gpu_ids)
    else:
        raise NotImplementedError('Generator model name [%s] is not recognized' % which_model_netG)
    netG.apply(weights_init)
    if len(gpu_ids) > 0:
        netG.cuda(gpu_ids[0])
    return netG


def define_D(input_nc, ndf, which_model_netD, n_layers_D=3, norm='batch', use_sigmoid=False, scale_factor=1,
             num_classes=2, gpu_ids=[]):
    if which_model_netD == 'basic':
        netD = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm=norm, use_sigmoid=use_sigmoid,
                                   scale_factor=scale_factor, num_classes=num_classes, gpu_ids=gpu_ids)
    elif which_model_netD == 'n_layers':
        netD = NLayerDiscriminator(input_nc, ndf, n_layers=n_layers_D, norm=norm, use_sigmoid=use_sigmoid,
                                   scale_factor=scale_factor, num_classes=num_classes, gpu_ids=gpu_ids)
    elif which_model_netD == 'dcgan':
        if scale_factor > 1:
            raise NotImplementedError("the dcgan discriminator has no Gaussian pre-filter (the reference fails on scale_factor > 1 too)")
        netD = DCGANDiscriminator(gpu_ids=gpu_ids, nc=input_nc, ndf=ndf)
    elif which_model_netD == 'n_layers_sep':
        raise NotImplementedError('Discriminator model name [%s] is not on the MI355X path yet' % which_model_netD)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % which_model_netD)
    netD.apply(weights_init)
    if scale_factor > 1:
        for param in netD.gauss_filter.parameters():
            sigma = int(scale_factor) // 2
            kw = 4 * sigma + 1
            param.data = torch.FloatTensor(init_gauss_filters(input_nc, kw, sigma))
    if len(gpu_ids) > 0:
        netD.cuda(gpu_ids[0])
    return netD

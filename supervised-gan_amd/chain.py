"""The layer-program core of the MI355X networks: `LayerSpec`, `ChainNet` (flat parameter / gradient buffers with the reference's
state_dict keys as strided views, derived weight copies, forward / backward programs over NHWC buffers), its autograd node, and the
grouped execution of several same-architecture chains as one launch per layer (`multi_forward`, `pack_flat`).  Split out of
networks.py, which re-exports everything here."""
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
    norm_key: Optional[str] = None   # state-dict prefix of the BatchNorm when it is not the next numbered child (CRN bilinear upsample block)
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

    def __init__(self, buf, rep=0, dev=None):
        self.buf, self.dev = buf, (buf.device if buf is not None else dev)      # buf None: every backward takes its sums from the step's pool
        self.rep = rep           # replica stride of the forward statistics AND of `buf` (they share one ops.stat_arena)
        self.rep_bwd = rep       # replica stride of what take() handed out last

    def take(self, n):
        buf, self.buf = self.buf, None
        if buf is None or buf.numel() < n:
            fresh = ops.stat_arena(n, self.dev)
            self.rep_bwd = ops.stat_rep(fresh)
            return fresh
        self.rep_bwd = self.rep
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
                parts = L.key.split(".")      # the norm is the next numbered child of the same nn.Sequential (unless the layer says otherwise)
                self._add_box(L.norm_key or ".".join(parts[:-1] + [str(int(parts[-1]) + 1)]), nb)
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
        return ops.with_packed(self._flat_t[L.w_off: L.w_off + n], self._pk_b[L.w_off: L.w_off + n], self._pk_bh[L.w_off: L.w_off + n])

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
            self._pk_bh = torch.zeros_like(self._flat)
        ops.pack_weights(self._flat, self._flat_t, self._pk_f, self._pk_b, self._conv_segments(), self._pk_bh)
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
            der = arena_p._sgan_derived = tuple(torch.zeros_like(arena_p) for _ in range(4))
        segs, stale = [], []
        for n in mates:
            off = n._arena[2]
            if getattr(n, "_flat_t", None) is None or n._flat_t.data_ptr() != der[0].data_ptr() + 4 * off:
                n._flat_t, n._pk_f, n._pk_b, n._pk_bh = (d[off: off + n._nflat] for d in der)
                n._wt_key = None
            key = n._derived_key()
            if n._wt_key != key:
                segs += n._conv_segments(off)
                stale.append((n, key))
        ops.pack_weights(arena_p, der[0], der[1], der[2], segs, der[3])
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
        rep = getattr(stats[-1], "rep", 0)      # the statistics live in an ops.stat_arena (replicated sums)
        if L.norm == "bn":
            g = self._flat[L.g_off: L.g_off + L.cout_s]
            be = self._flat[L.be_off: L.be_off + L.cout_s]
            return ops.norm_desc(st, g, be, count, BN_EPS, L.act, L.slope, 0, rep)
        return ops.norm_desc(st, None, None, count, IN_EPS, L.act, L.slope, 0, rep)

    def _norm_in(self, li, stats, count, drop):
        """_norm_of for a consumer that may read the materialised dropout tensor of layer li: only the activation is left to apply."""
        if li in drop:
            L = self.layers[li]
            return ops.norm_desc(None, None, None, count, 0.0, L.act, L.slope)
        return self._norm_of(li, stats, count)

    # ---- forward / backward programs ----------------------------------------------------------
    def _kept_arena(self, n_stats, device, zeroed=False):
        """Forward statistics of a KEPT forward (run_forward(keep=True)): their own allocation in the replicated layout of
        ops.stat_arena, outside the step's arena pool -- begin_step() of the NEXT step must not zero them (forward_pair).
        zeroed: the caller has cleared `self._kept_full` on this stream already."""
        full = getattr(self, "_kept_full", None)
        reps = ops.stat_replicas() if ops._STAT_REPLICATED else 1
        if full is None or full.numel() != reps * max(n_stats, 1) or full.device != device:
            assert not zeroed
            full = self._kept_full = torch.zeros(reps * max(n_stats, 1), dtype=torch.float64, device=device)
        elif not zeroed:
            ops.zero_multi([full])
        return full[:max(n_stats, 1)]

    def run_forward(self, x: torch.Tensor, update_running=True, keep=False):
        """x: [H, W, Cs] NHWC buffer.  Returns (outs, stats): raw conv outputs and per-layer stats.
        keep: remember this call's buffers as `self._kept` (input, outputs, statistics in an allocation of their own, backward sums
        taken from the pool at backward time) so that forward_pair() can later write another forward of this net into them."""
        ops.require_gpu(x, type(self).__name__)
        if self._flat.device != x.device:
            raise SganError(f"module parameters are on {self._flat.device}, input on {x.device}")
        H, W, Cs = x.shape
        assert Cs == self.layers[0].cin_s, (Cs, self.layers[0].cin_s)
        geo = self._geometry(H, W)
        final_act = self._take_call_act()
        n_stats = sum(2 * L.cout_s for L in self.layers if L.norm)
        if keep:
            assert not any(L.drop > 0 for L in self.layers), "a kept forward has no dropout state"
            arena = self._kept_arena(n_stats, x.device)
        else:
            # one zero-fill serves the forward statistics and the backward sums (second half, consumed by run_backward)
            arena = ops.stat_arena(2 * n_stats, x.device)
        rep = ops.stat_rep(arena)
        stats, o = [], 0
        for L in self.layers:
            if L.norm:
                stats.append(arena[o: o + 2 * L.cout_s])
                o += 2 * L.cout_s
            else:
                stats.append(None)
        stats.append(_BwdArena(None, rep, x.device) if keep else _BwdArena(arena[n_stats:], rep))
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
            ops.conv_fwd(desc, cur, in_norm, wt, b, out, final_act if last else ACT_NONE, stats[li], 0, rep)
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
                if L.norm == "bn":      # BatchNorm -> Dropout (the fcgan generator's blocks): the affine goes into the materialised tensor
                    raw = ops.norm_desc(stats[li], self._flat[L.g_off: L.g_off + L.cout_s], self._flat[L.be_off: L.be_off + L.cout_s],
                                        ho * wo, BN_EPS, ACT_NONE, 0.0, 0, rep)
                else:
                    raw = ops.norm_desc(stats[li], None, None, ho * wo, IN_EPS, ACT_NONE, 0.0, 0, rep)
                ops.norm_apply_fwd(out, raw, t, mask)
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
                    rl.append((stats[li], nb.running_mean, nb.running_var, nb.num_batches_tracked, L.cout, ho * wo, L.cout_s, rep))
            ops.bn_running_update(rl, BN_MOMENTUM)
        if keep:
            self._kept = dict(x=x, outs=outs, stats=stats, n_stats=n_stats, final_act=final_act)
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
        brep = stats[-1].rep_bwd
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
            wjob = [(desc, src, in_norm, dcur) + self._gwb(L)] if want_wgrad else None
            if li > 0:
                P = self.layers[li - 1]
                din = torch.empty((h, w, P.cout_s), dtype=torch.float32, device=dev)
                djob = [(desc, dcur, self._wt(L), din, src, in_norm, None if dropped else sums[li - 1], 0, False, True, brep)]
                dm = _dgrad_math(P, [dcur])
                if wjob:
                    ops.conv_bwd_grouped(djob, wjob, dm)      # both halves in one launch where the fused kernel covers the layer
                else:
                    with ops.math_scope(dm):
                        ops.conv_dgrad_grouped(djob)
                if dropped:      # din = d t * ReLU'(t); through the mask, with the two norm-backward sums of the masked gradient
                    raw_norm = self._norm_of(li - 1, stats, h * w)
                    ops.norm_apply_bwd_sums(din, outs[li - 1], raw_norm, sums[li - 1], drop[li - 1][1])      # adds to the first copy only
                    dg = self._gflat[P.g_off: P.g_off + P.cout_s] if (P.norm == "bn" and want_wgrad) else None
                    db = self._gflat[P.be_off: P.be_off + P.cout_s] if (P.norm == "bn" and want_wgrad) else None
                    ops.norm_bwd_apply(din, outs[li - 1], raw_norm, sums[li - 1], dg, db, 0, brep)
                elif P.norm:
                    dg = self._gflat[P.g_off: P.g_off + P.cout_s] if (P.norm == "bn" and want_wgrad) else None
                    db = self._gflat[P.be_off: P.be_off + P.cout_s] if (P.norm == "bn" and want_wgrad) else None
                    ops.norm_bwd_apply(din, src, in_norm, sums[li - 1], dg, db, 0, brep, publish_amax=_wants_amax(self.layers, li - 1))
                dcur = din
                continue
            if wjob:
                ops.conv_wgrad_grouped(wjob)
            if need_dx:
                dx = torch.empty((h, w, L.cin_s), dtype=torch.float32, device=dev)
                ops.conv_dgrad(desc, dcur, self._wt(L), dx, None, None, None, w_transposed=True)
        return dx


def _wants_amax(layers, li):
    """Does the backward-data launch that READS the gradient of layer li's output need fp16 planes?  Only where bf16 planes are not
    enough: its result is the gradient of a layer WITHOUT normalisation (see _dgrad_math) -- for the PatchGAN chains, the launch
    into the first conv's output.  Everywhere else bf16 planes stay (measured: fp16 planes cost the fused backward launches ~11 %,
    138 -> 149 us on the six-problem 128 -> 256 launch, for an accuracy the norm backward does not need)."""
    return li >= 1 and layers[li - 1].norm is None


def _dgrad_math(P, douts=()):
    """Arithmetic of the backward-data launch whose result is the gradient of layer P's output.  Round 3: when every gradient tensor
    the launch reads carries its published maximum (ops.norm_bwd_apply_multi(publish_amax=True)) it runs on fp16 planes scaled by
    it -- 11 + 11 significant bits, an fp32-equivalent product -- and the rule below is not needed (returns None).  Otherwise:  Behind a normalisation the
    result goes through the norm backward, which re-centres it with sums taken from the very same values: the 5e-6 element errors
    of the split products stay 5e-6.  Without one (the first PatchGAN layer) the result is used as is, and the layer's bias
    gradient sums it over every pixel -- terms that cancel to a small residual (the gradient that reaches it left a normalisation
    as a zero-sum field) while unbiased element errors do not: measured 7e-3 of the bias gradient against the fp64 reference,
    where the reference's own fp32 is at 7e-6.  Those launches run on the exact-fp32 kernel (one per discriminator pass)."""
    if douts and all(ops.has_amax(d) for d in douts):
        return None
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
        keep, net._keep_next = getattr(net, "_keep_next", False), False      # one-shot: set by the caller right before forward()
        outs, stats = net.run_forward(xb["chain_in"], keep=True) if keep else net.run_forward(xb["chain_in"])
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
class _AdoptFn(_ChainFn):
    """Autograd node over the buffers of net._kept after forward_pair() has written a forward of `x_logical` into them: no kernel
    runs here, the backward is _ChainFn's."""

    @staticmethod
    def forward(ctx, net: "ChainNet", x_logical, *params):
        kept = net._kept
        xb = net._prepare_input(x_logical)
        assert xb["chain_in"].data_ptr() == kept["x"].data_ptr(), "forward_pair() wrote the kept forward from another input buffer"
        ctx.net, ctx.xb, ctx.outs, ctx.stats = net, xb, kept["outs"], kept["stats"]
        ctx.want_wgrad = net.compute_param_grads and any(ctx.needs_input_grad[2:])
        ctx.need_dx = ctx.needs_input_grad[1]
        return ops.logical_view(kept["outs"][-1], net.layers[-1].cout)


def forward_pair(net: "ChainNet", x_a, x_b, arena_zeroed=False):
    """(net(x_a), net(x_b)) with ONE launch per layer (the two calls see the same weights: e.g. the re-draw that ends a training step
    and the forward() that opens the next one, models/fcgan_model.py:178-193).  net(x_a) is computed without autograd.  net(x_b)'s
    outputs and statistics are written INTO the buffers of the net's kept forward (run_forward(keep=True); x_b must be that call's
    input buffer) and come back under a fresh autograd node: whatever was built on those buffers -- a hipGraph that captured the
    backward of that forward -- sees a new forward without one having run on its own.  BatchNorm running statistics are updated for
    x_a, then for x_b, as two calls in that order would.  arena_zeroed: the caller has already cleared net._kept_full on this stream
    (ops.normal_fill_nhwc_pair does it in the launch that draws the two latents)."""
    kept = net._kept
    xa = net._prepare_input(x_a.detach())["chain_in"]
    xb = net._prepare_input(x_b.detach())["chain_in"]
    assert xb.data_ptr() == kept["x"].data_ptr() and tuple(xa.shape) == tuple(xb.shape)
    arena = net._kept_arena(kept["n_stats"], xb.device, arena_zeroed)      # zeroed here (one launch) unless the caller did
    stats_b, o = [], 0
    for L in net.layers:
        if L.norm:
            stats_b.append(arena[o: o + 2 * L.cout_s])
            o += 2 * L.cout_s
        else:
            stats_b.append(None)
    stats_b.append(_BwdArena(None, ops.stat_rep(arena), xb.device))
    stats_b[-1].final_act = net.final_act
    stats_b[-1].drop = {}
    kept["stats"], kept["final_act"] = stats_b, net.final_act
    with torch.no_grad():
        outs, _ = _grouped_forward([net, net], [xa, xb], given=[None, (kept["outs"], stats_b)])
    ya = net._wrap_output(ops.logical_view(outs[0][-1], net.layers[-1].cout))
    yb = net._wrap_output(_AdoptFn.apply(net, x_b, *list(net.model.parameters())))
    return ya, yb


def _same_architecture(a: "ChainNet", b: "ChainNet") -> bool:
    if len(a.layers) != len(b.layers) or a.final_act != b.final_act or getattr(a, "no_group", False) or getattr(b, "no_group", False):
        return False
    key = lambda L: (L.kind, L.k, L.stride, L.pad, L.cin, L.cout, L.bias, L.norm, L.act, L.slope)
    return all(key(x) == key(y) for x, y in zip(a.layers, b.layers))


def can_group(nets) -> bool:
    nets = list(nets)
    return 1 < len(nets) <= 8 and all(_same_architecture(nets[0], n) for n in nets[1:])


def _grouped_forward(nets, xs, given=None):
    """nets[j] applied to xs[j] ([H,W,Cs] buffers); per layer ONE grouped launch.  Returns per-job (outs, stats).
    given[j] = (outs, stats) of job j supplied by the caller (forward_pair: the buffers of a kept forward, statistics zeroed), None = fresh."""
    dev = xs[0].device
    J = len(nets)
    given = given or [None] * J
    geos = [n._geometry(x.shape[0], x.shape[1]) for n, x in zip(nets, xs)]
    per_job = sum(2 * L.cout_s for L in nets[0].layers if L.norm)
    fresh = [j for j in range(J) if given[j] is None]
    arena = ops.stat_arena(2 * per_job * len(fresh), dev)   # forward statistics | backward sums, in replicated copies
    rep = ops.stat_rep(arena)
    bwd = _BwdArena(arena[per_job * len(fresh):], rep)
    stats = []
    for j in range(J):
        if given[j] is not None:
            stats.append(given[j][1])
            continue
        st, o = [], fresh.index(j) * per_job
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
            out = given[j][0][li] if given[j] is not None else torch.empty((ho, wo, L.cout_s), dtype=torch.float32, device=dev)
            assert tuple(out.shape) == (ho, wo, L.cout_s)
            in_norm = net._norm_of(li - 1, stats[j], h * w) if li > 0 else None
            wt, b = net._wb(L)
            jobs.append((desc, cur[j], in_norm, wt, b, out, stats[j][li], 0, stats[j][-1].rep))
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
                    rl.append((stats[j][li], nb.running_mean, nb.running_var, nb.num_batches_tracked, L.cout, ho * wo, L.cout_s, stats[j][-1].rep))
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
    brep = stats[0][-1].rep_bwd
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
        wjobs = [(geos[j][li][0], srcs[j], norms[j], dcur[j]) + nets[j]._gwb(nets[j].layers[li]) for j in wj]
        pair0 = li == 0 and bool(wjobs) and wj == [j for j in range(J) if need_dx[j]]      # first layer, both gradients of the same jobs:
        if wjobs and li == 0 and not pair0:                                                  # one launch below (ops.conv_bwd_grouped)
            ops.conv_wgrad_grouped(wjobs)
        if li > 0:
            jobs, dins = [], []
            for j, net in enumerate(nets):
                Pv = net.layers[li - 1]
                desc, h, w, ho, wo = geos[j][li]
                din = torch.empty((h, w, Pv.cout_s), dtype=torch.float32, device=dev)
                dins.append(din)
                jobs.append((desc, dcur[j], net._wt(net.layers[li]), din, srcs[j], norms[j], sums[j][li - 1], 0, False, True, brep))
            dm = _dgrad_math(nets[0].layers[li - 1], dcur)
            if wjobs:
                ops.conv_bwd_grouped(jobs, wjobs, dm)
            else:
                if wjobs:
                    ops.conv_wgrad_grouped(wjobs)
                with ops.math_scope(dm):
                    ops.conv_dgrad_grouped(jobs)
            nb = []
            for j, net in enumerate(nets):
                Pv = net.layers[li - 1]
                if Pv.norm:
                    bn = Pv.norm == "bn" and want_wgrad[j]
                    dg = net._gflat[Pv.g_off: Pv.g_off + Pv.cout_s] if bn else None
                    db = net._gflat[Pv.be_off: Pv.be_off + Pv.cout_s] if bn else None
                    nb.append((dins[j], srcs[j], norms[j], sums[j][li - 1], dg, db, 0, brep))
                dcur[j] = dins[j]
            if nb:
                ops.norm_bwd_apply_multi(nb, publish_amax=_wants_amax(nets[0].layers, li - 1))
        else:
            dj = [j for j in range(J) if need_dx[j]]
            if dj:
                jobs = []
                for j in dj:
                    L = nets[j].layers[0]
                    desc, h, w, ho, wo = geos[j][0]
                    dxs[j] = torch.empty((h, w, L.cin_s), dtype=torch.float32, device=dev)
                    jobs.append((desc, dcur[j], nets[j]._wt(L), dxs[j], None, None, None, 0, False, True))
                if pair0:
                    ops.conv_bwd_grouped(jobs, wjobs)
                else:
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
        # largest problem first: the workgroups of a grouped launch are dealt out problem by problem, and where their durations differ
        # (backward-weight pixel ranges: 68 / 37 / 11 chunks for the three PatchGAN scales) the long ones should not start last
        def _work(i):
            n_, x_ = jobs[i]
            sf = float(getattr(n_, "scale_factor", 1) or 1)
            return -(x_.shape[-1] * x_.shape[-2]) / (sf * sf)
        grp = sorted(grp, key=_work)      # stable: equal sizes keep the caller's order
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

"""Thin torch-tensor -> pointer wrappers over the C ABI (include/sgan_hip.h).

Activation tensors are `[H, W, Cs]` fp32 CUDA tensors (NHWC, batch 1, Cs = stored channels, a
multiple of 4; `tensor.stride(1)` is the pixel stride so channel slices of wider buffers work).
Everything here launches on torch's current stream and never synchronises."""
import ctypes as C
import weakref

import torch

from . import _lib as L
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, CONV, CONVT  # noqa: F401


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


# Arithmetic of the MFMA conv kernels (sgan_conv_desc.math): "bf16x3" = split-bf16 (fp32-equivalent, ~2^-16 per product,
# 16/3 of the fp32 matrix rate; the default), "f32" = exact fp32 MFMA (the parity mode).  SGAN_MATH selects at start-up,
# set_math() at run time (every conv call stamps the current mode into its descriptor).
_MATH_NAMES = {"f32": L.MATH_F32, "fp32": L.MATH_F32, "bf16x3": L.MATH_BF16X3}
_math = _MATH_NAMES[__import__("os").environ.get("SGAN_MATH", "bf16x3").lower()]


_DGRAD_MATH = _MATH_NAMES.get(__import__("os").environ.get("SGAN_DGRAD_MATH", "").lower())      # diagnostics: backward-data only


def set_math(name: str):
    global _math
    _math = _MATH_NAMES[name.lower()]


def get_math() -> str:
    return "bf16x3" if _math == L.MATH_BF16X3 else "f32"


class math_scope:
    """`with ops.math_scope("f32"):` -- the conv calls inside run in that arithmetic mode (None: leave the current one)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _math
        self.prev = _math
        if self.name is not None:
            _math = _MATH_NAMES[self.name.lower()]

    def __exit__(self, *exc):
        global _math
        _math = self.prev
        return False


def with_packed(w: torch.Tensor, packed, packed_f16=None) -> torch.Tensor:
    """Tag a weight slice with the matching slice of the split 16-bit copy (sgan_pack_weights) for the job builders; packed_f16:
    the fp16-plane twin of the backward copy (backward-data with a known gradient maximum reads it)."""
    w._sgan_pk = packed
    w._sgan_pk16 = packed_f16
    return w


def _amax(t):
    """Device scalar max|t| when the producer published one (norm_bwd_apply*), else None."""
    a = getattr(t, "_sgan_amax", None)
    return a.data_ptr() if (a is not None and _math == L.MATH_BF16X3) else None


def _pk16(w):
    pk = getattr(w, "_sgan_pk16", None)
    return pk.data_ptr() if (pk is not None and _math == L.MATH_BF16X3) else None


def _pk(w):
    pk = getattr(w, "_sgan_pk", None)
    return pk.data_ptr() if (pk is not None and _math == L.MATH_BF16X3) else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise L.SganError(f"{what}: supervised_gan_amd kernels run on an MI355X (gfx950) only; got a {t.device} tensor. "
                          "There is no CPU fallback -- move the module and its inputs to cuda.")


def _act(t):
    assert t.dim() == 3 and t.stride(2) == 1 and t.dtype == torch.float32, (t.shape, t.stride(), t.dtype)
    return t


def _avail(t):
    """fp32 elements between a tensor's first element and the end of its storage."""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


def _fits(t, H, W, Cs, what):
    """The C ABI takes raw pointers: it cannot see how much memory stands behind them, so the one place that can -- this wrapper --
    checks every conv operand against its descriptor BEFORE anything is launched: an [H][W][Cs] NHWC tensor with pixel stride
    stride(1) must lie inside its storage.  (Round 2's memory access fault was exactly this: a descriptor that had become legal
    described a larger layer than the tensors handed in with it)"""
    _act(t)
    if t.shape[0] * t.shape[1] != H * W or t.shape[2] < Cs or (H * W - 1) * t.stride(1) + Cs > _avail(t):
        raise L.SganError(f"{what}: tensor {tuple(t.shape)} (pixel stride {t.stride(1)}, {_avail(t)} elements of storage) does not "
                          f"match the descriptor's {H}x{W}x{Cs}; nothing was launched")
    return t


def _check_fwd_jobs(jobs):
    for job in jobs:
        desc, x, _, w, _, out = job[:6]
        _fits(x, desc.Hin, desc.Win, desc.Cin, "conv_fwd x")
        _fits(out, desc.Hout, desc.Wout, desc.Cout, "conv_fwd out")
        _fits_w(w, desc, "conv_fwd w")


def _check_dgrad_jobs(jobs):
    for job in jobs:
        desc, dout, w, din, x = job[:5]
        _fits(dout, desc.Hout, desc.Wout, desc.Cout, "conv_dgrad dout")
        _fits(din, desc.Hin, desc.Win, desc.Cin, "conv_dgrad din")
        _fits_w(w, desc, "conv_dgrad w")
        if x is not None:
            _fits(x, desc.Hin, desc.Win, desc.Cin, "conv_dgrad x")


def _check_wgrad_jobs(jobs):
    for desc, x, _, dout, dw, _ in jobs:
        _fits(x, desc.Hin, desc.Win, desc.Cin, "conv_wgrad x")
        _fits(dout, desc.Hout, desc.Wout, desc.Cout, "conv_wgrad dout")
        _fits_w(dw, desc, "conv_wgrad dw")


def _fits_w(w, desc, what):
    need = desc.k * desc.k * desc.Cin * desc.Cout
    if w is None or _avail(w) < need:
        raise L.SganError(f"{what}: weight tensor holds {0 if w is None else _avail(w)} elements, the descriptor's k{desc.k} "
                          f"{desc.Cin}->{desc.Cout} layer needs {need}; nothing was launched")
    return w


def conv_desc(kind, k, stride, pad, Hin, Win, Cin_s, Hout, Wout, Cout_s, Cin=0, Cout=0):
    """Cin / Cout: logical channel counts (0 = unknown), a hint that lets kernels skip the zero padding channels."""
    return L.ConvDesc(kind, k, stride, pad, Hin, Win, Cin_s, Hout, Wout, Cout_s, Cin, Cout, _math)


def stat_replicas():
    """SGAN_STAT_REPLICAS of the loaded library."""
    return L.lib().sgan_stat_replicas()



_STAT_REPLICATED = __import__("os").environ.get("SGAN_NO_STAT_REPLICAS", "0") in ("", "0")      # diagnostics switch


class _ArenaPool:
    """The statistics arenas of a training step, zeroed by ONE launch at its start (round 2: one aten fill per network call, 6-9 per
    step).  A step asks for the same sequence of arenas every time, so the pool is a list walked by a cursor: begin_step() zeroes
    every slot handed out since the last call (sgan_zero_multi) and rewinds; stat_arena() takes the next slot when it is clean and
    of the right size, anything else (first step, a changed sequence, no begin_step at all) falls back to torch.zeros.  Each slot is
    its own allocation (a pool carved from one buffer measured +35 us per step in round 2: the arenas' atomics then share memory
    channels).  Slots stay alive for the life of the process; an autograd graph must not outlive the step it was built in."""
    MAX_SLOTS = 128

    def __init__(self):
        self.slots, self.clean, self.cur = [], [], 0

    def begin_step(self, also_zero=()):
        dirty = [i for i, c in enumerate(self.clean) if not c]
        extra = []
        for t in also_zero:      # e.g. the discriminators' gradient arena: cleared by the same launch instead of a fill of its own
            if t.is_contiguous() and t.data_ptr() % 16 == 0 and (t.numel() * t.element_size()) % 16 == 0:
                extra.append(t)
            else:
                t.zero_()
        if dirty or extra:
            zero_multi([self.slots[i] for i in dirty] + extra)
            for i in dirty:
                self.clean[i] = True
        self.cur = 0

    def take(self, total, device):
        if _NO_ARENA_POOL:
            return torch.zeros(total, dtype=torch.float64, device=device)
        i = self.cur
        if i < len(self.slots) and self.clean[i] and self.slots[i].numel() == total and self.slots[i].device == torch.device(device):
            self.clean[i] = False
            self.cur += 1
            return self.slots[i]
        t = torch.zeros(total, dtype=torch.float64, device=device)
        if i < self.MAX_SLOTS:
            if i < len(self.slots):
                self.slots[i], self.clean[i] = t, False
            else:
                self.slots.append(t)
                self.clean.append(False)
            self.cur += 1
        return t


_NO_ARENA_POOL = __import__("os").environ.get("SGAN_NO_ARENA_POOL", "0") not in ("", "0")      # diagnostics: one aten fill per arena
_ARENAS = _ArenaPool()


def begin_step(also_zero=()):
    """Call at the start of a training step (the trainers' optimize_parameters and the captured graph do): one launch zeroes every
    statistics arena the step is going to use -- and the tensors in `also_zero` (FusedAdam.take_zeroing(): the gradient buffers an
    optimizer would otherwise clear with a fill of its own in zero_grad())."""
    _ARENAS.begin_step(also_zero)


def stat_arena(n, device):
    """Zeroed fp64 statistics arena of `n` doubles kept in STAT_REPLICAS copies `n` apart (returns the first copy; pass
    rep_stride = stat_rep(arena) wherever a slice of it is written or read): the conv epilogues spread their same-address atomics
    over the copies."""
    n = max(n, 1)
    return _ARENAS.take((stat_replicas() if _STAT_REPLICATED else 1) * n, device)[:n]


def stat_rep(arena):
    """Replica stride of a stat_arena (0 when replication is switched off: one copy)."""
    return arena.numel() if _STAT_REPLICATED else 0


def norm_desc(stats=None, gamma=None, beta=None, count=1, eps=1e-5, act=ACT_NONE, slope=0.0, sq_stride=0, rep_stride=0):
    """None when the read is a plain one (no norm, no activation).  `sq_stride`: distance from a channel's sum to its
    sum of squares inside `stats` (0 = the channel count of the tensor being read; wider when `stats` is a slice);
    `rep_stride`: `stats` is a slice of a stat_arena of that length (0: a plain array)."""
    if stats is None and act == ACT_NONE:
        return None
    d = L.NormDesc(_ptr(stats).value, _ptr(gamma).value, _ptr(beta).value, int(count), float(eps), int(act), float(slope),
                   int(sq_stride), int(rep_stride) if stats is not None else 0)
    d._keep = (stats, gamma, beta)
    return d


def _nd(d):
    return C.byref(d) if d is not None else None


def _workspace(kib, device):
    """Split-K scratch for one call (torch's caching allocator; static inside a captured graph)."""
    if kib < 0:
        raise L.SganError(f"workspace query failed ({kib}): {L.lib().sgan_last_error().decode()}")
    return torch.empty(kib * 256, dtype=torch.float32, device=device) if kib > 0 else None


def conv_fwd(desc, x, in_norm, w, bias, out, out_act=ACT_NONE, out_stats=None, stats_sq=0, stats_rep=0):
    return conv_fwd_grouped([(desc, x, in_norm, w, bias, out, out_stats, stats_sq, stats_rep)], out_act)


def conv_dgrad(desc, dout, w, din, x=None, x_norm=None, bwd_sums=None, sums_sq=0, accumulate=False, w_transposed=False, sums_rep=0):
    """accumulate: din += result (a tensor with two consumers); sums_sq: see norm_desc; w_transposed: `w` is the
    [tap][Cin][Cout] copy made by pack_weights."""
    return conv_dgrad_grouped([(desc, dout, w, din, x, x_norm, bwd_sums, sums_sq, accumulate, w_transposed, sums_rep)])


def conv_wgrad(desc, x, in_norm, dout, dw, dbias):
    return conv_wgrad_grouped([(desc, x, in_norm, dout, dw, dbias)])


def _pn(d):
    return C.pointer(d) if d is not None else None


def conv_fwd_grouped(jobs, out_act=ACT_NONE):
    """jobs: list of (desc, x, in_norm, w, bias, out, out_stats[, stats_sq, stats_rep]) of the same layer type -> one launch."""
    arr = (L.ConvFwdJob * len(jobs))()
    for i, job in enumerate(jobs):
        desc, x, in_norm, w, bias, out, st = job[:7]
        desc.math = _math
        arr[i] = L.ConvFwdJob(C.pointer(desc), _ptr(_act(x)).value, x.stride(1), _pn(in_norm), _ptr(w).value, _ptr(bias).value,
                              _ptr(_act(out)).value, out.stride(1), _ptr(st).value, int(job[7]) if len(job) > 7 else 0, _pk(w),
                              int(job[8]) if (len(job) > 8 and st is not None) else 0)
    ws = _workspace(L.lib().sgan_conv_fwd_grouped(arr, len(jobs), out_act, None, -1, None), jobs[0][1].device)
    _check_fwd_jobs(jobs)      # after the size query: a descriptor the library rejects is reported in the library's words
    L.check(L.lib().sgan_conv_fwd_grouped(arr, len(jobs), out_act, _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream()),
            "sgan_conv_fwd_grouped")


def _dgrad_array(jobs):
    arr = (L.ConvDgradJob * len(jobs))()
    for i, job in enumerate(jobs):
        desc, dout, w, din, x, x_norm, sums = job[:7]
        desc.math = _DGRAD_MATH if _DGRAD_MATH is not None else _math
        arr[i] = L.ConvDgradJob(C.pointer(desc), _ptr(_act(dout)).value, dout.stride(1), _ptr(w).value, _ptr(_act(din)).value,
                                din.stride(1), _ptr(x).value, x.stride(1) if x is not None else 0, _pn(x_norm), _ptr(sums).value,
                                int(job[7]) if len(job) > 7 else 0, int(bool(job[8])) if len(job) > 8 else 0,
                                int(bool(job[9])) if len(job) > 9 else 0, _pk(w),
                                int(job[10]) if (len(job) > 10 and sums is not None) else 0, _amax(dout), _pk16(w))
    return arr


def _wgrad_array(jobs):
    arr = (L.ConvWgradJob * len(jobs))()
    for i, (desc, x, in_norm, dout, dw, dbias) in enumerate(jobs):
        desc.math = _math
        arr[i] = L.ConvWgradJob(C.pointer(desc), _ptr(_act(x)).value, x.stride(1), _pn(in_norm), _ptr(_act(dout)).value,
                                dout.stride(1), _ptr(dw).value, _ptr(dbias).value, _amax(dout))
    return arr


def conv_dgrad_grouped(jobs):
    """jobs: list of (desc, dout, w, din, x, x_norm, bwd_sums[, sums_sq, accumulate, w_transposed, sums_rep])."""
    arr = _dgrad_array(jobs)
    ws = _workspace(L.lib().sgan_conv_dgrad_grouped(arr, len(jobs), None, -1, None), jobs[0][1].device)
    _check_dgrad_jobs(jobs)
    L.check(L.lib().sgan_conv_dgrad_grouped(arr, len(jobs), _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream()),
            "sgan_conv_dgrad_grouped")


def conv_wgrad_grouped(jobs):
    """jobs: list of (desc, x, in_norm, dout, dw, dbias)."""
    arr = _wgrad_array(jobs)
    _check_wgrad_jobs(jobs)
    d0 = jobs[0][0]
    ws = _workspace(L.lib().sgan_conv_wgrad_grouped(arr, len(jobs), None, -1, None), jobs[0][1].device) if min(d0.Cin, d0.Cout) <= 4 else None
    L.check(L.lib().sgan_conv_wgrad_grouped(arr, len(jobs), _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream()),
            "sgan_conv_wgrad_grouped")


def conv_bwd_grouped(djobs, wjobs, dgrad_math=None):
    """A layer's backward-weight and backward-data (job lists as for the two calls above): one fused launch where
    sgan_conv_bwd_fused covers the layer, the two grouped launches otherwise.  dgrad_math: arithmetic of the backward-data half
    ("f32" / "bf16x3"; None = the current mode)."""
    dm = _DGRAD_MATH if _DGRAD_MATH is not None else (_MATH_NAMES[dgrad_math] if dgrad_math else _math)
    d0 = djobs[0][0]
    if d0.Cout == 4 and d0.Cout_logical == 1 and d0.kind == L.CONV and d0.stride == 1 and len(djobs) == len(wjobs) and djobs[0][4] is not None:
        _check_dgrad_jobs(djobs)
        _check_wgrad_jobs(wjobs)
        rc = L.lib().sgan_conv_head_bwd(_dgrad_array(djobs), _wgrad_array(wjobs), len(djobs), _stream())      # the one-channel head: one launch
        if rc == 0:
            return True
        if rc < 0:
            L.check(rc, "sgan_conv_head_bwd")
    if _math == L.MATH_BF16X3:
        _check_dgrad_jobs(djobs)
        _check_wgrad_jobs(wjobs)
        da, wa = _dgrad_array(djobs), _wgrad_array(wjobs)
        ws = _workspace(L.lib().sgan_conv_bwd_fused_ws(da, len(djobs), wa, len(wjobs), dm, None, -1, None), djobs[0][1].device)
        rc = L.lib().sgan_conv_bwd_fused_ws(da, len(djobs), wa, len(wjobs), dm, _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream())
        if rc == 0:
            return True
        if rc < 0:
            L.check(rc, "sgan_conv_bwd_fused_ws")
    if min(d0.Cin, d0.Cout) == 4 and len(djobs) == len(wjobs):      # a layer with a 4-channel side: its two launches in one grid
        _check_dgrad_jobs(djobs)
        _check_wgrad_jobs(wjobs)
        with math_scope(dgrad_math):
            rc = L.lib().sgan_conv_bwd_thin_pair(_dgrad_array(djobs), len(djobs), _wgrad_array(wjobs), len(wjobs), _stream())
        if rc == 0:
            return True
        if rc < 0:
            L.check(rc, "sgan_conv_bwd_thin_pair")
    conv_wgrad_grouped(wjobs)
    with math_scope(dgrad_math):
        conv_dgrad_grouped(djobs)
    return False


def transpose_weights(flat, flat_t, segs):
    """segs: list of (off, taps, cout_s, cin_s) conv ranges of the flat parameter buffer."""
    for i0 in range(0, len(segs), 64):
        part = segs[i0:i0 + 64]
        arr = (L.WtSeg * len(part))(*[L.WtSeg(int(o), int(t), int(co), int(ci)) for o, t, co, ci in part])
        L.check(L.lib().sgan_transpose_weights(_ptr(flat), _ptr(flat_t), arr, len(part), _stream()), "sgan_transpose_weights")


def pack_weights(flat, flat_t, pk_fwd, pk_bwd, segs, pk_bwd16=None):
    """Every derived weight copy of the conv ranges `segs` = [(off, taps, cout_s, cin_s)] of a flat parameter buffer in one
    launch per 64 ranges: fp32 transposed copy + the split 16-bit copies (sgan_pack_weights)."""
    for i0 in range(0, len(segs), 64):
        part = segs[i0:i0 + 64]
        arr = (L.WtSeg * len(part))(*[L.WtSeg(int(o), int(t), int(co), int(ci)) for o, t, co, ci in part])
        L.check(L.lib().sgan_pack_weights(_ptr(flat), _ptr(flat_t), _ptr(pk_fwd), _ptr(pk_bwd), _ptr(pk_bwd16), arr, len(part), _stream()),
                "sgan_pack_weights")


def norm_bwd_apply(dy, x, x_norm, bwd_sums, dgamma=None, dbeta=None, sums_sq=0, sums_rep=0, publish_amax=False):
    if sums_rep or publish_amax:
        return norm_bwd_apply_multi([(dy, x, x_norm, bwd_sums, dgamma, dbeta, sums_sq, sums_rep)], publish_amax)
    H, W, Cs = dy.shape
    L.check(L.lib().sgan_norm_bwd_apply(_ptr(_act(dy)), dy.stride(1), _ptr(_act(x)), x.stride(1), H * W, Cs, _nd(x_norm),
                                        _ptr(bwd_sums), int(sums_sq), _ptr(dgamma), _ptr(dbeta), _stream()), "sgan_norm_bwd_apply")


def norm_bwd_apply_multi(jobs, publish_amax=False):
    """jobs: list of (dy, x, x_norm, bwd_sums, dgamma, dbeta[, sums_sq, sums_rep]) -> one launch (<= 8 per launch).
    publish_amax: the kernel also leaves max|dy| of every result in a device scalar and tags the tensor with it (`dy._sgan_amax`): the
    backward-data / backward-weight calls that read THIS tensor object next then run on fp16 planes scaled by it (an fp32-equivalent
    product) instead of bf16 planes.  Only for callers that do not write dy again before it is consumed."""
    for i0 in range(0, len(jobs), 8):
        part = jobs[i0:i0 + 8]
        arr = (L.NormBwdJob * len(part))()
        am = None
        if publish_amax and _math == L.MATH_BF16X3 and not _NO_F16_BWD:
            am = stat_arena(len(part), part[0][0].device).view(torch.float32)      # one zeroed 8-byte slot per job (low word = the maximum)
        for i, job in enumerate(part):
            dy, x, x_norm, sums, dg, db = job[:6]
            H, W, Cs = dy.shape
            slot = am[2 * i: 2 * i + 1] if am is not None else None
            arr[i] = L.NormBwdJob(_ptr(_act(dy)).value, dy.stride(1), _ptr(_act(x)).value, x.stride(1), H * W, Cs, C.pointer(x_norm),
                                  _ptr(sums).value, int(job[6]) if len(job) > 6 else 0, _ptr(dg).value, _ptr(db).value,
                                  int(job[7]) if len(job) > 7 else 0, _ptr(slot).value)
            dy._sgan_amax = slot
        L.check(L.lib().sgan_norm_bwd_apply_multi(arr, len(part), _stream()), "sgan_norm_bwd_apply_multi")


_NO_F16_BWD = __import__("os").environ.get("SGAN_NO_F16_BWD", "0") not in ("", "0")      # diagnostics: bf16 planes in the backward pass as in round 2


def has_amax(t) -> bool:
    return getattr(t, "_sgan_amax", None) is not None and _math == L.MATH_BF16X3


def norm_apply_fwd(u, u_norm, t, mask=None, noise=None, sigma=0.0):
    """t = norm(u) * mask + sigma * noise  (U-Net up path: upnorm -> Dropout -> + Gaussian noise, into a concat slice)."""
    H, W, Cs = u.shape
    L.check(L.lib().sgan_norm_apply_fwd(_ptr(_act(u)), u.stride(1), _nd(u_norm), _ptr(mask), _ptr(noise), float(sigma),
                                        _ptr(_act(t)), t.stride(1), H * W, Cs, _stream()), "sgan_norm_apply_fwd")


def norm_apply_bwd_sums(dt, u, u_norm, bwd_sums, mask=None):
    """dt *= mask (in place), bwd_sums += (sum dt, sum dt * norm(u)) per channel."""
    H, W, Cs = dt.shape
    L.check(L.lib().sgan_norm_apply_bwd_sums(_ptr(_act(dt)), dt.stride(1), _ptr(mask), _ptr(_act(u)), u.stride(1), _nd(u_norm),
                                             _ptr(bwd_sums), H * W, Cs, _stream()), "sgan_norm_apply_bwd_sums")


def pad_reflect_fwd(x, x_norm, pad, out, mask=None):
    """out [H + 2 pad, W + 2 pad, C] = mask * act(norm(x)) at the reflected positions (nn.ReflectionPad2d after norm / act / dropout)."""
    H, W, Cs = x.shape
    assert out.shape == (H + 2 * pad, W + 2 * pad, Cs)
    L.check(L.lib().sgan_pad_reflect_fwd(_ptr(_act(x)), x.stride(1), H, W, Cs, _nd(x_norm), _ptr(mask), int(pad), _ptr(_act(out)),
                                         out.stride(1), _stream()), "sgan_pad_reflect_fwd")


def pad_reflect_bwd(dout, pad, din, x=None, x_norm=None, mask=None, bwd_sums=None, sums_sq=0):
    """din = act'(norm(x)) * mask * fold(dout); bwd_sums += (sum din, sum din * xhat)."""
    H, W, Cs = din.shape
    assert dout.shape == (H + 2 * pad, W + 2 * pad, Cs)
    L.check(L.lib().sgan_pad_reflect_bwd(_ptr(_act(dout)), dout.stride(1), H, W, Cs, int(pad), _ptr(x), x.stride(1) if x is not None else 0,
                                         _nd(x_norm), _ptr(mask), _ptr(_act(din)), din.stride(1), _ptr(bwd_sums), int(sums_sq), _stream()),
            "sgan_pad_reflect_bwd")


def bilinear_up2_fwd(x, out, out_stats=None, stats_sq=0):
    H, W, Cs = x.shape
    L.check(L.lib().sgan_bilinear_up2_fwd(_ptr(_act(x)), x.stride(1), H, W, Cs, _ptr(_act(out)), out.stride(1), _ptr(out_stats),
                                          int(stats_sq), _stream()), "sgan_bilinear_up2_fwd")


def bilinear_up2_bwd(dout, din):
    H, W, Cs = din.shape
    L.check(L.lib().sgan_bilinear_up2_bwd(_ptr(_act(dout)), dout.stride(1), H, W, Cs, _ptr(_act(din)), din.stride(1), _stream()),
            "sgan_bilinear_up2_bwd")


def _level_table(levels):
    ptrs = (C.c_void_p * 6)(*[t.data_ptr() if t is not None else None for t in levels])
    lds = (C.c_int32 * 6)(*[t.stride(1) if t is not None else 4 for t in levels])
    return ptrs, lds


def avgpool_pyramid_fwd(label, levels):
    """levels[s]: [H >> (s+1), W >> (s+1), 4] buffers (or channel slices), s = 0..5."""
    H, W, _ = label.shape
    ptrs, lds = _level_table(levels)
    L.check(L.lib().sgan_avgpool_pyramid_fwd(_ptr(_act(label)), label.stride(1), H, W, ptrs, lds, _stream()), "sgan_avgpool_pyramid_fwd")


def avgpool_pyramid_bwd(dlevels, dlabel, accumulate=False):
    H, W, _ = dlabel.shape
    ptrs, lds = _level_table(dlevels)
    L.check(L.lib().sgan_avgpool_pyramid_bwd(ptrs, lds, H, W, _ptr(_act(dlabel)), dlabel.stride(1), int(accumulate), _stream()),
            "sgan_avgpool_pyramid_bwd")


def dropout_mask(mask, p, seed, offset_dev=None, advance=True):
    L.check(L.lib().sgan_dropout_mask(_ptr(mask), mask.numel(), float(p), C.c_uint64(seed & (2 ** 64 - 1)), _ptr(offset_dev),
                                      int(bool(advance)), _stream()), "sgan_dropout_mask")


def rng_advance(offset_dev, by):
    L.check(L.lib().sgan_rng_advance(_ptr(offset_dev), C.c_uint64(int(by)), _stream()), "sgan_rng_advance")


IMAGE_LOSS_WS_BYTES = 2048   # SGAN_IMAGE_LOSS_WS_BYTES


def l1w_fwd(x, y, Creal, a, weights_dev, nweights, lam, loss_out, g):
    H, W, _ = x.shape
    ws = torch.empty(IMAGE_LOSS_WS_BYTES // 8, dtype=torch.float64, device=x.device)
    L.check(L.lib().sgan_l1w_fwd(_ptr(_act(x)), x.stride(1), _ptr(_act(y)), y.stride(1), H * W, Creal,
                                 _ptr(a), a.stride(1) if a is not None else 0, _ptr(weights_dev), nweights, float(lam),
                                 _ptr(loss_out), _ptr(_act(g)), g.stride(1), _ptr(ws), IMAGE_LOSS_WS_BYTES, _stream()), "sgan_l1w_fwd")


def ce_fwd(logits, Creal, label, const_label, class_w, acc, loss_out):
    """Class-weighted cross-entropy of an [H, W, Cs] logits buffer (sgan_ce_fwd).  acc: zeroed float64[2 + 1] scratch (sums | ticket)."""
    H, W, _ = logits.shape
    L.check(L.lib().sgan_ce_fwd(_ptr(_act(logits)), logits.stride(1), H * W, Creal, _ptr(label), int(const_label), _ptr(class_w),
                                _ptr(acc), C.c_void_p(acc.data_ptr() + 16), _ptr(loss_out), _stream()), "sgan_ce_fwd")


def ce_bwd(logits, Creal, label, const_label, class_w, acc, gout, dlogits):
    H, W, _ = logits.shape
    L.check(L.lib().sgan_ce_bwd(_ptr(_act(logits)), logits.stride(1), H * W, Creal, _ptr(label), int(const_label), _ptr(class_w),
                                _ptr(acc), _ptr(gout), _ptr(_act(dlogits)), dlogits.stride(1), _stream()), "sgan_ce_bwd")


def softmax_fwd(z, Creal, p):
    H, W, _ = z.shape
    L.check(L.lib().sgan_softmax_fwd(_ptr(_act(z)), z.stride(1), H * W, Creal, _ptr(_act(p)), p.stride(1), _stream()), "sgan_softmax_fwd")


def softmax_bwd(dp, p, Creal, dz):
    H, W, _ = p.shape
    L.check(L.lib().sgan_softmax_bwd(_ptr(_act(dp)), dp.stride(1), _ptr(_act(p)), p.stride(1), H * W, Creal, _ptr(_act(dz)), dz.stride(1),
                                     _stream()), "sgan_softmax_bwd")


def bce01_fwd(x, t, Creal, loss_out, g):
    H, W, _ = x.shape
    ws = torch.empty(IMAGE_LOSS_WS_BYTES // 8, dtype=torch.float64, device=x.device)
    L.check(L.lib().sgan_bce01_fwd(_ptr(_act(x)), x.stride(1), _ptr(_act(t)), t.stride(1), H * W, Creal, _ptr(loss_out),
                                   _ptr(_act(g)), g.stride(1), _ptr(ws), IMAGE_LOSS_WS_BYTES, _stream()), "sgan_bce01_fwd")


def scale(gout, g, dx):
    assert g.is_contiguous() and dx.is_contiguous()
    L.check(L.lib().sgan_scale(_ptr(gout), _ptr(g), _ptr(dx), g.numel(), _stream()), "sgan_scale")


def bn_running_update(layers, momentum=0.1):
    """layers: list of (stats, running_mean, running_var, num_batches_tracked, C, count[, sq_stride, rep_stride])."""
    arr = (L.BnRunningDesc * len(layers))()
    for i, job in enumerate(layers):
        st, rm, rv, nbt, c, cnt = job[:6]
        arr[i] = L.BnRunningDesc(st.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr() if nbt is not None else 0, c, cnt,
                                 int(job[6]) if len(job) > 6 else 0, int(job[7]) if len(job) > 7 else 0)
    L.check(L.lib().sgan_bn_running_update(arr, len(layers), momentum, _stream()), "sgan_bn_running_update")


def image_prep(img_u8, x0, y0, n, flip, rot, out=None):
    """img_u8: [H0, W0, 3] uint8 device tensor -> [n, n, 4] fp32 NHWC buffer in [-1, 1] (crop, flip, rot90, ToTensor, Normalize)."""
    require_gpu(img_u8, "image_prep")
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.shape[2] == 3 and img_u8.is_contiguous(), (img_u8.shape, img_u8.dtype)
    if out is None:
        out = torch.empty((n, n, 4), dtype=torch.float32, device=img_u8.device)
    L.check(L.lib().sgan_image_prep(_ptr(img_u8), img_u8.shape[0], img_u8.shape[1], int(x0), int(y0), int(n), int(bool(flip)), int(rot),
                                    _ptr(_act(out)), out.stride(1), out.shape[2], _stream()), "sgan_image_prep")
    return out


RESAMPLE = {"bilinear": 2, "bicubic": 3}      # Pillow's Image.BILINEAR / Image.BICUBIC


def image_resize(img_u8, wo, ho, resample):
    """img_u8 [H, W, C] uint8 device tensor -> [ho, wo, C] uint8: Image.resize((wo, ho), resample) bit for bit, on the device."""
    require_gpu(img_u8, "image_resize")
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 3 and img_u8.is_contiguous(), (img_u8.shape, img_u8.dtype)
    H, W, Cc = img_u8.shape
    f = RESAMPLE[resample] if isinstance(resample, str) else int(resample)
    lib = L.lib()
    need = int(lib.sgan_image_resize_workspace(H, W, Cc, int(ho), int(wo), f))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=img_u8.device)
    out = torch.empty((int(ho), int(wo), Cc), dtype=torch.uint8, device=img_u8.device)
    L.check(lib.sgan_image_resize(_ptr(img_u8), H, W, Cc, _ptr(out), int(ho), int(wo), f, _ptr(ws), need, _stream()), "sgan_image_resize")
    return out


def gauss_down_fwd(x, Creal, g, g_chan_stride, k, pad, s, out):
    H, W, Cs = x.shape
    Ho, Wo, _ = out.shape
    L.check(L.lib().sgan_gauss_down_fwd(_ptr(_act(x)), x.stride(1), H, W, Cs, Creal, _ptr(g), g_chan_stride, k, pad, s,
                                        _ptr(_act(out)), out.stride(1), Ho, Wo, _stream()), "sgan_gauss_down_fwd")


def gauss_down_bwd(dout, Creal, g, g_chan_stride, k, pad, s, din, accumulate=False):
    Ho, Wo, Cs = dout.shape
    H, W, _ = din.shape
    L.check(L.lib().sgan_gauss_down_bwd(_ptr(_act(dout)), dout.stride(1), Ho, Wo, Cs, Creal, _ptr(g), g_chan_stride, k, pad, s,
                                        _ptr(_act(din)), din.stride(1), H, W, int(bool(accumulate)), _stream()), "sgan_gauss_down_bwd")


def _gauss_jobs(jobs):
    """jobs: [(image, down, g, g_chan_stride, k, pad, s)] NHWC buffers of one channel count."""
    arr = (L.GaussJob * len(jobs))()
    for i, (img, down, g, gcs, k, pad, s) in enumerate(jobs):
        H, W, _ = img.shape
        Ho, Wo, _ = down.shape
        arr[i] = L.GaussJob(_ptr(_act(img)).value, img.stride(1), H, W, _ptr(_act(down)).value, down.stride(1), Ho, Wo,
                            _ptr(g).value, int(gcs), int(k), int(pad), int(s))
    return arr


def gauss_down_multi_fwd(jobs, Creal):
    for i0 in range(0, len(jobs), 4):
        part = jobs[i0: i0 + 4]
        L.check(L.lib().sgan_gauss_down_multi_fwd(_gauss_jobs(part), len(part), part[0][0].shape[2], Creal, _stream()),
                "sgan_gauss_down_multi_fwd")


def gauss_down_multi_bwd(jobs, Creal, accumulate=False):
    """All jobs share jobs[i][0], the image gradient, which receives the sum of their contributions."""
    for i0 in range(0, len(jobs), 4):
        part = jobs[i0: i0 + 4]
        L.check(L.lib().sgan_gauss_down_multi_bwd(_gauss_jobs(part), len(part), part[0][0].shape[2], Creal,
                                                  int(bool(accumulate or i0 > 0)), _stream()), "sgan_gauss_down_multi_bwd")


def gan_loss_fwd(logits, target, mode, loss_out, p_out=None):
    H, W, _ = logits.shape
    L.check(L.lib().sgan_gan_loss_fwd(_ptr(_act(logits)), logits.stride(1), H * W, float(target), mode, _ptr(loss_out),
                                      _ptr(p_out), _stream()), "sgan_gan_loss_fwd")


def gan_loss_bwd(logits, target, mode, gout, dlogits):
    H, W, _ = logits.shape
    L.check(L.lib().sgan_gan_loss_bwd(_ptr(_act(logits)), logits.stride(1), H * W, float(target), mode, _ptr(gout),
                                      _ptr(_act(dlogits)), dlogits.stride(1), _stream()), "sgan_gan_loss_bwd")


GAN_LOSS_WS_BYTES = 2048   # SGAN_GAN_LOSS_WS_BYTES
_loss_ws = {}
_unit_grads = []           # weak references to gradient tensors known to hold 1.0 (BaseModel._backward's cached root gradient)


def register_unit_grad(t):
    """`t` holds 1.0 and is never written: a fused loss node that receives this very tensor object as its upstream gradient hands
    out its precomputed unit-gradient result without a rescaling kernel."""
    _unit_grads[:] = [r for r in _unit_grads if r() is not None]
    _unit_grads.append(weakref.ref(t))


def is_unit_grad(t) -> bool:
    return any(r() is t for r in _unit_grads)


def _gan_loss_workspace(device):
    """Zero-initialised once per device; the kernel leaves its ticket counter at zero.  The fused loss runs on the trainer's main
    stream only (side-stream chains use the per-term loss)."""
    key = device.index
    ws = _loss_ws.get(key)
    if ws is None:
        ws = _loss_ws[key] = torch.zeros(GAN_LOSS_WS_BYTES // 8, dtype=torch.float64, device=device)
    return ws


def gan_loss_multi_fwd(logits, targets, weights, mode, each_out, total_out, dlogits=None):
    """dlogits (optional list of NHWC buffers): also receives d total / d logits_i for an upstream gradient of 1."""
    arr = (L.GanLossJob * len(logits))()
    for i, (lb, t, w) in enumerate(zip(logits, targets, weights)):
        d = dlogits[i] if dlogits is not None else None
        arr[i] = L.GanLossJob(_ptr(_act(lb)).value, lb.stride(1), lb.shape[0] * lb.shape[1], float(t), float(w),
                              _ptr(_act(d)).value if d is not None else None, d.stride(1) if d is not None else 0)
    ws = _gan_loss_workspace(each_out.device)
    L.check(L.lib().sgan_gan_loss_multi_fwd(arr, len(logits), mode, _ptr(each_out), _ptr(total_out), _ptr(ws), GAN_LOSS_WS_BYTES,
                                            _stream()), "sgan_gan_loss_multi_fwd")


def gan_loss_multi_bwd(logits, targets, weights, mode, gout, dlogits):
    arr = (L.GanLossJob * len(logits))()
    for i, (lb, t, w, d) in enumerate(zip(logits, targets, weights, dlogits)):
        arr[i] = L.GanLossJob(_ptr(_act(lb)).value, lb.stride(1), lb.shape[0] * lb.shape[1], float(t), float(w),
                              _ptr(_act(d)).value, d.stride(1))
    L.check(L.lib().sgan_gan_loss_multi_bwd(arr, len(logits), mode, _ptr(gout), _stream()), "sgan_gan_loss_multi_bwd")


def sigmoid_fwd(x, p):
    H, W, _ = x.shape
    L.check(L.lib().sgan_sigmoid_fwd(_ptr(_act(x)), x.stride(1), H * W, _ptr(_act(p)), p.stride(1), _stream()), "sgan_sigmoid_fwd")


def sigmoid_bwd(dp, p, dx):
    H, W, _ = p.shape
    L.check(L.lib().sgan_sigmoid_bwd(_ptr(_act(dp)), dp.stride(1), _ptr(_act(p)), p.stride(1), H * W, _ptr(_act(dx)),
                                     dx.stride(1), _stream()), "sgan_sigmoid_bwd")


def tanh_bwd(dy, y, dx):
    assert dy.is_contiguous() and y.is_contiguous() and dx.is_contiguous()
    L.check(L.lib().sgan_tanh_bwd(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream()), "sgan_tanh_bwd")


def add_act_fwd(a, b, out, act=ACT_TANH):
    """out = act(a + b) on contiguous padded NHWC buffers of one shape (the --use_residual tail of the generators)."""
    assert a.shape == b.shape == out.shape and a.is_contiguous() and b.is_contiguous() and out.is_contiguous()
    L.check(L.lib().sgan_add_act_fwd(_ptr(a), _ptr(b), _ptr(out), a.numel(), int(act), _stream()), "sgan_add_act_fwd")


def adam_multi(segs, lr_dev, beta1, beta2, eps, state_dev):
    """segs: list of (p, g, m, v, n) flat fp32 tensors (16-byte aligned)."""
    arr = (L.AdamSeg * len(segs))()
    for i, (p, g, m, v, n) in enumerate(segs):
        arr[i] = L.AdamSeg(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n)
    L.check(L.lib().sgan_adam_multi(arr, len(segs), _ptr(lr_dev), beta1, beta2, eps, _ptr(state_dev), _stream()), "sgan_adam_multi")


def adam_pack(p, g, m, v, lr_dev, beta1, beta2, eps, state_dev, flat_t, pk_f, pk_b, pk_b16, segs, zero_grads):
    """One launch: Adam over the flat segment (p, g, m, v) + the derived weight copies of its conv ranges
    `segs` = [(off, taps, cout_s, cin_s)] (offsets relative to p; sorted) + optional zeroing of the consumed gradients."""
    arr = (L.WtSeg * max(len(segs), 1))(*[L.WtSeg(int(o), int(t), int(co), int(ci)) for o, t, co, ci in segs])
    L.check(L.lib().sgan_adam_pack(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(lr_dev), beta1, beta2, eps, _ptr(state_dev),
                                   _ptr(flat_t), _ptr(pk_f), _ptr(pk_b), _ptr(pk_b16), arr, len(segs), int(bool(zero_grads)), _stream()), "sgan_adam_pack")


def zero_multi(bufs):
    """Zero a list of contiguous device tensors (byte sizes multiples of 16) in one launch per 64."""
    for i0 in range(0, len(bufs), 64):
        part = bufs[i0:i0 + 64]
        ptrs = (C.c_void_p * len(part))(*[t.data_ptr() for t in part])
        nb = (C.c_int64 * len(part))(*[t.numel() * t.element_size() for t in part])
        L.check(L.lib().sgan_zero_multi(ptrs, nb, len(part), _stream()), "sgan_zero_multi")


def sgd_multi(segs, lr_dev, momentum):
    """segs: list of (p, g, buf or None, n) flat fp32 tensors."""
    arr = (L.AdamSeg * len(segs))()
    for i, (p, g, m, n) in enumerate(segs):
        arr[i] = L.AdamSeg(p.data_ptr(), g.data_ptr(), m.data_ptr() if m is not None else None, None, n)
    L.check(L.lib().sgan_sgd_multi(arr, len(segs), _ptr(lr_dev), float(momentum), _stream()), "sgan_sgd_multi")


def normal_fill(dst, seed, offset_dev=None, advance=True):
    assert dst.is_contiguous() and dst.dtype == torch.float32
    L.check(L.lib().sgan_normal_fill(_ptr(dst), dst.numel(), C.c_uint64(seed & (2 ** 64 - 1)), _ptr(offset_dev), int(bool(advance)),
                                     _stream()), "sgan_normal_fill")


def normal_fill_nhwc(buf, C_real, seed, offset_dev=None, advance=True):
    """N(0, 1) values of a logical [1, C_real, H, W] tensor (the same values normal_fill gives its contiguous form), written into
    the padded NHWC buffer `buf` [H, W, Cs] the networks read: no layout kernel between the draw and the generator."""
    H, W, Cs = buf.shape
    assert buf.is_contiguous() and buf.dtype == torch.float32
    L.check(L.lib().sgan_normal_fill_nhwc(_ptr(buf), int(C_real), H, W, Cs, C.c_uint64(seed & (2 ** 64 - 1)), _ptr(offset_dev),
                                          int(bool(advance)), _stream()), "sgan_normal_fill_nhwc")


def normal_fill_nhwc_pair(buf_a, buf_b, C_real, seed, offset_dev, zero=None):
    """normal_fill_nhwc(buf_a) then normal_fill_nhwc(buf_b) -- same values, same advance of the stream -- as ONE launch that also
    zeroes the contiguous tensor `zero` (the statistics arena of the pass that reads the two latents)."""
    H, W, Cs = buf_a.shape
    assert buf_a.shape == buf_b.shape and buf_a.is_contiguous() and buf_b.is_contiguous() and buf_a.dtype == buf_b.dtype == torch.float32
    assert zero is None or zero.is_contiguous()
    L.check(L.lib().sgan_normal_fill_nhwc_pair(_ptr(buf_a), _ptr(buf_b), int(C_real), H, W, Cs, C.c_uint64(seed & (2 ** 64 - 1)),
                                               _ptr(offset_dev), _ptr(zero), zero.numel() * zero.element_size() if zero is not None else 0,
                                               _stream()), "sgan_normal_fill_nhwc_pair")


# ------------------------------------------------------------------------------------------------
# NCHW <-> NHWC boundary.  Tensors handed to user code are logical [1, C, H, W] *views* of our
# padded NHWC buffers (zero copy); a weak registry lets us recognise such a view (or its .detach())
# when it comes back, otherwise one strided-gather kernel converts the layout.
# ------------------------------------------------------------------------------------------------
_VIEW_REGISTRY = {}


def logical_view(buf: torch.Tensor, C_real: int) -> torch.Tensor:
    """[H, W, Cs] buffer -> logical [1, C_real, H, W] view; registered for zero-copy round trips."""
    v = buf.permute(2, 0, 1)[:C_real].unsqueeze(0)
    _VIEW_REGISTRY[buf.data_ptr()] = (weakref.ref(buf), C_real, buf.untyped_storage()._cdata, tuple(buf.shape))
    if len(_VIEW_REGISTRY) > 4096:
        for k in [k for k, e in _VIEW_REGISTRY.items() if e[0]() is None]:
            del _VIEW_REGISTRY[k]
    return v


def buffer_of(t: torch.Tensor):
    """The padded NHWC buffer a logical_view() tensor was made from (same memory), or None."""
    ent = _VIEW_REGISTRY.get(t.data_ptr())
    if ent is None or t.dim() != 4:
        return None
    buf = ent[0]()
    _, Cr, H, W = t.shape
    if (buf is not None and ent[1] == Cr and buf.shape[:2] == (H, W) and buf.data_ptr() == t.data_ptr()
            and t.stride(1) == 1 and t.stride(2) == buf.stride(0) and t.stride(3) == buf.stride(1)):
        return buf
    return None


def host_batch_to_nhwc(t: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """Pinned-host (or device) logical [1, C <= 4, H, W] fp32 batch -> the padded NHWC device buffer `out` [H, W, 4], by ONE gather
    kernel that reads the source where it lies: for a pinned host tensor that kernel is the H2D copy, on the current stream."""
    assert t.dim() == 4 and t.shape[0] == 1 and t.dtype == torch.float32 and t.stride(3) == 1, (t.shape, t.stride(), t.dtype)
    assert t.is_cuda or t.is_pinned(), "host batches must be pinned (page-locked) to be read by the device"
    _, Cr, H, W = t.shape
    assert out.shape == (H, W, 4) and Cr <= 4
    L.check(L.lib().sgan_to_nhwc(_ptr(t), t.stride(1), t.stride(2), 1, H, W, Cr, _ptr(out), out.stride(1), 4, _stream()), "sgan_to_nhwc")
    return out


def concat_nhwc(a, Ca, b, Cb):
    """Padded NHWC buffers a [H, W, >= Ca], b [H, W, >= Cb] -> [H, W, pad4(Ca + Cb)]: torch.cat of the logical tensors, one pass."""
    H, W, _ = a.shape
    assert b.shape[:2] == (H, W)
    out = torch.empty((H, W, pad4(Ca + Cb)), dtype=torch.float32, device=a.device)
    L.check(L.lib().sgan_concat_nhwc(_ptr(_act(a)), a.stride(1), int(Ca), _ptr(_act(b)), b.stride(1), int(Cb), H * W, _ptr(out), out.stride(1),
                                     out.shape[2], _stream()), "sgan_concat_nhwc")
    return out


def slice_nhwc(src, c0, Cn, out=None):
    """Channels [c0, c0 + Cn) of a padded NHWC buffer as a padded NHWC buffer of their own (`out`: write into this one)."""
    H, W, _ = src.shape
    if out is None:
        out = torch.empty((H, W, pad4(Cn)), dtype=torch.float32, device=src.device)
    assert out.shape[:2] == (H, W) and out.shape[2] >= pad4(Cn)
    L.check(L.lib().sgan_slice_nhwc(_ptr(_act(src)), src.stride(1), int(c0), int(Cn), H * W, _ptr(out), out.stride(1), out.shape[2], _stream()),
            "sgan_slice_nhwc")
    return out


def as_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Logical [1, C, H, W] tensor (any strides) -> padded NHWC buffer [H, W, pad4(C)]."""
    require_gpu(t, "as_nhwc")
    assert t.dim() == 4 and t.shape[0] == 1, f"batch 1 NCHW expected, got {tuple(t.shape)}"
    if t.dtype != torch.float32:
        raise L.SganError(f"fp32 expected, got {t.dtype}")
    _, Cr, H, W = t.shape
    Cs = pad4(Cr)
    ent = _VIEW_REGISTRY.get(t.data_ptr())
    if ent is not None:
        buf = ent[0]()
        if (buf is None and ent[3] == (H, W, Cs) and t.untyped_storage()._cdata == ent[2] and t.stride(1) == 1
                and t.stride(3) == Cs and t.stride(2) == W * Cs):
            # the buffer's Python object is gone (autograd hands out detached aliases of view outputs, ImagePool.query detaches)
            # but `t` still IS that buffer's storage at the registered address: the same memory under the buffer's shape
            buf = t.as_strided((H, W, Cs), (W * Cs, Cs, 1))
        if (buf is not None and ent[1] == Cr and buf.shape == (H, W, Cs) and buf.data_ptr() == t.data_ptr()
                and t.stride(1) == 1 and t.stride(2) == buf.stride(0) and t.stride(3) == buf.stride(1)):
            return buf
    out = torch.empty((H, W, Cs), device=t.device, dtype=torch.float32)
    L.check(L.lib().sgan_to_nhwc(_ptr(t), t.stride(1), t.stride(2), t.stride(3), H, W, Cr, _ptr(out), Cs, Cs, _stream()),
            "sgan_to_nhwc")
    return out

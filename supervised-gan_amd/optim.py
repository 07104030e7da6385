"""Fused Adam over the flat parameter storage of supervised_gan_amd networks.

Mirrors `torch.optim.Adam(params, lr=, betas=)` as the reference trainers use it
(models/fcgan_model.py:98-109: beta1 = opt.beta1 = 0.5, beta2 = 0.999, eps 1e-8, no weight decay):
same constructor call, `param_groups[i]['lr']` writable by the LR schedule, `zero_grad()`, `step()`.
One sgan_adam_multi launch updates every parameter of the optimizer; the step counter and the LR
live in device memory so that a captured hipGraph of the training step replays correctly."""
import torch

import os

from . import ops
from ._lib import SganError

_NO_ADAM_PACK = os.environ.get("SGAN_NO_ADAM_PACK", "0") not in ("", "0")      # diagnostics: the three-launch optimizer step of round 2


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, zero_grads_in_step=False):
        """zero_grads_in_step: step() leaves the gradients it consumed zeroed and the next zero_grad() is a no-op.  Only for
        optimizers whose gradients nobody writes between step() and zero_grad() (the generator's: the discriminators' are also
        written -- and discarded -- by the generator step, models/fcgan_model.py:165-176)."""
        params = list(params)
        self._zero_in_step = bool(zero_grads_in_step) and not _NO_ADAM_PACK
        self._grads_clean = False
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        self.param_groups = [{"params": params, "lr": float(lr), "betas": tuple(betas), "eps": float(eps)}]
        # map every parameter to its range in an arena (a network's flat storage, or the shared arena
        # several networks were packed into) and merge adjacent ranges
        ranges = {}
        self._nets = {}
        for p in params:
            seg = getattr(p, "_sgan_seg", None)
            if seg is None:
                raise SganError("FusedAdam only handles parameters of supervised_gan_amd networks "
                                "(use torch.optim.Adam for foreign parameters)")
            net, off, n = seg
            self._nets[id(net)] = net
            arena_p, arena_g, base = net._arena
            ranges.setdefault(arena_p.data_ptr(), (arena_p, arena_g, []))[2].append((base + off, n))
        self._segs = []   # (param arena, grad arena, off, n)
        for arena_p, arena_g, rs in ranges.values():
            rs.sort()
            cur_off, cur_n = rs[0]
            for off, n in rs[1:]:
                if off == cur_off + cur_n:
                    cur_n += n
                else:
                    self._segs.append((arena_p, arena_g, cur_off, cur_n))
                    cur_off, cur_n = off, n
            self._segs.append((arena_p, arena_g, cur_off, cur_n))
        self._state = None
        self._lr_host = None

    def _lazy_state(self):
        if self._state is not None:
            return
        dev = self._segs[0][0].device
        self._m = [torch.zeros(n, dtype=torch.float32, device=dev) for _, _, _, n in self._segs]
        self._v = [torch.zeros(n, dtype=torch.float32, device=dev) for _, _, _, n in self._segs]
        self._state = torch.zeros(4, dtype=torch.int32, device=dev)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)

    def reset_state(self):
        """Forget the moments and the step counter (as a freshly constructed optimizer)."""
        self._state = None
        self._lr_host = None
        self._grads_clean = False

    def sync_lr(self):
        """Push param_groups[0]['lr'] to the device scalar (call outside graph capture)."""
        self._lazy_state()
        lr = self.param_groups[0]["lr"]
        if lr != self._lr_host:
            self._lr_dev.fill_(lr)
            self._lr_host = lr

    def zero_grad(self, set_to_none=False):
        if self._grads_clean:       # step() zeroed what it consumed and nobody has written since
            self._grads_clean = False
            return
        for _, ag, off, n in self._segs:
            ag[off: off + n].zero_()

    def take_zeroing(self):
        """The gradient buffers the next zero_grad() would clear, for a caller that clears them itself RIGHT NOW on this stream
        (ops.begin_step(also_zero=...): the step's one zeroing launch); that zero_grad() then does nothing."""
        if self._grads_clean:
            return []
        self._grads_clean = True
        return [ag[off: off + n] for _, ag, off, n in self._segs]

    def _fused_plan(self):
        """(nets, conv ranges relative to the segment, derived buffers) when the whole optimizer is ONE arena segment whose networks
        keep their derived weight copies at the arena's offsets -- then Adam, the three copies and the gradient zeroing are one launch
        (sgan_adam_pack).  None: several segments (AdamGroups of networks with separate storage) -> sgan_adam_multi + lazy repack."""
        if len(self._segs) != 1 or _NO_ADAM_PACK:
            return None
        ap, ag, off, n = self._segs[0]
        nets = list(self._nets.values())
        if not all(hasattr(net, "_refresh_derived") and net._arena[0] is ap for net in nets):
            return None
        for net in nets:
            net._refresh_derived()       # storage exists and every copy is current (a no-op in steady state)
        bufs = {(net._flat_t.data_ptr() - 4 * net._arena[2], net._pk_f.data_ptr() - 4 * net._arena[2],
                 net._pk_b.data_ptr() - 4 * net._arena[2], net._pk_bh.data_ptr() - 4 * net._arena[2]) for net in nets}
        if len(bufs) != 1:               # the copies of the networks are not slices of one arena-wide buffer
            return None
        net0 = nets[0]
        base0 = net0._arena[2]
        der = getattr(ap, "_sgan_derived", None)
        if der is not None and net0._flat_t.data_ptr() == der[0].data_ptr() + 4 * base0:
            views = tuple(d[off: off + n] for d in der)
        elif len(nets) == 1 and base0 == 0:
            views = tuple(d[off: off + n] for d in (net0._flat_t, net0._pk_f, net0._pk_b, net0._pk_bh))
        else:
            return None
        segs = []
        for net in nets:
            for o, taps, co, ci in net._conv_segments(net._arena[2]):
                if o >= off and o + taps * co * ci <= off + n:
                    segs.append((o - off, taps, co, ci))
                elif o + taps * co * ci > off and o < off + n:
                    return None          # a conv weight only partly inside the segment
        segs.sort()
        if len(segs) > 64:
            return None
        return nets, segs, views

    def segments(self):
        """[(params, grads)] flat views -- what the data-parallel all-reduce works on."""
        return [(ap[off: off + n], ag[off: off + n]) for ap, ag, off, n in self._segs]

    @torch.no_grad()
    def step(self):
        self._lazy_state()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        g = self.param_groups[0]
        plan = self._fused_plan() if type(self) is FusedAdam else None
        if plan is not None:
            nets, csegs, (ft, pf, pb, pbh) = plan
            ap, ag, off, n = self._segs[0]
            ops.adam_pack(ap[off: off + n], ag[off: off + n], self._m[0], self._v[0], self._lr_dev, g["betas"][0], g["betas"][1], g["eps"],
                          self._state, ft, pf, pb, pbh, csegs, self._zero_in_step)
            for net in nets:                # the derived copies were written by the same launch: they are current
                net._wt_key = net._derived_key()
            self._grads_clean = self._zero_in_step
            return
        segs = [(ap[off: off + n], ag[off: off + n], m, v, n)
                for (ap, ag, off, n), m, v in zip(self._segs, self._m, self._v)]
        ops.adam_multi(segs, self._lr_dev, g["betas"][0], g["betas"][1], g["eps"], self._state)
        for net in self._nets.values():     # the transposed weight copy backward-data reads is stale now
            net._wt_epoch = getattr(net, "_wt_epoch", 0) + 1

    @property
    def step_count(self):
        return int(self._state[0].item()) if self._state is not None else 0


class FusedSGD(FusedAdam):
    """`torch.optim.SGD(params, lr=, momentum=)` over the flat storage (one sgan_sgd_multi launch): the second update rule of the
    boundary.  Same segment handling, LR-in-device-memory and derived-weight-copy invalidation as FusedAdam."""

    def __init__(self, params, lr=1e-3, momentum=0.0):
        super().__init__(params, lr=lr)
        self.param_groups[0]["momentum"] = float(momentum)

    def _lazy_state(self):
        if self._state is not None:
            return
        dev = self._segs[0][0].device
        mu = self.param_groups[0]["momentum"]
        self._m = [torch.zeros(n, dtype=torch.float32, device=dev) if mu else None for _, _, _, n in self._segs]
        self._state = torch.zeros(4, dtype=torch.int32, device=dev)
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)

    @torch.no_grad()
    def step(self):
        self._lazy_state()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        segs = [(ap[off: off + n], ag[off: off + n], m, n) for (ap, ag, off, n), m in zip(self._segs, self._m)]
        ops.sgd_multi(segs, self._lr_dev, self.param_groups[0]["momentum"])
        for net in self._nets.values():
            net._wt_epoch = getattr(net, "_wt_epoch", 0) + 1


class AdamGroups:
    """`torch.optim.Adam([{'name':, 'params':, 'lr':}, ...], betas=)` as the two-stage trainers build it
    (models/twostage_cycle_model.py:141-144): one FusedAdam per named group, stepped together."""

    def __init__(self, groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.optimizers = []
        self.param_groups = []
        for g in groups:
            o = FusedAdam(g["params"], lr=g.get("lr", lr), betas=betas, eps=eps)
            o.param_groups[0]["name"] = g.get("name")
            self.optimizers.append(o)
            self.param_groups.append(o.param_groups[0])

    def zero_grad(self, set_to_none=False):
        for o in self.optimizers:
            o.zero_grad()

    def take_zeroing(self):
        return [t for o in self.optimizers for t in o.take_zeroing()]

    def step(self):
        for o in self.optimizers:
            o.step()

    def sync_lr(self):
        for o in self.optimizers:
            o.sync_lr()

    def reset_state(self):
        for o in self.optimizers:
            o.reset_state()

    def segments(self):
        return [s for o in self.optimizers for s in o.segments()]

    @property
    def step_count(self):
        return self.optimizers[0].step_count

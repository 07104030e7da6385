"""Helpers shared by the `-m gpu` parity tests: NCHW (CPU, torch reference) <-> padded NHWC (GPU)."""
import torch

from supervised_gan_amd.ops import pad4


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def to_buf(t, dev="cuda"):
    """[1, C, H, W] CPU tensor -> [H, W, pad4(C)] GPU buffer (zero padded)."""
    _, C, H, W = t.shape
    buf = torch.zeros(H, W, pad4(C), dtype=torch.float32)
    buf[..., :C] = t[0].permute(1, 2, 0)
    return buf.to(dev)


def from_buf(buf, C):
    """[H, W, Cs] GPU buffer -> [1, C, H, W] CPU tensor."""
    return buf.detach().cpu()[..., :C].permute(2, 0, 1).unsqueeze(0).contiguous()


def master_weight(w, transposed, dev="cuda"):
    """torch conv weight ([Cout,Cin,k,k], or [Cin,Cout,k,k] for ConvTranspose2d) -> master
    [k*k][Cout_s][Cin_s] flat GPU tensor."""
    if transposed:
        cin, cout, k, _ = w.shape
        wm = w.permute(2, 3, 1, 0)      # [k,k,Cout,Cin]
    else:
        cout, cin, k, _ = w.shape
        wm = w.permute(2, 3, 0, 1)
    m = torch.zeros(k, k, pad4(cout), pad4(cin), dtype=torch.float32)
    m[:, :, :cout, :cin] = wm
    return derived_copies(m.reshape(-1).to(dev), k, pad4(cout), pad4(cin))[0]


def derived_copies(wm, k, cout_s, cin_s):
    """(wm, wt): the master weight tagged with its split-bf16 forward copy, and the transposed copy [tap][Cin][Cout] tagged with
    the split-bf16 backward copy -- what ChainNet._wb / _wt hand to the conv calls (one sgan_pack_weights launch)."""
    from supervised_gan_amd import ops
    wt, pf, pb, pbh = torch.zeros_like(wm), torch.zeros_like(wm), torch.zeros_like(wm), torch.zeros_like(wm)
    ops.pack_weights(wm, wt, pf, pb, [(0, k * k, cout_s, cin_s)], pbh)
    ops.with_packed(wm, pf)
    ops.with_packed(wt, pb, pbh)
    wm._sgan_wt = wt
    return wm, wt


def from_master(m, k, cin, cout, transposed):
    m = m.detach().cpu().view(k, k, pad4(cout), pad4(cin))[:, :, :cout, :cin]
    return (m.permute(3, 2, 0, 1) if transposed else m.permute(2, 3, 0, 1)).contiguous()


def pad_vec(v, dev="cuda"):
    out = torch.zeros(pad4(v.numel()), dtype=torch.float32)
    out[: v.numel()] = v
    return out.to(dev)


def stats_of(x, dev="cuda"):
    """(sum, sumsq) over H,W of a [1,C,H,W] tensor as the kernels accumulate them: double[2*Cs]."""
    C = x.shape[1]
    Cs = pad4(C)
    st = torch.zeros(2 * Cs, dtype=torch.float64)
    xd = x.double()
    st[:C] = xd.sum((0, 2, 3))
    st[Cs: Cs + C] = (xd * xd).sum((0, 2, 3))
    return st.to(dev)

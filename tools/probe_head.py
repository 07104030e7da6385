"""Timing of the PatchGAN-head backward (sg_head_bwd_kernel) against the two generic launches it replaces (tuning instrument)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from supervised_gan_amd import ops, _lib
from hip_utils import derived_copies


def timeit(fn, n=20):
    """GPU time per call: n calls captured into one hipGraph (no host launch overhead), replayed."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


def jobs(sizes, C=256, k=4, p=2):
    dj, wj = [], []
    for H in sizes:
        Ho = H + 2 * p - k + 1
        desc = ops.conv_desc(0, k, 1, p, H, H, C, Ho, Ho, 4, C, 1)
        x = torch.randn(H, H, C, device="cuda"); r = torch.zeros(Ho, Ho, 4, device="cuda"); r[..., 0].normal_()
        w = torch.randn(k * k * 4 * C, device="cuda") * 0.05
        wm, wt = derived_copies(w, k, 4, C)
        wt._wm = wm
        dw = torch.zeros_like(w); db = torch.zeros(4, device="cuda")
        st = torch.zeros(2 * C, dtype=torch.float64, device="cuda"); st[C:] = H * H
        nrm = ops.norm_desc(st, None, None, H * H, 1e-5, 2, 0.2)
        sums = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
        din = torch.empty(H, H, C, device="cuda")
        dj.append((desc, r, wt, din, x, nrm, sums, 0, False, True, 0))
        wj.append((desc, x, nrm, r, dw, db))
    return dj, wj


if __name__ == "__main__":
    C = int(os.environ.get("PROBE_C", "256"))
    cases = (("n=6", [66, 34, 18] * 2), ("n=3", [66, 34, 18]))
    if os.environ.get("PROBE_SIZES"):      # e.g. PROBE_SIZES=66,66,34,34
        sz = [int(v) for v in os.environ["PROBE_SIZES"].split(",")]
        cases = ((f"C{C} {sz}", sz),)
    for name, sizes in cases:
        dj, wj = jobs(sizes, C)
        os.environ.pop("SGAN_NO_HEAD_BWD", None)
        t = timeit(lambda: ops.conv_bwd_grouped(dj, wj))
        print(f"head {name}  one launch   {t:7.1f} us  {_lib.lib().sgan_last_kernel().decode()}")
        fj = [(d[0], d[4], d[5], wm_, None, torch.empty(d[0].Hout, d[0].Wout, 4, device="cuda"), None, 0, 0) for d, wm_ in zip(dj, [j[2]._wm for j in dj])]
        tfw = timeit(lambda: ops.conv_fwd_grouped(fj, 0))
        print(f"head {name}  forward      {tfw:7.1f} us  {_lib.lib().sgan_last_kernel().decode()}")
        if os.environ.get("PROBE_GENERIC"):
            t1 = timeit(lambda: ops.conv_dgrad_grouped(dj)); k1 = _lib.lib().sgan_last_kernel().decode()
            t2 = timeit(lambda: ops.conv_wgrad_grouped(wj)); k2 = _lib.lib().sgan_last_kernel().decode()
            print(f"head {name}  generic      {t1:7.1f} us {k1}  +  {t2:7.1f} us {k2}")

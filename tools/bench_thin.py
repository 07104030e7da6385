"""Timing of the thin-layer weight-gradient launches of the fcgan step (tuning instrument)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from supervised_gan_amd import ops, _lib

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

def jobs_conv(kind, k, s, p, cin, cout, sizes, norm, lcin=0, lcout=0):
    jobs = []
    for H in sizes:
        Ho = (H - 1) * s - 2 * p + k if kind else (H + 2 * p - k) // s + 1
        desc = ops.conv_desc(kind, k, s, p, H, H, cin, Ho, Ho, cout, lcin or cin, lcout or cout)
        x = torch.randn(H, H, cin, device="cuda"); r = torch.randn(Ho, Ho, cout, device="cuda")
        dw = torch.zeros(k * k * cin * cout, device="cuda"); db = torch.zeros(cout, device="cuda")
        nrm = None
        if norm:
            st = torch.zeros(2 * cin, dtype=torch.float64, device="cuda"); st[cin:] = H * H
            nrm = ops.norm_desc(st, None, None, H * H, 1e-5, 2, 0.2)
        jobs.append((desc, x, nrm, r, dw, db))
    return jobs

cases = {"D0 4->32 n=6": jobs_conv(0, 4, 2, 2, 4, 32, [512, 256, 128] * 2, False, lcin=2),
         "Dhead 256->4 n=6": jobs_conv(0, 4, 1, 2, 256, 4, [66, 34, 18] * 2, True, lcout=1),
         "G5 T 32->4": jobs_conv(1, 4, 2, 1, 32, 4, [256], True, lcout=2),
         "D0 4->32 n=3": jobs_conv(0, 4, 2, 2, 4, 32, [512, 256, 128], False, lcin=2),
         "Dhead 256->4 n=3": jobs_conv(0, 4, 1, 2, 256, 4, [66, 34, 18], True, lcout=1)}
if __name__ == "__main__":
  for want in sys.argv[1:] or ["1024"]:
      w, mp = want.split(":") if ":" in want else (want, "256")
      os.environ["SGAN_THIN_WANT"] = w; os.environ["SGAN_THIN_MINPIX"] = mp
      for name, jobs in cases.items():
          t = timeit(lambda: ops.conv_wgrad_grouped(jobs))
          print(f"want {w:>5s} minpix {mp:>4s}  {name:18s} {t:8.1f} us  {_lib.lib().sgan_last_kernel().decode()}")
  os.environ["SGAN_NO_THIN_WGRAD"] = "1"
  for name, jobs in cases.items():
      t = timeit(lambda: ops.conv_wgrad_grouped(jobs))
      print(f"old kernel            {name:18s} {t:8.1f} us  {_lib.lib().sgan_last_kernel().decode()}")

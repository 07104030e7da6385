"""Do two under-filled kernels on two captured streams overlap in a hipGraph replay?  (tuning probe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from supervised_gan_amd import ops
from hip_utils import derived_copies

ops.set_math("bf16x3")
os.environ["SGAN_IGEMM3P"] = "0"
k, cin, cout, H = 4, 256, 256, 16


def make():
    w = torch.randn(k * k * cout * cin, device="cuda") * 0.05
    wm, wt = derived_copies(w, k, cout, cin)
    x = torch.randn(H, H, cin, device="cuda"); y = torch.empty(2 * H, 2 * H, cout, device="cuda")
    desc = ops.conv_desc(1, k, 2, 1, H, H, cin, 2 * H, 2 * H, cout)
    return lambda: ops.conv_fwd(desc, x, None, wm, None, y), (w, wm, wt, x, y, desc)


f1, k1 = make(); f2, k2 = make()
side = torch.cuda.Stream()


def run(parallel, n=10):
    cur = torch.cuda.current_stream()
    for _ in range(n):
        if parallel:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                f2()
            f1()
            cur.wait_stream(side)
        else:
            f1(); f2()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for parallel in (False, True):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(parallel); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            run(parallel)
        t = min(timed(g.replay) for _ in range(5))
        te = min(timed(lambda: run(parallel)) for _ in range(5))
    print(f"parallel={parallel}: graph replay {t:.1f} us, eager {te:.1f} us for 10 x 2 launches")

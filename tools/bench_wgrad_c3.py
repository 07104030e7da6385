import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from supervised_gan_amd import ops, _lib
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_thin import timeit, jobs_conv
cases = {"c3 128->256 n=6": jobs_conv(0, 4, 1, 2, 128, 256, [65, 33, 17] * 2, True),
         "c3 128->256 n=3": jobs_conv(0, 4, 1, 2, 128, 256, [65, 33, 17], True),
         "c2 64->128 n=6": jobs_conv(0, 4, 2, 2, 64, 128, [129, 65, 33] * 2, True)}
for want in sys.argv[1:]:
    os.environ["SGAN_WGRAD_WANT"] = want
    for name, jobs in cases.items():
        t = timeit(lambda: ops.conv_wgrad_grouped(jobs))
        print(f"want {want:>5s} {name:18s} {t:8.1f} us  {_lib.lib().sgan_last_kernel().decode()}")

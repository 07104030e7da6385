"""Diagnostics: per-workgroup phase times of sg_wgrad_thin_kernel from a -DSGTHIN_STAMP build (tools/stamp_thin.sh)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from supervised_gan_amd import _lib, ops  # noqa: E402
sys.argv = sys.argv[:1]
import bench_thin as BT  # noqa: E402

lib = _lib.lib()
lib.sgan_debug_stamps_thin.argtypes = [C.c_void_p, C.c_int]
for name, jobs in BT.cases.items():
    if "G5" in name:
        jobs = [j[:5] + (None,) for j in jobs]      # the fcgan generator's last layer has no bias
    z = np.zeros(8 * 4096, dtype=np.uint64)
    for _ in range(5):
        ops.conv_wgrad_grouped(jobs)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 4096, dtype=np.uint64)
    assert lib.sgan_debug_stamps_thin(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(4096, 8)
    st = st[(st[:, 0] > 0) & (st[:, 4] > 0)].astype(np.int64)
    d = np.diff(st[:, :5], axis=1)
    print(name, "workgroups stamped (tile workgroups that ran to the end):", len(st))
    for i, n in enumerate(["prologue", "pixel loop", "LDS combine", "atomics"]):
        print(f"  {n:14s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f} cycles")
    rt0 = st[:, 6].min()
    start, end = (st[:, 6] - rt0) / 100.0, (st[:, 7] - rt0) / 100.0
    print(f"  start (us) p50 {np.median(start):.1f} max {start.max():.1f}; end max {end.max():.1f}; WG duration median {np.median(end - start):.1f} us")

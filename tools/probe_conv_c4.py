"""Isolated timing of the first PatchGAN conv (2 -> 32 channels, k4 s2 p2, three scales, fake + real: six problems in one launch), 8 launches per hipGraph replay:
python tools/probe_conv_c4.py    (SGAN_NO_C4=1: the generic exact-fp32 kernel; SGAN_C4_RB=1|2|4: row blocks per wave)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from supervised_gan_amd import ops, _lib
from hip_utils import master_weight, pad_vec, to_buf
g = torch.Generator().manual_seed(1)
jobs = []
w = master_weight(torch.randn(32, 2, 4, 4, generator=g) * 0.05, False)
b = pad_vec(torch.randn(32, generator=g))
for H in (512, 256, 128, 512, 256, 128):
    Ho = (H + 4 - 4) // 2 + 1
    x = to_buf(torch.randn(1, 2, H, H, generator=g))
    out = torch.empty(Ho, Ho, 32, device="cuda")
    jobs.append((ops.conv_desc(0, 4, 2, 2, H, H, 4, Ho, Ho, 32, 2, 32), x, None, w, b, out, None))
ops.conv_fwd_grouped(jobs)
print(_lib.lib().sgan_last_kernel().decode())
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(8):
        ops.conv_fwd_grouped(jobs)
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 8 * 1e3)
print(f"{best:.1f} us per launch (6 problems)")

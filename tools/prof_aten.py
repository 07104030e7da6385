"""Which ATen kernels the fcgan step still launches, with the Python lines that asked for them (tuning instrument)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

args = argparse.Namespace(n_update_G=2, skip_wasted_D_wgrad=False, no_d_streams=False, no_group=False)
torch.cuda.set_device(0)
model = bench.build_model(args, 0)
ring = bench.synthetic_ring(4, 0, torch.device("cuda", 0))
for i in range(3):
    model.set_input(ring[i])
    model.optimize_parameters()
torch.cuda.synchronize()
import collections, traceback
_cnt = collections.Counter()
def _spy(name):
    orig = getattr(torch, name)
    def f(*a, **k):
        fr = traceback.extract_stack(limit=3)[0]
        _cnt[(name, fr.filename.split("/")[-1], fr.lineno)] += 1
        return orig(*a, **k)
    setattr(torch, name, f)
for nm in ("zeros", "ones_like", "zeros_like", "empty"):
    _spy(nm)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.set_input(ring[3])
    model.optimize_parameters()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    if not any(k.device_time > 0 for k in [ev] ) and ev.device_time_total <= 0:
        continue
    st = [s for s in (ev.stack or []) if "supervised" in s or "bench" in s][:2]
    key = (ev.name, " <- ".join(s.split("/")[-1] for s in st))
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1
    r[1] += ev.device_time_total
for (name, st), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:3d} {us:8.1f} us  {name:28s} {st}")
print(_cnt)

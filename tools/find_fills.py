"""Which Python lines launch aten fill / copy / elementwise kernels during one eager fcgan step (diagnostic; the hipGraph replays the
same launches): python tools/find_fills.py"""
import argparse, collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.utils._python_dispatch import TorchDispatchMode

ns = argparse.Namespace(n_update_G=2, skip_wasted_D_wgrad=False, no_d_streams=False, no_group=False)
torch.cuda.set_device(0)
m = bench.build_model(ns, 0)
ring = bench.synthetic_ring(4, 0, torch.device("cuda", 0))
for i in range(3):
    m.set_input(ring[i]); m.optimize_parameters()
seen = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("fill", "zero", "copy", "ones", "add", "mul", "clone", "cat", "index")):
            fr = [f for f in traceback.extract_stack() if "supervised" in f.filename and "_python_dispatch" not in f.filename]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:])
            seen[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    m.set_input(ring[3]); m.optimize_parameters()
torch.cuda.synchronize()
for (n, w), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d} {n:40s} {w}")

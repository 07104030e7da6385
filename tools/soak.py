"""Soak run of a graphed step (fcgan, cgan or twostage_cycle): N steps, losses must stay finite and device memory flat (diagnostic)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from supervised_gan_amd.graph_step import GraphedStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--workload", default="fcgan", choices=["fcgan", "cgan", "twostage_cycle"])
a = ap.parse_args()
args = argparse.Namespace(n_update_G=2, skip_wasted_D_wgrad=False, no_d_streams=False, no_group=False)
torch.cuda.set_device(0)
model = {"fcgan": bench.build_model, "cgan": bench.build_cgan, "twostage_cycle": bench.build_twostage}[a.workload](args, 0)
ring = bench.synthetic_ring(64, 0, torch.device("cuda", 0))
gs = GraphedStep(model)
gs.capture(ring[0])
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated()
for i in range(a.steps):
    gs.step(ring[i % len(ring)])
    if i % 250 == 0 or i == a.steps - 1:
        e = model.get_current_errors()
        ok = all(v == v and abs(v) < 1e6 for v in e.values())
        print(i, {k: round(v, 4) for k, v in e.items()}, "mem MiB", torch.cuda.memory_allocated() >> 20, flush=True)
        assert ok, e
torch.cuda.synchronize()
assert torch.cuda.memory_allocated() <= mem0 + (160 << 20), (mem0, torch.cuda.memory_allocated())   # the 50-image pool is 100 MiB
nets = [model.netG1, model.netG2, model.netF2] + model.netD1 + model.netD2 if a.workload == "twostage_cycle" else [model.netG] + model.netD
assert all(torch.isfinite(n._flat).all() for n in nets)
print("soak OK")

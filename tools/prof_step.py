"""A few eager fcgan steps for rocprofv3 (per-dispatch kernel trace)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse, torch
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--n_update_G", type=int, default=2)
ap.add_argument("--skip_wasted_D_wgrad", action="store_true"); ap.add_argument("--no_d_streams", action="store_true")
a = ap.parse_args()
m = bench.build_model(a, 0)
ring = bench.synthetic_ring(4, 0, torch.device("cuda", 0))
for i in range(2 + a.steps):
    if i == 2:
        torch.cuda.synchronize(); print("PROFILE_STEPS_BEGIN", flush=True)
    m.set_input(ring[i % 4]); m.optimize_parameters()
torch.cuda.synchronize()

"""cProfile of the eager fcgan step: where the host time goes (diagnostic)."""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

args = argparse.Namespace(n_update_G=2, skip_wasted_D_wgrad=False, no_d_streams=False, no_group=False)
torch.cuda.set_device(0)
model = bench.build_model(args, 0)
ring = bench.synthetic_ring(8, 0, torch.device("cuda", 0))
for i in range(5):
    model.set_input(ring[i % 8])
    model.optimize_parameters()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(40):
    model.set_input(ring[i % 8])
    model.optimize_parameters()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)

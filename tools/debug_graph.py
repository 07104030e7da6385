import faulthandler, sys, os, time
faulthandler.dump_traceback_later(int(os.environ.get("DUMP_AFTER", "70")), exit=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
t0 = time.time()
def log(*a):
    print(f"[{time.time()-t0:7.2f}s]", *a, flush=True)
import sgan_oracle as O
from test_hip_step import build_model, real3
small = len(sys.argv) > 1 and sys.argv[1] == "small"
cfg = O.FCGANConfig(ngf=8, ndf=8, noiseSize=2) if small else O.FCGANConfig()
m = build_model(cfg, 0)
m.noise_source = None
log("model built")
data = {"A": real3(cfg, 0).cuda(), "A_paths": ["x"]}
for i in range(3):
    m.set_input(data); m.optimize_parameters()
torch.cuda.synchronize(); log("eager steps ok", m.get_current_errors())
from supervised_gan_amd.graph_step import GraphedFCGANStep
from supervised_gan_amd import ops
gs = GraphedFCGANStep(m, warmup_steps=1)
# capture piecewise with logging
m.set_input(data); m.optimize_parameters(); m.optimizer_D.sync_lr(); m.optimizer_G.sync_lr(); torch.cuda.synchronize()
H = cfg.fineSize
gs.fake_for_D = torch.zeros((H, H, 4), device="cuda")
m._pool_override = ops.logical_view(gs.fake_for_D, 2)
gA = torch.cuda.CUDAGraph()
log("begin capture A")
with torch.cuda.graph(gA):
    m.forward()
log("captured A")
gA.replay(); torch.cuda.synchronize(); log("replayed A")
fakeA = m.fake
fns = []
for item in gs._program():
    if not isinstance(item, str):
        fns += item
log("begin capture B with", len(fns), "callables")
gB = torch.cuda.CUDAGraph()
with torch.cuda.graph(gB, pool=gA.pool()):
    for f in fns:
        f()
log("captured B")
gB.replay(); torch.cuda.synchronize(); log("replayed B", m.get_current_errors())
for it in range(20):
    gA.replay()
    q = m.fake_pool.query(fakeA)
    gs.fake_for_D.copy_(ops.as_nhwc(q))
    gB.replay()
torch.cuda.synchronize(); log("20 replays ok", m.get_current_errors())
t1 = time.time()
for it in range(50):
    gA.replay()
    q = m.fake_pool.query(fakeA)
    gs.fake_for_D.copy_(ops.as_nhwc(q))
    gB.replay()
torch.cuda.synchronize(); log("50 steps: ms/step", (time.time() - t1) * 1e3 / 50, m.get_current_errors())

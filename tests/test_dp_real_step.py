"""A REAL data-parallel step on the GPU path, two ranks: the discriminator gradients each rank holds after the gradient
all-reduce equal the mean of the two ranks' own (un-reduced) gradients of the same fcgan D step, and both ranks hold the same
bits.  With >= 2 GPUs the ranks take one card each over RCCL (backend "nccl"); on the one-GPU box they share the card and the
collective runs over gloo -- the same GradAverager / flat-segment path either way (models/networks.py:536-539 is the
reference's seam: nn.parallel.data_parallel)."""
import os
import sys

import pytest
import torch

from test_dp_gloo import ROOT, by_value, run_ranks

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import sgan_oracle as O
        from supervised_gan_amd import dist as sdist
        from test_hip_step import build_model, real3
        multi = torch.cuda.device_count() >= world
        dev = rank if multi else 0
        torch.cuda.set_device(dev)
        sdist.init_from_env(backend="nccl" if multi else "gloo")
        cfg = O.FCGANConfig(ngf=8, ndf=8, noiseSize=2, n_update_G=1)
        m = build_model(cfg, 10 * rank, extra=("--gpu_ids", str(dev)))      # own latent stream per rank
        sdist.broadcast_parameters([m.netG] + m.netD)
        m.set_input({"A": real3(cfg, rank), "A_paths": ["synthetic"]})        # own sample per rank
        m.forward()
        m.optimizer_D.zero_grad()
        m.backward_D()
        seg = m.optimizer_D.segments()
        assert len(seg) == 1
        local = seg[0][1].detach().clone()
        avg = sdist.GradAverager()
        avg(m.optimizer_D)
        torch.cuda.synchronize()
        q.put((rank, by_value({"local": local, "synced": seg[0][1], "backend": torch.distributed.get_backend(),
                               "bytes": avg.bytes, "weights": torch.cat([d._flat.detach() for d in m.netD])})))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except BaseException:      # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise


@pytest.mark.timeout(600)
def test_dp_step_gradients_are_the_mean_of_the_rank_gradients():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need an MI355X; no CUDA/HIP device is visible")
    res = run_ranks(_worker, 2, timeout=400)
    a, b = res[0], res[1]
    assert torch.equal(a["weights"], b["weights"])                       # broadcast
    assert float((a["local"] - b["local"]).abs().max()) > 0               # the ranks really saw different samples
    mean = (a["local"].double() + b["local"].double()) / 2
    scale = float(mean.abs().max())
    assert float((a["synced"].double() - mean).abs().max()) <= 1e-6 * scale
    assert torch.equal(a["synced"], b["synced"])                          # every rank steps with the same bits
    assert a["bytes"] == a["local"].numel() * 4
    print("backend:", a["backend"], "ranks on", "separate GPUs" if torch.cuda.device_count() >= 2 else "one shared GPU")

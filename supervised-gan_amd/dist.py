"""Data parallelism for the G-step / D-step loop: one process per GPU, RCCL over xGMI.

The reference's only multi-GPU mechanism is single-process nn.parallel.data_parallel
(models/networks.py:536-539,844-847), which cannot split its batchSize=1.  Here every rank runs the
whole bs=1 step on its own sample (own latent stream, own ImagePool, per-replica BatchNorm statistics
exactly as data_parallel replicas would have) and the ranks exchange one thing: the flat gradient
buffer of the optimizer about to step, averaged with a single all-reduce (BCELoss is a mean, so the
global-batch gradient is the mean of the per-rank gradients).  fcgan: 3 x 693,729 discriminator
gradients in one grouped call (8.3 MB fp32), 1,772,448 generator gradients (7.1 MB) per G update."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC between the ranks of a host (read when HSA starts)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SGAN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_parameters(nets, src=0):
    """Identical initial weights on every rank (flat storage, BN buffers, gauss filters)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for net in nets:
        flat = getattr(net, "_flat", None)
        if flat is not None:
            dist.broadcast(flat, src)
            if hasattr(net, "invalidate_derived"):   # c10d writes the buffer without moving a version counter
                net.invalidate_derived()
        for b in net.buffers():
            dist.broadcast(b, src)
        for p in getattr(net, "_extra_parameters", lambda: [])():
            dist.broadcast(p.data, src)


class GradAverager:
    """callable(optimizer): all-reduce(mean) of the optimizer's flat gradient segments."""

    def __init__(self):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.calls = 0
        self.bytes = 0

    def __call__(self, optimizer):
        if self.world == 1:
            return
        for _, g in optimizer.segments():     # fcgan: exactly one segment per optimizer (pack_flat)
            if dist.get_backend() == "nccl":
                dist.all_reduce(g, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
                g.mul_(1.0 / self.world)
            self.bytes += g.numel() * 4
        self.calls += 1

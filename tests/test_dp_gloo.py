"""Data-parallel path on CPU: world_size 2, gloo.  The kernels need a GPU, the N>1 host logic does not:
identical initial weights after broadcast, one contiguous gradient segment per optimizer (pack_flat),
and all-reduced gradients == mean of the per-rank gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def by_value(out):
    """Tensors in a result dict as numpy arrays: pickled by value.  A torch tensor on an mp queue travels as a shared-memory
    handle that dies with the sending rank, which may have exited before the parent reads it."""
    return {k: v.detach().cpu().numpy() if torch.is_tensor(v) else v for k, v in out.items()}


RENDEZVOUS_ERRORS = ("address already in use", "eaddrinuse", "connection refused", "connection reset", "failed to bind")


def _worker(rank, world, port, q):
    """Everything a rank does; any exception travels to the parent as ("error", traceback) -- a crash is never swallowed."""
    try:
        _rank_body(rank, world, port, q)
    except BaseException:      # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))
        raise


def _rank_body(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from supervised_gan_amd import dist as sdist
    from supervised_gan_amd import networks as N
    from supervised_gan_amd.optim import FusedAdam
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                     # different init per rank on purpose
    nets = [N.define_D(2, 8, "n_layers", n_layers_D=3, norm="instance", use_sigmoid=True, scale_factor=s) for s in (1, 2, 4)]
    G = N.define_G(2, 0, 8, "fcgan", "instance", False, n_layers_G=5, use_fcn=True, noise_nc=8)
    N.pack_flat(nets)
    sdist.broadcast_parameters([G] + nets)
    opt_D = FusedAdam([p for d in nets for p in d.model.parameters()], lr=2e-4, betas=(0.5, 0.999))
    opt_G = FusedAdam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    assert len(opt_D.segments()) == 1 and len(opt_G.segments()) == 1       # one all-reduce per optimizer
    assert opt_D.segments()[0][1].numel() == sum(d._nflat for d in nets)
    # rank-dependent "gradients" written through the parameter .grad views
    for i, p in enumerate(p for d in nets for p in d.model.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    avg = sdist.GradAverager()
    avg(opt_D)
    avg(opt_G)
    out = {
        "w": torch.cat([d._flat.clone() for d in nets] + [G._flat.clone()]),
        "g": [float(p.grad.flatten()[0]) for d in nets for p in d.model.parameters()],
        "bytes": avg.bytes, "calls": avg.calls,
        "gauss": nets[1].state_dict()["gauss_filter.0.weight"].clone(),
    }
    q.put((rank, by_value(out)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def run_ranks(worker, world, timeout=180):
    """Start `world` spawned ranks on a fresh loopback port and collect one result dict per rank.  Retried ONCE, and only when a
    rank reports a bind / rendezvous error (the port race of a busy host); any other failure -- a worker exception, a non-zero
    exit code, a missing result -- fails the test with the worker's traceback."""
    ctx = mp.get_context("spawn")
    for attempt in range(2):
        port = _free_port()
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res, lost = {}, None
        try:
            for _ in range(world):
                r, out = q.get(timeout=timeout)
                res[r] = {k: torch.from_numpy(v) if isinstance(v, np.ndarray) else v for k, v in out.items()}
                if "error" in out:
                    break
        except Exception as e:      # noqa: BLE001  (queue.Empty: a rank died without a word)
            lost = repr(e)
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
        errors = [o["error"] for o in res.values() if "error" in o]
        if errors:
            if attempt == 0 and all(any(k in e.lower() for k in RENDEZVOUS_ERRORS) for e in errors):
                continue
            pytest.fail("worker failed:\n" + "\n".join(errors))
        assert len(res) == world and all(p.exitcode == 0 for p in procs), (sorted(res), [p.exitcode for p in procs], lost)
        return res
    pytest.fail("rendezvous failed twice")


@pytest.mark.timeout(300)
def test_two_rank_gloo_broadcast_and_grad_average():
    world = 2
    ctx = mp.get_context("spawn")
    res = run_ranks(_worker, world)
    assert torch.equal(res[0]["w"], res[1]["w"])                     # broadcast: identical weights
    assert torch.equal(res[0]["gauss"], res[1]["gauss"])
    n = len(res[0]["g"])
    expect = [1.5 * (i + 1) for i in range(n)]                        # mean of (1, 2) * (i + 1)
    assert res[0]["g"] == expect and res[1]["g"] == expect
    assert res[0]["calls"] == 2 and res[0]["bytes"] == res[0]["w"].numel() * 4

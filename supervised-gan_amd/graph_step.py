"""hipGraph capture of FCGANModel.optimize_parameters (models/fcgan_model.py:178-193) and
CGANModel.optimize_parameters (models/cgan_model.py:212-226).

A bs=1 step is a few hundred short kernels; launched eagerly from Python the host is the bottleneck.
The step is therefore captured once into hipGraphs and replayed:

    graph A : forward()                      latent fill + G forward -> static `fake` (cgan: cat(real_A, fake_B))
    host    : ImagePool.query(fake)          the reference's python-random history policy, one D2D copy
    graph B : D step, then the G step(s)     zero_grad / backward / Adam, loss scalars left on device

Everything that changes from step to step lives in device memory the kernels read and advance
themselves (Adam step counter and LR, Philox offset, BatchNorm num_batches_tracked), so replays are
faithful.  With data parallelism the gradient all-reduce is not captured: graph B is cut at the
two/three synchronisation points and RCCL runs between the pieces on the same stream.

Prefetch (fcgan with n_update_G > 1): the re-draw that ends a step and the forward() that opens the next one are two generator
passes over the same weights, each a chain of launches too small to fill the card (~100 us for 4 GFLOP).  The graphed step runs
them as ONE two-problem pass at the end of graph B (FCGANModel.sample_noise_and_prefetch -> chain.forward_pair) which writes the
second problem into the buffers of the forward the captured backward was built on; graph A disappears.  Same latents in the same
order, same BatchNorm running-statistics updates in the same order.  SGAN_NO_G_PREFETCH=1 switches it off."""
import os

import torch

from . import ops


class GraphedStep:
    """Works on any trainer exposing optimizer_D/G, backward_D/G, forward, sample_noise, fake_pool, `_pool_source()`
    (what the reference feeds ImagePool.query) and `_pool_override`; or, for trainers with their own step structure
    (the two-stage models), `graph_spec()` -> dict(pools, sources, set_overrides, program)."""

    def __init__(self, model, warmup_steps=2):
        self.m = model
        opt = model.opt
        assert hasattr(model, "graph_spec") or opt.n_update_D == 1, "graphed step supports n_update_D == 1 (every README recipe)"
        assert opt.batchSize == 1
        self._captured = False
        self._warmup_steps = warmup_steps
        self._prefetch = (not hasattr(model, "graph_spec") and hasattr(model, "prefetch_supported") and model.prefetch_supported()
                          and os.environ.get("SGAN_NO_G_PREFETCH", "0") in ("", "0"))

    # the step cut into capturable segments; "sync_D"/"sync_G" are the data-parallel hand-off points
    def _program(self):
        m, o = self.m, self.m.opt
        prog = [[m.optimizer_D.zero_grad, m.backward_D], "sync_D", [m.optimizer_D.step]]
        for _ in range(o.n_update_G):
            prog[-1] += [m.optimizer_G.zero_grad, m.backward_G]
            prog += ["sync_G", [m.optimizer_G.step]]
            if o.n_update_G > 1:
                prog[-1].append(m.sample_noise)
        if self._prefetch:
            prog[-1][-1] = m.sample_noise_and_prefetch
        return prog

    def _begin(self):
        """Start of a step: the arenas' zeroing launch, which also clears the gradient buffers optimizer_D.zero_grad() is about to."""
        opt_d = getattr(self.m, "optimizer_D", None)
        ops.begin_step(opt_d.take_zeroing() if (hasattr(opt_d, "take_zeroing") and not hasattr(self.m, "graph_spec")) else ())

    def _spec(self):
        m = self.m
        if hasattr(m, "graph_spec"):
            return m.graph_spec()
        return dict(pools=[m.fake_pool], sources=lambda: [m._pool_source()],
                    set_overrides=lambda views: setattr(m, "_pool_override", views[0]), program=self._program())

    def capture(self, example_input):
        m = self.m
        assert getattr(m, "noise_source", None) is None, "graphed step draws its latents on the device"
        spec = self._spec()
        for _ in range(self._warmup_steps - (1 if self._prefetch else 0)):   # lazy state (optimizer moments, caches) must exist before capture
            m.set_input(example_input)
            m.optimize_parameters()
        if self._prefetch:
            # the last warm-up step runs the graph's own program eagerly: the arena pool then holds the sequence of arenas the
            # capture is going to ask for, and the step ends with the first two-problem pass (the next forward() is in place)
            # ON the stream the capture will use: autograd runs a node's backward on the stream its forward ran on, and the capture's
            # first backward walks the node this pass leaves behind (a backward hopping to the default stream mid-capture crashed
            # hipStreamEndCapture)
            m._prefetch = True
            m.set_input(example_input)
            self._cap_stream = torch.cuda.Stream(device=m.device)
            self._cap_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._cap_stream):
                self._begin()
                m.forward()
                for item in self._program():
                    if isinstance(item, list):
                        for f in item:
                            f()
                    elif m.grad_sync is not None:
                        m.grad_sync(m.optimizer_D if item == "sync_D" else m.optimizer_G)
            torch.cuda.current_stream().wait_stream(self._cap_stream)
        for name in ("optimizer_D", "optimizer_D1", "optimizer_D2", "optimizer_G"):
            if hasattr(m, name):
                getattr(m, name).sync_lr()
        torch.cuda.synchronize()
        if self._prefetch:
            m.adopt_prefetched()
        self.pools = spec["pools"]
        shapes = [tuple(t.shape) for t in spec["sources"]()]
        self.fake_for_D = [torch.zeros((h, w, ops.pad4(nc)), dtype=torch.float32, device=m.device) for (_, nc, h, w) in shapes]
        spec["set_overrides"]([ops.logical_view(buf, sh[1]) for buf, sh in zip(self.fake_for_D, shapes)])
        # with a process group alive its watchdog thread polls CUDA events: only the capturing thread is held to the
        # capture rules then (PyTorch's recipe for graphs next to NCCL)
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        self._mode = "thread_local" if dist_on else "global"
        if dist_on:
            torch.distributed.barrier()
            torch.cuda.synchronize()
        if self._prefetch:
            self.gA = None
            self._fakeA = spec["sources"]()      # the kept forward's output: every replay's two-problem pass rewrites it in place
            pool = None
            merged = [self._begin]
        else:
            self.gA = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gA, capture_error_mode=self._mode):
                self._begin()      # the statistics arenas of the whole step (graph A and every piece of graph B), one launch
                m.forward()
                self._fakeA = spec["sources"]()
            pool = self.gA.pool()
            merged = []
        self.segs = []
        for item in spec["program"]:
            if isinstance(item, tuple):          # ("sync", optimizer): data-parallel hand-off point
                if m.grad_sync is not None:
                    self.segs.append(("graph", self._capture(merged, pool, self._mode)))
                    self.segs.append(("sync", item[1]))
                    merged = []
            elif isinstance(item, str):
                if m.grad_sync is not None:
                    self.segs.append(("graph", self._capture(merged, pool, self._mode)))
                    self.segs.append(("sync", m.optimizer_D if item == "sync_D" else m.optimizer_G))
                    merged = []
            else:
                merged += item
        if merged:
            self.segs.append(("graph", self._capture(merged, pool, self._mode)))
        self._captured = True
        torch.cuda.synchronize()
        if self._prefetch:
            self._fake_last = m.fake      # the re-drawn sample of the step just finished: what `fake` is between steps

    def _capture(self, fns, pool, mode="global"):
        g = torch.cuda.CUDAGraph()
        if pool is None:
            pool = getattr(self, "_pool", None)
        with torch.cuda.graph(g, pool=pool, stream=getattr(self, "_cap_stream", None), capture_error_mode=mode):
            for f in fns:
                f()
        self._pool = g.pool()
        return g

    def step(self, data=None):
        """One training step == model.optimize_parameters() (set_input first when data is given)."""
        m = self.m
        if data is not None:
            m.set_input(data)
        if self._prefetch:
            m.adopt_prefetched()      # host attributes only: the forward itself ran at the end of the previous step
        else:
            self.gA.replay()
        for pool, src, buf in zip(self.pools, self._fakeA, self.fake_for_D):      # the reference's query order
            # a copy KERNEL on the step's stream: Tensor.copy_ of a contiguous tensor is hipMemcpyAsync, which on this stack starts
            # ~100 us after the work queued before it (measured: profiles/r02 timeline), a bubble in every step
            q = ops.as_nhwc(pool.query(src))
            ops.slice_nhwc(q, 0, q.shape[2], out=buf)
        for kind, obj in self.segs:
            if kind == "graph":
                obj.replay()
            else:
                m.grad_sync(obj)
        if self._prefetch:
            m.fake, m.noise = self._fake_last, m._noise_alt


GraphedFCGANStep = GraphedStep

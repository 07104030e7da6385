#!/usr/bin/env python3
"""Headline benchmark: fcgan training-step images/sec (deconv G + 3x PatchGAN D, 512x512, bs=1/GPU)
on N MI355X, with the dominant kernel's roofline fraction and a CPU baseline timed in the same run.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = FCGANModel.optimize_parameters() of the README fcgan recipe (README.md:33: n_update_D=1,
n_update_G=2, pool_size=50) on one synthetic 3x512x512 batch taken from a ring of 64 pinned host tensors: the
H2D copy of the batch is inside the timed region, on the step's stream; data loading is not.
With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this process starts the N ranks itself, one
child process per GPU, before it touches any GPU, and relays rank 0's line.
Rank 0 prints ONE JSON line (contract in the task statement / DESIGN.md section "Measurement")."""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between the ranks of this host

import torch  # noqa: E402

# SURVEY 8(d) / BASELINE.md section 3: conv MACs per forward pass of the reference nets
GMAC_G, GMAC_D = 1.9483, 4.7386


# cgan (BASELINE configs[2]): unet_256 ngf64 forward 23.86 GMAC, D n_layers 3 + 4 (ndf 64, scale 1) 13.77 + 11.74
GMAC_G_CGAN, GMAC_D_CGAN = 23.86, 25.51


def flops_per_image(n_update_G, workload="fcgan"):
    g, d = (GMAC_G, GMAC_D) if workload == "fcgan" else (GMAC_G_CGAN, GMAC_D_CGAN)
    g_f, g_b, d_f, d_b = (3, 2, 4, 4) if n_update_G == 2 else (1, 1, 3, 3)
    return 2.0 * 1e9 * (g_f * g + 2 * g_b * g + d_f * d + 2 * d_b * d)


def build_twostage(args, rank):
    """README.md:18 (BASELINE configs[4]): G1 fcgan ngf32 (z 8x4x4 -> 256^2 labels), bilinear x2, G2 crn ngf64 bilinear 2-layer
    blocks, F2 unet_128 nff32, D1 n_layers 3 3 scale 1 2, D2 n_layers 3 4 3 4 scale 1 1 2 2, no dropout."""
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    argv = ["--name", "bench", "--model", "twostage_cycle", "--which_direction", "AtoB", "--dataset_mode", "single",
            "--loadSize", "1024", "--fineSize", "512", "--transform_1to2", "bilinear_2", "--batchSize", "1", "--input_nc", "2",
            "--output_nc", "1", "--which_channel", "rg_b", "--which_model_netG1", "fcgan", "--n_layers_G1", "5", "--ngf1", "32",
            "--which_model_netD1", "n_layers", "--n_layers_D1", "3", "3", "--ndf1", "32", "--scale_factor1", "1", "2",
            "--lambda_D1", "0.5", "0.4", "--which_model_netG2", "crn", "--ngf2", "64", "--upsample_mode2", "bilinear",
            "--n_layers_CRN_block2", "2", "--which_model_netF2", "unet_128", "--nff2", "32", "--which_model_netD2", "n_layers",
            "--n_layers_D2", "3", "4", "3", "4", "--ndf2", "64", "--scale_factor2", "1", "1", "2", "2",
            "--lambda_D2", "0.3", "0.3", "0.2", "0.2", "--lambda_A", "10", "--lambda_B", "10", "--lambda_A_cycle", "5",
            "--lambda_fake_cycle", "1", "--noise_nc1", "8", "--noiseSize1", "4", "--noise_nc2", "8", "--noiseSize2", "8",
            "--norm", "instance", "--no_dropout1", "--no_dropout2", "--n_update_G", "1", "--no_lsgan1", "--manualSeed", str(rank),
            "--gpu_ids", str(torch.cuda.current_device()), "--checkpoints_dir", "/tmp/sgan_bench_ckpt"]
    if args.skip_wasted_D_wgrad:
        argv.append("--skip_wasted_D_wgrad")
    opt = TrainOptions().parse(argv, save=False, verbose=False)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):      # stdout carries exactly one JSON line
        return create_model(opt)


def build_cgan(args, rank):
    """README.md:38: unet_256 ngf64 with dropout + Gaussian noise, D n_layers 3 4 ndf64 scale 1 1,
    lambda_D .5 .5, lambda_A 10, L1 weights 2 4, no_lsgan, n_update_G 2, which_channel rg_b."""
    from supervised_gan_amd.models import create_model
    from supervised_gan_amd.options import TrainOptions
    argv = ["--name", "bench", "--model", "cgan", "--which_direction", "AtoB", "--dataset_mode", "single",
            "--loadSize", "1024", "--fineSize", "512", "--batchSize", "1", "--input_nc", "2", "--output_nc", "1",
            "--which_model_netG", "unet_256", "--ngf", "64", "--which_model_netD", "n_layers", "--n_layers_D", "3", "4",
            "--ndf", "64", "--scale_factor", "1", "1", "--lambda_D", "0.5", "0.5", "--lambda_A", "10", "--noise_nc", "8",
            "--noiseSize", "4", "--norm", "instance", "--n_update_G", str(args.n_update_G), "--weights", "2", "4",
            "--no_lsgan", "--manualSeed", str(rank), "--add_gaussian_noise", "--which_channel", "rg_b",
            "--gpu_ids", str(torch.cuda.current_device()), "--checkpoints_dir", "/tmp/sgan_bench_ckpt"]
    if args.skip_wasted_D_wgrad:
        argv.append("--skip_wasted_D_wgrad")
    opt = TrainOptions().parse(argv, save=False, verbose=False)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        return create_model(opt)


def build_model(args, rank):
    from supervised_gan_amd.fcgan_model import FCGANModel
    from supervised_gan_amd.options import TrainOptions
    argv = ["--name", "bench", "--model", "fcgan", "--which_direction", "A", "--dataset_mode", "single",
            "--loadSize", "512", "--fineSize", "512", "--batchSize", "1", "--input_nc", "2",
            "--which_model_netG", "deconv", "--n_layers_G", "5", "--ngf", "32", "--which_model_netD", "n_layers",
            "--n_layers_D", "3", "3", "3", "--ndf", "32", "--scale_factor", "1", "2", "4",
            "--lambda_D", "0.5", "0.4", "0.1", "--noise_nc", "8", "--noiseSize", "8", "--norm", "instance",
            "--no_dropout", "--n_update_G", str(args.n_update_G), "--no_lsgan", "--which_channel", "rg",
            "--manualSeed", str(rank), "--gpu_ids", str(torch.cuda.current_device()),
            "--checkpoints_dir", "/tmp/sgan_bench_ckpt"]
    if args.skip_wasted_D_wgrad:
        argv.append("--skip_wasted_D_wgrad")
    if args.no_d_streams:
        argv.append("--no_d_streams")
    if getattr(args, "no_group", False):
        argv.append("--no_group")
    opt = TrainOptions().parse(argv, save=False, verbose=False)
    torch.manual_seed(0)          # identical initial weights on every rank (also broadcast below)
    m = FCGANModel()
    m.initialize(opt)
    return m


def synthetic_ring(n, rank, device, where="hbm"):
    """64 pre-generated batches.  where="hbm" (the headline: inputs resident in HBM when the timed region starts; set_input() is one
    gather kernel that picks the channels into the padded NHWC input buffer) or "host" (PINNED host memory, SURVEY 8d: the same
    kernel then reads the batch over PCIe inside the timed region -- reported beside the headline as `host_input`)."""
    g = torch.Generator().manual_seed(123 + rank)
    ring = [{"A": torch.rand(1, 3, 512, 512, generator=g) * 2 - 1, "A_paths": ["synthetic"]} for _ in range(n)]
    for b in ring:
        b["A"] = b["A"].pin_memory() if where == "host" else b["A"].to(device)
    return ring


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU, RCCL between them) and
    relay rank 0's JSON line.  Runs BEFORE this process touches a GPU (it never does: it only waits), so no process that has
    initialised HIP is ever re-executed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=600 if rcs[0] == 0 else 20))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(-9)
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"bench.py: rank exit codes {rcs}")


def profile_kernels(model, ring, reps=3, workload="fcgan"):
    """Per-kernel-instantiation time and algorithmic flops of the conv kernels, measured live.

    Every sgan_conv_* call of one eager training step is recorded (descriptor + tensors), then the whole list
    is re-issued back to back `reps` times on one stream with the library's launch timing on (sgan_profile_*:
    two HIP events recorded inside the library around each main conv kernel launch).  With the queue kept full
    that is the kernel-only duration rocprofv3 --kernel-trace reports.  FLOP per launch = 2*pixels*Cin*Cout*k^2
    from the descriptor (logical channels)."""
    import ctypes
    from supervised_gan_amd import _lib, ops
    lib = _lib.lib()
    calls = []

    def flops(desc):   # stored 4 channels <- logical channels: fcgan 2-channel image / 1-channel logits; cgan 2-channel label
        # (U-Net, pad 1), 3-channel pair (discriminators, pad 2), 1-channel image and logits
        if desc.Cin_logical and desc.Cout_logical:      # the networks fill the logical-channel hint of every descriptor
            cin, cout = desc.Cin_logical, desc.Cout_logical
        elif workload == "fcgan":
            cin = desc.Cin if desc.Cin > 4 else 2
            cout = desc.Cout if desc.Cout > 4 else (2 if desc.kind == 1 else 1)
        else:
            cin = desc.Cin if desc.Cin > 4 else (2 if desc.pad == 1 else 3)
            cout = desc.Cout if desc.Cout > 4 else 1
        pix = desc.Hout * desc.Wout if desc.kind == 0 else desc.Hin * desc.Win
        return 2.0 * pix * cin * cout * desc.k * desc.k

    def abytes(desc, kind, has_x=True):
        """Algorithmic HBM bytes of one problem: every operand tensor once (stored channels, fp32).  fwd: in + out; dgrad: dOut + dX
        (+ the forward tensor its activation derivative reads); wgrad: x + dOut; bwd (one fused launch): dOut + x + dX."""
        i, o = 4.0 * desc.Hin * desc.Win * desc.Cin, 4.0 * desc.Hout * desc.Wout * desc.Cout
        if kind == "fwd":
            return i + o
        if kind == "dgrad":
            return o + i + (i if has_x else 0.0)
        if kind == "wgrad":
            return i + o
        return o + 2.0 * i

    depth = [0]      # conv_fwd / conv_dgrad may delegate to their grouped form: record the outermost call only

    def wrap(name, grouped):
        orig = getattr(ops, name)

        def f(first, *a, **k):
            if depth[0] == 0:
                fl = sum(flops(j[0]) for j in first) if grouped else flops(first)
                kd = "fwd" if "fwd" in name else ("dgrad" if "dgrad" in name else "wgrad")
                ab = sum(abytes(j[0], kd, kd != "dgrad" or j[4] is not None) for j in first) if grouped else abytes(first, kd)
                calls.append((orig, (first,) + a, k, fl, ab))
            depth[0] += 1
            try:
                return orig(first, *a, **k)
            finally:
                depth[0] -= 1
        setattr(ops, name, f)
        return orig

    def wrap_pair(name):      # conv_bwd_grouped(djobs, wjobs): ONE record when it took the fused launch, else its two inner calls record themselves
        orig = getattr(ops, name)

        def f(djobs, wjobs, *a, **k):
            if depth[0] > 0:
                return orig(djobs, wjobs, *a, **k)
            fused = orig(djobs, wjobs, *a, **k)
            if fused:
                calls.append((orig, (djobs, wjobs) + a, k, sum(flops(j[0]) for j in djobs) + sum(flops(j[0]) for j in wjobs),
                              sum(abytes(j[0], "bwd") for j in djobs)))
            return fused
        setattr(ops, name, f)
        return orig

    names = {"conv_fwd": False, "conv_dgrad": False, "conv_wgrad": False,
             "conv_fwd_grouped": True, "conv_dgrad_grouped": True, "conv_wgrad_grouped": True}
    saved_streams, model._streams = getattr(model, "_streams", []), []     # one stream: kernels are timed one at a time
    try:
        for i in range(2):      # untimed: code-object loads and allocator growth are not kernel time
            model.set_input(ring[i % len(ring)])
            model.optimize_parameters()
        origs = {n: wrap(n, gr) for n, gr in names.items()}
        origs["conv_bwd_grouped"] = wrap_pair("conv_bwd_grouped")
        try:
            model.set_input(ring[2 % len(ring)])
            model.optimize_parameters()
        finally:
            for n_, o in origs.items():
                setattr(ops, n_, o)
        torch.cuda.synchronize()
        # Each recorded call is captured NREP times back to back into a hipGraph and replayed: elapsed / NREP is the call's device
        # time including the ~1.5 us dependent-launch boundary and any second kernel it launches (split-K epilogue, thin-wgrad
        # reduce) -- slightly ABOVE the kernel duration rocprofv3 --kernel-trace reports, never below it (round 1 subtracted an
        # event-pair overhead from per-launch events and came out 6 % optimistic).
        NREP = 8
        agg, per_call = {}, []
        for fn, a, k, fl, ab in calls:
            fn(*a, **k)
            nm = lib.sgan_last_kernel().decode()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(NREP):
                    fn(*a, **k)
            g.replay()
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / NREP)      # ms per call
            del g
            a_ = agg.setdefault(nm, [0, 0.0, 0.0, 0.0])
            a_[0] += reps
            a_[1] += best * reps
            a_[2] += fl * reps
            a_[3] += ab * reps
            per_call.append((nm, best * 1e3))
        if os.environ.get("SGAN_BENCH_CALLS"):     # tuning aid: one line per conv call of the step
            with open(os.environ["SGAN_BENCH_CALLS"], "w") as f:
                for ci, ((fn, a, k, fl, ab), (nm, us)) in enumerate(zip(calls, per_call)):
                    first = a[0]
                    descs = [j[0] for j in first] if isinstance(first, list) else [first]
                    shape = " ".join(f"{'T' if d.kind else 'C'}k{d.k}s{d.stride} {d.Cin}->{d.Cout} {d.Hin}x{d.Win}->{d.Hout}x{d.Wout}" for d in descs[:2])
                    f.write(f"{ci:3d} {getattr(fn, '__name__', '?'):20s} {nm:36s} n={len(descs)} {fl / 1e9:8.3f} GF {us:8.1f} us {fl / us / 1e6:6.1f} TF  {shape}\n")
    finally:
        model._streams = saved_streams
    calls.clear()
    return {k: {"launches_per_step": v[0] / reps, "avg_us": 1e3 * v[1] / v[0], "ms_per_step": v[1] / reps,
                "tflops": v[2] / (v[1] * 1e-3) / 1e12, "gflop_per_launch": v[2] / v[0] / 1e9,
                "algorithmic_bytes_per_launch": int(v[3] / v[0])} for k, v in agg.items()}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_cgan(n_update_G, budget_s=25.0):
    """The CPU oracle's cgan step (same README config) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sgan_oracle as O
    cores = int(os.environ.get("SGAN_CPU_THREADS", min(16, os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    cfg = O.CGANConfig(**dict(O.CGAN_README, n_update_G=n_update_G))
    m = O.CGANOracle(cfg, seed=0)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
    m.set_input(x[:, :2].contiguous(), x[:, 2:3].contiguous())
    m.optimize_parameters()                          # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 1 or (time.perf_counter() - t0 < budget_s and n < 20):
        m.optimize_parameters()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} steps of the same cgan unet_256 512x512 bs=1 n_update_G={n_update_G} workload after 1 warm-up step, "
                      f"torch {torch.__version__} fp32 CPU oracle, {cores} threads"}


def cpu_baseline(n_update_G, budget_s=20.0):
    """The CPU oracle (fp32 restatement of the reference step, pinned against reference goldens)
    timed on this box's host cores on the same workload; bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sgan_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cfg = O.FCGANConfig(n_update_G=n_update_G)
    o = O.FCGANOracle(cfg, seed=0)
    g = torch.Generator().manual_seed(5)
    o.noise_iter = iter(lambda: torch.randn(1, 8, 8, 8, generator=g), None)
    real = [torch.rand(1, 2, 512, 512, generator=g) * 2 - 1 for _ in range(4)]
    # thread-count sweep (SURVEY 8d asks for the host's cores, stated): this bs=1 workload stops scaling long before a 256-thread
    # host is full (measured on the EPYC 9575F box: 16 threads 6.3 images/s, 32: 3.2, 64: 1.1, 256: 0.006), so time one step at
    # 8 / 16 / 32 / 64 / all threads, stop at the first count that is slower than the best so far, and keep the fastest
    forced = os.environ.get("SGAN_CPU_THREADS")
    cands = [int(forced)] if forced else sorted({c for c in (8, 16, 32, 64, avail) if c <= avail} or {avail})
    sweep = {}
    torch.set_num_threads(cands[0])
    o.optimize_parameters(real[0])          # warm-up (allocator, oneDNN primitive caches)
    for c in cands:
        torch.set_num_threads(c)
        o.optimize_parameters(real[1])
        ts = time.perf_counter()
        o.optimize_parameters(real[2])
        sweep[c] = 1.0 / (time.perf_counter() - ts)
        if sweep[c] < 0.9 * max(sweep.values()):      # past the knee: more threads only get slower (256 threads: minutes per step)
            break
    cores = max(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    o.optimize_parameters(real[3])
    t0 = time.perf_counter()
    n = 0
    while True:
        o.optimize_parameters(real[n % 4])
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 40:
            break
    dt = time.perf_counter() - t0
    # a 1-thread figure beside the all-cores one (two steps, ~1.5 s each)
    torch.set_num_threads(1)
    o.optimize_parameters(real[0])
    t1 = time.perf_counter()
    for i in range(2):
        o.optimize_parameters(real[i])
    dt1 = (time.perf_counter() - t1) / 2
    torch.set_num_threads(cores)
    return {"value": n / dt, "unit": "images/sec", "cores": cores, "cores_available": avail, "kind": "port",
            "sample": f"{n} steps of the same fcgan 512x512 bs=1 n_update_G={n_update_G} workload after warm-up, "
                      f"torch {torch.__version__} fp32 CPU oracle, {cores} threads (fastest of the sweep)",
            "thread_sweep_images_per_sec": {str(k): round(v, 3) for k, v in sweep.items()},
            "cpu_model": _cpu_model(), "value_1thread": 1.0 / dt1}


def committed_profile_of(kernel):
    """(HBM bytes per launch from the newest profiles/r*_pmc_traffic.json, average launch us and file name from the newest
    profiles/r*_bench_kernel_stats.csv) of `kernel`, template variants merged the way sgan_last_kernel() names them."""
    import csv
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from kernel_names import short
    except Exception:      # noqa: BLE001
        return None, None, None
    traffic = us = src = None
    try:
        pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]))["kernels"]
        traffic = pm[kernel]["hbm_bytes_per_launch"] if kernel in pm else None
    except Exception:      # noqa: BLE001
        pass
    try:
        f = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))[-1]
        calls = tot = 0
        for r in csv.DictReader(open(f)):
            if short(r["Name"]) == kernel:
                calls += int(r["Calls"])
                tot += float(r["TotalDurationNs"])
        if calls:
            us, src = tot / calls / 1e3, os.path.basename(f)
    except Exception:      # noqa: BLE001
        pass
    return traffic, us, src


def main():
    if os.environ.get("SGAN_BENCH_WATCHDOG"):      # debugging aid: dump every thread's stack and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SGAN_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n_update_G", type=int, default=2, help="README recipe uses 2 (README.md:33)")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into hipGraphs")
    ap.add_argument("--skip_wasted_D_wgrad", action="store_true")
    ap.add_argument("--no_d_streams", action="store_true")
    ap.add_argument("--no_group", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_kernel_profile", action="store_true")
    ap.add_argument("--workload", default="fcgan", choices=["fcgan", "cgan", "twostage_cycle"],
                    help="fcgan = the headline metric (BASELINE configs[1]); cgan = BASELINE configs[2], twostage_cycle = configs[4], "
                         "each reported under its own metric name")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # no launcher: be the launcher (nothing has touched a GPU yet)
        return spawn_ranks(args.gpus)
    from supervised_gan_amd import dist as sdist
    rank, world, local = sdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("SGAN_FORCE_DEVICE"):     # rehearsal of N ranks on a 1-GPU box (gloo): all ranks share one card
        local = int(os.environ["SGAN_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    cgan = args.workload == "cgan"
    two = args.workload == "twostage_cycle"
    if two:
        args.no_cpu_baseline = True
        model = build_twostage(args, rank)
        sdist.broadcast_parameters([model.netG1, model.netG2, model.netF2] + model.netD1 + model.netD2)
    else:
        model = build_cgan(args, rank) if cgan else build_model(args, rank)
        sdist.broadcast_parameters([model.netG] + model.netD)
    ring = synthetic_ring(64, rank, device)

    kern = None
    if rank == 0 and not args.no_kernel_profile:     # a few un-synchronised steps on rank 0 only (no collective inside) ...
        kern = profile_kernels(model, ring, workload=args.workload)
    if kern is not None:                             # ... whose optimizer moments must not leak into the measured run
        for name in ("optimizer_D", "optimizer_D1", "optimizer_D2", "optimizer_G"):
            if hasattr(model, name):
                getattr(model, name).reset_state()
    if world > 1:                                    # ... and every rank is handed rank 0's weights again before the timed run
        nets = [model.netG1, model.netG2, model.netF2] + model.netD1 + model.netD2 if two else [model.netG] + model.netD
        sdist.broadcast_parameters(nets)
        model.grad_sync = sdist.GradAverager()

    if os.environ.get("SGAN_FORCE_SEGMENTS") and model.grad_sync is None:     # diagnostic: the N > 1 graph cuts without the collectives
        model.grad_sync = lambda optimizer: None
    if args.eager:
        def step(i):
            model.set_input(ring[i % len(ring)])
            model.optimize_parameters()
    else:
        from supervised_gan_amd.graph_step import GraphedStep
        gs = GraphedStep(model)
        gs.capture(ring[0])

        def step(i):
            gs.step(ring[i % len(ring)])

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    if getattr(model, "grad_sync", None) is not None and hasattr(model.grad_sync, "bytes"):
        model.grad_sync.bytes = model.grad_sync.calls = 0      # count the timed steps only
    host_busy = 0.0
    # one event per step boundary on the step's stream: per-step device time without a synchronisation inside the timed region
    # (the headline stays the wall clock over all K steps; the median says whether a single late launch moved it)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        th = time.perf_counter()
        step(args.warmup + i)
        marks[i + 1].record()
        host_busy += time.perf_counter() - th
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    gs_bytes, gs_calls = getattr(getattr(model, "grad_sync", None), "bytes", 0), getattr(getattr(model, "grad_sync", None), "calls", 0)      # of the timed steps
    # the same steps with the batches in pinned host memory (round 2's headline regime: the input gather reads over PCIe inside the step)
    host_ring = synthetic_ring(8, rank, device, where="host")
    kh = min(args.steps, 100)
    for s_ in host_ring:      # untimed: the first device access of a freshly pinned buffer is not PCIe time (2.05 vs 2.87 ms/step seen without)
        if args.eager:
            model.set_input(s_); model.optimize_parameters()
        else:
            gs.step(s_)
    barrier()
    th0 = time.perf_counter()
    for i in range(kh):
        s_ = host_ring[i % len(host_ring)]
        if args.eager:
            model.set_input(s_); model.optimize_parameters()
        else:
            gs.step(s_)
    barrier()
    dt_host = (time.perf_counter() - th0) / kh
    if os.environ.get("SGAN_BENCH_HOST"):      # diagnostic: is the host (graph launches, pool policy) or the device the longer leg?
        print(f"[bench] host enqueue {host_busy / args.steps * 1e3:.3f} ms/step of {dt / args.steps * 1e3:.3f} ms/step", file=sys.stderr)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    errs = model.get_current_errors()
    assert all(v == v and abs(v) < 1e4 for v in errs.values()), errs     # finite losses after the run
    rccl = None
    if world > 1:      # what the collective layer itself reports: ranks that took part in a real all-reduce, bytes reduced per step
        one = torch.ones(1, device=device)
        torch.distributed.all_reduce(one)
        gsync = model.grad_sync
        rccl = {"backend": torch.distributed.get_backend(), "rccl_ranks": int(round(float(one.item()))),
                "world_size": torch.distributed.get_world_size(),
                "allreduce_bytes_per_step": int(gs_bytes / max(args.steps, 1)),
                "allreduce_calls_per_step": gs_calls / max(args.steps, 1)}

    if rank == 0:
        from supervised_gan_amd import ops as _ops
        # storage and accumulation are fp32 in both modes; "bf16x3" = every product as three bf16 MFMAs (an fp32-equivalent result)
        dtype_label = "f32 (bf16x3 MFMA)" if _ops.get_math() == "bf16x3" else "f32"
        ips = world * args.steps / dt
        fl = 990e9 if two else flops_per_image(args.n_update_G, args.workload)      # twostage_cycle ~ 990 GFLOP / image
        workload = ("fcgan deconv-G(ngf32, z 8x8x8) + 3x PatchGAN-D(ndf32, n_layers 3, scale 1/2/4) 512x512 bs=1, "
                    f"n_update_D=1 n_update_G={args.n_update_G}, Adam, pool 50 (BASELINE configs[1] shape, fp32 compute); "
                    "synthetic batches resident in HBM (ring of 64)")
        if cgan:
            workload = ("cgan unet_256-G(ngf64, 2->1 ch, dropout, gaussian noise) + PatchGAN-D n_layers 3 and 4 (ndf64, scale 1 1) "
                        f"512x512 bs=1, GAN + weighted L1 (lambda_A 10, weights 2 4), n_update_D=1 n_update_G={args.n_update_G}, "
                        "Adam, pool 50 (BASELINE configs[2], fp32 compute)")
        if two:
            workload = ("twostage_cycle: G1 fcgan(ngf32) + bilinear x2 + G2 crn(ngf64, bilinear, 2-layer blocks) + F2 unet_128(nff32) + "
                        "D1 n_layers 3 3 (scale 1 2) + D2 n_layers 3 4 3 4 (scale 1 1 2 2) 512x512 bs=1, one D1, D2 and G update per "
                        "step, Adam (BASELINE configs[4], fp32 compute)")
        out = {
            "metric": ("train-step images/sec, twostage_cycle 512x512 bs=1/GPU" if two else
                       "train-step images/sec, cgan unet_256 512x512 bs=1/GPU" if cgan else "train-step images/sec, fcgan 512x512 bs=1/GPU"),
            "value": ips, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "ms_per_step_median": ms_median,
            "host_input": {"ms_per_step": 1e3 * dt_host, "value": world / dt_host, "steps": kh,
                           "note": "same step with the batch in pinned host memory: the input gather kernel reads it over PCIe inside the timed region"}, "ms_per_step_min": per_step[0],
            "ms_per_step_max": per_step[-1], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": workload,
                       "parallelism": f"dp{world}", "global_batch": world, "hip_graph": not args.eager,
                       "skip_wasted_D_wgrad": bool(args.skip_wasted_D_wgrad),
                       "gflop_per_image_reference_executed": fl / 1e9,
                       "achieved_tflops_per_gpu_reference_executed": fl * ips / world / 1e12},
            "losses": {k: round(v, 5) for k, v in errs.items()},
        }
        if rccl:
            out["collective"] = rccl
        if kern:
            dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
            split = "igemm3" in dom or "wgrad3" in dom or dom == "sg_bwd_fused_kernel"      # split-bf16 kernel: 3 bf16 MFMA flops issued per useful flop
            # MI355X_MICROARCH.md: fp32 matrix peak 157.3 TFLOP/s; bf16 dense MFMA peak 2500 TFLOP/s (never the 2:1-sparsity figure)
            peak = 2500.0 if split else 157.3
            traffic, rp_us, rp_src = None, None, None
            if args.workload == "fcgan":      # the committed passes were taken on the fcgan step: only its launch mix matches them
                traffic, rp_us, rp_src = committed_profile_of(dom)
            # achieved = ALGORITHMIC (useful) FLOP/s; the two extra MFMA passes of the split-bf16 emulation are cost, not work
            useful = kern[dom]["tflops"]
            issued = useful * (3.0 if split else 1.0)
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": useful, "peak": peak,
                               "unit": "TFLOP/s", "frac": useful / peak, "traffic": traffic,
                               "traffic_algorithmic": kern[dom].get("algorithmic_bytes_per_launch"),      # every operand tensor of a launch once
                               "achieved_useful": useful, "achieved_issued": issued, "frac_issued": issued / peak,
                               "mfma_flops_issued_per_useful_flop": 3 if split else 1,
                               "frac_of_fp32_matrix_peak": kern[dom]["tflops"] / 157.3,
                               "avg_launch_us": kern[dom]["avg_us"], "gflop_per_launch": kern[dom]["gflop_per_launch"],
                               # the same kernel in the committed rocprofv3 --kernel-trace --stats summary of this command (profiles/):
                               # average launch duration there, and the fraction it gives
                               "rocprof_avg_launch_us": rp_us, "rocprof_source": rp_src,
                               "rocprof_frac": (kern[dom]["gflop_per_launch"] / rp_us * 1e3 / peak) if rp_us else None,      # GFLOP / us = PFLOP/s; useful FLOP
                               "launches_per_step": kern[dom]["launches_per_step"],
                               "measured": "every conv call of one step captured 8x back to back into a hipGraph and replayed: device time "
                                           "per call (HIP events on the replay stream), launch boundary and any second kernel of the call included"}
            out["kernels"] = {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in sorted(kern.items())}
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline_cgan(args.n_update_G) if cgan else cpu_baseline(args.n_update_G)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- voxels/s of the XLSTM-HVED forward+backward hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp16|fp32] [--size 128] [--no-graph] [--no-cpu]

One process per GPU.  Under torch.distributed.run RANK/LOCAL_RANK/WORLD_SIZE are read from the environment; a plain
`python bench.py --gpus N` (no launcher) spawns its own N worker processes before anything touches a GPU.  A step = one forward + backward of XLSTM_HVED (train mode, all 4 modalities, recon=True, loss of
SURVEY.md 8(d)) on one synthetic 1x4x128^3 patch per rank, plus for N>1 the flat RCCL all-reduce of the generator's
gradients (data parallel; weak scaling).  Forward+backward are captured once into a hipGraph and replayed (the
all-reduce is issued eagerly after each replay, on the same stream); W warm-up steps, then exactly K timed steps
between barrier + synchronize pairs; the slowest rank's time is used.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant conv kernel of the step (by total time), timed per launch with HIP events on the launch stream
                during an instrumented pass over the same steps; algorithmic bytes/flops from the launch's shapes.
  cpu_baseline  the CPU oracle (port of the reference path; the reference sources do not travel to the GPU box)
                timed on the best of a small thread-count sweep on a bounded sample of the same workload; it runs in a
                child process next to the GPU legs (a 256-thread host; the GPU timed region is one graph launch per step).
  modes         ms/step and voxels/s of the same step in the other storage modes (fp32 = the parity mode, fp16 = the
                reference's AMP dtype, bf16), each with the deviation class the test suite measured for it (`parity`, text) and
                with THIS run's measurement (`parity_measured`: Dice deviation and mask flips of the mode's 128^3 segmentation
                of the trained-like parity case against the mask the real reference produced, tests/golden/make_mask_128.py).
  config3       BASELINE config 3: N = 2 at 128^3, per-sample modality dropout drawn per step from the 15 subsets, one hipGraph.
"""
import subprocess
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
FP32_VALU_PEAK_TFLOPS = 157.3  # vector fp32 peak: the fp32 (parity mode) conv kernels are FMA kernels on the VALU
BF16_MFMA_PEAK_TFLOPS = 2500.0 # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md): the 16-bit conv kernels are implicit GEMMs on MFMA
LOSS_SCALE_FP16 = 65536.0      # torch.cuda.amp.GradScaler's initial scale (the reference's AMP, train.py:207)
# deviation class of each storage mode from the fp32 reference path (tests/test_gpu_network.py, measured on MI355X at 64^3 / 128^3;
# yardstick = the reference's own fp16-autocast deviation on the same weights / inputs, tests/golden/amp_yardstick.json)
# "trained-like": the real reference trained 300 CPU steps on smooth synthetic patches (tests/golden/make_trained_like.py)
STORAGE_NOTE = {
    "bf16": "bf16 activation storage / fp32 arithmetic", "fp16": "fp16 activation storage / fp32 arithmetic",
    "fp32": "fp32 activation storage / fp32 arithmetic",
    "fp32_mfma": "fp32 activation storage / 3^3 convs as two-term fp16 split on the matrix cores, fp32 accumulation",
}
MODE_PARITY = {
    "fp32": "parity mode: seg |d| <= 8e-4, Dice deviation <= 1e-5 vs the CPU oracle at 128^3 on random-init weights, all parameter "
            "gradients within 2.1e-3 of the largest; 0 mask flips of 6.3 M on trained-like weights (SURVEY 8c tolerances 5e-3 / 1e-4)",
    "fp32_mfma": "parity mode on the matrix cores (fp32 storage; 3^3 quad-channel convs as two-term fp16 split MFMA, 7^3 gate convs and "
                 "weight gradients with operands rounded once to fp16, loss scale 65536): seg |d| <= 9.5e-4, Dice deviation <= 1.3e-5 vs "
                 "the CPU oracle at 128^3 on random-init weights, all parameter gradients within 2.1e-3 of the largest, relative L2 "
                 "3.3e-3; trained-like weights: Dice deviation 1.5e-5, 7 mask flips of 6.3 M "
                 "(tests/test_gpu_network.py::test_fp32_full_size_128_{vs,backward_vs}_oracle[split_mfma], "
                 "::test_storage_modes_on_trained_like_weights)",
    "fp16": "Dice deviation 5.5e-4 at 128^3 on trained-like weights (reference fp16-AMP: 8.3e-3); random-init weights: seg rel-L2 "
            "0.010, Dice deviation 3.6e-3 (reference fp16-AMP: 0.111, 4.0e-2)",
    "bf16": "Dice deviation 3.9e-3 at 128^3 on trained-like weights (reference bf16-AMP: 4.9e-2); random-init weights: seg rel-L2 "
            "0.072, Dice deviation 2.5e-2 (reference fp16-AMP: 0.111, 4.0e-2; bf16-AMP: 0.179, 6.7e-2)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32", "fp32_mfma"],
                    help="activation storage; fp32_mfma = fp32 storage with the quad-channel 3^3 convs on the matrix cores through "
                         "the two-term fp16 split (ops.set_fp32_mfma), run under the fp16 loss scale")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--inner", type=int, default=0,
                    help="passes per timed step (0 = sized so that the timed region lasts --min-timed-s; 1 = the plain K-step loop)")
    ap.add_argument("--min-timed-s", type=float, default=5.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and issue the gradient all-reduce even with one rank (exercises the N>1 code path)")
    ap.add_argument("--extras", action="store_true", help="also time forward-only and the shared-encoder two-forward step")
    ap.add_argument("--no-modes", action="store_true", help="skip timing the other storage modes")
    ap.add_argument("--level-streams", action="store_true", help="A/B: the coarse latent-path chains on side streams (ops.set_level_streams; eager launches only: use with --no-graph)")
    ap.add_argument("--no-config3", action="store_true", help="skip the BASELINE config 3 leg (N = 2, per-step modality dropout)")
    ap.add_argument("--no-trainstep", action="store_true", help="skip timing the whole training step (train.py:208-296)")
    ap.add_argument("--wgrad-overlap", action="store_true",
                    help="launch the weight-gradient kernels on a second HIP stream (measured on MI355X: no gain, 9.88-10.7 ms "
                         "vs 9.96 ms -- the hipGraph's cross-stream dependencies cost what the overlap saves; off by default)")
    ap.add_argument("--no-wgrad-defer", action="store_true",
                    help="launch every weight gradient where backward reaches it instead of batching them at the end of backward")
    ap.add_argument("--wgrad-batch", type=int, default=12, help="weight-gradient calls forked to the side stream per batch")
    ap.add_argument("--wgrad-flush-streams", type=int, default=1,
                    help="HIP streams the deferred weight-gradient flush deals its independent problems to (ops.set_wgrad_flush_streams)")
    ap.add_argument("--wgrad-early-flush", action="store_true",
                    help="A/B: launch the decoder's deferred weight gradients on a side stream when backward reaches the deep levels "
                         "(ops.set_wgrad_early_flush; measured slower)")
    ap.add_argument("--xh-option", action="append", default=[], metavar="KEY=VALUE",
                    help="xh_set_option(KEY, VALUE) before the run (A/B tests of launch plans; include/xlstm_hved.h lists the keys)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # child process of the N=1 run
    ap.add_argument("--force-spawn", action="store_true",
                    help="take the self-spawning path of a launcher-less `--gpus N` run even for N = 1 (tests)")
    return ap.parse_args()


def bench_loss(seg, mu, lv, rec):
    """seg.mean() + rec.mean() + sum_l (mu_l.mean() + logvar_l.mean()) -- reaches every parameter the
    reference's training loss reaches.  Each mean is one HIP reduction pass (losses.mean_of), not a cast + ATen reduce."""
    from xlstm_hved_amd.losses import sum_of_means
    return sum_of_means([seg, rec] + [t for ab in zip(mu, lv) for t in ab])


CPU_BASELINE_THREADS = (16, 32)     # pinned (profiles/cpu_baseline_calibration.json: stock CPU conv3d stops scaling between them)


def cpu_baseline(size, batch, threads=CPU_BASELINE_THREADS):
    """Times the CPU oracle (functional restatement of the reference path, stock torch ops, fp32) on this host: one FULL-SIZE
    fwd+bwd step at each of the PINNED thread counts, both reported, `value` = the faster.  (Rounds 1-5 chose the count from a
    64^3 sweep that raced the GPU legs' launch threads -- the driver saw 16 threads / 8.5e4 where the builder saw 32 / 6.2e4; stock
    CPU conv3d stops scaling long before a 256-thread host is full: 256 threads took 771 s per 128^3 step.)  A warm-up step at
    64^3 per count (thread pool, allocator) is untimed.  Never touches the GPU (runs in a child process of the N=1 bench)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xlstm_hved_oracle as O
    import xlstm_hved_amd as X
    ncpu = os.cpu_count() or 1
    torch.manual_seed(1)
    model = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    model.apply(X.init_weights)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)

    def run(s):
        x = torch.rand(batch, 4, s, s, s, generator=g)
        eps = [torch.randn(batch, 2 ** l, s >> (l + 1), s >> (l + 1), s >> (l + 1), generator=g) for l in range(4)]
        sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
        t0 = time.perf_counter()
        prob, _, mu, lv, rec = O.xlstm_hved_forward(sd, x, 14, eps_list=eps, training=True, reference_cost=True)
        O.bench_loss(prob, mu, lv, rec).backward()
        return time.perf_counter() - t0
    counts = sorted({min(c, ncpu) for c in threads})
    times = {}
    for c in counts:
        torch.set_num_threads(c)
        run(min(size, 64))                        # untimed: thread pool + allocator warm-up at this count
        times[c] = run(size)
    cores = min(times, key=times.get)
    per = {str(c): {"seconds": round(t, 3), "voxels_per_s": batch * size ** 3 / t} for c, t in times.items()}
    return {"value": batch * size ** 3 / times[cores], "unit": "voxels/s", "cores": cores, "kind": "port",
            "threads_pinned": list(counts), "per_thread_count": per, "concurrent_with_gpu_legs": True,
            "sample": f"one fwd+bwd step of {batch}x4x{size}^3 fp32 through oracle/xlstm_hved_oracle.py (torch {torch.__version__} "
                      f"CPU ops) at each pinned thread count of a {ncpu}-thread host: "
                      + ", ".join(f"{c} threads {t:.2f} s" for c, t in times.items())
                      + f"; value = the faster ({cores} threads); timed in a child process next to the GPU legs of this run"}


def spawn_workers(args):
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes of this script (one per GPU, RCCL
    rendezvous on 127.0.0.1) BEFORE this process touches a GPU, relay rank 0's JSON line, exit with the worst code."""
    import socket
    n_dev = torch.cuda.device_count()             # does not initialise the GPU runtime
    if n_dev < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {n_dev} device(s) visible\n")
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.size, args.batch)))
        return
    if (args.gpus > 1 or args.force_spawn) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_workers(args))
    # stdout must carry exactly ONE line (the JSON).  Native libraries (RCCL prints a version banner at communicator
    # creation, flushed at exit) write to file descriptor 1 too, so fd 1 is pointed at stderr for the whole run and the
    # JSON is written to a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cpu_proc = None
    if rank == 0 and world == 1 and not args.no_cpu:
        # the CPU baseline runs NEXT TO the GPU legs in its own process (it never touches the GPU): the device is busy while
        # the driver samples it, and the run is not 40 s of idle GPU followed by 3 s of work
        cpu_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--size", str(args.size),
                                     "--batch", str(args.batch)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    import xlstm_hved_amd as X
    from xlstm_hved_amd import ops
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
    DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp32_mfma": torch.float32}
    dtype = DT[args.dtype]
    ops.set_fp32_mfma(args.dtype == "fp32_mfma")              # process-wide switch, read when a conv is launched (= at capture)
    S, B = args.size, args.batch

    torch.manual_seed(1)                                      # same weights on every rank
    model = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    model.apply(X.init_weights)
    model = model.to(dev).train()
    model.noise_state(dev)
    model.seed_noise(20260 + rank)                             # reparameterisation noise: per rank (seed = base + rank)
    params = [p for p in model.parameters()]
    g = torch.Generator(device="cpu").manual_seed(1 + rank)   # per-rank synthetic patch
    x = torch.rand(B, 4, S, S, S, generator=g).to(dev, dtype)
    grads = X.parallel.FlatGrads(params)                       # p.grad = views of one flat fp32 bucket
    for kv in args.xh_option:
        k_, v_ = kv.split("=")
        X._lib.check(X._lib.load().xh_set_option(int(k_), int(v_)), "xh_set_option")
    ops.set_level_streams(args.level_streams)
    ops.set_wgrad_early_flush(args.wgrad_early_flush)
    ops.set_wgrad_flush_streams(args.wgrad_flush_streams)
    ops.set_wgrad_overlap(args.wgrad_overlap, args.wgrad_batch)
    ops.set_wgrad_defer(not args.no_wgrad_defer and not args.wgrad_overlap)   # weight gradients batched at the end of backward           # weight gradients on a second HIP stream, joined once per step

    def make_compute(xin, split=False):
        # fp16 storage: the activation gradients need the caller's loss scaling, as the reference's GradScaler provides
        # (train.py:207,265-268; initial scale 65536); the unscale of the fp32 parameter gradients is part of the step.
        # split (fp32_mfma): fp32 storage, but the activation gradients enter the matrix cores as fp16 pairs -- same scale
        scale = LOSS_SCALE_FP16 if (xin.dtype == torch.float16 or split) else 1.0
        # the backward pass is seeded with a resident scalar (the loss scale, or 1): d(scale * loss) / d(loss), without the
        # multiply launch and without the ones-fill autograd issues for an implicit seed
        seed = torch.full((), scale, dtype=torch.float32, device=xin.device)

        def compute():
            grads.zero()
            seg, (mu, lv), rec = model(xin, [14], recon=True)
            loss = bench_loss(seg, mu, lv, rec[0])
            loss.backward(seed if loss.dim() == 0 else seed.view(loss.shape))
            ops.join_wgrad_stream()                           # the weight-gradient branch rejoins the step here
            if scale != 1.0:
                grads.flat.mul_(1.0 / scale)
        return compute
    compute = make_compute(x, args.dtype == "fp32_mfma")

    def step():
        compute()
        if use_dist:
            grads.all_reduce(world)                             # one in-place RCCL all-reduce of the bucket

    def sync_all():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (eager) + capture ------------------------------------------------------------------
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        ops.prepare_capture()                                 # weight-pack job table of the step's convolutions, fan-in block
        graph = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread may query events while this thread captures; that must not abort the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if use_dist else "global"):
            compute()                                           # the collective stays outside the capture

    def run_graph():
        graph.replay()
        if use_dist:
            grads.all_reduce(world)
    run = run_graph if graph is not None else step
    for _ in range(args.warmup):
        run()
    sync_all()
    # Inner repeats: the driver's K may be small (20 steps = 0.08 s of GPU time, invisible to its utilisation sampler), so
    # each of the K timed steps replays the pass `inner` times, sized from a short probe so that the timed region lasts
    # >= --min-timed-s; ms_per_step and value are per PASS (one forward + backward [+ all-reduce]), inner_repeats is reported.
    inner = max(1, args.inner)
    if args.inner == 0:
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        sync_all()
        probe = torch.tensor([(time.perf_counter() - t0) / 3.0], device=dev, dtype=torch.float64)
        if use_dist:
            import torch.distributed as dist
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)        # every rank must choose the same count
        inner = max(1, min(1000, int(-(-args.min_timed_s // (probe.item() * max(args.steps, 1))))))
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(inner):
            run()
    sync_all()
    dt = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    passes = args.steps * inner
    ms = dt / passes * 1e3
    value = world * B * S ** 3 / (dt / passes)

    out = {
        "metric": "voxels/sec fwd+bwd, 4-modality 128^3 patch", "value": value, "unit": "voxels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "inner_repeats": inner, "timed_region_s": dt,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"XLSTM_HVED fwd+bwd, {B}x4x{S}^3 patch per GPU, f_maps=4 'ilc' (train.py:142-143), "
                               f"train mode, subset [14], recon=True, {STORAGE_NOTE[args.dtype]}, "
                               f"random-init weights, {'hipGraph replay' if graph is not None else 'eager'}",
                   "parallelism": f"dp{world}", "per_gpu_batch": B, "global_batch": B * world},
    }
    out["parity"] = MODE_PARITY[args.dtype] + "  [test-suite figures; this run's own measurement: parity_measured]"
    if rank == 0 and world == 1 and not args.no_modes:
        # the same step in the other storage modes (same weights, same patch, hipGraph replay, no collective), so the
        # parity-mode throughput is measured in the same run as the headline
        modes = {args.dtype: {"ms_per_step": ms, "voxels_per_s": B * S ** 3 / (ms * 1e-3), "parity": MODE_PARITY[args.dtype]}}
        for name, dt_ in DT.items():
            if name == args.dtype:
                continue
            ops.set_fp32_mfma(name == "fp32_mfma")
            try:
                ms_m = time_graph(make_compute(x.to(dt_), name == "fp32_mfma"), args.steps, args.warmup, thread_local=use_dist)
            finally:
                ops.set_fp32_mfma(args.dtype == "fp32_mfma")
            modes[name] = {"ms_per_step": ms_m, "voxels_per_s": B * S ** 3 / (ms_m * 1e-3), "parity": MODE_PARITY[name]}
        out["modes"] = modes
        # measured, not quoted: every mode's mask on the 128^3 parity case against the reference's own mask
        try:
            mp = measured_parity(dev, DT)
            out["parity_measured"] = mp
            for name in modes:
                modes[name]["parity_measured"] = mp["modes"].get(name)
        except Exception as e:
            out["parity_measured_error"] = repr(e)[:200]
    if rank == 0 and world == 1 and not args.no_config3:
        try:
            out["config3"] = config3_leg(model, grads, dev, dtype, S, max(10, min(args.steps, 40)), 3)
        except Exception as e:
            out["config3_error"] = repr(e)[:200]
    if rank == 0 and world == 1 and not args.no_config3 and S == 128:
        # BASELINE config 5 on one GPU: sliding-window inference of one whole volume, fp16 as that config names
        try:
            out["tiler"] = tiler_leg(model, dev)
        except Exception as e:
            out["tiler_error"] = repr(e)[:200]
    if args.extras and rank == 0:
        out["extras"] = extras(model, x, grads, args.steps)
    if cpu_proc is not None:
        cpu_out, _ = cpu_proc.communicate()
        try:
            out["cpu_baseline"] = json.loads(cpu_out.decode().strip().splitlines()[-1])
        except Exception as e:                                  # the GPU result must survive a CPU-leg failure
            out["cpu_baseline"] = {"value": None, "unit": "voxels/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    # ---- roofline of the dominant kernel family: per-launch HIP events on the launch stream.  LAST, after the CPU leg has
    # finished: the instrumented pass is eager, and a host that is busy with the oracle's threads enqueues late enough for the
    # GPU to idle inside a bracket (the batched weight-gradient bracket read 369 us next to the CPU leg, 282 us under rocprof)
    if rank == 0 and not args.no_roofline:
        out["roofline"] = roofline_pass(compute, ops, min(args.steps, 5), dtype)      # rank-local: no collective inside
        try:
            out["roofline"]["hbm_stream_measured"] = stream_probe(dev)
        except Exception as e:
            out["roofline"]["hbm_stream_measured"] = {"error": repr(e)[:200]}
    if not args.no_trainstep:
        # the whole training step, LAST: TrainStep re-points the generator's .grad at its own bucket.  With
        # N > 1 every rank runs it data-parallel (TrainStep(group=...): two graphs, the two bucket all-reduces between / after)
        try:
            if use_dist:
                import torch.distributed as dist
                grp = dist.group.WORLD
            else:
                grp = None
            leg = train_step_leg(model, x, max(6, min(args.steps, 30)), group=grp, world=world,
                                 mfma_roofline=(rank == 0 and world == 1 and not args.no_roofline))   # (its eager pass would issue
                                                                                                         # rank 0's collectives alone)
            if rank == 0:
                out.update(leg)
        except Exception as e:                                  # must not cost the headline line
            out["train_step_error"] = repr(e)[:200]
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        import torch.distributed as dist
        dist.barrier()                                          # rank 0 arrives after its roofline pass
        dist.destroy_process_group()


def measured_parity(dev, modes, keep_mode=False):
    """MEASURED in this run, per storage mode: the thresholded segmentation of the 128^3 parity case against the mask the REAL
    reference produced for it (tests/golden/mask_trained_like_128.npz, written by tests/golden/make_mask_128.py from the imported
    reference; the CPU oracle reproduces it with 0 flips).  Case: weights_trained_like.npz (the reference trained for 300 CPU
    steps), tests/synth_blobs.blob_case(8, 1, 128), all four modalities, eval mode, posterior mean.  Dice as metrics.py:85-107
    (threshold 0.5, eps 1e-6), per region channel; `dice_dev` = the largest |Dice - 1| of the three."""
    import numpy as np
    import xlstm_hved_amd as X
    from xlstm_hved_amd import ops
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_blobs as SB
    gold = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gold, "mask_trained_like_128.npz"))
    shape = tuple(int(v) for v in z["shape"])
    ref = torch.from_numpy(np.unpackbits(z["bits"])[: int(np.prod(shape))].reshape(shape).astype(np.bool_)).to(dev)
    w = np.load(os.path.join(gold, "weights_trained_like.npz"))
    m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}, strict=True)
    m = m.to(dev).eval()
    x, _ = SB.blob_case(8, 1, 128)
    x = x.to(dev)
    was = ops._FP32_MFMA[0]
    out = {}
    try:
        for name, dt_ in modes.items():
            if not keep_mode:                                   # (keep_mode: the caller has set the arithmetic / storage policy)
                ops.set_fp32_mfma(name == "fp32_mfma")
            with torch.no_grad():
                seg = m(x.to(dt_), [14], recon=True, valid=True)[0]
            got = seg.float() > 0.5
            inter = (got & ref).sum((0, 2, 3, 4)).double()
            den = got.sum((0, 2, 3, 4)).double() + ref.sum((0, 2, 3, 4)).double()
            dice = (2 * inter + 1e-6) / (den + 1e-6)
            out[name] = {"dice_dev": float((dice - 1).abs().max()), "dice": [float(v) for v in dice],
                         "mask_flips": int((got != ref).sum()), "mask_voxels": int(ref.numel()),
                         "meets_1e-4": bool(float((dice - 1).abs().max()) <= 1e-4)}
    finally:
        ops.set_fp32_mfma(was)
    return {"case": "trained-like weights (reference trained 300 CPU steps), synthetic blob patch 1x4x128^3, subset [14], eval, posterior "
                    "mean; target = the REAL reference's fp32 mask (tests/golden/mask_trained_like_128.npz)",
            "positives_per_channel": [int(v) for v in z["pos"]], "modes": out}


def config3_leg(model, grads, dev, dtype, size, nsteps, warmup):
    """BASELINE config 3 (RA_HVED.py:513-520,588-594): N = 2 at 128^3 with per-sample modality dropout, the two
    samples' subsets drawn anew for every step from the 15 subsets, forward + backward through ONE captured hipGraph.  The graph
    holds `model(x, [14], instance_missing=True, recon=True)` -- the reference's own detection of dropped modalities
    (`x.sum((2,3,4)) == 0`, RA_HVED.py:515) runs on the device inside the graph, PoE2 takes the resulting per-sample mask -- so a
    step only rewrites the input buffer: the full patch times a (2, 4) keep mask (what the reference's data pipeline hands over)."""
    import xlstm_hved_amd as X
    from xlstm_hved_amd import ops
    g = torch.Generator(device="cpu").manual_seed(3)
    xfull = torch.rand(2, 4, size, size, size, generator=g).to(dev, dtype)
    xin = xfull.clone()
    table = torch.tensor([[1.0 if c in X.SUBSETS_MODALITIES[k] else 0.0 for c in range(4)] for k in range(15)], device=dev).to(dtype)
    n_all = warmup + nsteps + 2
    draws = torch.randint(0, 15, (n_all, 2), generator=g)
    keeps = [table[draws[i].to(dev)].view(2, 4, 1, 1, 1) for i in range(n_all)]       # built before the timed region
    scale = LOSS_SCALE_FP16 if dtype == torch.float16 else 1.0
    seed = torch.full((), scale, dtype=torch.float32, device=dev)

    def compute():
        grads.zero()
        seg, (mu, lv), rec = model(xin, [14], instance_missing=True, recon=True)
        bench_loss(seg, mu, lv, rec[0]).backward(seed)
        ops.join_wgrad_stream()
        if scale != 1.0:
            grads.flat.mul_(1.0 / scale)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.mul(xfull, keeps[-1], out=xin)
        compute()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    from xlstm_hved_amd import ops as _ops
    _ops.prepare_capture()
    with torch.cuda.graph(graph):
        compute()
    for i in range(warmup):
        torch.mul(xfull, keeps[i], out=xin)
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + nsteps):
        torch.mul(xfull, keeps[i], out=xin)                   # this step's dropout pattern (inside the timed region)
        graph.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / nsteps * 1e3
    finite = bool(torch.isfinite(grads.flat).all())
    return {"workload": f"XLSTM_HVED fwd+bwd, 2x4x{size}^3 per GPU, instance_missing=True, both samples' modality subsets drawn per step "
                        f"from the 15 subsets (seeded), one captured hipGraph, input mask rewritten per step inside the timed region",
            "ms_per_step": ms, "voxels_per_s": 2 * size ** 3 / (ms * 1e-3), "steps": nsteps, "dtype": str(dtype).replace("torch.", ""),
            "subsets_first_steps": draws[warmup:warmup + 6].tolist(), "gradients_finite": finite}


def train_step_leg(model, x, nsteps, group=None, world=1, mfma_roofline=False):
    """the reference's whole optimisation step (train.py:208-296 without the optimizer updates) -- two
    shared-encoder generator forwards, the HIP loss epilogues, three passes of the ks=4 Discriminator of train.py:146 (csrc/dconv.hip
    implicit GEMMs on the matrix cores) and both backward passes -- captured ONCE into a hipGraph and replayed with a different
    modality subset every step (train.py:222-223), the subset entering as a device mask.  With a process group the step is
    data parallel: two graphs, the generator bucket's all-reduce on a communication stream under the discriminator passes, the
    discriminator bucket's (44.3 MB) after them; the slowest rank's time is reported."""
    import xlstm_hved_amd as X
    from xlstm_hved_amd.train_step import TrainStep
    torch.manual_seed(2)
    disc = X.Discriminator(in_channels=7, ks=4, strides=[1, 2, 2, 2])            # train.py:146
    disc.apply(X.init_weights)
    disc = disc.to(x.device)
    ts = TrainStep(model, disc, storage=x.dtype, group=group, world_size=world if group is not None else None)
    mask = (torch.rand(x.shape[0], 3, *x.shape[2:], device=x.device) > 0.7).float()
    xf = x.float()
    ts.capture(xf, mask)
    subsets = [[3], [6], [12], [0], [9], [13], [14]]

    def sync():
        if group is not None:
            import torch.distributed as dist
            dist.barrier(group)
        torch.cuda.synchronize()
    for i in range(3):
        ts.replay(xf, mask, subsets[i % len(subsets)], update=False)
    sync()
    t0 = time.perf_counter()
    for i in range(nsteps):
        ts.replay(xf, mask, subsets[i % len(subsets)], update=False)
    sync()
    dt = time.perf_counter() - t0
    if group is not None:
        import torch.distributed as dist
        t = torch.tensor([dt], device=x.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dt = t.item()
    ms = dt / nsteps * 1e3
    gflop_d = 561.4 * x.shape[0] * (x.shape[2] / 128.0) ** 3       # one ks=4 discriminator forward (tools/microbench_disc.py)
    res = {"train_step_graph_ms": ms, "train_step_ranks": world,
           "train_step_note": ("train.py:208-296 without the optimizer updates, ONE captured hipGraph replayed with a new modality "
                               f"subset per step: 2 generator forwards (shared encoder) + Dice/MSE/KLD/LSGAN epilogues + 3 passes of "
                               f"Discriminator(in_channels=7, ks=4, strides=[1,2,2,2]) (~{gflop_d:.0f} GFLOP per forward pass) + both "
                               f"backward passes; {nsteps} replays timed, subset mask and inputs rewritten before each"
                               + ("" if group is None else f"; data parallel over {world} rank(s): two graphs, generator bucket "
                                  f"({ts.grads.flat.numel() * 4 / 1e6:.2f} MB) all-reduced on a communication stream under the "
                                  f"discriminator passes, discriminator bucket ({ts.grads_d.flat.numel() * 4 / 1e6:.1f} MB) after them"))}
    if mfma_roofline:
        try:
            res["roofline_mfma"] = mfma_roofline_pass(ts, xf, mask)
        except Exception as e:
            res["roofline_mfma_error"] = repr(e)[:200]
    return res


class GpuDelay:
    """A device-side delay that keeps the WHOLE chip busy (a stream of HBM-bound element-wise passes over a 256 MB buffer, like
    the step itself): an eager instrumented step queued behind it runs back to back as in a graph replay.  Behind a one-thread
    spin kernel (torch.cuda._sleep) the GPU sits idle for tens of milliseconds, power management lowers the clocks, and the
    first brackets of the step read up to 1.8x long (measured: 48 us instead of 27 for the dominant conv instance)."""

    def __init__(self):
        self.junk = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
        for _ in range(5):
            self.junk.add_(1.0)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(50):
            self.junk.add_(1.0)
        c1.record()
        torch.cuda.synchronize()
        self.pass_ms = max(c0.elapsed_time(c1) / 50.0, 1e-3)

    def __call__(self, ms):
        for _ in range(max(1, int(ms / self.pass_ms))):
            self.junk.add_(1.0)

    def empty_pair_ms(self, delay_ms):
        """Median cost of an empty event pair under the same queued conditions (subtracted from every bracket)."""
        self(delay_ms)
        empties = []
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            empties.append((e0, e1))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in empties)[len(empties) // 2]


def mfma_roofline_pass(ts, x, mask, nsteps=3):
    """The MFMA-bound objects of the path: every implicit-GEMM launch of the Discriminator inside the REAL training
    step (forward of the fake and the real sample, data gradients, weight gradients; csrc/dconv.hip) bracketed by HIP events on
    the launch stream, the eager step queued behind a device-side delay like roofline_pass.  flops = 2 * (conv output voxels) *
    ks^3 * Cin * Cout with the REAL channel counts (7 input channels, not the padded 8; the 512 -> 1 conv's padding to 32 is not
    counted).  The dominant instance by total time is the object; the others are listed."""
    from xlstm_hved_amd import disc as D
    recs = []
    ev = lambda: torch.cuda.Event(enable_timing=True)
    orig = (D._conv_into, D._conv, D._wgrad)

    def vox(sp):
        return sp[0] * sp[1] * sp[2]

    def conv_into(y, x_, wp, bias, mode, stride, n, sp_in, sp_out, cs, cn, **kw):
        e0, e1 = ev(), ev()
        e0.record()
        r = orig[0](y, x_, wp, bias, mode, stride, n, sp_in, sp_out, cs, cn, **kw)
        e1.record()
        ks = kw.get("ks", 3)
        recs.append((f"dconv_cl forward {min(cs, 7) if cs == 8 else cs}->{cn} s{stride} @{'x'.join(map(str, sp_out))}", e0, e1,
                     2.0 * n * vox(sp_out) * ks ** 3 * (7 if cs == 8 else cs) * cn))
        return r

    def conv(x_, wp, bias, mode, stride, n, sp_in, sp_out, cs, cn, **kw):
        e0, e1 = ev(), ev()
        e0.record()
        r = orig[1](x_, wp, bias, mode, stride, n, sp_in, sp_out, cs, cn, **kw)
        e1.record()
        ks = kw.get("ks", 3)
        o = sp_out if mode == 0 else sp_in                    # conv-output extents (a data gradient's SOURCE)
        cs_r, cn_r = (1 if cs == 32 else cs), (7 if cn == 8 else cn)
        recs.append((f"dconv_cl {'forward' if mode == 0 else 'data gradient'} {cn_r if mode else cs_r}{'<-' if mode else '->'}"
                     f"{cs_r if mode else cn_r} s{stride} @{'x'.join(map(str, o))}" + (" +mask/bias epilogue" if kw.get("mask") is not None else ""),
                     e0, e1, 2.0 * n * vox(o) * ks ** 3 * cs_r * cn_r))
        return r

    def wgrad(x_, dy, stride, n, sp_in, sp_out, cs, cn, ks=3, gs=None, **kw):
        e0, e1 = ev(), ev()
        e0.record()
        r = orig[2](x_, dy, stride, n, sp_in, sp_out, cs, cn, ks=ks, gs=gs, **kw)
        e1.record()
        cs_r, cn_r = (7 if cs == 8 else cs), (1 if cn == 32 else cn)
        recs.append((f"dwgrad_cl {cs_r}->{cn_r} s{stride} @{'x'.join(map(str, sp_out))}" + ("" if kw.get("dwp") is not None else " (+ zero fill of the packed gradient)"), e0, e1,
                     2.0 * n * vox(sp_out) * ks ** 3 * cs_r * cn_r))
        return r
    delay = GpuDelay()
    keep = ts.keep_mask([6], x.shape[0])
    D._conv_into, D._conv, D._wgrad = conv_into, conv, wgrad
    try:
        t_h = time.perf_counter()
        ts.compute(x, mask, keep)
        host_ms = (time.perf_counter() - t_h) * 1e3
        torch.cuda.synchronize()
        recs.clear()
        delay_ms = min(2.0 * host_ms + 30.0, 900.0)
        overhead = delay.empty_pair_ms(delay_ms)
        for _ in range(nsteps):
            delay(delay_ms)
            ts.compute(x, mask, keep)
        torch.cuda.synchronize()
    finally:
        D._conv_into, D._conv, D._wgrad = orig
    agg = {}
    for name, e0, e1, fl in recs:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += max(e0.elapsed_time(e1) - overhead, 1e-3)
        a[2] += fl
    rows = {k: {"launches_per_step": v[0] / nsteps, "avg_launch_us": v[1] / v[0] * 1e3, "tflops": v[2] / v[1] / 1e9,
                "frac_of_dense_peak": v[2] / v[1] / 1e9 / BF16_MFMA_PEAK_TFLOPS} for k, v in agg.items()}
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    name, (cnt, ms_sum, fl_sum) = dom
    tfl = fl_sum / ms_sum / 1e9
    total_ms = sum(v[1] for v in agg.values()) / nsteps
    return {"bound": "mfma", "achieved": tfl, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / BF16_MFMA_PEAK_TFLOPS,
            "kernel": name, "launches_per_step": cnt / nsteps, "avg_launch_us": ms_sum / cnt * 1e3,
            "algorithmic_flops_per_launch": fl_sum / cnt, "traffic": None,
            "discriminator_gemm_ms_per_step": total_ms,
            "discriminator_gemm_tflops_overall": sum(v[2] for v in agg.values()) / sum(v[1] for v in agg.values()) / 1e9,
            "instances": dict(sorted(rows.items(), key=lambda kv: -kv[1]["avg_launch_us"] * kv[1]["launches_per_step"])),
            "timing": "HIP events on the launch stream around every implicit-GEMM launch of the Discriminator inside the eager "
                      "training step, queued behind a device-side delay; minus the median empty event pair",
            "event_pair_overhead_us": overhead * 1e3, "host_enqueue_ms_per_step": host_ms}


def time_graph(fn, nsteps, warmup=3, thread_local=False):
    """ms per replay of `fn` captured once into a hipGraph (one untimed eager pass first)."""
    from xlstm_hved_amd import ops as ops_
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    ops_.prepare_capture()                                   # weight-pack job table of the step's convolutions, fan-in block
    with torch.cuda.graph(g, capture_error_mode="thread_local" if thread_local else "global"):
        fn()
    for _ in range(warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nsteps * 1e3


def extras(model, x, grads, nsteps):
    """Not part of the headline metric: forward-only time (SURVEY 8(d) asks for it) and the training step's two
    forwards (`[14]` + a random subset, train.py:224-225) with and without the shared encoder (forward_shared)."""
    def timed(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        from xlstm_hved_amd import ops as _ops
        _ops.prepare_capture()
        with torch.cuda.graph(g):
            fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / nsteps * 1e3

    def fwd_only():
        with torch.no_grad():
            model(x, [14], recon=True)

    def two_plain():
        grads.zero()
        a = model(x, [14], recon=True)
        b = model(x, [6], recon=True)
        (bench_loss(a[0], a[1][0], a[1][1], a[2][0]) + bench_loss(b[0], b[1][0], b[1][1], b[2][0])).backward()

    def two_shared():
        grads.zero()
        a, b = model.forward_shared(x, [dict(subset_idx_list=[14]), dict(subset_idx_list=[6])], recon=True)
        (bench_loss(a[0], a[1][0], a[1][1], a[2][0]) + bench_loss(b[0], b[1][0], b[1][1], b[2][0])).backward()
    res = {"forward_only_ms": timed(fwd_only), "two_forwards_fwd_bwd_ms": timed(two_plain),
           "two_forwards_shared_encoder_fwd_bwd_ms": timed(two_shared)}
    # the reference's whole optimisation step (train.py:208-296) -- two shared-encoder generator forwards, the
    # HIP loss epilogues, three discriminator passes (channels-last implicit-GEMM MFMA convolutions) and both backward passes
    try:
        import xlstm_hved_amd as X
        from xlstm_hved_amd.train_step import TrainStep
        torch.manual_seed(2)
        disc = X.Discriminator(in_channels=7, ks=4, strides=[1, 2, 2, 2])            # train.py:146
        disc.apply(X.init_weights)
        disc = disc.to(x.device)
        ts = TrainStep(model, disc, storage=x.dtype)
        mask = (torch.rand(x.shape[0], 3, *x.shape[2:], device=x.device) > 0.7).float()
        keep6 = ts.keep_mask([6], x.shape[0])

        def train_step():
            ts.compute(x, mask, keep6)
        for _ in range(2):
            train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(3, nsteps // 4)):
            train_step()
        torch.cuda.synchronize()
        res["train_step_eager_ms"] = (time.perf_counter() - t0) / max(3, nsteps // 4) * 1e3
        res["train_step_note"] = ("train.py:208-296 without the optimizer updates: 2 generator forwards (shared encoder) + Dice/MSE/KLD/"
                                  "LSGAN epilogues on HIP + 3 Discriminator passes (csrc/dconv.hip implicit GEMMs) + both backward passes")
        try:
            res["train_step_graph_ms"] = time_graph(train_step, max(3, nsteps // 2))
        except Exception as e:
            res["train_step_graph_error"] = repr(e)[:200]
    except Exception as e:                                    # an extras failure must not cost the headline line
        res["train_step_error"] = repr(e)[:200]
    # one 240 x 240 x 155 volume, 128^3 windows every 64 voxels (18 windows), posterior mean, eval mode
    from xlstm_hved_amd.inference import eval_overlap_volume
    vol = torch.rand(1, 4, 240, 240, 155, device=x.device).to(x.dtype)
    model.eval()
    try:
        for use_graph in (False, True):
            eval_overlap_volume(model, vol, 14, use_graph=use_graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eval_overlap_volume(model, vol, 14, use_graph=use_graph)
            torch.cuda.synchronize()
            res["tiler_240x240x155_18_windows_ms" + ("_graph" if use_graph else "_eager")] = (time.perf_counter() - t0) * 1e3
    finally:
        model.train()
    return res


def tiler_leg(model, dev, reps=3):
    """One 240 x 240 x 155 volume (BraTS extent), 128^3 windows every 64 voxels = 18 windows, posterior mean, eval mode, fp16 storage,
    the window forward replayed from a captured hipGraph (xlstm_hved_amd.inference.eval_overlap_volume).  Volume resident in HBM."""
    from xlstm_hved_amd.inference import eval_overlap_volume, window_list
    vol = torch.rand(1, 4, 240, 240, 155, device=dev).half()
    was = model.training
    model.eval()
    try:
        eval_overlap_volume(model, vol, 14, use_graph=True)           # captures the window forward
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            prob = eval_overlap_volume(model, vol, 14, use_graph=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
    finally:
        model.train(was)
    nwin = len(window_list((240, 240, 155), (128, 128, 128), (64, 64, 64)))
    return {"workload": "sliding-window inference, 1x4x240x240x155 volume, %d windows of 128^3 every 64 voxels, subset 14, fp16" % nwin,
            "ms_per_volume": ms, "volume_voxels_per_s": 240 * 240 * 155 / (ms * 1e-3), "windows": nwin,
            "output_finite": bool(torch.isfinite(prob).all())}


def stream_probe(dev):
    """What plain streaming reaches on this device with 1 GiB tensors (nothing stays in the memory-side cache): the practical
    ceiling next to `peak` (the nominal 8 TB/s every fraction in `roofline` is quoted against).  Stock ATen kernels, 10 launches each."""
    n = 2 ** 29
    a = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)

    def t(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    r = {"copy": 2 * n * 2 / t(lambda: b.copy_(a)) / 1e9, "add_2r1w": 3 * n * 2 / t(lambda: torch.add(a, b, out=b)) / 1e9,
         "fill": n * 2 / t(lambda: b.zero_()) / 1e9, "unit": "GB/s",
         "what": "1 GiB bf16 tensors: b.copy_(a), torch.add(a, b, out=b), b.zero_(); bytes moved / time"}
    del a, b
    return r


def roofline_pass(step, ops, nsteps, dtype):
    """Per-launch timing of the conv kernels inside the real step.

    Every xh_conv3d_fwd (forward convs + stride-1 data gradients) and xh_conv3d_wgrad call of `nsteps` steps is
    bracketed by HIP events on torch's current stream (= the stream the C ABI launches on).  The pass runs eagerly,
    so each step is queued behind a ~60 ms device-side delay: the host finishes enqueueing the step while the GPU
    waits, and the kernels then execute back to back exactly as in the hipGraph replay (without it the GPU idles
    between launches and the clocks drop).  Launches are grouped by the kernel template instance the library
    reports (xh_last_conv_kernel, the spelling rocprofv3 prints); the instance with the largest total time is the
    dominant kernel.  Algorithmic work per launch comes from the launch's shapes: bytes = every input and output
    element once at its storage size (+ fp32 weights / weight gradients), flops = 2*out_elements*k^3*Cin/groups."""
    records = []
    overlap_was = ops._WG["on"]
    ops.set_wgrad_overlap(False)          # per-launch brackets need every conv launch on the one (current) stream
    orig_fwd, orig_wg = ops.conv3d, ops.conv3d_wgrad
    esz = 4 if dtype == torch.float32 else 2

    def ev():
        return torch.cuda.Event(enable_timing=True)

    c1_meta = []                                              # k = 1 convs recorded between conv1x1_collect() and conv1x1_flush()

    def timed_fwd(xa, xb, weights, biases, **kw):
        collected = ops._C1_COLLECT[0] is not None and kw["k"] == 1      # not launched here: ops.conv1x1_flush issues it (timed_c1_flush)
        n_before = len(ops._C1_COLLECT[0]) if collected else 0
        e0, e1 = ev(), ev()
        if not collected:
            e0.record()
        res = orig_fwd(xa, xb, weights, biases, **kw)
        if not collected:
            e1.record()
        y = res[0] if isinstance(res, tuple) else res         # (y, sc, sh, mean, rstd) when the norm finalisation is fused
        bc = kw.get("bcast", 0)                               # broadcast operand: the tensor read has a quarter of the logical channels
        cin = xa.shape[1] * (bc if bc and not kw.get("transposed") else 1) + (xb.shape[1] if xb is not None else 0)
        k, groups = kw["k"], kw.get("groups", 1)
        in_el = xa.numel() + (xb.numel() if xb is not None else 0)
        cout_l = kw["cout"]
        out_sp = tuple(y.shape[2:]) if y is not None else tuple(xa.shape[2:])
        vox = xa.shape[0] * out_sp[0] * out_sp[1] * out_sp[2]
        y_el = y.numel() if y is not None else 0              # (a data gradient that only sums stores nothing)
        e_el = (kw["e"][0].numel() if bc else cout_l * vox) if kw.get("epi", 0) == 1 else 0     # epi 1 also reads the saved activation once
        nbytes = (in_el + y_el + e_el) * esz + sum(w.numel() for w in weights) * 4
        flops = 2.0 * cout_l * vox * k ** 3 * cin / groups
        shape = (f"k{k} s{kw.get('stride', 1)} g{groups} {cin}->{cout_l} @{'x'.join(map(str, out_sp))}"
                 + (" dgrad" if kw.get("transposed") else "") + (" bcast" if bc else ""))
        if collected and ops._C1_COLLECT[0] is not None and len(ops._C1_COLLECT[0]) > n_before:
            c1_meta.append((nbytes, flops, shape))
            return res
        if collected:                                          # the call was not taken by the collector after all: launched, unbracketed
            return res
        records.append((ops.last_conv_kernel(), e0, e1, nbytes, flops, shape, 1))
        return res

    orig_c1_flush = ops.conv1x1_flush

    def timed_c1_flush():
        """The collected k = 1 convs of the latent path: ONE bracket around the multi-problem launch(es) that carry them."""
        metas = c1_meta[:]
        c1_meta.clear()
        e0, e1 = ev(), ev()
        e0.record()
        r = orig_c1_flush()
        e1.record()
        if metas:
            nl = -(-len(metas) // 4)
            records.append((ops.last_conv_kernel(), e0, e1, sum(m[0] for m in metas), sum(m[1] for m in metas),
                            f"{len(metas)} k1 convs in {nl} launch(es): " + "; ".join(m[2] for m in metas), nl))
        return r

    pending_meta = []

    def timed_wg(xa, xb, dy, dws, dbs, **kw):
        cin = xa.shape[1] * (kw.get("bcast", 0) or 1) + (xb.shape[1] if xb is not None else 0)
        k, groups = kw["k"], kw.get("groups", 1)
        in_el = xa.numel() + (xb.numel() if xb is not None else 0)
        nbytes = (in_el + dy.numel()) * esz + sum(w.numel() for w in dws) * 4
        flops = 2.0 * dy.numel() * k ** 3 * cin / groups
        shape = f"k{k} s{kw.get('stride', 1)} g{groups} {cin}->{dy.shape[1]} @{'x'.join(map(str, dy.shape[2:]))} wgrad"
        if kw.get("side") and ops._WG["defer"]:
            # deferred: launched by the batch flush at the end of backward (timed_flush); remember what it will carry
            # the class decides which launch of xh_conv3d_wgrad_batch carries the problem (mirrors its grouping rule):
            # quad-channel kernel per input-quad count (8 problems per launch), else the implicit-GEMM kernel per volume
            # class (7 per launch), else one ordinary launch each
            cg, og, w_ = cin // groups, dy.shape[1] // groups, xa.shape[-1]
            k3 = k == 3 and kw.get("stride", 1) == 1 and esz == 2
            if cg == 1 and og == 1 and groups % 4 == 0:          # depthwise: groups of 4 with diagonal blocks (wgrad_q4 plan)
                cg = og = 4
            q4 = (k3 and w_ % 32 == 0 and cg % 4 == 0 and og % 4 == 0 and cg <= 48 and og <= 48 and xa.shape[2] >= 4
                  and xa.shape[3] >= 4 and (xa.shape[1] % 4 == 0 or kw.get("bcast")))
            mfma = k3 and (w_ % 32 == 0 or w_ in (8, 16)) and cg >= 4
            qs = cg // 4
            qs = qs if qs <= 3 else 3 if qs % 3 == 0 else 2 if qs % 2 == 0 else 1      # input quads per unit (wgrad_q4 plan)
            s2 = k == 3 and kw.get("stride", 1) == 2 and ((dy.shape[1] // groups) % 4 == 0 or dy.shape[1] // groups == 2) and w_ % (16 // esz) == 0
            tiny = (k == 3 and kw.get("stride", 1) == 1 and groups == 1 and (cin, dy.shape[1]) in ((1, 2), (2, 1))
                    and w_ % (16 // esz) == 0 and kw.get("pre") is None)
            k7 = k == 7 and esz == 2 and cin == 4 and dy.shape[1] == 2 and groups == 1 and w_ % 32 == 0 and kw.get("pre") is None
            k1 = k == 1 and kw.get("stride", 1) == 1 and (dy.shape[2] * dy.shape[3] * dy.shape[4]) % (16 // esz) == 0
            # rows of 32 / 64 / 128 voxels, H a multiple of 8: the full-row kernel (conv3d_wgrad_q5.hip), 12 problems per launch
            q5 = q4 and w_ in (32, 64, 128) and xa.shape[3] % 8 == 0
            cls = ("q5" if q5 else f"q4_{qs}" if q4 else ("big" if dy.shape[2] * dy.shape[3] * dy.shape[4] >= (1 << 20) else "small")
                   if mfma else "k1" if k1 else "s2" if s2 else "k7" if k7 else "tiny" if tiny else "rest")
            pending_meta.append((cls, nbytes, flops, shape))
            return orig_wg(xa, xb, dy, dws, dbs, **kw)
        e0, e1 = ev(), ev()
        e0.record()
        r = orig_wg(xa, xb, dy, dws, dbs, **kw)
        e1.record()
        records.append((ops.last_conv_kernel(), e0, e1, nbytes, flops, shape, 1))
        return r

    # ---- the norm / element-wise family, measured in THIS run: every call of the non-conv stage entry points (statistics,
    # normalisation forward / backward, activation backward, pooling, trilinear up-sampling and adjoint, gates, DuSE, skip-return
    # tail, PoE, loss reductions, fills) bracketed the same way.  A call is one launch (a few are two).
    ELT = ("moments moments2 norm_finalize affine_act in_affine_act bn_affine_act bn_affine_act2 norm_bwd_fused2 act_bwd_reduce "
           "norm_bwd_coef norm_bwd_apply norm_bwd_fused in_bwd_apply in_bwd_apply2 maxpool2 maxpool2_bwd upsample upsample_bwd "
           "upsample2x_in_act upsample2x_bwd_act_reduce add act_bwd channel_pool channel_pool_bwd gate gate_bwd channel_pool2 "
           "channel_pool2_bwd gate2 gate2_bwd gate_maxpool gate_maxpool_bwd duse_gate duse_gate_bwd duse_gate_fc rank1_add_fc "
           "rank1_add skr_tail skr_tail_bn skr_tail_bwd duse_fc_fwd duse_fc_bwd poe_fwd poe_bwd poe_fwd_multi poe_bwd_multi "
           "compose_multi pair_sums lincomb loss_finalize kld_fwd kld_bwd nested_weight multi_sum multi_fill fill "
           "in_affine_act_multi act_bwd_reduce_multi in_bwd_apply_multi upsample2x_in_act_multi upsample2x_bwd_act_reduce_multi").split()
    ELT = [n_ for n_ in ELT if hasattr(ops, n_)]
    elt_orig = {n_: getattr(ops, n_) for n_ in ELT}
    elt_records = []

    def elt_wrap(name_, fn_):
        def timed(*a_, **k_):
            e0, e1 = ev(), ev()
            e0.record()
            r_ = fn_(*a_, **k_)
            e1.record()
            elt_records.append((name_, e0, e1))
            return r_
        return timed
    elt_timed = {n_: elt_wrap(n_, f_) for n_, f_ in elt_orig.items()}

    def elt_install(on):
        for n_ in ELT:
            setattr(ops, n_, elt_timed[n_] if on else elt_orig[n_])

    orig_flush = ops._flush_deferred

    def timed_flush():
        """The deferred weight gradients, one bracket per kernel class: the k=3 MFMA problems of a volume class share
        launches (xh_conv3d_wgrad_batch packs 7 per launch), the k=1 problems too (20 per launch) and the stride-2 ones (4 per launch), the rest run one by one."""
        calls, metas = ops._WG["deferred"], pending_meta[:]
        ops._WG["deferred"] = []
        pending_meta.clear()
        if len(calls) != len(metas):                         # something bypassed the wrapper: do not attribute
            ops._WG["deferred"] = calls
            return orig_flush()
        for cls in ("q5", "q4_1", "q4_2", "q4_3", "k1", "s2", "k7", "tiny", "small", "big"):
            grp = [(c, m) for c, m in zip(calls, metas) if m[0] == cls]
            if not grp:
                continue
            ops._WG["deferred"] = [c for c, _ in grp]
            e0, e1 = ev(), ev()
            e0.record()
            orig_flush()
            e1.record()
            nl = -(-len(grp) // (12 if cls == "q5" else 8 if cls.startswith("q4") else 20 if cls == "k1" else 4 if cls in ("s2", "k7") else 8 if cls == "tiny" else 7))
            if cls == "k7":
                nl = 2 * nl if len(grp) > 1 else 2            # partial sums + second-stage reduction
            records.append((ops.last_conv_kernel(), e0, e1, sum(m[1] for _, m in grp), sum(m[2] for _, m in grp),
                            f"{len(grp)} {cls if cls in ('k1', 'k7') else 'k3'} weight gradients ({cls}) in {nl} launch(es): "
                            + "; ".join(m[3] for _, m in grp), nl))
        for c, m in zip(calls, metas):
            if m[0] != "rest":
                continue
            ops._WG["deferred"] = [c]
            e0, e1 = ev(), ev()
            e0.record()
            orig_flush()
            e1.record()
            records.append((ops.last_conv_kernel(), e0, e1, m[1], m[2], m[3], 1))
    gpu_delay = GpuDelay()                   # (see the class: keeps the whole chip busy so the clocks stay up)
    # the delay must outlast the host's enqueue time of one instrumented step (else the GPU catches up and a bracket
    # also spans host launch latency): time one instrumented enqueue, then wait 1.5x that (+20 ms), at most 600 ms
    ops.conv3d, ops.conv3d_wgrad, ops._flush_deferred, ops.conv1x1_flush = timed_fwd, timed_wg, timed_flush, timed_c1_flush
    elt_install(True)
    try:
        t_h = time.perf_counter()
        step()
        host_ms = (time.perf_counter() - t_h) * 1e3
        torch.cuda.synchronize()
    finally:
        ops.conv3d, ops.conv3d_wgrad, ops._flush_deferred, ops.conv1x1_flush = orig_fwd, orig_wg, orig_flush, orig_c1_flush
        elt_install(False)
    records.clear()
    elt_records.clear()
    delay_ms = min(2.0 * host_ms + 30.0, 600.0)
    # an event pair costs a few microseconds of its own (two marker packets): measure empty brackets under the same
    # queued conditions and subtract their median from every bracket
    overhead_ms = gpu_delay.empty_pair_ms(delay_ms)
    ops.conv3d, ops.conv3d_wgrad, ops._flush_deferred, ops.conv1x1_flush = timed_fwd, timed_wg, timed_flush, timed_c1_flush
    elt_install(True)
    whole = []
    try:
        for _ in range(nsteps):
            gpu_delay(delay_ms)
            w0, w1 = ev(), ev()
            w0.record()
            step()
            w1.record()
            whole.append((w0, w1))
        torch.cuda.synchronize()
    finally:
        ops.conv3d, ops.conv3d_wgrad, ops._flush_deferred, ops.conv1x1_flush = orig_fwd, orig_wg, orig_flush, orig_c1_flush
        elt_install(False)
    # element-wise family: per entry point, the n-th call of each step across the steps -> median, like the conv brackets
    elt_by = {}
    for name_, e0, e1 in elt_records:
        elt_by.setdefault(name_, []).append(max(e0.elapsed_time(e1) - overhead_ms, 5e-4))
    elt_tab, elt_ms, elt_calls = {}, 0.0, 0
    for name_, ts_ in elt_by.items():
        per = len(ts_) // nsteps
        if per and len(ts_) % nsteps == 0:
            tot = sum(sorted(ts_[st * per + j] for st in range(nsteps))[nsteps // 2] for j in range(per))
        else:
            per, tot = len(ts_) / nsteps, sum(ts_) / nsteps
        elt_tab[name_] = {"calls_per_step": per, "ms_per_step": tot}
        elt_ms += tot
        elt_calls += per
    # every (kernel instance, shape) occurs a fixed number of times per step; the n-th occurrence of each step forms one
    # sample group whose MEDIAN over the steps is taken (a host hiccup that lets the GPU catch up inflates single brackets)
    per_call = {}
    seen = {}
    for name, e0, e1, nbytes, flops, shape, nl in records:
        key = (name, shape)
        seen[key] = seen.get(key, 0) + 1
        per_call.setdefault(key, []).append((max(e0.elapsed_time(e1) - overhead_ms * nl, 1e-3), nbytes, flops, nl))
    agg = {}
    shape_us = {}                                             # mean bracket per launch of a (kernel, shape), us
    for (name, shape), calls in per_call.items():
        per_step = len(calls) // nsteps if len(calls) % nsteps == 0 else 0
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, {}])
        if per_step:
            for j in range(per_step):                         # j-th occurrence within a step, across the steps
                ts = sorted(calls[st * per_step + j][0] for st in range(nsteps))
                a[1] += ts[len(ts) // 2] * nsteps
        else:
            a[1] += sum(c[0] for c in calls)
        a[0] += sum(c[3] for c in calls)                      # launches (a batch bracket spans several)
        a[2] += sum(c[1] for c in calls)
        a[3] += sum(c[2] for c in calls)
        a[4][shape] = a[4].get(shape, 0) + sum(c[3] for c in calls)
        shape_us[(name, shape)] = sum(c[0] for c in calls) / max(sum(c[3] for c in calls), 1) * 1e3
    ops.set_wgrad_overlap(overlap_was)
    total_ms = sum(a[1] for a in agg.values())
    step_ms = sorted(a.elapsed_time(b) for a, b in whole)[len(whole) // 2]      # one instrumented step, back to back on the device
    n_brackets = len(records) / nsteps
    name, (cnt, ms_sum, bytes_sum, flops_sum, shapes) = max(agg.items(), key=lambda kv: kv[1][1])
    avg_ms, nbytes, flops = ms_sum / cnt, bytes_sum / cnt, flops_sum / cnt
    gbs = nbytes / (avg_ms * 1e-3) / 1e9
    tfl = flops / (avg_ms * 1e-3) / 1e12
    mfma = "mfma" in name or "q4" in name or "q5" in name   # the quad-channel kernels are MFMA kernels too
    peak_tf = BF16_MFMA_PEAK_TFLOPS if mfma else FP32_VALU_PEAK_TFLOPS
    # roofline: the kernel is HBM-bound when its arithmetic intensity is below peak_flops / peak_bandwidth
    if flops / nbytes < peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9):
        r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    else:
        r = {"bound": "mfma", "achieved": tfl, "peak": peak_tf, "unit": "TFLOP/s", "frac": tfl / peak_tf,
             "note": None if mfma else "fp32 FMA on the vector ALU (157.3 TFLOP/s); this instance does not use MFMA"}
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            t = json.load(f)
        if t.get("dtype") == {torch.float32: "fp32", torch.bfloat16: "bf16", torch.float16: "fp16"}[dtype]:
            ks = t.get("kernels", {})
            if name in ks:
                traffic = ks[name]["hbm_bytes_per_launch"]
            else:
                # a class of this run may be several template instances of the capture (the name of the class drops trailing
                # arguments, e.g. the broadcast-operand flag of conv3_q4w_kernel): the launch-weighted mean over them
                inst = [v for k, v in ks.items() if name.endswith(">") and k.startswith(name[:-1] + ",")]
                if inst:
                    traffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in inst) / sum(v["launches"] for v in inst)
    tmeta = {}
    if traffic is not None:
        # how old the committed capture is: the commit it was taken from, commits since (None where there is no .git, e.g. on a
        # gpurun box) and whether the kernel sources of THIS tree are the ones it was captured with (content hash)
        age = None
        if t.get("captured_commit"):
            try:
                age = int(subprocess.check_output(["git", "-C", ROOT, "rev-list", "--count", t["captured_commit"] + "..HEAD"],
                                                  stderr=subprocess.DEVNULL).decode())
            except Exception:
                age = None
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hashlib
            h = hashlib.sha1()
            croot = os.path.join(ROOT, "xlstm-hved_amd", "csrc")
            for f_ in sorted(os.listdir(croot)):
                if f_.endswith((".hip", ".h")):
                    h.update(f_.encode())
                    h.update(open(os.path.join(croot, f_), "rb").read())
            same = h.hexdigest() == t.get("csrc_sha1")
        except Exception:
            same = None
        tmeta = {"traffic_captured_commit": t.get("captured_commit"), "traffic_captured_date": t.get("captured_date"),
                 "traffic_age_commits": age, "traffic_kernel_sources_unchanged": same}
    r.update(tmeta)
    r.update({"traffic": traffic,
              "traffic_source": None if traffic is None else "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command; "
                                                              "a committed capture, not a measurement of this run: see traffic_*)",
              "kernel": name, "launches_per_step": cnt / nsteps, "avg_launch_us": avg_ms * 1e3,
              "shapes": {k: v / nsteps for k, v in shapes.items()},
              "shape_avg_us": {k: round(shape_us[(name, k)], 1) for k in shapes},
              "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops,
              "arithmetic_intensity_flop_per_byte": flops / nbytes, "tflops": tfl,
              "share_of_conv_time": ms_sum / total_ms, "conv_time_per_step_ms": total_ms / nsteps,
              # what the per-kernel fraction above does not see: everything of the step that is NOT a bracketed conv launch
              # (norm / element-wise passes, ViL, PoE, loss, packs, fills) = the instrumented step minus its conv brackets minus
              # the event pairs; the rocprof family split of the same step is in profiles/ (r06z_timeline.txt)
              "instrumented_step_ms": step_ms,
              "non_conv_ms_per_step": max(step_ms - total_ms / nsteps - (n_brackets + elt_calls) * overhead_ms, 0.0),
              # the same family, measured two ways: in THIS run (event brackets around every non-conv stage entry point of the
              # instrumented steps) and from the rocprofv3 kernel trace of this command committed under profiles/ (a file, not a
              # measurement of this run -- the key says so)
              "elementwise_in_run": {"ms_per_step": elt_ms, "calls_per_step": elt_calls,
                                     "top": dict(sorted(elt_tab.items(), key=lambda kv: -kv[1]["ms_per_step"])[:10]),
                                     "how": "HIP-event brackets around every norm / element-wise / PoE / loss entry point of ops.py "
                                            "inside the instrumented steps, minus the median empty pair"},
              "elementwise_from_committed_profile": elementwise_from_profile(),
              "other_conv_kernels": {k: {"launches_per_step": v[0] / nsteps, "avg_launch_us": v[1] / v[0] * 1e3,
                                          "GBps": v[2] / v[1] / 1e6, "frac_of_hbm_peak": v[2] / v[1] / 1e6 / HBM_PEAK_GBS,
                                          "shapes": {a_: b_ / nsteps for a_, b_ in v[4].items()},
                                          "shape_avg_us": {a_: round(shape_us[(k, a_)], 1) for a_ in v[4]}}
                                     for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[1:12]},
              "event_pair_overhead_us": overhead_ms * 1e3, "host_enqueue_ms_per_step": host_ms, "device_delay_ms": delay_ms,
              "timing": "HIP events on the launch stream around each launch of the real step (queued behind a device-side "
                        "delay so launches run back to back as in the graph replay), minus the median empty event pair; weight "
                        "fragments are prepacked once per step (xh_conv3d_prepack), so a bracket holds the conv launch alone"})
    k7 = {k: v for k, v in agg.items() if "conv7_mfma" in k or "conv7_as" in k}
    if k7:
        cnt7 = sum(v[0] for v in k7.values())
        ms7 = sum(v[1] for v in k7.values())
        fl7 = sum(v[3] for v in k7.values())
        r["gate_conv7"] = {"bound": "mfma", "kernel": "conv7_as_kernel (AttenModule2's composed 7^3 gate conv, forward + data gradient)",
                           "achieved": fl7 / ms7 / 1e9, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": fl7 / ms7 / 1e9 / BF16_MFMA_PEAK_TFLOPS, "launches_per_step": cnt7 / nsteps,
                           "ms_per_step": ms7 / nsteps,
                           "note": "flops counted on the composed 4 -> 2 (2 -> 4) channel conv: 2 * out * 343 * Cin"}
    return r


def elementwise_from_profile():
    """The norm / element-wise family's share of a step, from the rocprofv3 kernel trace of this command committed under
    profiles/ (tools/dump_step.py classifies every launch of one replayed step); None when no summary is present."""
    path = os.path.join(ROOT, "profiles", "step_families.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return {"ms": t.get("norm_elementwise_ms"), "launches": t.get("norm_elementwise_launches"), "launches_per_step": t.get("launches"),
            "source": t.get("source")}


if __name__ == "__main__":
    main()

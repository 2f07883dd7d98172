#!/usr/bin/env python3
"""bench.py -- voxels/s of the XLSTM-HVED forward+backward hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--size 128] [--no-graph] [--no-cpu]

One process per GPU (for N>1 launch with torch.distributed.run; RANK/LOCAL_RANK/WORLD_SIZE are read from the
environment).  A step = one forward + backward of XLSTM_HVED (train mode, all 4 modalities, recon=True, loss of
SURVEY.md 8(d)) on one synthetic 1x4x128^3 patch per rank, plus for N>1 the flat RCCL all-reduce of the generator's
gradients (data parallel; weak scaling).  The step is captured once into a hipGraph and replayed; W warm-up replays,
then exactly K timed replays between barrier + synchronize pairs; the slowest rank's time is used.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel family of the step, timed per launch with HIP events on the launch stream during an
                instrumented (non-graph) pass over the same K steps; algorithmic bytes/flops from the launch's shapes.
  cpu_baseline  the CPU oracle (port of the reference path; the reference sources do not travel to the GPU box)
                timed on the host cores on the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
FP32_VALU_PEAK_TFLOPS = 157.3  # vector fp32 peak (the conv kernels of this round are fp32-FMA bound, not MFMA)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def bench_loss(seg, mu, lv, rec):
    """reaches every parameter the reference's training loss reaches."""
    loss = seg.float().mean() + rec.float().mean()
    for a, b in zip(mu, lv):
        loss = loss + a.float().mean() + b.float().mean()
    return loss


def cpu_baseline(size, batch, steps_budget_s=30.0):
    """Times the CPU oracle (functional restatement of the reference path, stock torch ops, fp32) on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xlstm_hved_oracle as O
    import xlstm_hved_amd as X
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    model = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    model.apply(X.init_weights)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    # bounded sample: one 1x4x64^3 warm-up, then as many full-size steps as fit the budget (at least one)
    def run(s):
        x = torch.rand(batch, 4, s, s, s, generator=g)
        eps = [torch.randn(batch, 2 ** l, s >> (l + 1), s >> (l + 1), s >> (l + 1), generator=g) for l in range(4)]
        sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
        t0 = time.perf_counter()
        prob, _, mu, lv, rec = O.xlstm_hved_forward(sd, x, 14, eps_list=eps, training=True)
        O.bench_loss(prob, mu, lv, rec).backward()
        return time.perf_counter() - t0
    run(min(size, 64))
    times = []
    t_all = time.perf_counter()
    while not times or (time.perf_counter() - t_all + times[-1] < steps_budget_s and len(times) < 3):
        times.append(run(size))
    best = min(times)
    return {"value": batch * size ** 3 / best, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} fwd+bwd step(s) of {batch}x4x{size}^3 fp32 through oracle/xlstm_hved_oracle.py "
                      f"(torch {torch.__version__} CPU ops, {cores} threads), best {best:.2f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    import xlstm_hved_amd as X
    from xlstm_hved_amd import ops
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)      # RCCL over xGMI
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    S, B = args.size, args.batch

    torch.manual_seed(1)                                      # same weights on every rank
    model = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    model.apply(X.init_weights)
    model = model.to(dev).train()
    params = [p for p in model.parameters()]
    g = torch.Generator(device="cpu").manual_seed(1 + rank)   # per-rank synthetic patch
    x = torch.rand(B, 4, S, S, S, generator=g).to(dev, dtype)
    grads = X.parallel.FlatGrads(params)                       # p.grad = views of one flat fp32 bucket

    def step():
        grads.zero()
        seg, (mu, lv), rec = model(x, [14], recon=True)
        bench_loss(seg, mu, lv, rec[0]).backward()
        if world > 1:
            grads.all_reduce(world)                             # one in-place RCCL all-reduce of the bucket

    def sync_all():
        if world > 1:
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
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
    run = graph.replay if graph is not None else step
    for _ in range(args.warmup):
        run()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    value = world * B * S ** 3 / (dt / args.steps)

    # ---- roofline of the dominant kernel family: per-launch HIP events on the launch stream ------------
    roof = None
    if rank == 0 and not args.no_roofline:
        roof = roofline_pass(step, ops, min(args.steps, 5), dtype)

    out = {
        "metric": "voxels/sec fwd+bwd, 4-modality 128^3 patch", "value": value, "unit": "voxels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"XLSTM_HVED fwd+bwd, {B}x4x{S}^3 patch per GPU, f_maps=4 'ilc' (train.py:142-143), "
                               f"train mode, subset [14], recon=True, {args.dtype} activation storage / fp32 arithmetic, "
                               f"random-init weights, {'hipGraph replay' if graph is not None else 'eager'}",
                   "parallelism": f"dp{world}", "per_gpu_batch": B, "global_batch": B * world},
    }
    if roof is not None:
        out["roofline"] = roof
    if rank == 0 and not args.no_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(S, B)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def roofline_pass(step, ops, nsteps, dtype):
    """Instrumented eager pass: every xh_conv3d_fwd launch (forward convs and stride-1 data gradients: the dominant
    kernel family) is bracketed by HIP events on torch's current stream = the launch stream.  Algorithmic work per
    launch comes from the launch's own shapes: bytes = input + output elements x storage size (+ fp32 weights),
    flops = 2 * out_elements * k^3 * Cin/groups."""
    records = []
    orig = ops.conv3d
    esz = 2 if dtype == torch.bfloat16 else 4

    def timed(xa, xb, weights, biases, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(xa, xb, weights, biases, **kw)
        e1.record()
        cin = xa.shape[1] + (xb.shape[1] if xb is not None else 0)
        k, groups = kw["k"], kw.get("groups", 1)
        in_el = xa.numel() + (xb.numel() if xb is not None else 0)
        e_el = in_el if kw.get("epi", 0) == 1 and False else 0
        if kw.get("epi", 0) == 1:
            e_el = y.numel()
        nbytes = (in_el + y.numel() + e_el) * esz + sum(w.numel() for w in weights) * 4
        flops = 2.0 * y.numel() * k ** 3 * cin / groups
        key = f"k{k} s{kw.get('stride', 1)} g{groups} {cin}->{y.shape[1]} @{tuple(y.shape[2:])}" + (" dgrad" if kw.get("transposed") else "")
        records.append((key, e0, e1, nbytes, flops))
        return y
    ops.conv3d = timed
    try:
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
    finally:
        ops.conv3d = orig
    agg = {}
    for key, e0, e1, nbytes, flops in records:
        a = agg.setdefault(key, [0, 0.0, nbytes, flops])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    total_ms = sum(a[1] for a in agg.values())
    # dominant = the shape class with the largest share of conv time
    key, (cnt, ms_sum, nbytes, flops) = max(agg.items(), key=lambda kv: kv[1][1])
    avg_ms = ms_sum / cnt
    gbs = nbytes / (avg_ms * 1e-3) / 1e9
    tfl = flops / (avg_ms * 1e-3) / 1e12
    ai = flops / nbytes
    # the class is fp32-FMA (vector ALU) bound when its arithmetic intensity exceeds peak_flops/peak_bw
    bound = "hbm" if ai < FP32_VALU_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9) else "valu"
    if bound == "hbm":
        r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    else:
        r = {"bound": "mfma", "achieved": tfl, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / FP32_VALU_PEAK_TFLOPS,
             "note": "fp32 FMA on the vector ALU / f32-MFMA rate (157.3 TFLOP/s); bf16 MFMA is not used by this kernel yet"}
    r.update({"traffic": None, "kernel": "conv_fwd_kernel (xh_conv3d_fwd) " + key, "launches_per_step": cnt // nsteps,
              "avg_launch_us": avg_ms * 1e3, "algorithmic_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops,
              "hbm_equiv_GBps": gbs, "share_of_conv_time": ms_sum / total_ms,
              "timing": "HIP events on the launch stream around each launch, eager pass over the same steps"})
    return r


if __name__ == "__main__":
    main()

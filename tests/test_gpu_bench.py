"""-m gpu: bench.py end to end (the driver's contract): exactly one JSON line on stdout with the metric, roofline and
cpu_baseline objects.  Small sizes so it finishes in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--size", "64"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["steps"] == 3 and j["n_gpus"] == 1 and j["value"] > 0
    assert j["roofline"]["bound"] in ("hbm", "mfma") and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
    assert set(j["modes"]) == {"bf16", "fp16", "fp32", "fp32_mfma"} and all(m["ms_per_step"] > 0 for m in j["modes"].values())
    # measured in the run, not quoted: each mode's 128^3 mask against the real reference's mask; the two fp32-storage arithmetics
    # meet north_star's Dice tolerance
    pm = j["parity_measured"]["modes"]
    assert set(pm) == {"bf16", "fp16", "fp32", "fp32_mfma"}
    assert pm["fp32"]["dice_dev"] <= 1e-4 and pm["fp32_mfma"]["dice_dev"] <= 1e-4, pm
    assert pm["fp16"]["dice_dev"] <= pm["bf16"]["dice_dev"] + 1e-3
    assert j["config3"]["ms_per_step"] > 0 and j["config3"]["gradients_finite"]
    assert j["roofline"]["elementwise_in_run"]["ms_per_step"] > 0 and j["roofline"]["elementwise_in_run"]["calls_per_step"] > 50
    # the practical streaming ceiling measured in the run, next to the nominal peak the fractions are quoted against
    hs = j["roofline"]["hbm_stream_measured"]
    assert 1000 < hs["copy"] < 8000 and 1000 < hs["add_2r1w"] < 8000, hs


def test_bench_force_dist_rccl_allreduce_with_graph_capture():
    """The N>1 code path on one GPU: RCCL communicator, thread-local hipGraph capture of the step, the flat-bucket
    all-reduce issued after every replay (bench.py --force-dist)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--size", "64", "--force-dist",
                        "--no-cpu", "--no-roofline", "--no-modes"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29563"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["config"]["parallelism"] == "dp1"


def test_bench_self_spawn_path_relays_one_json_line():
    """`python bench.py --gpus N` without a launcher spawns its workers itself; exercised here with one worker (--force-spawn)
    going through the RCCL path (--force-dist): rank 0's single JSON line must come out of the parent's stdout."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-spawn", "--force-dist", "--steps", "2",
                        "--warmup", "1", "--size", "64", "--no-cpu", "--no-roofline", "--no-modes"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["n_gpus"] == 1


def test_bench_gpus_n_without_launcher_fails_cleanly_when_devices_are_missing():
    """`python bench.py --gpus 2` (no torchrun) spawns its own workers; on a 1-GPU box it must exit non-zero with a message
    and without touching the device (the driver runs the same form on an 8-GPU node)."""
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0", "--size", "64",
                        "--no-cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "device(s) visible" in r.stderr and not r.stdout.strip()


def test_tiler_through_an_initialised_rccl_group():
    """eval_overlap_volume(..., world=1, group=...) with a real `nccl` (RCCL) process group: the window shard + the
    all-reduce of (sum, count) run through the collective and give the ungrouped result."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["XH_ROOT"]); sys.path.insert(0, os.path.join(os.environ["XH_ROOT"], "tests"))
import xlstm_hved_amd as X
from gpu_common import load
from xlstm_hved_amd.inference import eval_overlap_volume
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.load_state_dict(load("weights_seed1")); m = m.to(dev).eval()
torch.manual_seed(2); x = torch.rand(1, 4, 40, 32, 41, device=dev)
a = eval_overlap_volume(m, x, 5, (32, 32, 32), (8, 8, 8), batch_size=2)
b = eval_overlap_volume(m, x, 5, (32, 32, 32), (8, 8, 8), batch_size=2, rank=0, world=1, group=dist.group.WORLD)
g = dist.new_group([0])
c = eval_overlap_volume(m, x, 5, (32, 32, 32), (8, 8, 8), batch_size=2, rank=0, world=2, group=g)   # shard 0 of 2, reduced over a 1-rank group
torch.cuda.synchronize()
assert torch.equal(a, b), (a - b).abs().max()
assert torch.isfinite(c[:, :, :32]).all()
dist.destroy_process_group()
print("TILER_RCCL_OK")
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29564", XH_ROOT=ROOT))
    assert r.returncode == 0 and "TILER_RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])

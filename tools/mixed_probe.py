"""Mixed storage policy (ops.set_mixed_storage) next to the uniform modes: mask flips / Dice deviation of the 128^3 parity case
and ms per captured fwd+bwd step of bench.py's workload.
usage: python tools/mixed_probe.py [--steps 50] [--modes bf16,fp16,fp32_mfma,mixed_fp16,mixed_bf16]"""
import argparse, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import xlstm_hved_amd as X
from xlstm_hved_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--modes", default="bf16,fp16,fp32_mfma,mixed_fp16,mixed_bf16")
ap.add_argument("--no-time", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
MODES = {"bf16": (torch.bfloat16, False, None), "fp16": (torch.float16, False, None), "fp32": (torch.float32, False, None),
         "fp32_mfma": (torch.float32, True, None), "mixed_fp16": (torch.float32, True, torch.float16),
         "mixed_bf16": (torch.float32, True, torch.bfloat16)}


def setmode(name):
    dt, split, mixed = MODES[name]
    ops.set_fp32_mfma(split)
    ops.set_mixed_storage(mixed)
    return dt


res = {}
for name in a.modes.split(","):
    dt = setmode(name)
    mp = B.measured_parity(dev, {name: dt}, keep_mode=True)["modes"][name]
    res[name] = {"flips": mp["mask_flips"], "dice_dev": mp["dice_dev"]}
    print(name, res[name], flush=True)

if not a.no_time:
    torch.manual_seed(1)
    model = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    model.apply(X.init_weights)
    model = model.to(dev).train()
    grads = X.parallel.FlatGrads(list(model.parameters()))
    ops.set_wgrad_defer(True)
    x = torch.rand(1, 4, 128, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
    for name in a.modes.split(","):
        dt = setmode(name)
        xin = x.to(dt)
        scale = 1.0 if dt == torch.bfloat16 else B.LOSS_SCALE_FP16
        seed = torch.full((), scale, dtype=torch.float32, device=dev)

        def compute():
            grads.zero()
            seg, (mu, lv), rec = model(xin, [14], recon=True)
            B.bench_loss(seg, mu, lv, rec[0]).backward(seed)
            ops.join_wgrad_stream()
            if scale != 1.0:
                grads.flat.mul_(1.0 / scale)
        ms = B.time_graph(compute, a.steps, 5)
        res[name]["ms_per_step"] = ms
        print(name, f"{ms:.3f} ms/step", flush=True)
setmode("bf16")
print(json.dumps(res))

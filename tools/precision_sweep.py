"""Where do the mask flips of 16-bit storage come from?  

The 128^3 parity case of bench.py (trained-like weights, blob patch 8, subset [14], eval, posterior mean; target = the real
reference's fp32 mask, tests/golden/mask_trained_like_128.npz) is run in fp32 STORAGE (`fp32` vector kernels or `fp32_mfma`),
and the activation tensors the ops wrappers hand back are rounded IN PLACE to a 16-bit format right after the launch that wrote
them -- one tensor at a time, then groups, then "everything except ...".  A tensor whose rounding alone moves the mask is one a
mixed storage policy has to keep in fp32; tensors that move nothing can stay 16-bit.

usage: python tools/precision_sweep.py [--arith fp32|fp32_mfma] [--out gpurun_out/precision_sweep.json]
"""
import argparse, json, os, sys, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import xlstm_hved_amd as X
import synth_blobs as SB

ops = X.ops
DEV = torch.device("cuda:0")
FMT = {"bf16": torch.bfloat16, "fp16": torch.float16}

STATE = {"seq": 0, "labels": [], "round": None, "fmt": torch.bfloat16, "seen": set(), "stats": []}


def _tensors(r, acc):
    if isinstance(r, torch.Tensor):
        acc.append(r)
    elif isinstance(r, (list, tuple)):
        for t in r:
            _tensors(t, acc)
    return acc


def wrap(name, fn):
    def w(*args, **kw):
        r = fn(*args, **kw)
        outs = _tensors(r, [])
        for k in ("out", "into", "y"):
            if k in kw:
                _tensors(kw[k], outs)
        done = set()
        for t in outs:
            if t.dim() != 5 or t.dtype != torch.float32 or not t.is_cuda or t.data_ptr() in done:
                continue
            done.add(t.data_ptr())
            i = STATE["seq"]
            STATE["seq"] += 1
            if STATE["round"] is None:
                STATE["labels"].append(f"{name}:{'x'.join(map(str, t.shape[1:]))}")
                # how far the stored values sit from zero in units of their spread: what a floating-point format pays for
                v = t.float()
                mu, sd = v.mean((2, 3, 4)), v.std((2, 3, 4)) + 1e-30
                STATE["stats"].append((float((mu.abs() / sd).mean()), float((mu.abs() / sd).max()),
                                       float((v.abs().amax((2, 3, 4)) / sd).mean())))
            elif STATE["round"](i):
                if t.is_contiguous():
                    t.copy_(t.to(STATE["fmt"]))
                else:
                    t.copy_(t.to(STATE["fmt"]).float())
        return r
    return w


SKIP = {"new_like", "zeros_f32", "zeros_f64", "zeros_red", "fan_block", "last_conv_kernel", "red_arena_reset", "join_wgrad_stream",
        "prepack_all", "begin_capture_scope"}
for k, v in list(vars(ops).items()):
    if isinstance(v, types.FunctionType) and not k.startswith("_") and not k.startswith("set_") and k not in SKIP and v.__module__ == ops.__name__:
        setattr(ops, k, wrap(k, v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arith", default="fp32_mfma")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "precision_sweep.json"))
    a = ap.parse_args()
    gold = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gold, "mask_trained_like_128.npz"))
    shape = tuple(int(v) for v in z["shape"])
    ref = torch.from_numpy(np.unpackbits(z["bits"])[: int(np.prod(shape))].reshape(shape).astype(np.bool_)).to(DEV)
    w = np.load(os.path.join(gold, "weights_trained_like.npz"))
    m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    m.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}, strict=True)
    m = m.to(DEV).eval()
    x, _ = SB.blob_case(8, 1, 128)
    x = x.to(DEV)
    ops.set_fp32_mfma(a.arith == "fp32_mfma")

    def run(sel, fmt="bf16", dt=torch.float32):
        nonlocal x
        STATE["seq"], STATE["round"], STATE["fmt"] = 0, sel, FMT[fmt]
        with torch.no_grad():
            seg = m(x.to(dt), [14], recon=True, valid=True)[0]
        got = seg.float() > 0.5
        inter = (got & ref).sum((0, 2, 3, 4)).double()
        den = got.sum((0, 2, 3, 4)).double() + ref.sum((0, 2, 3, 4)).double()
        dice = (2 * inter + 1e-6) / (den + 1e-6)
        return int((got != ref).sum()), float((dice - 1).abs().max())

    base = run(None)
    labels = list(STATE["labels"])
    n = len(labels)
    print(f"{a.arith}: {n} activation tensors; unrounded: flips {base[0]} dice_dev {base[1]:.2e}", flush=True)
    res = {"arith": a.arith, "baseline": base, "tensors": []}
    for fmt in ("bf16", "fp16"):
        res[f"all_{fmt}"] = run(lambda i: True, fmt)
        print(f"every tensor rounded to {fmt}: flips {res[f'all_{fmt}'][0]} dice_dev {res[f'all_{fmt}'][1]:.2e}", flush=True)
    # the true 16-bit storage modes (operand / weight rounding inside the convs included)
    ops.set_fp32_mfma(False)
    for fmt in ("bf16", "fp16"):
        STATE["round"] = lambda i: False
        res[f"storage_{fmt}"] = run(lambda i: False, fmt, FMT[fmt])
        print(f"{fmt} STORAGE mode: flips {res[f'storage_{fmt}'][0]} dice_dev {res[f'storage_{fmt}'][1]:.2e}", flush=True)
    ops.set_fp32_mfma(a.arith == "fp32_mfma")
    for fmt in ("bf16", "fp16"):                          # the INPUT patch rounded, every tensor of the network fp32
        xs = x
        x = xs.to(FMT[fmt]).float()
        res[f"input_{fmt}"] = run(lambda i: False, fmt)
        x = xs
        print(f"input patch rounded to {fmt}, fp32 storage: flips {res[f'input_{fmt}'][0]} dice_dev {res[f'input_{fmt}'][1]:.2e}", flush=True)
    one = {}
    for fmt in ("bf16", "fp16"):
        one[fmt] = [run(lambda i, j=j: i == j, fmt) for j in range(n)]
    for j in range(n):
        res["tensors"].append({"i": j, "label": labels[j], "bf16": one["bf16"][j], "fp16": one["fp16"][j]})
        st = STATE["stats"][j]
        res["tensors"][-1]["mean_over_std"] = st
        print(f"{j:3d} {labels[j]:48s} bf16 {one['bf16'][j][0]:5d}  fp16 {one['fp16'][j][0]:5d}   |mean|/std avg {st[0]:6.2f} max {st[1]:6.2f}  max|x|/std {st[2]:6.1f}", flush=True)
    # cumulative: keep the k most sensitive tensors in fp32, round every other one
    for fmt in ("bf16", "fp16"):
        order = sorted(range(n), key=lambda j: -one[fmt][j][0])
        cum = []
        for k in (0, 1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
            if k > n:
                break
            keep = set(order[:k])
            f = run(lambda i: i not in keep, fmt)
            cum.append({"keep_fp32": k, "flips": f[0], "dice_dev": f[1]})
            print(f"{fmt}: top-{k} sensitive tensors kept fp32, the rest rounded: flips {f[0]} dice_dev {f[1]:.2e}", flush=True)
        res[f"cumulative_{fmt}"] = cum
        res[f"order_{fmt}"] = order
        # prefix / suffix policies in launch order: everything from tensor j on stays fp32
        suf = []
        for j in list(range(0, n, max(1, n // 24))) + [n]:
            f = run(lambda i, j=j: i < j, fmt)
            suf.append({"fp32_from": j, "flips": f[0], "dice_dev": f[1]})
            print(f"{fmt}: tensors [0,{j}) rounded, [{j},{n}) fp32: flips {f[0]} dice_dev {f[1]:.2e}", flush=True)
        res[f"suffix_{fmt}"] = suf
    # policies a mixed storage mode can actually take: the network has ONE narrow waist -- everything the decoders see of the
    # encoders passes through the DRB outputs (-> PoE) and the 16^3 skip feature (-> ViL), RA_HVED.py:569-626 -- so "encoder half
    # fp32, decoder half 16-bit" needs casts of tiny tensors only
    first_poe = next(j for j, l in enumerate(labels) if l.startswith("poe_fwd"))
    skr = {j for j in range(first_poe) if labels[j].startswith("skr_tail")}
    for j in sorted(skr):                                # the ResBlock / attention tensors in front of each skr_tail output
        k = j - 1
        while k >= 0 and labels[k].startswith("conv3d:") and labels[k].split(":")[1].split("x")[0] == labels[j - 1].split(":")[1].split("x")[0] and j - k <= 4:
            skr.add(k)
            k -= 1
    last = n - 1
    pol = {
        "decoder_half": lambda i: i >= first_poe,
        "decoder_half_seg_out_fp32": lambda i: first_poe <= i < last,
        "decoder_half+skr_internals": lambda i: i >= first_poe or i in skr,
        "decoder_half+skr_internals_seg_out_fp32": lambda i: (i >= first_poe or i in skr) and i != last,
        "encoder_half": lambda i: i < first_poe,
        "128^3_tensors_only": lambda i: labels[i].endswith("128x128x128"),
        "all_but_128^3_and_64^3": lambda i: not (labels[i].endswith("128x128x128") or labels[i].endswith("64x64x64")),
    }
    res["policies"] = {"first_poe": first_poe, "skr_internals": sorted(skr)}
    for fmt in ("bf16", "fp16"):
        for name, sel in pol.items():
            f = run(sel, fmt)
            res["policies"][f"{name}.{fmt}"] = f
            print(f"policy {name} rounded to {fmt}: flips {f[0]} dice_dev {f[1]:.2e}", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

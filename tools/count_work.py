"""Algorithmic work of one XLSTM_HVED forward as this build executes it (SURVEY.md 8(d) counting rule: per conv launch
flops = 2 * out.numel() * k^3 * Cin/groups, elements = in.numel() + out.numel()), next to the reference's module-level count.

    python tools/count_work.py [--size 128]        (needs the GPU: the product path has no CPU fallback)
"""
import argparse, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
a = ap.parse_args()
S = a.size
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
orig = ops.conv3d


def counted(xa, xb, weights, biases, **kw):
    res = orig(xa, xb, weights, biases, **kw)
    y = res[0] if isinstance(res, tuple) else res
    cin = xa.shape[1] + (xb.shape[1] if xb is not None else 0)
    k, g = kw["k"], kw.get("groups", 1)
    t = tot[f"k{k}" + (" depthwise" if g == cin and g > 1 else "") + (" stride 2" if kw.get("stride", 1) == 2 else "")]
    t[0] += 1
    t[1] += 2.0 * y.numel() * k ** 3 * cin / g
    t[2] += xa.numel() + (xb.numel() if xb is not None else 0) + y.numel()
    return res


ops.conv3d = counted
torch.manual_seed(1)
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
m.apply(X.init_weights)
m = m.cuda().train()
x = torch.rand(1, 4, S, S, S, device="cuda").bfloat16()
with torch.no_grad():
    m(x, [14], recon=True)
torch.cuda.synchronize()
vox = S ** 3
gf = sum(v[1] for v in tot.values()) / 1e9
el = sum(v[2] for v in tot.values()) / 1e6
print(f"forward at {S}^3: {sum(v[0] for v in tot.values())} conv launches, {gf:.2f} GFLOP = {gf * 1e9 / vox:.0f} flop/voxel, "
      f"{el:.1f} M conv in+out elements = {2 * el * 1e6 / vox:.1f} B/voxel at bf16")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:22s} {v[0]:3d} launches  {v[1] / 1e9:8.2f} GFLOP  {v[2] / 1e6:8.1f} M elements")
print("reference (SURVEY 8d, 169 Conv3d modules at 128^3): 93.19 GFLOP = 44 436 flop/voxel, 1 037.6 M elements = 989.5 B/voxel;\n"
      "the build runs the skip-return attention once instead of 4x, applies the 7^3 grouped + 1x1 pairs as one composed 7^3 conv\n"
      "and the 4 modality streams as grouped launches, so both its launch count and its work are lower for the same function.")

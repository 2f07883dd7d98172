"""Bisects a deviation of the eval forward against the fp64 reference fixture by switching specialised kernels off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from gpu_common import load
import xlstm_hved_amd as X
g = load("net32_subsets_eval")
m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS); m.load_state_dict(load("weights_seed1"), strict=True); m = m.cuda().eval()
x = g["x2"][:1].cuda()
lib = X._lib.load()
for name, mask, arena in (("all on", 0, True), ("no dw slide", 1, True), ("no up2x", 2, True), ("neither", 3, True), ("no arena", 0, False)):
    lib.xh_set_option(2, mask)
    X.ops._ARENA_DOUBLES = (1 << 19) if arena else 0
    with torch.no_grad():
        for k in (2, 14):
            seg = m(x, [k], recon=True, valid=True)[0]
            e = (seg.flatten().cpu()[g["idx_seg"]].double() - g[f"seg_{k}"]).abs()
            print(f"{name:12s} subset {k}: max {e.max().item():.2e} mean {e.mean().item():.2e}")

"""Quad-channel weight-gradient kernel at the 128^3 shapes, with / without its global atomics tail (ablation bit 2048)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
S = int(os.environ.get("XH_S", "128"))
ROT = int(os.environ.get("XH_ROT", "1"))      # operand sets walked round-robin (> 1: from HBM instead of the last-level cache)
for (cin, cout, g) in [(4, 4, 1), (16, 16, 4), (12, 4, 1), (8, 8, 1)]:
    sets = [(torch.randn(1, cin, S, S, S, device="cuda").bfloat16(), torch.randn(1, cout, S, S, S, device="cuda").bfloat16()) for _ in range(ROT)]
    tick = [0]
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    dws = [torch.zeros(cout // g, cin // g, 3, 3, 3, device="cuda") for _ in range(g)]
    dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
    def call():
        tick[0] += 1
        x, dy = sets[tick[0] % ROT]
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=(sc, sh, 0.01))
    line = f"wgrad {cin}->{cout} g{g} @{S}^3:"
    for m, nm in [(0, "full (2 x 64 tiles)"), (524288, "4 x 32 tiles"), (2048, "no global atomics"), (524288 + 2048, "4 x 32, no atomics")]:
        L.load().xh_set_option(1, m)
        line += f" {nm} {bench(call):.1f} us |"
    L.load().xh_set_option(1, 0)
    print(line, ops.last_conv_kernel(), flush=True)

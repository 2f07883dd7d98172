"""Stride-2 k3 convs of the DRBs (grouped, 2 or 4 output channels per group): forward, data gradient, weight gradient, with the
weight gradient's workgroup cap (xh_set_option(12, n))."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
for (cin, cout, g, S) in [(16, 8, 4, 128), (32, 16, 4, 64), (64, 32, 4, 32), (4, 2, 1, 128)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    dy = torch.randn(1, cout, S // 2, S // 2, S // 2, device="cuda").bfloat16()
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
    dws = [torch.zeros_like(w) for w in ws]
    dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
    fwd = lambda: ops.conv3d(x, None, ws, None, k=3, cout=cout, stride=2, groups=g, pre=(sc, sh, 0.01))
    wg = lambda: ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, stride=2, groups=g, pre=(sc, sh, 0.01))
    red = torch.zeros(1, cin, 2, dtype=torch.float64, device="cuda")
    dg = lambda: ops.conv3d_dgrad_s2(dy, ws, cin=cin, in_spatial=(S, S, S), groups=g, e=(x, None, sc, sh, 0.01), red=red)
    line = f"s2 {cin}->{cout} g{g} @{S}^3: fwd {bench(fwd):.1f} us ({ops.last_conv_kernel()[:24]}) dgrad {bench(dg):.1f} us"
    for cap in (256, 512, 1024, 2048, 4096):
        L.load().xh_set_option(12, cap)
        line += f" | wgrad cap{cap} {bench(wg):.1f} us"
    L.load().xh_set_option(12, 512)
    print(line, ops.last_conv_kernel()[:30], flush=True)

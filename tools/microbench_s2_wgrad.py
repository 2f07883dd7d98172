"""Stride-2 k3 weight gradients at the step's shapes (the DRB convs, groups of 4): each alone, then the multi launch of all four."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops
data = []
for (cin, cout, g, S) in [(16, 8, 4, 128), (32, 16, 4, 64), (64, 32, 4, 32), (128, 64, 4, 16)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    dy = torch.randn(1, cout, S // 2, S // 2, S // 2, device="cuda").bfloat16()
    pre = (torch.rand(1, cin, device="cuda") + 0.5, torch.randn(1, cin, device="cuda"), 0.01)
    dws = [torch.zeros(cout // g, cin // g, 3, 3, 3, device="cuda") for _ in range(g)]
    dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
    data.append((x, dy, pre, g, dws, dbs))
    call = lambda: ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, stride=2, groups=g, pre=pre)
    t = bench(call)
    print(f"s2 wgrad {cin}->{cout} g{g} @{S}^3: {t:.1f} us [{ops.last_conv_kernel()}]", flush=True)
def batch():
    ops.set_wgrad_defer(True)
    for x, dy, pre, g, dws, dbs in data:
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, stride=2, groups=g, pre=pre, side=True)
    ops.join_wgrad_stream()
    ops.set_wgrad_defer(False)
for _ in range(2):
    print(f"batch of 4: {bench(batch, n=8):.1f} us", flush=True)

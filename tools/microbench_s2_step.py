"""Stride-2 k3 convs at the shapes of the 128^3 step (five streams): forward and data gradient.  python tools/microbench_s2_step.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops
for (cin, cout, g, S) in [(20, 10, 5, 128), (40, 20, 5, 64), (80, 40, 5, 32), (160, 80, 5, 16)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    dy = torch.randn(1, cout, S // 2, S // 2, S // 2, device="cuda").bfloat16()
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
    fwd = lambda: ops.conv3d(x, None, ws, None, k=3, cout=cout, stride=2, groups=g, pre=(sc, sh, 0.01))
    dg = lambda: ops.conv3d_dgrad_s2(dy, ws, cin=cin, in_spatial=(S, S, S), groups=g)
    t1 = bench(fwd); k1 = ops.last_conv_kernel()
    t2 = bench(dg); k2 = ops.last_conv_kernel()
    print(f"s2 {cin}->{cout} g{g} @{S}^3: fwd {t1:.1f} us [{k1}]  dgrad {t2:.1f} us [{k2}]", flush=True)

"""Vector (non-MFMA) weight-gradient kernel at the shapes of the step that reach it, against its workgroup target (key 13)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
for (cin, cout, g, S, st) in [(1, 2, 1, 128, 1), (1, 2, 1, 64, 1), (1, 2, 1, 32, 1), (64, 32, 4, 32, 2), (128, 64, 4, 16, 2), (32, 16, 4, 64, 2)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    so = S // st
    dy = torch.randn(1, cout, so, so, so, device="cuda").bfloat16()
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
    dws = [torch.zeros_like(w) for w in ws]
    dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
    wg = lambda: ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, stride=st, groups=g)
    line = f"wgrad k3 s{st} {cin}->{cout} g{g} @{S}^3:"
    for cap in (256, 512, 1024, 2048, 4096):
        L.load().xh_set_option(13, cap)
        line += f" | target {cap}: {bench(wg):.1f} us"
    L.load().xh_set_option(13, 512)
    print(line, ops.last_conv_kernel()[:40], flush=True)

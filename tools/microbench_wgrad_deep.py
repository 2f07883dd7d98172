import sys, os
sys.path.insert(0, "/root/repo")
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops
dt = torch.bfloat16
R = lambda *s: torch.randn(*s, device="cuda").to(dt)
for cin, cout, g, S in [(32, 32, 32, 16), (80, 160, 5, 16), (80, 80, 5, 16), (128, 128, 4, 16), (16, 16, 1, 32)]:
    x = R(1, cin, S, S, S); dy = R(1, cout, S, S, S)
    dw = [torch.zeros(cout, cin // g, 3, 3, 3, device="cuda")]
    db = [torch.zeros(cout, device="cuda")]
    t = bench(lambda: ops.conv3d_wgrad(x, None, dy, dw, db, k=3, groups=g))
    print(f"wgrad k3 g{g} {cin}->{cout} @{S}^3: {t:6.1f} us [{ops.last_conv_kernel()}]", flush=True)

"""Which conv shapes should the quad-channel kernel take?  Forward (norm prologue + moments) and data gradient of the 64^3 / 32^3
shapes of the step with the per-group channel limit (xh_set_option(11, n)) at 12 (default) and 24."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
for (cin, cout, g, S) in [(24, 8, 1, 64), (8, 24, 1, 64), (16, 16, 1, 32), (48, 16, 1, 32), (16, 8, 1, 64), (16, 16, 1, 64), (32, 32, 4, 32), (32, 64, 4, 32), (64, 32, 4, 32)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    dy = torch.randn(1, cout, S, S, S, device="cuda").bfloat16()
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
    red2 = torch.zeros(1, cin, 2, dtype=torch.float64, device="cuda")
    fwd = lambda: ops.conv3d(x, None, ws, None, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red)
    dgr = lambda: ops.conv3d(dy, None, ws, None, k=3, cout=cin, groups=g, transposed=True, epi=1, e=(x, None, sc, sh, 0.01), red=red2)
    line = f"{cin}->{cout} g{g} @{S}^3:"
    for lim in (12, 24, 48):
        L.load().xh_set_option(11, lim)
        ops.set_prepack(True)
        tf = bench(fwd); kf = ops.last_conv_kernel().split("<")[0]
        td = bench(dgr); kd = ops.last_conv_kernel().split("<")[0]
        line += f" | limit {lim}: fwd {tf:5.1f} us ({kf[6:]}) dgrad {td:5.1f} us ({kd[6:]})"
    L.load().xh_set_option(11, 48)
    print(line, flush=True)

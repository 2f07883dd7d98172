"""One k3 weight-gradient problem, N plain launches (for rocprofv3 --pmc passes).  python3 tools/prof_wq4.py cin cout groups [n]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
ops = X.ops
cin, cout, g = (int(v) for v in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
S = int(os.environ.get("XH_S", "128"))
x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
dy = torch.randn(1, cout, S, S, S, device="cuda").bfloat16()
sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
dws = [torch.zeros(cout // g, cin // g, 3, 3, 3, device="cuda") for _ in range(g)]
dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
for _ in range(n):
    ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=(sc, sh, 0.01))
torch.cuda.synchronize()
print(ops.last_conv_kernel())

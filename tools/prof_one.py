"""One conv-family call repeated a few times, for rocprofv3 --pmc passes:  python tools/prof_one.py wgrad|fwd|dgrad cin cout groups [S]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
ops = X.ops
kind, cin, cout, g = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
S = int(sys.argv[5]) if len(sys.argv) > 5 else 128
x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
dy = torch.randn(1, cout, S, S, S, device="cuda").bfloat16()
sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
dws = [torch.zeros(cout // g, cin // g, 3, 3, 3, device="cuda") for _ in range(g)]
dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
red2 = torch.zeros(1, cin, 2, dtype=torch.float64, device="cuda")
for _ in range(5):
    if kind == "wgrad":
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=(sc, sh, 0.01))
    elif kind == "fwd":
        ops.conv3d(x, None, ws, None, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red)
    else:
        ops.conv3d(dy, None, ws, None, k=3, cout=cin, groups=g, transposed=True, epi=1, e=(x, None, sc, sh, 0.01), red=red2)
torch.cuda.synchronize()
print(ops.last_conv_kernel())

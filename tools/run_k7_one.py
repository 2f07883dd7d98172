"""One configuration of the 7^3 gate conv at 128^3 for counter passes: python3 tools/run_k7_one.py <option 24 value> [dgrad]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
ops = X.ops
X._lib.load().xh_set_option(24, int(sys.argv[1]))
S = 128
dg = len(sys.argv) > 2
x = torch.randn(1, 2 if dg else 4, S, S, S, device="cuda").bfloat16()
w = torch.randn(2, 4, 7, 7, 7, device="cuda") * 0.05
b = torch.randn(2, device="cuda")
for _ in range(6):
    if dg:
        ops.conv3d(x, None, [w], None, k=7, cout=4, transposed=True)
    else:
        ops.conv3d(x, None, [w], [b], k=7, cout=2, act=ops.ACT_SIGMOID)
torch.cuda.synchronize()

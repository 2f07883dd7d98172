"""The 7^3 gate conv's weight gradient at 128^3 for counter passes: python3 tools/run_k7w_one.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
ops = X.ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
x = torch.randn(1, 4, S, S, S, device="cuda").bfloat16()
dy = torch.randn(1, 2, S, S, S, device="cuda").bfloat16()
dw = torch.zeros(2, 4, 7, 7, 7, device="cuda"); db = torch.zeros(2, device="cuda")
for _ in range(6):
    ops.conv3d_wgrad(x, None, dy, [dw], [db], k=7)
torch.cuda.synchronize()

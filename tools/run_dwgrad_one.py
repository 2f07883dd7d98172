"""One weight gradient of a stride-2 k = 4 discriminator conv for counter passes:
python3 tools/run_dwgrad_one.py <cin> <cout> <source extent> [option 14 value]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import disc as D
cin, cout, sp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X._lib.load().xh_set_option(14, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
so = (sp + 2 - 4) // 2 + 1
x = torch.randn(2, sp, sp, sp, cin, device="cuda").bfloat16()
dy = torch.randn(2, so, so, so, cout, device="cuda").bfloat16()
for _ in range(6):
    D._wgrad(x, dy, 2, 2, (sp,) * 3, (so,) * 3, cin, cout, ks=4)
torch.cuda.synchronize()

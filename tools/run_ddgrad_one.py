"""One stride-2 k = 4 data gradient of the discriminator (with the mask / channel-sum epilogue of the 64 <- 128 layer) for counter
passes: python3 tools/run_ddgrad_one.py <cin> <cout> <source extent of the forward> [option 14 value]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import disc as D
cin, cout, sp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X._lib.load().xh_set_option(14, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
so = (sp + 2 - 4) // 2 + 1
NB = 2
dy = torch.randn(NB, so, so, so, cout, device="cuda").bfloat16()
w = torch.randn(cout, cin, 4, 4, 4, device="cuda") * 0.05
wpt = D._pack(w, 1, cout, cin, torch.bfloat16)
mask = torch.randn(NB, sp, sp, sp, cin, device="cuda").bfloat16()
red = torch.zeros(NB, cin, 2, dtype=torch.float64, device="cuda")
for _ in range(6):
    D._conv(dy, wpt, None, 1, 2, NB, (so,) * 3, (sp,) * 3, cout, cin, ks=4, red=red, mask=mask)
torch.cuda.synchronize()

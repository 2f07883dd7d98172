"""One forward conv of the discriminator for counter passes: python3 tools/run_dfwd_one.py <cin> <cout> <stride> <source extent> [batch] [option 14]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import disc as D
cin, cout, s, sp = (int(v) for v in sys.argv[1:5])
NB = int(sys.argv[5]) if len(sys.argv) > 5 else 1
X._lib.load().xh_set_option(14, int(sys.argv[6]) if len(sys.argv) > 6 else 0)
so = (sp + 2 - 4) // s + 1
x = torch.randn(NB, sp, sp, sp, cin, device="cuda").bfloat16()
w = torch.randn(cout, cin, 4, 4, 4, device="cuda") * 0.05
wp = D._pack(w, 0, cout, cin, torch.bfloat16)
red = torch.zeros(NB, cout, 2, dtype=torch.float64, device="cuda")
for _ in range(6):
    D._conv(x, wp, None, 0, s, NB, (sp,) * 3, (so,) * 3, cin, cout, red=red, ks=4)
torch.cuda.synchronize()

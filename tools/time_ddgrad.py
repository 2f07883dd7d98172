"""Stride-2 k = 4 data gradient of the discriminator, timed with and without its mask / channel-sum epilogue:
python3 tools/time_ddgrad.py <cin> <cout> <source extent of the forward> [batch] [option 14 value]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import disc as D
cin, cout, sp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 1
X._lib.load().xh_set_option(14, int(sys.argv[5]) if len(sys.argv) > 5 else 0)
so = (sp + 2 - 4) // 2 + 1
dy = torch.randn(NB, so, so, so, cout, device="cuda").bfloat16()
w = torch.randn(cout, cin, 4, 4, 4, device="cuda") * 0.05
wpt = D._pack(w, 1, cout, cin, torch.bfloat16)
mask = torch.randn(NB, sp, sp, sp, cin, device="cuda").bfloat16()
red = torch.zeros(NB, cin, 2, dtype=torch.float64, device="cuda")


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


flop = 2.0 * NB * so ** 3 * 64 * cin * cout
for name, kw in (("plain", {}), ("mask", dict(mask=mask)), ("mask + sums", dict(mask=mask, red=red))):
    us = t(lambda: D._conv(dy, wpt, None, 1, 2, NB, (so,) * 3, (sp,) * 3, cout, cin, ks=4, **kw))
    print(f"{cin} <- {cout} @{sp}^3 N={NB} {name:12s} {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)

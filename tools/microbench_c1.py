"""1x1 convolution forward at 128^3 / 64^3 with and without the output-moments epilogue, against the workgroup cap (key 10)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
for (cin, cout, S) in [(4, 4, 128), (8, 8, 64), (16, 16, 32), (1, 4, 128)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    w = [torch.randn(cout, cin, 1, 1, 1, device="cuda")]
    b = [torch.randn(cout, device="cuda")]
    red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
    for epi in (0, 2):
        call = lambda: ops.conv3d(x, None, w, b, k=1, cout=cout, epi=epi, red=red if epi else None)
        line = f"1x1 {cin}->{cout} @{S}^3 epi{epi}:"
        for cap in (0, 256, 512, 1024, 4096):
            L.load().xh_set_option(10, cap)
            t = bench(call)
            line += f" cap{cap} {t:.1f} us ({(cin + cout) * S ** 3 * 2 / t / 1e3:.0f} GB/s) |"
        L.load().xh_set_option(10, 0)
        print(line, ops.last_conv_kernel(), flush=True)

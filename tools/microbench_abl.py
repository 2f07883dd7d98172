import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
ops = X.ops; L = X._lib
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout, g, S) in [(16, 16, 4, 128), (4, 4, 1, 128)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
    bs = [torch.randn(cout // g, device="cuda") for _ in range(g)]
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
    for mask, name in [(0, "full"), (1, "no weight pack"), (2, "no staging"), (4, "no mfma loop"), (8, "no stores"), (3, "no pack+stage"), (15, "nothing"), (11, "only mfma"), (14, "only pack"), (13, "only stage")]:
        L.load().xh_set_option(1, mask)
        t = bench(lambda: ops.conv3d(x, None, ws, bs, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red))
        print(f"{cin}->{cout} g{g}: {name:16s} {t:7.1f} us")
    L.load().xh_set_option(1, 0)

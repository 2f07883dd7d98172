"""Times the k3 conv forward (vector vs MFMA kernel) at the 128^3 shapes of the network."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
ops = X.ops
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (cin, cout, g, S) in [(16, 16, 4, 128), (12, 4, 1, 128), (4, 4, 1, 128), (16, 32, 4, 64), (24, 8, 1, 64)]:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16()
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
    bs = [torch.randn(cout // g, device="cuda") for _ in range(g)]
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    res = {}
    for mfma in (False, True):
        ops.set_mfma(mfma)
        red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
        res[mfma] = bench(lambda: ops.conv3d(x, None, ws, bs, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red))
    ops.set_mfma(True)
    nbytes = (cin + cout) * S ** 3 * 2
    flops = 2 * cout * S ** 3 * 27 * cin / g
    print(f"{cin}->{cout} g{g} @{S}^3: vector {res[False]:.1f} us, mfma {res[True]:.1f} us ({nbytes / res[True] / 1e3:.0f} GB/s algorithmic, {flops / res[True] / 1e6:.1f} TFLOP/s useful)")

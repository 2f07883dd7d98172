"""Times the fused gate + max-pool (+ moments) kernels against the three / two separate launches at the level-1..3 shapes of
the 128^3 step (hipGraph replays)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import ops

def bench(fn, n=10, reps=5):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

dt = torch.bfloat16
for c, sp in ((16, 128), (32, 64), (64, 32)) if __name__ == "__main__" else ():
    x = torch.randn(1, c, sp, sp, sp, device="cuda").to(dt)
    a = torch.sigmoid(torch.randn(1, 1, sp, sp, sp, device="cuda")).to(dt)
    red = torch.zeros(1, c, 2, dtype=torch.float64, device="cuda")
    y = ops.gate_maxpool(x, a, red)
    dy = torch.randn_like(y)
    g = ops.gate(x, a)
    def sep_f():
        gg = ops.gate(x, a); yy = ops.maxpool2(gg); ops.moments(yy, red, 0)
    def sep_b():
        dg = ops.maxpool2_bwd(g, dy); ops.gate_bwd(x, a, dg)
    print(f"C={c} @{sp}^3: fwd fused {bench(lambda: ops.gate_maxpool(x, a, red)):6.1f} us vs separate {bench(sep_f):6.1f} us;  "
          f"bwd fused {bench(lambda: ops.gate_maxpool_bwd(x, a, dy)):6.1f} us vs separate {bench(sep_b):6.1f} us")

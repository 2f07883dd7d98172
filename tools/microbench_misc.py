"""GPU-time microbenchmarks (hipGraph replay) of the non-MFMA kernels at network shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import xlstm_hved_amd as X
from microbench_conv import bench
ops = X.ops
S = 128
for C in (4, 16):
    x = torch.randn(1, C, S, S, S, device="cuda").bfloat16()
    w = [torch.randn(C, 1, 3, 3, 3, device="cuda") * 0.1]
    sc = torch.rand(1, C, device="cuda") + 0.5; sh = torch.randn(1, C, device="cuda")
    red = torch.zeros(1, C, 2, dtype=torch.float64, device="cuda")
    nb = 2 * x.numel() * 2
    t = bench(lambda: ops.conv3d(x, None, w, None, k=3, cout=C, groups=C))
    print(f"dw3 {C}ch plain: {t:.1f} us ({nb / t / 1e6:.2f} TB/s)")
    t = bench(lambda: ops.conv3d(x, None, w, None, k=3, cout=C, groups=C, pre=(sc, sh, 0.0)))
    print(f"dw3 {C}ch pre: {t:.1f} us")
    t = bench(lambda: ops.conv3d(x, None, w, None, k=3, cout=C, groups=C, epi=2, red=red))
    print(f"dw3 {C}ch epi2: {t:.1f} us")
    t = bench(lambda: ops.conv3d(x, None, w, None, k=3, cout=C, groups=C, transposed=True, epi=1, e=(x, None, sc, sh, 0.0), red=red))
    print(f"dw3 {C}ch dgrad epi1: {t:.1f} us")
    dw = [torch.zeros(C, 1, 3, 3, 3, device="cuda")]
    t = bench(lambda: ops.conv3d_wgrad(x, None, x, dw, None, k=3, groups=C))
    print(f"dw3 wgrad {C}ch: {t:.1f} us")
    y = torch.empty(1, C, 2 * S // 2, S, S, device="cuda").bfloat16()
    xs = torch.randn(1, C, S // 2, S // 2, S // 2, device="cuda").bfloat16()
    t = bench(lambda: ops.upsample(xs, (S, S, S)))
    print(f"upsample fwd {C}ch 64->128: {t:.1f} us ({(xs.numel() + x.numel()) * 2 / t / 1e6:.2f} TB/s)")
    t = bench(lambda: ops.upsample_bwd(x, (S // 2, S // 2, S // 2)))
    print(f"upsample bwd {C}ch: {t:.1f} us")
    t = bench(lambda: ops.maxpool2(x))
    print(f"maxpool fwd {C}ch: {t:.1f} us")
    s1 = torch.rand(1, 1, S, S, S, device="cuda").bfloat16()
    t = bench(lambda: ops.gate_bwd(x, s1, x))
    print(f"gate_bwd {C}ch: {t:.1f} us")
# k7
P = torch.randn(1, 4, S, S, S, device="cuda").bfloat16()
w7 = [torch.randn(2, 4, 7, 7, 7, device="cuda") * 0.01]; b7 = [torch.zeros(2, device="cuda")]
t = bench(lambda: ops.conv3d(P, None, w7, b7, k=7, cout=2, act=3))
print(f"k7 4->2 fwd 128^3: {t:.1f} us ({2 * 2 * S**3 * 343 * 4 / t / 1e6:.1f} TFLOP/s)")
E = torch.randn(1, 2, S, S, S, device="cuda").bfloat16()
t = bench(lambda: ops.conv3d(E, None, w7, None, k=7, cout=4, transposed=True))
print(f"k7 dgrad 2->4: {t:.1f} us")
dw7 = [torch.zeros(2, 4, 7, 7, 7, device="cuda")]; db7 = [torch.zeros(2, device="cuda")]
t = bench(lambda: ops.conv3d_wgrad(P, None, E, dw7, db7, k=7))
print(f"k7 wgrad: {t:.1f} us")

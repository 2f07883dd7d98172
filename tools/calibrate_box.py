"""Calibrates the GPU box: CU count, clocks, device-copy bandwidth, a streaming moments pass, launch latency."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
p = torch.cuda.get_device_properties(0)
print("device:", p.name, "CUs:", p.multi_processor_count, "mem GB:", p.total_memory / 2**30, "clock MHz:", getattr(p, "clock_rate", 0) / 1e3,
      "L2 MB:", getattr(p, "L2_cache_size", 0) / 2**20)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 512, 2048):
    x = torch.empty(mb * 2**20 // 4, device="cuda", dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy {mb} MB: {2 * mb * 2**20 / t / 1e12:.2f} TB/s (read+write)")
    t = timeit(lambda: x.sum())
    print(f"sum  {mb} MB: {mb * 2**20 / t / 1e12:.2f} TB/s (read)")
    del x, y
import xlstm_hved_amd as X
ops = X.ops
for C, S in ((16, 128), (4, 128), (64, 128)):
    x = torch.randn(1, C, S, S, S, device="cuda").bfloat16()
    red = torch.zeros(1, C, 2, dtype=torch.float64, device="cuda")
    t = timeit(lambda: ops.moments(x, red))
    print(f"xh_moments {C}ch {S}^3 bf16 ({x.numel() * 2 / 2**20:.0f} MB): {t * 1e6:.1f} us -> {x.numel() * 2 / t / 1e12:.2f} TB/s")
    sc = torch.ones(1, C, device="cuda"); sh = torch.zeros(1, C, device="cuda")
    y = torch.empty_like(x)
    t = timeit(lambda: ops.affine_act(x, sc, sh, 2, 0.01, out=y))
    print(f"xh_affine_act {C}ch: {t * 1e6:.1f} us -> {2 * x.numel() * 2 / t / 1e12:.2f} TB/s")
x = torch.zeros(16, device="cuda")
t = timeit(lambda: x.add_(1), n=200)
print(f"tiny kernel launch+run (eager): {t * 1e6:.1f} us")

"""What a plain streaming kernel reaches on this MI355X with data that does not fit the 256 MB memory-side cache: the practical
ceiling next to the 8 TB/s the roofline fractions are quoted against.  python3 tools/probe/hbm_stream.py"""
import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for gib in (0.125, 1, 4):
    n = int(gib * 2 ** 30) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    tc = t(lambda: y.copy_(x))
    ts = t(lambda: x.sum(dtype=torch.float32))
    tf = t(lambda: y.zero_())
    ta = t(lambda: torch.add(x, y, out=y))
    print(f"{gib:6.3f} GiB tensors: copy {2 * n * 2 / tc / 1e12:5.2f} TB/s (read + write)   sum {n * 2 / ts / 1e12:5.2f} TB/s (read)   "
          f"fill {n * 2 / tf / 1e12:5.2f} TB/s (write)   a + b -> b {3 * n * 2 / ta / 1e12:5.2f} TB/s (2 reads + 1 write)", flush=True)

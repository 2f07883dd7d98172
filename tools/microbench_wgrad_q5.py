"""Full-row weight-gradient kernel (conv3d_wgrad_q5.hip) against the tile kernel (conv3d_wgrad_q4.hip, xh_set_option(21, 0)) at the
step's shapes, operands rotated so they come from HBM (XH_ROT sets), each problem alone; then the whole end-of-backward batch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
ROT = int(os.environ.get("XH_ROT", "6"))
shapes = [(16, 16, 4, 128), (12, 4, 1, 128), (4, 4, 1, 128), (8, 8, 2, 128), (4, 12, 1, 128), (24, 8, 1, 64), (8, 8, 1, 64), (32, 32, 4, 64), (16, 16, 2, 64)]
for (cin, cout, g, S) in shapes:
    sets = [(torch.randn(1, cin, S, S, S, device="cuda").bfloat16(), torch.randn(1, cout, S, S, S, device="cuda").bfloat16()) for _ in range(ROT)]
    tick = [0]
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    dws = [torch.zeros(cout // g, cin // g, 3, 3, 3, device="cuda") for _ in range(g)]
    dbs = [torch.zeros(cout // g, device="cuda") for _ in range(g)]
    def call():
        tick[0] += 1
        x, dy = sets[tick[0] % ROT]
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=(sc, sh, 0.01))
    mb = (cin + cout) * S ** 3 * 2 / 1e6
    line = f"wgrad {cin}->{cout} g{g} @{S}^3 ({mb:.0f} MB):"
    for on in (1, 0):
        L.load().xh_set_option(21, on)
        for abl, nm in ((0, ""), (2048, " no atomics")):
            L.load().xh_set_option(1, abl)
            t = bench(call)
            line += f" {'full-row' if on else 'tile'}{nm} {t:.1f} us ({mb / t / 1e3 * 1e3 / 1e3:.2f} TB/s) |"
    L.load().xh_set_option(1, 0); L.load().xh_set_option(21, 1)
    print(line, flush=True)
# the batch of one step (bench shapes), deferred flush
# the 24 quad-channel problems of one 128^3 step (bench.py's roofline object lists them)
allp = [(8, 8, 2, 128), (4, 4, 4, 128), (4, 4, 1, 128), (4, 4, 1, 128), (16, 16, 4, 128), (16, 16, 4, 128), (12, 4, 1, 128), (12, 4, 1, 128),
        (8, 8, 8, 64), (20, 40, 5, 64), (20, 20, 5, 64), (16, 16, 2, 64), (8, 8, 1, 64), (8, 8, 1, 64), (24, 8, 1, 64), (24, 8, 1, 64),
        (16, 16, 16, 32), (32, 32, 2, 32), (16, 16, 1, 32), (16, 16, 1, 32), (40, 80, 5, 32), (40, 40, 5, 32), (48, 16, 1, 32), (48, 16, 1, 32)]
data = []
for cin, cout, g, S in allp:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16(); dy = torch.randn(1, cout, S, S, S, device="cuda").bfloat16()
    data.append((x, dy, (torch.rand(1, cin, device="cuda") + 0.5, torch.randn(1, cin, device="cuda"), 0.01), g,
                 [torch.zeros(cout // (g if g <= 4 else 1), cin // g, 3, 3, 3, device="cuda") for _ in range(g if g <= 4 else 1)],
                 [torch.zeros(cout // (g if g <= 4 else 1), device="cuda") for _ in range(g if g <= 4 else 1)]))
def batch():
    ops.set_wgrad_defer(True)
    for x, dy, pre, g, dws, dbs in data:
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=pre, side=True)
    ops.join_wgrad_stream()
    ops.set_wgrad_defer(False)
tot = sum((c + o) * S ** 3 * 2 for c, o, g, S in allp) / 1e6
for on in (1, 0):
    L.load().xh_set_option(21, on)
    t = bench(batch, n=4)
    print(f"batch of {len(allp)} problems ({tot:.0f} MB): {'full-row' if on else 'tile'} {t:.1f} us ({tot / t:.2f} TB/s)", flush=True)
L.load().xh_set_option(21, 1)

"""Full-row weight gradient on rows of 64 voxels: units of one quad of each operand (xh_set_option(28, 0)) against units of two input /
two output quads (28, 1 | 2 | 3), each 64^3 problem of the step alone (operands rotated through HBM), then the step's end-of-backward
batch of 128^3 + 64^3 problems."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
ROT = int(os.environ.get("XH_ROT", "6"))
shapes = [(12, 4, 1, 128), (24, 8, 1, 64), (8, 8, 1, 64), (16, 16, 2, 64), (20, 40, 5, 64), (20, 20, 5, 64), (32, 32, 4, 64)]
for (cin, cout, g, S) in shapes:
    sets = [(torch.randn(1, cin, S, S, S, device="cuda").bfloat16(), torch.randn(1, cout, S, S, S, device="cuda").bfloat16()) for _ in range(ROT)]
    tick = [0]
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    nw = g if g <= 4 else 1
    dws = [torch.zeros(cout // nw, cin // g, 3, 3, 3, device="cuda") for _ in range(nw)]
    dbs = [torch.zeros(cout // nw, device="cuda") for _ in range(nw)]
    def call():
        tick[0] += 1
        x, dy = sets[tick[0] % ROT]
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=(sc, sh, 0.01))
    mb = (cin + cout) * S ** 3 * 2 / 1e6
    line = f"wgrad {cin}->{cout} g{g} @{S}^3 ({mb:.0f} MB):"
    for uq in (0, 1, 2, 3, 7):
        L.load().xh_set_option(28, uq)
        t = bench(call)
        line += f" uq{uq} {t:.1f} us ({mb / t / 1e3:.2f} TB/s) |"
    L.load().xh_set_option(28, 7)
    print(line, flush=True)
allp = [(8, 8, 2, 128), (4, 4, 4, 128), (4, 4, 1, 128), (4, 4, 1, 128), (16, 16, 4, 128), (16, 16, 4, 128), (12, 4, 1, 128), (12, 4, 1, 128),
        (8, 8, 8, 64), (20, 40, 5, 64), (20, 20, 5, 64), (16, 16, 2, 64), (8, 8, 1, 64), (8, 8, 1, 64), (24, 8, 1, 64), (24, 8, 1, 64)]
data = []
for cin, cout, g, S in allp:
    x = torch.randn(1, cin, S, S, S, device="cuda").bfloat16(); dy = torch.randn(1, cout, S, S, S, device="cuda").bfloat16()
    nw = g if g <= 4 else 1
    data.append((x, dy, (torch.rand(1, cin, device="cuda") + 0.5, torch.randn(1, cin, device="cuda"), 0.01), g,
                 [torch.zeros(cout // nw, cin // g, 3, 3, 3, device="cuda") for _ in range(nw)],
                 [torch.zeros(cout // nw, device="cuda") for _ in range(nw)]))
def batch():
    ops.set_wgrad_defer(True)
    for x, dy, pre, g, dws, dbs in data:
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=pre, side=True)
    ops.join_wgrad_stream()
    ops.set_wgrad_defer(False)
tot = sum((c + o) * S ** 3 * 2 for c, o, g, S in allp) / 1e6
for uq in (0, 3, 7, 0, 3, 7):
    L.load().xh_set_option(28, uq)
    t = bench(batch, n=4)
    print(f"batch of {len(allp)} problems ({tot:.0f} MB): uq{uq} {t:.1f} us ({tot / t / 1e3:.2f} TB/s)", flush=True)
L.load().xh_set_option(28, 7)

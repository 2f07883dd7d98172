"""DuSE adjust convs (modules/DuSFE.py:141-144): 1 -> 2 channels k=3 (LDS-tiled vector kernels) vs the depthwise pair on a duplicated
squeeze channel (sliding-window kernels): forward (+ sigmoid), data gradient, weight gradient; hipGraph replays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from xlstm_hved_amd import ops
from tools.microbench_gmp import bench  # noqa

dt = torch.bfloat16
for sp in (128, 64, 32):
    w = torch.randn(2, 1, 3, 3, 3, device="cuda") * 0.2
    b = torch.randn(2, device="cuda")
    c1 = torch.randn(1, 1, sp, sp, sp, device="cuda").to(dt)
    c2 = c1.repeat(1, 2, 1, 1, 1).contiguous()
    dy = torch.randn(1, 2, sp, sp, sp, device="cuda").to(dt)
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    res = []
    for x, g in ((c1, 1), (c2, 2)):
        f = bench(lambda: ops.conv3d(x, None, [w], [b], k=3, cout=2, groups=g, act=ops.ACT_SIGMOID)); kf = ops.last_conv_kernel()
        d = bench(lambda: ops.conv3d(dy, None, [w], None, k=3, cout=g, groups=g, transposed=True)); kd = ops.last_conv_kernel()
        wg = bench(lambda: ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3, groups=g)); kw = ops.last_conv_kernel()
        res.append((f, d, wg, kf, kd, kw))
    for tag, r in zip(("1->2 dense   ", "2ch depthwise"), res):
        print(f"@{sp}^3 {tag}: fwd {r[0]:6.1f} us  dgrad {r[1]:6.1f} us  wgrad {r[2]:6.1f} us   [{r[3]} | {r[4]} | {r[5]}]")

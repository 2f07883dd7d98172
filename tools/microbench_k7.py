"""Times AttenModule2's composed 7^3 gate conv (forward 4->2, data gradient 2->4, weight gradient) per volume size."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import xlstm_hved_amd as X
from microbench_conv import bench
ops = X.ops
L = X._lib.load()
for S in (32, 64, 128):
    x = torch.randn(1, 4, S, S, S, device="cuda").bfloat16()
    w = torch.randn(2, 4, 7, 7, 7, device="cuda") * 0.05
    b = torch.randn(2, device="cuda")
    dy = torch.randn(1, 2, S, S, S, device="cuda").bfloat16()
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    t_f = bench(lambda: ops.conv3d(x, None, [w], [b], k=7, cout=2, act=ops.ACT_SIGMOID))
    t_d = bench(lambda: ops.conv3d(dy, None, [w], None, k=7, cout=4, transposed=True))
    t_w = bench(lambda: ops.conv3d_wgrad(x, None, dy, [dw], [db], k=7))
    L.xh_set_option(24, 0)                            # the output-stationary kernel of rounds 2 - 4
    t_f0 = bench(lambda: ops.conv3d(x, None, [w], [b], k=7, cout=2, act=ops.ACT_SIGMOID))
    t_d0 = bench(lambda: ops.conv3d(dy, None, [w], None, k=7, cout=4, transposed=True))
    L.xh_set_option(24, 2)                            # input-stationary, weight fragments in registers
    t_f2 = bench(lambda: ops.conv3d(x, None, [w], [b], k=7, cout=2, act=ops.ACT_SIGMOID))
    t_d2 = bench(lambda: ops.conv3d(dy, None, [w], None, k=7, cout=4, transposed=True))
    L.xh_set_option(24, 1)
    print(f"k7 @{S}^3: fwd {t_f:.1f} us (output-stationary {t_f0:.1f}, B in registers {t_f2:.1f}), dgrad {t_d:.1f} us ({t_d0:.1f}, {t_d2:.1f}), wgrad {t_w:.1f} us", flush=True)

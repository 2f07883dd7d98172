"""The deep-level launches of the step that run on a few dozen workgroups (tools/dump_step.py: 10 - 30 us each for kilobytes of
data): k1 convs with output moments, the fused gate / max-pool and skip-return backward kernels, channel pools, upsample
backward, stride-2 convs.  hipGraph-captured, 20 calls per replay.   python tools/microbench_deep.py [c1|gmp|pool|s2|up ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib
which = set(sys.argv[1:]) or {"c1", "gmp", "pool", "s2", "up"}
dt = torch.bfloat16
R = lambda *s: torch.randn(*s, device="cuda").to(dt)

if "c1" in which:
    for cin, cout, S in [(8, 8, 64), (16, 16, 32), (1, 4, 64), (2, 8, 32), (4, 16, 16), (8, 32, 8), (4, 4, 128)]:
        x = R(1, cin, S, S, S)
        w = [torch.randn(cout, cin, 1, 1, 1, device="cuda") * 0.1]; b = [torch.randn(cout, device="cuda")]
        red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
        t2 = bench(lambda: ops.conv3d(x, None, w, b, k=1, cout=cout, epi=2, red=red))
        k2 = ops.last_conv_kernel()
        t0 = bench(lambda: ops.conv3d(x, None, w, b, k=1, cout=cout))
        print(f"k1 {cin:2d}->{cout:2d} @{S:3d}^3: moments {t2:6.1f} us, plain {t0:6.1f} us  [{k2}]", flush=True)

if "gmp" in which:
    for C, S in [(16, 128), (32, 64), (64, 32)]:
        x = R(1, C, S, S, S); g = torch.rand(1, 1, S, S, S, device="cuda").to(dt)
        red = torch.zeros(1, C, 2, dtype=torch.float64, device="cuda")
        dy = R(1, C, S // 2, S // 2, S // 2)
        tf = bench(lambda: ops.gate_maxpool(x, g, red))
        tb = bench(lambda: ops.gate_maxpool_bwd(x, g, dy))
        print(f"gate_maxpool C={C} @{S}^3: fwd {tf:6.1f} us, bwd {tb:6.1f} us", flush=True)

if "pool" in which:
    for ca, cb, S in [(32, 16, 32), (16, 8, 64), (8, 4, 128)]:
        a = R(1, ca, S, S, S); b_ = R(1, cb, S, S, S)
        tf = bench(lambda: ops.channel_pool2(a, b_))
        p = ops.channel_pool2(a, b_); dp = R(*p.shape)
        tb = bench(lambda: ops.channel_pool2_bwd(a, b_, dp))
        print(f"channel_pool2 {ca}+{cb} @{S}^3: fwd {tf:6.1f} us, bwd {tb:6.1f} us", flush=True)

if "up" in which:
    for C, S in [(32, 16), (16, 32), (8, 64), (4, 128)]:
        dy = R(1, C, S, S, S)
        tb = bench(lambda: ops.upsample_bwd(dy, (S // 2, S // 2, S // 2)))
        x = R(1, C, S // 2, S // 2, S // 2)
        tf = bench(lambda: ops.upsample(x, (S, S, S)))
        print(f"upsample2x C={C} ->{S}^3: fwd {tf:6.1f} us, bwd {tb:6.1f} us", flush=True)

if "s2" in which:
    for cin, cout, g, S in [(16, 8, 4, 128), (32, 16, 4, 64), (64, 32, 4, 32), (128, 64, 4, 16)]:
        x = R(1, cin, S, S, S)
        ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
        bs = [torch.randn(cout // g, device="cuda") for _ in range(g)]
        sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
        tf = bench(lambda: ops.conv3d(x, None, ws, bs, k=3, cout=cout, stride=2, groups=g, pre=(sc, sh, 0.01)))
        kf = ops.last_conv_kernel()
        dy = R(1, cout, S // 2, S // 2, S // 2)
        tb = bench(lambda: ops.conv3d_dgrad_s2(dy, ws, cin=cin, in_spatial=(S, S, S), groups=g))
        print(f"k3 s2 {cin}->{cout} g{g} @{S}^3: fwd {tf:6.1f} us [{kf}], dgrad {tb:6.1f} us [{ops.last_conv_kernel()}]", flush=True)

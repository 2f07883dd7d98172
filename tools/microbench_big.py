"""128^3-class k3 convolutions as the training step issues them: forward (producer norm + LeakyReLU on the way in, output
moments in the epilogue) and data gradient (transposed weights, leaky'-masked norm-backward sums in the epilogue), with the
MFMA kernels' ablation masks.  hipGraph-captured, 20 calls per replay.   python tools/microbench_big.py [--abl]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import xlstm_hved_amd as X
from tools.microbench_conv import bench
ops = X.ops; L = X._lib

S = int(os.environ.get("XH_S", "128"))
abl = "--abl" in sys.argv
masks = [(0, "full")] + ([(4, "no mfma"), (4096, "setup only"), (8192, "staging only")] if abl else [])
for kv in filter(None, os.environ.get("XH_OPTS", "").split(",")):      # e.g. XH_OPTS=19=0 (no persistent kernel), 19=2,1=16384
    k_, v_ = kv.split("=")
    L.load().xh_set_option(int(k_), int(v_))
BASE_ABL = int(os.environ.get("XH_ABL", "0"))
ROT = int(os.environ.get("XH_ROT", "1"))      # buffer sets walked round-robin: > 1.3 sets of 201 MB no longer sit in the 256 MB last-level cache
SHAPES = [(4, 4, 1), (16, 16, 4), (12, 4, 1), (8, 8, 1)] if S >= 128 else [(8, 8, 1), (20, 20, 5), (20, 40, 5), (24, 8, 1), (16, 16, 2)]
for (cin, cout, g) in SHAPES[:int(os.environ.get("XH_NSHAPE", "5"))]:
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3, device="cuda") * 0.1 for _ in range(g)]
    bs = [torch.randn(cout // g, device="cuda") for _ in range(g)]
    sc = torch.rand(1, cin, device="cuda") + 0.5; sh = torch.randn(1, cin, device="cuda")
    esc = torch.rand(1, cin, device="cuda") + 0.5; esh = torch.randn(1, cin, device="cuda")
    red = torch.zeros(1, cout, 2, dtype=torch.float64, device="cuda")
    red2 = torch.zeros(1, cin, 2, dtype=torch.float64, device="cuda")
    sets = [(torch.randn(1, cin, S, S, S, device="cuda").bfloat16(), torch.randn(1, cout, S, S, S, device="cuda").bfloat16()) for _ in range(ROT)]
    tick = [0]
    def rot(f):
        def call():
            tick[0] += 1
            return f(*sets[tick[0] % ROT])
        return call
    fwd = rot(lambda x, dy: ops.conv3d(x, None, ws, bs, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red))
    fwd0 = rot(lambda x, dy: ops.conv3d(x, None, ws, bs, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=0))
    dgr = rot(lambda x, dy: ops.conv3d(dy, None, ws, None, k=3, cout=cin, groups=g, transposed=True, epi=1, e=(x, None, esc, esh, 0.01), red=red2))
    for name, call, nb in [("fwd epi2", fwd, (cin + cout)), ("fwd epi0", fwd0, (cin + cout)), ("dgrad epi1", dgr, (cout + 2 * cin))]:
        line = f"{cin}->{cout} g{g} @{S}^3 {name:10s}"
        for m, nm in masks:
            L.load().xh_set_option(1, m | BASE_ABL)
            t = bench(call)
            line += f" | {nm} {t:6.1f} us" + (f" ({nb * S ** 3 * 2 / t / 1e3:.0f} GB/s)" if m == 0 else "")
        L.load().xh_set_option(1, BASE_ABL)
        print(line + f"  [{ops.last_conv_kernel()}]", flush=True)

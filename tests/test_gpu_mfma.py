"""-m gpu: the 16-bit MFMA implicit-GEMM conv kernels (bf16 and fp16 storage) against the vector kernels (same 16-bit inputs)
and stock fp32 ops -- at small shapes (512-thread instances) and at the 128^3 shapes of BASELINE config 2, where the
256-thread `big` instances that carry the headline run."""
import pytest
import torch

from gpu_common import l2_err

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402

DEV = "cuda"
CASES = [
    dict(cin=16, cout=16, groups=4, sp=(8, 16, 32)),          # the four modality encoders batched (4->4 x4)
    dict(cin=16, cout=32, groups=4, sp=(8, 8, 64)),           # 4->8 x4: two groups per block set
    dict(cin=32, cout=64, groups=4, sp=(5, 9, 32)),           # 8->16 x4: one group per set, ragged D/H
    dict(cin=64, cout=128, groups=4, sp=(4, 8, 32)),          # 16->32 x4: two 16-wide output tiles per set
    dict(cin=12, cout=4, groups=1, sp=(8, 8, 32), split=4),   # decoder conv on a virtual concat, CINP=12
    dict(cin=24, cout=8, groups=1, sp=(4, 8, 32), split=8),   # CINP=24
    dict(cin=4, cout=4, groups=1, sp=(6, 8, 32)),             # CINP=4 (two 8-byte fragment reads)
    dict(cin=8, cout=8, groups=1, sp=(4, 8, 32)),
    dict(cin=64, cout=128, groups=4, sp=(8, 8, 16)),          # 16-wide volumes (level 3 of a 128^3 patch): TW=16 tiles
    dict(cin=16, cout=16, groups=1, sp=(6, 9, 16)),
    dict(cin=16, cout=16, groups=16, sp=(5, 8, 32)),          # depthwise: dedicated no-LDS kernel (not MFMA)
    dict(cin=48, cout=16, groups=1, sp=(8, 8, 32), split=16), # > 24 input channels: split-K launches with fp32 partials
    dict(cin=96, cout=32, groups=1, sp=(4, 8, 16), split=32),
    dict(cin=28, cout=8, groups=1, sp=(4, 8, 32)),            # uneven split (16 + 12)
    dict(cin=128, cout=64, groups=4, sp=(4, 8, 16)),          # grouped, 32 channels per group: split-K inside each group
    dict(cin=64, cout=128, groups=4, sp=(8, 8, 8)),           # 8-wide volumes (level 4): wgrad K steps of 4 rows x 8 voxels
    dict(cin=32, cout=16, groups=1, sp=(5, 7, 8)),            # ... ragged D/H
    dict(cin=32, cout=16, groups=1, sp=(5, 11, 16)),          # 16-wide, ragged H: wgrad K steps of 2 rows x 16 voxels
    dict(cin=4, cout=12, groups=1, sp=(11, 13, 64)),          # quad-channel kernel: three output quads, ragged D/H tiles, two W tiles
    dict(cin=8, cout=12, groups=1, sp=(5, 9, 32)),            # ... two input quads through the same LDS tile
    dict(cin=12, cout=12, groups=1, sp=(9, 10, 32), split=8), # ... virtual concat split on a quad boundary
    dict(cin=24, cout=24, groups=2, sp=(8, 8, 32), split=12), # ... grouped, 12 channels per group
]


DTYPES = [torch.bfloat16, torch.float16]
# relative-L2 bands: bf16 rounds operands to 2^-9, fp16 to 2^-12
TOL = {torch.bfloat16: dict(y=8e-3, dx=5e-2, dw=3e-2), torch.float16: dict(y=1.5e-3, dx=8e-3, dw=5e-3)}


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", CASES)
def test_mfma_conv_forward_backward(cfg, dtype):
    torch.manual_seed(11)
    n, cin, cout, g = 2, cfg["cin"], cfg["cout"], cfg["groups"]
    x = (torch.randn((n, cin) + cfg["sp"]) * 1.5 + 0.3).to(dtype)
    nw = g if g <= 4 else 1                             # the C ABI takes one weight pointer, or one per group (<= 4)
    ws = [torch.randn(cout // nw, cin // g, 3, 3, 3) * (2.0 / (27 * cin // g)) ** 0.5 for _ in range(nw)]
    bs = [torch.randn(cout // nw) for _ in range(nw)]
    wgt = torch.randn((n, cout) + cfg["sp"])

    def run(mfma):
        X.ops.set_mfma(mfma)
        try:
            xg = x.to(DEV).requires_grad_(True)
            wg = [w.to(DEV).requires_grad_(True) for w in ws]
            bg = [b.to(DEV).requires_grad_(True) for b in bs]
            xa, xb = (xg[:, :cfg["split"]], xg[:, cfg["split"]:]) if "split" in cfg else (xg, None)
            y = X.functional.in_lrelu_conv(xa, xb, wg, bg, 1, g)
            (y.float() * wgt.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            return y.detach().float().cpu(), xg.grad.float().cpu(), [w.grad.cpu() for w in wg], [b.grad.cpu() for b in bg]
        finally:
            X.ops.set_mfma(True)
    y1, dx1, dw1, db1 = run(True)
    y0, dx0, dw0, db0 = run(False)
    # stock fp32 ops on the same bf16-representable input
    xo = x.float().requires_grad_(True)
    wo = [w.clone().requires_grad_(True) for w in ws]
    bo = [b.clone().requires_grad_(True) for b in bs]
    h = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(xo, eps=1e-5), 0.01)
    yo = torch.nn.functional.conv3d(h, torch.cat(wo, 0), torch.cat(bo, 0), padding=1, groups=g)
    (yo * wgt).sum().backward()
    # forward: MFMA rounds the normalised activations and the weights to bf16 (2^-9 each) on top of the output rounding
    e = dict(y_vs_stock=l2_err(y1, yo), y_vs_vector=l2_err(y1, y0), dx_vs_stock=l2_err(dx1, xo.grad), dx_vs_vector=l2_err(dx1, dx0))
    print(cfg, dtype, {k: f"{v:.2e}" for k, v in e.items()})
    t = TOL[dtype]
    assert e["y_vs_stock"] < t["y"] and e["y_vs_vector"] < t["y"], e
    assert e["dx_vs_stock"] < t["dx"] and e["dx_vs_vector"] < t["dx"], e
    gmax = max(w.grad.abs().max() for w in wo)
    for a, b in zip(dw1 + db1, [w.grad for w in wo] + [b.grad for b in bo]):
        assert l2_err(a, b) < t["dw"] or (a - b).abs().max() < t["dw"] * 0.7 * gmax


def test_quad_channel_kernel_is_selected():
    """Convs with <= 48 channels per group on 32-multiple rows take conv3_q4_kernel (forward and data gradient); the others keep
    the plain implicit-GEMM kernels; xh_set_option(2, 16) switches the quad-channel path off (A/B runs)."""
    lib = X._lib.load()
    x = torch.randn(1, 16, 8, 8, 32, device=DEV).bfloat16()
    w4 = [torch.randn(4, 4, 3, 3, 3, device=DEV) for _ in range(4)]
    X.ops.conv3d(x, None, w4, None, k=3, cout=16, groups=4)
    assert "conv3_q4_kernel<0, 0, 0, false, false" in X.ops.last_conv_kernel()      # (+ ", 2>": planes per workgroup)
    X.ops.conv3d(x.half(), None, w4, None, k=3, cout=16, groups=4, pre=(torch.ones(1, 16, device=DEV), torch.zeros(1, 16, device=DEV), 0.01),
                 epi=2, red=torch.zeros(1, 16, 2, dtype=torch.float64, device=DEV))
    assert "conv3_q4_kernel<1, 1, 2, false, false" in X.ops.last_conv_kernel()
    x64 = torch.randn(1, 64, 8, 8, 32, device=DEV).bfloat16()
    X.ops.conv3d(x64, None, [torch.randn(16, 64, 3, 3, 3, device=DEV)], None, k=3, cout=16)      # > 48 channels per group
    assert "conv3_mfma_kernel" in X.ops.last_conv_kernel()
    x16 = torch.randn(1, 16, 8, 8, 16, device=DEV).bfloat16()
    X.ops.conv3d(x16, None, [torch.randn(16, 16, 3, 3, 3, device=DEV)], None, k=3, cout=16)      # 16-wide rows
    assert "conv3_mfma_kernel" in X.ops.last_conv_kernel()
    lib.xh_set_option(2, 16)
    try:
        X.ops.conv3d(x, None, w4, None, k=3, cout=16, groups=4)
        assert "conv3_q4_kernel" not in X.ops.last_conv_kernel()
    finally:
        lib.xh_set_option(2, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_quad_channel_kernel_planes_per_workgroup(dtype):
    """conv3_q4_kernel walks 8, 4 or 2 output planes per workgroup (xh_set_option(17, n): fewer planes while the launch has
    fewer than n workgroups -- the 64^3 / 32^3 levels of a step).  The three tilings give the same outputs bit for bit (the
    accumulation order of a voxel does not depend on the tile) and the same statistics to fp64 round-off: forward with the
    producer's norm on load + output moments, data gradient with the activation mask + norm-backward sums; ragged D and H."""
    torch.manual_seed(5)
    lib = X._lib.load()
    n, cin, cout, sp = 2, 12, 12, (11, 13, 64)
    x = torch.randn((n, cin) + sp, device=DEV).to(dtype)
    dy = torch.randn((n, cout) + sp, device=DEV).to(dtype)
    w = [torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.1]
    b = [torch.randn(cout, device=DEV)]
    sc, sh = torch.rand(n, cin, device=DEV) + 0.5, torch.randn(n, cin, device=DEV)
    outs = {}
    # workgroups of this launch with 8 / 4 planes each: 2 x 2 tiles x 3 output quads x 2 samples x (2 | 3) = 48 | 72
    for name, thr in (("8", 0), ("4", 60), ("2", 512)):
        lib.xh_set_option(17, thr)
        lib.xh_set_option(20, 0)                      # the 32-wide tile kernel under test (64-wide rows default to conv3_q4w_kernel)
        try:
            red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
            y = X.ops.conv3d(x, None, w, b, k=3, cout=cout, pre=(sc, sh, 0.01), epi=2, red=red)
            kf = X.ops.last_conv_kernel()
            red2 = torch.zeros(n, cin, 2, dtype=torch.float64, device=DEV)
            dx = X.ops.conv3d(dy, None, w, None, k=3, cout=cin, transposed=True, epi=1, e=(x, None, sc, sh, 0.01), red=red2)
            kb = X.ops.last_conv_kernel()
        finally:
            lib.xh_set_option(17, 512)
            lib.xh_set_option(20, 3)
        assert kf.endswith(f"true, {name}>") and kb.endswith(f"true, {name}>"), (name, kf, kb)
        outs[name] = (y.float(), red.clone(), dx.float(), red2.clone())
    for name in ("4", "2"):
        assert torch.equal(outs[name][0], outs["8"][0]) and torch.equal(outs[name][2], outs["8"][2])
        for i in (1, 3):
            a, r = outs[name][i], outs["8"][i]
            assert (a - r).abs().max().item() <= 1e-5 * r.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("transposed", [False, True], ids=["fwd", "dgrad"])
def test_depthwise_conv_through_quad_channel_kernel(transposed, dtype):
    """A depthwise k3 conv with a multiple of 4 channels on 32-multiple rows runs on the matrix cores as groups of 4 channels
    with diagonal 4 x 4 weight blocks (conv3_q4_kernel); xh_set_option(2, 128) puts it back on the sliding-window vector
    kernel.  Same result up to the 16-bit rounding of the weights; bias included; H not a multiple of the tile."""
    torch.manual_seed(8)
    lib = X._lib.load()
    n, c, sp = 2, 8, (8, 34, 64)
    x = torch.randn((n, c) + sp, device=DEV).to(dtype)
    ws = [(torch.randn(c, 1, 3, 3, 3, device=DEV) * 0.3).to(dtype).float()]
    bs = None if transposed else [torch.randn(c, device=DEV)]
    call = lambda: X.ops.conv3d(x, None, ws, bs, k=3, cout=c, groups=c, transposed=transposed).float()
    y = call()
    assert "conv3_q4w_kernel" in X.ops.last_conv_kernel()          # (rows of 64 voxels: the full-row variant of the quad-channel kernel)
    lib.xh_set_option(2, 128)
    try:
        y_vec = call()
        assert "conv3_q4" not in X.ops.last_conv_kernel()
    finally:
        lib.xh_set_option(2, 0)
    wf = torch.cat(ws, 0)
    if transposed:
        ref = torch.nn.functional.conv_transpose3d(x.float(), wf, None, padding=1, groups=c)
    else:
        ref = torch.nn.functional.conv3d(x.float(), wf, torch.cat(bs, 0), padding=1, groups=c)
    tol = 4e-3 if dtype == torch.bfloat16 else 6e-4
    assert l2_err(y, ref) < tol and l2_err(y_vec, ref) < tol
    if not transposed:
        # weight / bias gradient: the diagonal of the 4 x 4 blocks the quad-channel weight-gradient kernel accumulates
        dy = torch.randn_like(x)
        grads = []
        for opt in (0, 128):
            lib.xh_set_option(2, opt)
            try:
                dw, db = torch.zeros_like(ws[0]), torch.zeros(c, device=DEV)
                X.ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3, groups=c)
                grads.append((dw, db, X.ops.last_conv_kernel()))
            finally:
                lib.xh_set_option(2, 0)
        assert ("conv3_wgrad_q4" in grads[0][2] or "conv3_wgrad_q5" in grads[0][2]) and "q4" not in grads[1][2] and "q5" not in grads[1][2]
        xr = x.float().requires_grad_(False)
        wr = ws[0].clone().requires_grad_(True)
        br = torch.zeros(c, device=DEV, requires_grad=True)
        (torch.nn.functional.conv3d(xr, wr, br, padding=1, groups=c) * dy.float()).sum().backward()
        for dw, db, _ in grads:
            assert l2_err(dw, wr.grad) < 2e-3 and l2_err(db, br.grad) < 2e-3


@pytest.mark.parametrize("cfg", [dict(cin=16, cout=16, groups=4, sp=(8, 16, 32)), dict(cin=16, cout=16, groups=1, sp=(6, 8, 32)),
                                 dict(cin=48, cout=16, groups=1, sp=(8, 8, 32))], ids=["q4", "gemm", "splitk"])
def test_prepacked_fragments_match_and_never_go_stale(cfg):
    """ops.prepack_all() packs every registered conv's weight fragments in one launch (xh_conv3d_prepack); a conv then runs
    without its own pack launch.  Same bits as packing in front of the conv; a weight changed after the prepack (version
    counter) or an old epoch makes the conv pack for itself again instead of reading stale fragments."""
    torch.manual_seed(5)
    n, cin, cout, g = 2, cfg["cin"], cfg["cout"], cfg["groups"]
    x = torch.randn((n, cin) + cfg["sp"], device=DEV).bfloat16()
    nw = g if g <= 4 else 1
    ws = [torch.randn(cout // nw, cin // g, 3, 3, 3, device=DEV) * 0.1 for _ in range(nw)]
    call = lambda: X.ops.conv3d(x, None, ws, None, k=3, cout=cout, groups=g).float()
    X.ops.set_prepack(False)
    try:
        ref = call()
    finally:
        X.ops.set_prepack(True)
    y0 = call()                                   # first sighting: registers, packs on its own
    X.ops.prepack_all()
    y1 = call()                                   # prepacked
    assert torch.equal(ref, y0) and torch.equal(ref, y1)
    for w in ws:
        w.mul_(-2.0)                              # in-place update, no prepack_all: the version counters give it away
    y2 = call()
    assert torch.allclose(y2, -2.0 * ref, rtol=2e-2, atol=1e-2 * ref.abs().max().item())
    X.ops.prepack_all()
    assert torch.equal(call(), y2)
    # the ABI flag itself: ws_packed = 1 with fragments packed by xh_conv3d_prepack for OTHER weights reads those fragments
    X.ops.set_prepack(False)
    try:
        assert torch.equal(call(), y2)
    finally:
        X.ops.set_prepack(True)


@pytest.mark.parametrize("case", ["q4", "gemm", "k1", "dw3"])
def test_statistics_fan_in_matches_direct_atomics_and_cleans_up(case):
    """Launches with >= 256 workgroups per (sample, channel block) sum their epilogue statistics through the two-level fan-in
    of csrc/fanin.h (arena replicas + counters, collected by the last workgroup) instead of same-line atomics.  Same sums as
    the direct path (xh_set_option(2, 64)) and as the stored output; repeated launches keep accumulating correctly, i.e.
    every launch leaves the arena zero (a stale counter or replica would show up in the next launch's totals)."""
    torch.manual_seed(9)
    lib = X._lib.load()
    if case == "q4":
        x = torch.randn(2, 8, 64, 64, 64, device=DEV).bfloat16()
        w, kw = [torch.randn(4, 4, 3, 3, 3, device=DEV) * 0.1 for _ in range(2)], dict(k=3, cout=8, groups=2)
    elif case == "gemm":
        x = torch.randn(1, 16, 64, 64, 64, device=DEV).bfloat16()
        w, kw = [torch.randn(16, 16, 3, 3, 3, device=DEV) * 0.05], dict(k=3, cout=16)
    elif case == "k1":
        x = torch.randn(1, 4, 128, 128, 128, device=DEV).bfloat16()
        w, kw = [torch.randn(8, 4, 1, 1, 1, device=DEV)], dict(k=1, cout=8)
    else:
        x = torch.randn(1, 4, 128, 128, 128, device=DEV).bfloat16()
        w, kw = [torch.randn(4, 1, 3, 3, 3, device=DEV) * 0.2], dict(k=3, cout=4, groups=4)
    n, cout = x.shape[0], kw["cout"]

    def run(reps):
        red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
        for _ in range(reps):
            y = X.ops.conv3d(x, None, w, None, epi=2, red=red, **kw)
        torch.cuda.synchronize()
        return y, red
    y, red1 = run(1)
    _, red3 = run(3)
    lib.xh_set_option(2, 64)
    try:
        _, red_direct = run(1)
    finally:
        lib.xh_set_option(2, 0)
    ys = y.double()
    ref = torch.stack((ys.sum((2, 3, 4)), (ys * ys).sum((2, 3, 4))), -1)
    scale = torch.stack((ys.abs().sum((2, 3, 4)), (ys * ys).sum((2, 3, 4))), -1)
    assert ((red1 - ref).abs() / scale).max().item() < 1e-6
    assert ((red_direct - ref).abs() / scale).max().item() < 1e-6
    assert ((red3 - 3 * ref).abs() / scale).max().item() < 3e-6


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(1, 4, 12, 20, 32), (2, 4, 9, 16, 64), (1, 4, 5, 7, 32), (1, 4, 23, 16, 32), (1, 4, 64, 64, 64)])
def test_k7_gate_conv_mfma_vs_vector_vs_stock(shape, dtype):
    """AttenModule2's composed 7^3 conv (4 pooled channels -> 2 sigmoid gates) and its data gradient (2 -> 4) on the
    Toeplitz-in-H MFMA kernel, against the vector kernel and stock fp32 ops on the same bf16-representable input."""
    torch.manual_seed(3)
    x = torch.randn(shape).to(dtype)
    w = torch.randn(2, 4, 7, 7, 7) * 0.05
    b = torch.randn(2) * 0.1
    g = torch.randn((shape[0], 2) + shape[2:])

    def run(mfma):
        X.ops.set_mfma(mfma)
        try:
            xg = x.to(DEV).requires_grad_(True)
            wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
            y = X.functional.conv(xg, [wg], [bg], act=X.ops.ACT_SIGMOID)
            name = X.ops.last_conv_kernel()
            (y.float() * g.to(DEV)).sum().backward()
            torch.cuda.synchronize()
            return y.detach().float().cpu(), xg.grad.float().cpu(), wg.grad.cpu(), bg.grad.cpu(), name
        finally:
            X.ops.set_mfma(True)
    X._lib.load().xh_set_option(24, 3)                 # the input-stationary kernel on every volume (default: >= 2^20 voxels only)
    try:
        y1, dx1, dw1, db1, name1 = run(True)
    finally:
        X._lib.load().xh_set_option(24, 1)
    y0, dx0, dw0, db0, name0 = run(False)
    assert "conv7_as_kernel" in name1 and "conv7_" not in name0
    # the input-stationary kernel (round 5, default) against the output-stationary one it replaces (xh_set_option(24, 0)): the same
    # products, summed in another order
    X._lib.load().xh_set_option(24, 0)
    try:
        y2, dx2, _, _, name2 = run(True)
    finally:
        X._lib.load().xh_set_option(24, 1)
    assert "conv7_mfma_kernel" in name2
    tol2 = 4e-3 if dtype == torch.bfloat16 else 5e-4                 # one rounding of the 16-bit outputs
    assert l2_err(y1, y2) < tol2 and l2_err(dx1, dx2) < tol2, (l2_err(y1, y2), l2_err(dx1, dx2))
    xo, wo, bo = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = torch.sigmoid(torch.nn.functional.conv3d(xo, wo, bo, padding=3))
    (yo * g).sum().backward()
    e = dict(y_vs_stock=l2_err(y1, yo), y_vs_vector=l2_err(y1, y0), dx_vs_stock=l2_err(dx1, xo.grad), dx_vs_vector=l2_err(dx1, dx0),
             dw_vs_stock=l2_err(dw1, wo.grad), db_vs_stock=l2_err(db1, bo.grad))
    print(shape, dtype, {k: f"{v:.2e}" for k, v in e.items()})
    k = 1.0 if dtype == torch.bfloat16 else 0.2
    assert e["y_vs_stock"] < 8e-3 * k and e["y_vs_vector"] < 8e-3 * k, e
    assert e["dx_vs_stock"] < 2e-2 * k and e["dx_vs_vector"] < 2e-2 * k, e
    assert e["dw_vs_stock"] < 2e-2 * k and e["db_vs_stock"] < 2e-2 * k, e


# ---------------------------------------------------------------------------------------------------------------------
# Full-size (BASELINE config 2: 128^3) instances.  Volumes of >= 2^20 voxels select the 256-thread `big` instances of the
# k=3 forward / data-gradient / weight-gradient MFMA kernels; these are the launches that carry the headline number, so each
# is checked here on its own against stock fp32 conv3d (CPU) on the same 16-bit-rounded inputs -- forward with the fused
# InstanceNorm+LeakyReLU prologue and the output-moments epilogue (epi 2), data gradient with the norm-backward epilogue
# (epi 1, inside InLreluConv.backward), weight and bias gradients.
BIG = [
    dict(cin=4, cout=4, groups=1),                 # decoder level 0 second conv, skr encoder; gemm path: conv3_mfma_kernel<F, 4, 256, 32, 8>
    dict(cin=12, cout=4, groups=1, split=4),       # <F, 12, 256, 32, 8>: decoder level 0 first conv on the virtual concat
    dict(cin=16, cout=16, groups=4),               # <F, 16, 256, 32, 8>: the four modality encoders, level 0
]


@pytest.mark.parametrize("path", ["q4", "gemm"])
@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", BIG, ids=lambda c: f"{c['cin']}to{c['cout']}g{c['groups']}")
def test_mfma_conv_full_size_128_vs_stock(cfg, dtype, path):
    """path q4: the quad-channel W-Toeplitz kernel that carries these shapes in production; path gemm: the plain implicit-GEMM
    `big` instances (quad-channel path switched off), which remain the kernels for denser groups at this volume class."""
    X._lib.load().xh_set_option(2, 0 if path == "q4" else 16 | 32)     # bit 4: forward / data gradient, bit 5: weight gradient
    try:
        _full_size_128(cfg, dtype, path)
    finally:
        X._lib.load().xh_set_option(2, 0)


def _full_size_128(cfg, dtype, path):
    torch.manual_seed(21)
    torch.set_num_threads(min(32, torch.get_num_threads() * 4, __import__("os").cpu_count() or 1))
    S = 128
    n, cin, cout, g = 1, cfg["cin"], cfg["cout"], cfg["groups"]
    x = (torch.randn(n, cin, S, S, S) * 1.5 + 0.3).to(dtype)
    nw = g if g <= 4 else 1
    ws = [torch.randn(cout // nw, cin // g, 3, 3, 3) * (2.0 / (27 * cin // g)) ** 0.5 for _ in range(nw)]
    bs = [torch.randn(cout // nw) for _ in range(nw)]
    wgt = torch.randn(n, cout, S, S, S)
    xg = x.to(DEV).requires_grad_(True)
    wg = [w.to(DEV).requires_grad_(True) for w in ws]
    bg = [b.to(DEV).requires_grad_(True) for b in bs]
    xa, xb = (xg[:, :cfg["split"]], xg[:, cfg["split"]:]) if "split" in cfg else (xg, None)
    y, red = X.functional.in_lrelu_conv(xa, xb, wg, bg, 1, g, out_stats=True)
    k_fwd = X.ops.last_conv_kernel()
    if path == "q4":
        big_ok = lambda k: "conv3_q4w_kernel" in k            # rows of 128 voxels: the full-row variant of the quad-channel kernel
    else:
        big_ok = lambda k: "conv3_mfma_kernel" in k and ", 256, 32, 8, 2>" in k
    assert big_ok(k_fwd), k_fwd
    (y.float() * wgt.to(DEV)).sum().backward()
    k_bwd = X.ops.last_conv_kernel()              # the data gradient is the last conv launch of InLreluConv.backward
    assert big_ok(k_bwd), k_bwd
    torch.cuda.synchronize()
    xo = x.float().requires_grad_(True)
    wo = [w.clone().requires_grad_(True) for w in ws]
    bo = [b.clone().requires_grad_(True) for b in bs]
    h = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(xo, eps=1e-5), 0.01)
    yo = torch.nn.functional.conv3d(h, torch.cat(wo, 0), torch.cat(bo, 0), padding=1, groups=g)
    (yo * wgt).sum().backward()
    t = TOL[dtype]
    e = dict(y=l2_err(y, yo), dx=l2_err(xg.grad, xo.grad))
    # epilogue 2: channel sums of the STORED output (what the next InstanceNorm will see)
    ys = y.detach().double().cpu()
    s0, s1 = ys.sum((2, 3, 4)), (ys * ys).sum((2, 3, 4))
    e["sum"] = ((red[..., 0].cpu() - s0).abs() / ys.abs().sum((2, 3, 4))).max().item()
    e["sumsq"] = ((red[..., 1].cpu() - s1).abs() / s1).max().item()
    print(cfg, dtype, k_fwd, {k: f"{v:.2e}" for k, v in e.items()})
    assert e["y"] < t["y"] and e["dx"] < t["dx"], e
    assert e["sum"] < 1e-6 and e["sumsq"] < 1e-6, e      # quads are summed in fp32, everything above them in fp64
    gmax = max(w.grad.abs().max() for w in wo)
    for a, b in zip([w.grad.cpu() for w in wg] + [b.grad.cpu() for b in bg], [w.grad for w in wo] + [b.grad for b in bo]):
        assert l2_err(a, b) < t["dw"] or (a - b).abs().max() < t["dw"] * 0.7 * gmax, (l2_err(a, b), a.shape)


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
def test_wgrad_mfma_full_size_128_kernel_name(dtype):
    """The weight gradient of a few-channel conv at 128^3 goes through the full-row kernel conv3_wgrad_q5_multi_kernel (round 5;
    conv3_wgrad_q4_multi_kernel with xh_set_option(21, 0)); with the quad-channel
    kernels switched off it is conv3_wgrad_mfma_kernel<F, 4, 256> (both checked numerically by the test above through
    InLreluConv.backward); this pins the selection so a plan change cannot silently drop the coverage."""
    x = torch.randn(1, 4, 128, 128, 128, device=DEV).to(dtype)
    dy = torch.randn(1, 4, 128, 128, 128, device=DEV).to(dtype)
    dw, db = torch.zeros(4, 4, 3, 3, 3, device=DEV), torch.zeros(4, device=DEV)
    X.ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3)
    name = X.ops.last_conv_kernel()
    assert "conv3_wgrad_q5_multi_kernel" in name, name
    X._lib.load().xh_set_option(21, 0)
    try:
        X.ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3)
        assert "conv3_wgrad_q4_multi_kernel" in X.ops.last_conv_kernel()
    finally:
        X._lib.load().xh_set_option(21, 1)
    X._lib.load().xh_set_option(2, 32)
    try:
        X.ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3)
        name = X.ops.last_conv_kernel()
    finally:
        X._lib.load().xh_set_option(2, 0)
    assert "conv3_wgrad_mfma_kernel" in name and name.endswith(", 4, 256>"), name


@pytest.mark.parametrize("w", [16, 8])
def test_wgrad_mfma_narrow_volume_kernel_name(w):
    """16- and 8-wide volumes (levels 3 and 4 of a 128^3 patch) take the MFMA weight-gradient kernel too (2 / 4 rows per
    K step); checked numerically by test_mfma_conv_forward_backward."""
    x = torch.randn(1, 16, 8, w, w, device=DEV).bfloat16()
    dy = torch.randn(1, 16, 8, w, w, device=DEV).bfloat16()
    dw, db = torch.zeros(16, 16, 3, 3, 3, device=DEV), torch.zeros(16, device=DEV)
    X.ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3)
    name = X.ops.last_conv_kernel()
    assert "conv3_wgrad_mfma_kernel" in name, name
    ref = torch.nn.grad.conv3d_weight(x.float(), dw.shape, dy.float(), padding=1)
    assert l2_err(dw.cpu(), ref.cpu()) < 2e-3
    assert l2_err(db.cpu(), dy.float().sum((0, 2, 3, 4)).cpu()) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", [dict(n=2, cin=4, cout=4, g=1, sp=(9, 6, 32)), dict(n=1, cin=12, cout=8, g=1, sp=(17, 11, 64)),
                                 dict(n=1, cin=16, cout=32, g=4, sp=(8, 8, 32)), dict(n=2, cin=24, cout=8, g=2, sp=(5, 9, 32)),
                                 dict(n=1, cin=24, cout=16, g=1, sp=(9, 8, 32)), dict(n=1, cin=16, cout=48, g=1, sp=(8, 5, 64)),
                                 dict(n=1, cin=20, cout=4, g=1, sp=(8, 8, 32))],
                         ids=["4to4", "12to8", "16to32g4", "24to8g2", "24to16_chunks3", "16to48_chunks2", "20to4_chunks1"])
def test_wgrad_quad_channel_kernel_vs_stock(cfg, dtype):
    """conv3_wgrad_q4_multi_kernel on its own (raw x, no input transform; then with the InstanceNorm + LeakyReLU transform):
    weight and bias gradients against torch.nn.grad on the same 16-bit inputs; ragged D / H, several W tiles, batch 2."""
    torch.manual_seed(31)
    n, cin, cout, g = cfg["n"], cfg["cin"], cfg["cout"], cfg["g"]
    x = torch.randn((n, cin) + cfg["sp"], device=DEV).to(dtype)
    dy = torch.randn((n, cout) + cfg["sp"], device=DEV).to(dtype)
    nw = g if g <= 4 else 1
    for pre in (None, (torch.rand(n, cin, device=DEV) + 0.5, torch.randn(n, cin, device=DEV), 0.01)):
        dws = [torch.zeros(cout // nw, cin // g, 3, 3, 3, device=DEV) for _ in range(nw)]
        dbs = [torch.zeros(cout // nw, device=DEV) for _ in range(nw)]
        X._lib.load().xh_set_option(21, 0)             # H % 8 == 0 cases would take the full-row kernel (tested below)
        try:
            X.ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=pre)
        finally:
            X._lib.load().xh_set_option(21, 1)
        assert "conv3_wgrad_q4_multi_kernel" in X.ops.last_conv_kernel()
        xf = x.float()
        if pre is not None:
            xf = torch.nn.functional.leaky_relu(xf * pre[0][:, :, None, None, None] + pre[1][:, :, None, None, None], 0.01)
            xf = xf.to(dtype).float()                  # the kernel rounds the transformed operand to the storage format
        ref = torch.nn.grad.conv3d_weight(xf, (cout, cin // g, 3, 3, 3), dy.float(), padding=1, groups=g)
        tol = 2e-3 if dtype == torch.bfloat16 else 5e-4
        assert l2_err(torch.cat(dws, 0).cpu(), ref.cpu()) < tol
        assert l2_err(torch.cat(dbs, 0).cpu(), dy.float().sum((0, 2, 3, 4)).cpu()) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
def test_k7_gate_conv_full_size_128_vs_stock(dtype):
    """conv7_mfma_kernel<F,4,2> (forward), <F,2,4> (data gradient) and conv7_wgrad_mfma_kernel<F> at 128^3 against stock fp32
    conv3d on the CPU (11.5 GFLOP: a few seconds)."""
    torch.manual_seed(23)
    torch.set_num_threads(min(32, __import__("os").cpu_count() or 1))
    S = 128
    x = torch.randn(1, 4, S, S, S).to(dtype)
    w = torch.randn(2, 4, 7, 7, 7) * 0.05
    b = torch.randn(2) * 0.1
    g = torch.randn(1, 2, S, S, S)
    xg = x.to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = X.functional.conv(xg, [wg], [bg], act=X.ops.ACT_SIGMOID)
    assert "conv7_as_kernel" in X.ops.last_conv_kernel()
    (y.float() * g.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    xo, wo, bo = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = torch.sigmoid(torch.nn.functional.conv3d(xo, wo, bo, padding=3))
    (yo * g).sum().backward()
    e = dict(y=l2_err(y, yo), dx=l2_err(xg.grad, xo.grad), dw=l2_err(wg.grad, wo.grad), db=l2_err(bg.grad, bo.grad))
    print("k7 128^3", dtype, {k: f"{v:.2e}" for k, v in e.items()})
    k = 1.0 if dtype == torch.bfloat16 else 0.2
    assert e["y"] < 8e-3 * k and e["dx"] < 2e-2 * k and e["dw"] < 2e-2 * k and e["db"] < 2e-2 * k, e


NB_CASES = [
    dict(cin=4, mid=4, cout=4, groups=1, sp=(32, 32, 32)),            # decoder level 0 shapes, small launch: 2 / 4-plane tiles
    dict(cin=16, mid=16, cout=16, groups=4, sp=(16, 24, 64)),         # the four encoder streams: quad per group, two W tiles
    dict(cin=12, mid=4, cout=4, groups=1, sp=(11, 13, 32), split=4),  # virtual concat in front, ragged D / H tiles (ownership)
    dict(cin=24, mid=8, cout=8, groups=1, sp=(8, 16, 32), split=8),   # two input quads through the same LDS tile (MULTI)
    dict(cin=16, mid=32, cout=32, groups=4, sp=(8, 8, 32)),           # 4 -> 8 -> 8 per stream: two quads per group
    dict(cin=4, mid=4, cout=4, groups=1, sp=(64, 64, 128), n=1),      # >= 512 workgroups: the 8-plane instance
    dict(cin=16, mid=16, cout=16, groups=4, sp=(8, 8, 16)),           # 16-wide rows: not on the quad-channel kernel -> the pass runs
]


@pytest.mark.parametrize("dtype", DTYPES + [torch.float32], ids=["bf16", "f16", "f32"])
@pytest.mark.parametrize("cfg", NB_CASES)
def test_norm_backward_folded_into_the_data_gradient_vs_separate_pass(cfg, dtype):
    """DoubleConv backward (buildingblocks.py:464-507): the InstanceNorm backward between the two data gradients applied by the
    second one on load (xh_conv_desc.pre == 2, functional._NB_PENDING) against the separate xh_in_bwd_apply pass: input and
    parameter gradients agree to the rounding of one storage value (both paths round the same fp32 expression once; fp32 storage
    and shapes the quad-channel kernel does not take run the pass either way and must agree exactly)."""
    from xlstm_hved_amd import functional as Fn
    torch.manual_seed(13)
    n, cin, mid, cout, g = cfg.get("n", 2), cfg["cin"], cfg["mid"], cfg["cout"], cfg["groups"]
    split = cfg.get("split")
    x = (torch.randn((n, cin) + cfg["sp"]) * 1.5 + 0.3).to(dtype)
    mk = lambda co, ci: ([torch.randn(co // g, ci // g, 3, 3, 3) * (2.0 / (27 * ci // g)) ** 0.5 for _ in range(g)], [torch.randn(co // g) for _ in range(g)])
    (w1, b1), (w2, b2) = mk(mid, cin), mk(cout, mid)
    wgt = torch.randn((n, cout) + cfg["sp"])

    def run(fold):
        X.ops.set_norm_bwd_fold(fold)
        try:
            xg = x.to(DEV).requires_grad_(True)
            ps = [[t.to(DEV).requires_grad_(True) for t in l] for l in (w1, b1, w2, b2)]
            xa, xb = (xg[:, :split], xg[:, split:]) if split else (xg, None)
            y1, st1 = Fn.in_lrelu_conv(xa, xb, ps[0], ps[1], 1, g, out_stats=True, drop_bias=True)
            y2 = Fn.in_lrelu_conv(y1, None, ps[2], ps[3], 1, g, in_stats=st1, sole_consumer=True)
            (y2.float() * wgt.to(DEV)).sum().backward()
            X.ops.join_wgrad_stream()                   # raises if a hand-over was left untaken
            torch.cuda.synchronize()
            return y2.detach(), xg.grad, [t.grad for l in ps for t in l], X.ops.last_conv_kernel()
        finally:
            X.ops.set_norm_bwd_fold(True)
    ya, dxa, ga, _ = run(False)
    yb, dxb, gb, kern = run(True)
    assert torch.equal(ya, yb)
    exact = dtype == torch.float32 or cfg["sp"][2] % 32 != 0
    tol = 2e-6 if exact else (6e-3 if dtype == torch.bfloat16 else 8e-4)      # (exact: up to the order of the fp64 / fp32 atomics)
    e_dx = l2_err(dxb, dxa)
    gscale = max(t.abs().max().item() for t in ga)
    e_g = max((p - q).abs().max().item() for p, q in zip(gb, ga)) / gscale
    print(f"norm-backward fold {cfg} {dtype}: dx rel-L2 {e_dx:.2e}, parameter gradients {e_g:.2e} of the largest")
    assert e_dx <= tol and e_g <= tol, (e_dx, e_g)


@pytest.mark.parametrize("n", [1, 2], ids=["n1_bn_folded", "n2"])
@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
@pytest.mark.parametrize("c,sp", [(4, (16, 16, 32)), (8, (8, 16, 32)), (16, (8, 8, 32))])
def test_skip_return_attention_with_composed_depthwise_pointwise_convs(c, sp, dtype, n):
    """ResBlock(lkdw=True) of the skip-return attention (sa_modules/sa_module.py:79-137): each DWConvNorm's depthwise 3^3 conv and
    pointwise 1x1 conv applied as ONE dense 3^3 conv with weights pw o dw (Fn.ComposeAll sep jobs; gradients scattered back to the
    two parameters at the end of the backward pass) against the two-conv form in fp32: attention map, input gradient and all
    parameter gradients within the storage format's band -- and closer than the two-conv form in the same format (one rounding
    of the intermediate tensor less)."""
    from xlstm_hved_amd import functional as Fn
    from xlstm_hved_amd.blocks import SkipReturnAttention
    torch.manual_seed(17)
    x = torch.randn((n, c) + sp)              # n == 1: the two BatchNorm finalisations ride inside the second conv / the tail pass
    wgt = torch.randn((n, 1) + sp)
    mod0 = SkipReturnAttention(c)
    mod0.apply(X.init_weights)
    sd = {k: v.clone() for k, v in mod0.state_dict().items()}

    def run(dt, composed):
        m = SkipReturnAttention(c)
        m.load_state_dict(sd)
        m = m.to(DEV).train()
        xg = x.to(DEV, dt).requires_grad_(True)
        if composed:
            outs = Fn.ComposeAll.apply(([], [], False, (c, c)), *m.compose_params())
            m.__dict__["_pre"] = (outs[0], outs[1])
        a = m(xg, steps=1)
        (a.float() * wgt.to(DEV)).sum().backward()
        X.ops.join_wgrad_stream()
        torch.cuda.synchronize()
        bufs = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items() if "running_" in k}
        return a.detach().float(), xg.grad.float(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, bufs
    (a0, dx0, g0, b0), (a1, dx1, g1, b1), (a2, dx2, g2, b2) = run(torch.float32, False), run(dtype, False), run(dtype, True)
    bdev = max((b2[k] - b0[k]).abs().max().item() for k in b0)
    assert bdev <= (2e-2 if dtype == torch.bfloat16 else 3e-3), bdev          # running statistics (16-bit t1 / t2 vs fp32)
    assert g0.keys() == g2.keys(), (sorted(g0), sorted(g2))
    gs = max(v.abs().max().item() for v in g0.values())
    e1 = (l2_err(a1, a0), l2_err(dx1, dx0), max((g1[k] - g0[k]).abs().max().item() for k in g0) / gs)
    e2 = (l2_err(a2, a0), l2_err(dx2, dx0), max((g2[k] - g0[k]).abs().max().item() for k in g0) / gs)
    print(f"skip-return attention c={c} {dtype}: two convs (a, dx, params) {e1[0]:.2e} {e1[1]:.2e} {e1[2]:.2e}; composed {e2[0]:.2e} {e2[1]:.2e} {e2[2]:.2e}")
    # bands: the format's rounding through two BatchNorm backward passes (measured c = 4, bf16: two convs 1.7e-3 / 0.106 / 0.171,
    # composed 1.6e-3 / 0.090 / 0.143), and never worse than the two-conv form by more than noise
    band = (2e-2, 0.25, 0.35) if dtype == torch.bfloat16 else (3e-3, 0.1, 0.15)
    assert all(a_ <= b_ for a_, b_ in zip(e2, band)), (e2, band)
    assert all(e2[i] <= 1.25 * e1[i] + 5e-3 for i in range(3)), (e1, e2)


Q4S_CASES = [
    dict(cin=4, cout=4, groups=1, sp=(9, 11, 32)),             # single quad, ragged D / H tiles
    dict(cin=16, cout=16, groups=4, sp=(8, 16, 64)),           # the four encoder streams, two W tiles
    dict(cin=12, cout=4, groups=1, sp=(8, 8, 32), split=4),    # decoder conv on a virtual concat: three input quads
    dict(cin=4, cout=12, groups=1, sp=(6, 8, 32)),             # three output quads
    dict(cin=20, cout=40, groups=5, sp=(4, 8, 32)),            # five groups (the skip stream riding along), 4 -> 8 each
    dict(cin=4, cout=4, groups=1, sp=(32, 64, 128), n=1),      # a launch that takes the 4-plane tiles
]


@pytest.mark.parametrize("cfg", Q4S_CASES)
def test_fp32_storage_through_split_fp16_matrix_cores_vs_fp32_vector_kernels(cfg):
    """ops.set_fp32_mfma(True): fp32 tensors through conv3_q4s_kernel (x = hi + 2^-11 lo in fp16, three MFMA products per fp32
    product, fp32 accumulation) against the fp32 FMA kernels on the same inputs: SingleConv 'ilc' forward with its fused
    InstanceNorm + LeakyReLU and output moments, and the data gradient with the norm-backward epilogue.  ~22 significand bits per
    product: relative L2 at the 1e-6 level (fp32 round-off is 6e-8; fp16 operands alone would read 3e-4).  The weight / bias
    gradients of the mode come from conv3_wgrad_q4_multi_kernel<2, ...> / conv3_wgrad_q5_multi_kernel<2>: fp32 loads, operands rounded ONCE to fp16, fp32
    accumulation -- a sum over every voxel of independently rounded products, so the 3e-4 per-product error averages down."""
    from xlstm_hved_amd import functional as Fn
    torch.manual_seed(19)
    n, cin, cout, g = cfg.get("n", 2), cfg["cin"], cfg["cout"], cfg["groups"]
    split = cfg.get("split")
    x = torch.randn((n, cin) + cfg["sp"]) * 1.5 + 0.3
    nw = g
    ws = [torch.randn(cout // nw, cin // g, 3, 3, 3) * (2.0 / (27 * cin // g)) ** 0.5 for _ in range(nw)]
    bs = [torch.randn(cout // nw) for _ in range(nw)]
    wgt = torch.randn((n, cout) + cfg["sp"])

    def run(on):
        X.ops.set_fp32_mfma(on)
        try:
            xg = x.to(DEV).requires_grad_(True)
            wg = [w.to(DEV).requires_grad_(True) for w in ws]
            bg = [b.to(DEV).requires_grad_(True) for b in bs]
            xa, xb = (xg[:, :split], xg[:, split:]) if split else (xg, None)
            y, st = Fn.in_lrelu_conv(xa, xb, wg, bg, 1, g, out_stats=True)
            (y * wgt.to(DEV)).sum().backward()
            X.ops.join_wgrad_stream()
            torch.cuda.synchronize()
            out = (y.detach().cpu(), st.detach().cpu().clone(), xg.grad.cpu(), [w.grad.cpu() for w in wg] + [torch.cat([b.grad.cpu() for b in bg])])
            # which kernel a weight gradient of this shape and dtype takes (the backward above ends on its data gradient)
            scr = [torch.zeros_like(w) for w in wg]
            X.ops.conv3d_wgrad(xg.detach()[:, :split] if split else xg.detach(), xg.detach()[:, split:] if split else None, wgt.to(DEV), scr, None,
                               k=3, groups=g)
            return out + (X.ops.last_conv_kernel(),)
        finally:
            X.ops.set_fp32_mfma(False)
    y0, s0, dx0, dw0, _ = run(False)
    y1, s1, dx1, dw1, kern = run(True)
    e = dict(y=l2_err(y1, y0), dx=l2_err(dx1, dx0), st=l2_err(s1, s0), dw=max(l2_err(a_, b_) for a_, b_ in zip(dw1, dw0)))
    print(cfg, {k: f"{v:.2e}" for k, v in e.items()}, kern)
    assert "wgrad_q4_multi_kernel<2" in kern or "wgrad_q5_multi_kernel<2" in kern, kern   # (full-row kernel: H % 8 == 0 shapes, round 6)
    assert e["y"] < 3e-6 and e["dx"] < 6e-6 and e["st"] < 1e-6 and e["dw"] < 4e-4, e


Q4P_CASES = [
    dict(cin=16, cout=16, groups=4, sp=(24, 40, 64), n=1),            # four streams: runs of tiles cross output quads
    dict(cin=12, cout=4, groups=1, sp=(20, 24, 32), n=2, split=4),    # decoder conv on a virtual concat: three input quads per tile, 2 samples
    dict(cin=4, cout=12, groups=1, sp=(9, 17, 32), n=1),              # ragged D and H tiles
    dict(cin=8, cout=8, groups=1, sp=(16, 16, 64), n=1),              # two input quads, two output quads
    dict(cin=4, cout=4, groups=1, sp=(64, 64, 128), n=1),             # 1 024 tiles on the chip
]


@pytest.mark.parametrize("occ3", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", Q4P_CASES)
def test_persistent_pipelined_q4_kernel_equals_the_per_tile_kernel(cfg, dtype, occ3):
    """conv3_q4p_kernel (resident workgroups walking runs of tiles, the next stage's loads in flight under the matrix phase and
    epilogue of the current one; xh_set_option(19, 2) forces it; two or three workgroups per CU) against conv3_q4_kernel on the same operands: forward with the
    producer's norm + LeakyReLU from explicit scale / shift AND from raw sums (fused finalisation), output moments; data gradient
    with the leaky'-masked norm-backward sums; plain.  Same products in the same order: the outputs must agree BIT FOR BIT; the
    fp64 statistics differ in summation order only."""
    lib = X._lib.load()
    torch.manual_seed(23)
    n, cin, cout, g = cfg["n"], cfg["cin"], cfg["cout"], cfg["groups"]
    split = cfg.get("split")
    x = (torch.randn((n, cin) + cfg["sp"]) * 1.3 + 0.2).to(DEV, dtype)
    dy = torch.randn((n, cout) + cfg["sp"]).to(DEV, dtype)
    ws = [(torch.randn(cout // g, cin // g, 3, 3, 3) * (2.0 / (27 * cin // g)) ** 0.5).to(DEV) for _ in range(g)]
    bs = [torch.randn(cout // g).to(DEV) for _ in range(g)]
    sc, sh = (torch.rand(n, cin) + 0.5).to(DEV), torch.randn(n, cin).to(DEV)
    cnt = cfg["sp"][0] * cfg["sp"][1] * cfg["sp"][2]
    xf = x.float()
    raw = torch.stack([xf.sum((2, 3, 4)), (xf * xf).sum((2, 3, 4))], -1).double().contiguous()
    xa, xb = (x[:, :split], x[:, split:]) if split else (x, None)

    def run(mode):
        X._lib.check(lib.xh_set_option(19, mode), "xh_set_option")
        X._lib.check(lib.xh_set_option(17, 0), "xh_set_option")          # 8 output planes per tile whatever the launch size
        X._lib.check(lib.xh_set_option(20, 0), "xh_set_option")          # (rows of 64 / 128 voxels would take the full-row kernel)
        X._lib.check(lib.xh_set_option(1, 32768 if (occ3 and mode) else 0), "xh_set_option")
        try:
            out = {}
            red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
            out["fwd"] = X.ops.conv3d(xa, xb, ws, bs, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red)
            out["fwd_red"], out["k_fwd"] = red, X.ops.last_conv_kernel()
            red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
            y, fsc, fsh, fm, fr = X.ops.conv3d(xa, xb, ws, bs, k=3, cout=cout, groups=g, in_stats=(raw, cnt, 0.01), epi=2, red=red)
            out["fin"], out["fin_red"], out["fin_sc"], out["fin_sh"], out["fin_m"], out["fin_r"] = y, red, fsc, fsh, fm, fr
            out["k_fin"] = X.ops.last_conv_kernel()
            out["plain"] = X.ops.conv3d(xa, xb, ws, bs, k=3, cout=cout, groups=g)
            red = torch.zeros(n, cin, 2, dtype=torch.float64, device=DEV)
            out["dgrad"] = X.ops.conv3d(dy, None, ws, None, k=3, cout=cin, groups=g, transposed=True, epi=1,
                                        e=(xa, xb, sc, sh, 0.01), red=red)
            out["dgrad_red"], out["k_dgrad"] = red, X.ops.last_conv_kernel()
            torch.cuda.synchronize()
            return out
        finally:
            lib.xh_set_option(19, 1)
            lib.xh_set_option(17, 512)
            lib.xh_set_option(20, 3)
            lib.xh_set_option(1, 0)
    a_, b_ = run(0), run(2)
    assert "conv3_q4_kernel" in a_["k_fwd"] and "conv3_q4p_kernel" in b_["k_fwd"] and "conv3_q4p_kernel" in b_["k_dgrad"], (a_["k_fwd"], b_["k_fwd"])
    for k_ in ("fwd", "fin", "plain", "dgrad", "fin_sc", "fin_sh", "fin_m", "fin_r"):
        assert torch.equal(a_[k_], b_[k_]), (k_, (a_[k_].float() - b_[k_].float()).abs().max().item(), a_["k_fin"], b_["k_fin"])
    for k_ in ("fwd_red", "fin_red", "dgrad_red"):
        d_ = (a_[k_] - b_[k_]).abs().max().item() / max(a_[k_].abs().max().item(), 1e-30)
        assert d_ < 1e-9, (k_, d_)


Q4W_CASES = [
    dict(cin=16, cout=16, groups=4, sp=(12, 24, 128), n=1),           # the four encoder streams
    dict(cin=12, cout=4, groups=1, sp=(9, 17, 128), n=2, split=4),    # virtual concat, three input quads, ragged D / H tiles, 2 samples
    dict(cin=4, cout=12, groups=1, sp=(8, 8, 128), n=1),              # one tile per plane quad
    dict(cin=8, cout=8, groups=1, sp=(16, 16, 128), n=1),             # two input quads, two output quads
    dict(cin=4, cout=4, groups=1, sp=(32, 40, 128), n=1),
    dict(cin=20, cout=40, groups=5, sp=(6, 8, 128), n=1),             # five groups, 4 -> 8 channels each
    dict(cin=20, cout=20, groups=5, sp=(16, 24, 64), n=1),            # rows of 64 voxels: one 64-voxel N tile per row
    dict(cin=24, cout=8, groups=1, sp=(10, 9, 64), n=2, split=8),     # 64-wide, six input quads, ragged tiles, 2 samples
    dict(cin=8, cout=24, groups=1, sp=(8, 16, 64), n=1),
    dict(cin=4, cout=4, groups=1, sp=(4, 8, 64), n=3),                # the smallest volume the kernel takes: one tile, 3 samples
    dict(cin=8, cout=4, groups=1, sp=(5, 9, 128), n=1),               # one plane and one row beyond a tile
    dict(cin=24, cout=8, groups=2, sp=(8, 16, 128), n=1),             # data gradient 4 -> 12 per group: workgroups of three output quads (round 6)
    dict(cin=12, cout=4, groups=1, sp=(8, 24, 64), n=2, split=8),     # the same on rows of 64 voxels, e from two sources, 2 samples
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", Q4W_CASES)
def test_full_row_q4_kernel_equals_the_32_wide_tile_kernel(cfg, dtype):
    """conv3_q4w_kernel (rows of 128 / 64 voxels: 4 x 8 x W tiles of 8 waves, no W halo, full-line loads; default) against
    conv3_q4_kernel (xh_set_option(20, 0)) on the same operands: forward with the producer's norm + LeakyReLU from explicit
    scale / shift and from raw sums (fused finalisation), output moments; data gradient with the leaky'-masked norm-backward sums;
    plain.  Same products in the same order per output value: the outputs must agree BIT FOR BIT; the statistics differ in the
    grouping of the fp32 lane partials (32 stored values each, other tile shape) and in fp64 summation order."""
    lib = X._lib.load()
    torch.manual_seed(29)
    n, cin, cout, g = cfg["n"], cfg["cin"], cfg["cout"], cfg["groups"]
    split = cfg.get("split")
    x = (torch.randn((n, cin) + cfg["sp"]) * 1.3 + 0.2).to(DEV, dtype)
    dy = torch.randn((n, cout) + cfg["sp"]).to(DEV, dtype)
    ws = [(torch.randn(cout // g, cin // g, 3, 3, 3) * (2.0 / (27 * cin // g)) ** 0.5).to(DEV) for _ in range(g)]
    bs = [torch.randn(cout // g).to(DEV) for _ in range(g)]
    sc, sh = (torch.rand(n, cin) + 0.5).to(DEV), torch.randn(n, cin).to(DEV)
    cnt = cfg["sp"][0] * cfg["sp"][1] * cfg["sp"][2]
    xf = x.float()
    raw = torch.stack([xf.sum((2, 3, 4)), (xf * xf).sum((2, 3, 4))], -1).double().contiguous()
    xa, xb = (x[:, :split], x[:, split:]) if split else (x, None)

    def run(wide):
        X._lib.check(lib.xh_set_option(20, wide), "xh_set_option")
        X._lib.check(lib.xh_set_option(19, 0), "xh_set_option")
        try:
            out = {}
            red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
            out["fwd"] = X.ops.conv3d(xa, xb, ws, bs, k=3, cout=cout, groups=g, pre=(sc, sh, 0.01), epi=2, red=red)
            out["fwd_red"], out["k_fwd"] = red, X.ops.last_conv_kernel()
            red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
            y, fsc, fsh, fm, fr = X.ops.conv3d(xa, xb, ws, bs, k=3, cout=cout, groups=g, in_stats=(raw, cnt, 0.01), epi=2, red=red)
            out["fin"], out["fin_red"], out["fin_sc"], out["fin_sh"], out["fin_m"], out["fin_r"] = y, red, fsc, fsh, fm, fr
            out["plain"] = X.ops.conv3d(xa, xb, ws, None, k=3, cout=cout, groups=g)
            red = torch.zeros(n, cin, 2, dtype=torch.float64, device=DEV)
            out["dgrad"] = X.ops.conv3d(dy, None, ws, None, k=3, cout=cin, groups=g, transposed=True, epi=1,
                                        e=(xa, xb, sc, sh, 0.01), red=red)
            out["dgrad_red"], out["k_dgrad"] = red, X.ops.last_conv_kernel()
            torch.cuda.synchronize()
            return out
        finally:
            lib.xh_set_option(20, 3)
            lib.xh_set_option(19, 1)
    a_, b_ = run(0), run(3)
    assert "conv3_q4_kernel" in a_["k_fwd"] and "conv3_q4w_kernel" in b_["k_fwd"] and "conv3_q4w_kernel" in b_["k_dgrad"], (a_["k_fwd"], b_["k_fwd"])
    for k_ in ("fwd", "fin", "plain", "dgrad", "fin_sc", "fin_sh", "fin_m", "fin_r"):
        assert torch.equal(a_[k_], b_[k_]), (k_, (a_[k_].float() - b_[k_].float()).abs().max().item())
    for k_ in ("fwd_red", "fin_red", "dgrad_red"):
        d_ = (a_[k_] - b_[k_]).abs().max().item() / max(a_[k_].abs().max().item(), 1e-30)
        assert d_ < 1e-6, (k_, d_)


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", [dict(n=2, cin=4, cout=4, g=1, sp=(9, 11, 64)), dict(n=1, cin=16, cout=16, g=4, sp=(8, 9, 128)),
                                 dict(n=1, cin=12, cout=4, g=1, sp=(10, 6, 128)), dict(n=1, cin=24, cout=8, g=1, sp=(6, 7, 64)),
                                 dict(n=1, cin=20, cout=40, g=5, sp=(5, 8, 64))],
                         ids=["4to4_w64", "16to16g4_w128", "12to4_w128", "24to8_w64_chunks2", "20to40g5_w64"])
def test_wgrad_quad_channel_wide_tiles_vs_narrow_tiles_and_stock(cfg, dtype):
    """Rows of 64 / 128 voxels: conv3_wgrad_q4_multi_kernel walks 2 x 64 tiles (full 128-byte lines; WgQ4.wide, the default there)
    instead of 4 x 32 (xh_set_option(1, 524288)).  Both against torch.nn.grad on the same 16-bit inputs and against each other (the
    same products, summed in another order: fp32 round-off); odd H, ragged depth segments, the norm + LeakyReLU input transform."""
    lib = X._lib.load()
    torch.manual_seed(37)
    n, cin, cout, g = cfg["n"], cfg["cin"], cfg["cout"], cfg["g"]
    x = torch.randn((n, cin) + cfg["sp"], device=DEV).to(dtype)
    dy = torch.randn((n, cout) + cfg["sp"], device=DEV).to(dtype)
    nw = g if g <= 4 else 1
    pre = (torch.rand(n, cin, device=DEV) + 0.5, torch.randn(n, cin, device=DEV), 0.01)
    res = {}
    for name, abl in (("wide", 0), ("narrow", 524288)):
        lib.xh_set_option(1, abl)
        lib.xh_set_option(21, 0)                       # the full-row kernel (conv3d_wgrad_q5.hip) would take the H % 8 == 0 cases
        try:
            dws = [torch.zeros(cout // nw, cin // g, 3, 3, 3, device=DEV) for _ in range(nw)]
            dbs = [torch.zeros(cout // nw, device=DEV) for _ in range(nw)]
            X.ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=pre)
            assert "conv3_wgrad_q4_multi_kernel" in X.ops.last_conv_kernel()
            torch.cuda.synchronize()
            res[name] = (torch.cat(dws, 0).cpu(), torch.cat(dbs, 0).cpu())
        finally:
            lib.xh_set_option(1, 0)
            lib.xh_set_option(21, 1)
    xf = torch.nn.functional.leaky_relu(x.float() * pre[0][:, :, None, None, None] + pre[1][:, :, None, None, None], 0.01).to(dtype).float()
    ref = torch.nn.grad.conv3d_weight(xf, (cout, cin // g, 3, 3, 3), dy.float(), padding=1, groups=g).cpu()
    tol = 2e-3 if dtype == torch.bfloat16 else 5e-4
    for name in res:
        assert l2_err(res[name][0], ref) < tol, (name, l2_err(res[name][0], ref))
        assert l2_err(res[name][1], dy.float().sum((0, 2, 3, 4)).cpu()) < 1e-4, name
    assert l2_err(res["wide"][0], res["narrow"][0]) < 2e-6 and l2_err(res["wide"][1], res["narrow"][1]) < 2e-6


@pytest.mark.parametrize("sp", [(9, 12, 32), (16, 16, 64)])
def test_k7_gate_weight_gradient_fp32_storage_on_the_matrix_cores(sp):
    """ops.set_fp32_mfma(True): the 7^3 gate conv's weight / bias gradient for fp32 tensors through conv7_wgrad_mfma_kernel<2> (fp32
    loads, operands rounded once to fp16, fp32 accumulation) against the fp32 FMA kernel on the same inputs.  A sum over every
    voxel of independently rounded products: relative L2 at the 1e-4 level (the parity mode's gradient band is 5e-3)."""
    torch.manual_seed(41)
    x = torch.randn((2, 4) + sp, device=DEV)
    dy = torch.randn((2, 2) + sp, device=DEV)

    def run(on):
        X.ops.set_fp32_mfma(on)
        try:
            dw, db = [torch.zeros(2, 4, 7, 7, 7, device=DEV)], [torch.zeros(2, device=DEV)]
            X.ops.conv3d_wgrad(x, None, dy, dw, db, k=7)
            torch.cuda.synchronize()
            return dw[0].cpu(), db[0].cpu(), X.ops.last_conv_kernel()
        finally:
            X.ops.set_fp32_mfma(False)
    w0, b0, k0 = run(False)
    w1, b1, k1 = run(True)
    assert "conv7_wgrad_mfma_kernel<2>" in k1 and "mfma" not in k0, (k0, k1)
    ref = torch.nn.grad.conv3d_weight(x.cpu(), (2, 4, 7, 7, 7), dy.cpu(), padding=3)
    e0, e1 = l2_err(w0, ref), l2_err(w1, ref)
    print(sp, f"fp32 FMA {e0:.2e}  fp16 operands {e1:.2e}  bias {l2_err(b1, b0):.2e}")
    assert e0 < 1e-5 and e1 < 1e-3 and l2_err(b1, b0) < 1e-3


@pytest.mark.parametrize("sp", [(9, 12, 32), (16, 16, 64), (23, 16, 32)])
def test_k7_gate_conv_fp32_storage_on_the_matrix_cores(sp):
    """ops.set_fp32_mfma(True): the composed 7^3 gate conv (4 -> 2 + sigmoid) and its data gradient (2 -> 4) for fp32 tensors through
    conv7_as_kernel<2, ...> (fp32 loads and stores, operands rounded ONCE to fp16 on their way into LDS, fp32 accumulation) against
    the fp32 FMA kernel and stock fp32 conv3d on the same inputs: an output is a sum of 1 372 products of independently rounded
    operands -- relative L2 at the 2e-4 level behind the sigmoid."""
    torch.manual_seed(43)
    x = torch.randn((2, 4) + sp, device=DEV)
    w = torch.randn(2, 4, 7, 7, 7, device=DEV) * 0.05
    b = torch.randn(2, device=DEV) * 0.1
    g = torch.randn((2, 2) + sp, device=DEV)

    def run(on):
        X.ops.set_fp32_mfma(on)
        try:
            y = X.ops.conv3d(x, None, [w], [b], k=7, cout=2, act=X.ops.ACT_SIGMOID)
            k1 = X.ops.last_conv_kernel()
            dx = X.ops.conv3d(g, None, [w], None, k=7, cout=4, transposed=True)
            k2 = X.ops.last_conv_kernel()
            torch.cuda.synchronize()
            return y.cpu(), dx.cpu(), k1, k2
        finally:
            X.ops.set_fp32_mfma(False)
    y0, dx0, k0a, k0b = run(False)
    y1, dx1, k1a, k1b = run(True)
    assert "conv7_as_kernel<2" in k1a and "conv7_as_kernel<2" in k1b and "conv7" not in k0a, (k0a, k1a, k1b)
    yo = torch.sigmoid(torch.nn.functional.conv3d(x.cpu(), w.cpu(), b.cpu(), padding=3))
    dxo = torch.nn.functional.conv_transpose3d(g.cpu(), w.cpu(), padding=3)
    e = (l2_err(y0, yo), l2_err(y1, yo), l2_err(dx0, dxo), l2_err(dx1, dxo))
    print(sp, "fp32 FMA y %.2e, fp16 operands y %.2e; dx %.2e / %.2e" % e)
    assert e[0] < 1e-5 and e[2] < 1e-5 and e[1] < 5e-4 and e[3] < 1e-3


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", [dict(n=2, cin=4, cout=4, g=1, sp=(9, 16, 64)), dict(n=1, cin=16, cout=16, g=4, sp=(21, 8, 128)),
                                 dict(n=1, cin=12, cout=4, g=1, sp=(10, 16, 128), split=4), dict(n=1, cin=24, cout=8, g=1, sp=(6, 8, 64), split=16),
                                 dict(n=2, cin=20, cout=40, g=5, sp=(5, 8, 64)), dict(n=1, cin=8, cout=8, g=8, sp=(12, 24, 128)),
                                 dict(n=1, cin=4, cout=12, g=1, sp=(4, 8, 128)), dict(n=2, cin=16, cout=32, g=4, sp=(8, 8, 32)),
                                 dict(n=1, cin=48, cout=16, g=1, sp=(32, 32, 32), split=32), dict(n=1, cin=16, cout=16, g=16, sp=(19, 16, 32)),
                                 dict(n=1, cin=24, cout=8, g=1, sp=(7, 16, 64), split=12), dict(n=2, cin=16, cout=16, g=2, sp=(5, 8, 64)),
                                 dict(n=1, cin=8, cout=4, g=1, sp=(9, 8, 64)), dict(n=1, cin=8, cout=8, g=8, sp=(6, 8, 64))],
                         ids=["4to4_w64_n2", "16to16g4_w128_ragged_segments", "12to4_w128_two_sources", "24to8_w64_two_sources", "20to40g5_w64_n2",
                              "depthwise8_w128", "4to12_w128_four_planes", "16to32g4_w32_n2", "48to16_w32_two_sources", "depthwise16_w32",
                              "24to8_w64_unit_across_the_sources", "16to16g2_w64_n2_two_by_two_quads", "8to4_w64_two_input_quads", "depthwise8_w64"])
def test_wgrad_full_row_kernel_vs_tile_kernel_and_stock(cfg, dtype):
    """conv3_wgrad_q5_multi_kernel (rows of 32 / 64 / 128 voxels, H a multiple of 8: 8-row full-row tiles, dY staged once, the kw shift
    applied at fragment-read time) against conv3_wgrad_q4_multi_kernel (xh_set_option(21, 0)) and torch.nn.grad on the same 16-bit
    inputs: the same products summed in another order (fp32 round-off between the two kernels); batch 2, groups, several input
    quads per group (units that re-stage dY), the skip | x two-source input, depthwise as groups of 4, depth segments that do not
    divide D, the InstanceNorm + LeakyReLU input transform and the bias gradient."""
    lib = X._lib.load()
    torch.manual_seed(41)
    n, cin, cout, g = cfg["n"], cfg["cin"], cfg["cout"], cfg["g"]
    x = torch.randn((n, cin) + cfg["sp"], device=DEV).to(dtype)
    dy = torch.randn((n, cout) + cfg["sp"], device=DEV).to(dtype)
    xa, xb = (x, None) if "split" not in cfg else (x[:, :cfg["split"]].contiguous(), x[:, cfg["split"]:].contiguous())
    nw = g if g <= 4 else 1
    for pre in (None, (torch.rand(n, cin, device=DEV) + 0.5, torch.randn(n, cin, device=DEV), 0.01)):
        res = {}
        for name, on in (("full", 1), ("tile", 0)):
            lib.xh_set_option(21, on)
            lib.xh_set_option(23, 1)                   # rows of 32 voxels too (the default since round 6: see conv3d_wgrad_q5.hip)
            try:
                dws = [torch.zeros(cout // nw, cin // g, 3, 3, 3, device=DEV) for _ in range(nw)]
                dbs = [torch.zeros(cout // nw, device=DEV) for _ in range(nw)]
                X.ops.conv3d_wgrad(xa, xb, dy, dws, dbs, k=3, groups=g, pre=pre)
                kn = X.ops.last_conv_kernel()
                assert ("conv3_wgrad_q5_multi_kernel" if on else "conv3_wgrad_q4_multi_kernel") in kn, kn
                torch.cuda.synchronize()
                res[name] = (torch.cat(dws, 0).cpu(), torch.cat(dbs, 0).cpu())
            finally:
                lib.xh_set_option(21, 1)
        xf = x.float()
        if pre is not None:
            xf = torch.nn.functional.leaky_relu(xf * pre[0][:, :, None, None, None] + pre[1][:, :, None, None, None], 0.01).to(dtype).float()
        ref = torch.nn.grad.conv3d_weight(xf, (cout, cin // g, 3, 3, 3), dy.float(), padding=1, groups=g).cpu()
        tol = 2e-3 if dtype == torch.bfloat16 else 5e-4
        assert l2_err(res["full"][0], ref) < tol, l2_err(res["full"][0], ref)
        assert l2_err(res["full"][1], dy.float().sum((0, 2, 3, 4)).cpu()) < 1e-4
        assert l2_err(res["full"][0], res["tile"][0]) < 2e-6 and l2_err(res["full"][1], res["tile"][1]) < 2e-6


@pytest.mark.parametrize("cfg", [(24, 8, 1, (12, 16, 64), 16), (16, 16, 2, (9, 8, 64), None), (20, 40, 5, (6, 8, 64), None), (8, 4, 1, (8, 24, 64), None),
                                 (12, 4, 1, (11, 16, 128), 4), (24, 8, 2, (5, 8, 128), None), (24, 4, 1, (4, 8, 128), 8),
                                 (48, 16, 1, (9, 16, 32), 32), (40, 80, 5, (6, 8, 32), None)],
                         ids=["24to8", "16to16g2", "20to40g5", "8to4", "12to4_w128_three_input_quads", "24to8g2_w128", "24to4_w128_two_chunks_of_three",
                              "48to16_w32", "40to80g5_w32"])
def test_wgrad_full_row_units_of_two_quads_equal_single_quad_units(cfg):
    """Rows of 64 and of 32 voxels (round 6): a unit of conv3_wgrad_q5_multi_kernel stages two input and / or two output quads once and multiplies
    every pair (xh_set_option(28, ...) bits 0 and 1) instead of one quad of each per unit (28, 0: dY re-staged per input quad, x per
    output quad); rows of 128 voxels: three input quads against one staging of dY, rounds of one plane (bit 2).  Same products, same
    order inside a pair: the plans agree to the order of the fp32 atomics."""
    lib = X._lib.load()
    torch.manual_seed(47)
    cin, cout, g, sp, split = cfg
    x = torch.randn((2, cin) + sp, device=DEV).bfloat16()
    dy = torch.randn((2, cout) + sp, device=DEV).bfloat16()
    xa, xb = (x, None) if split is None else (x[:, :split].contiguous(), x[:, split:].contiguous())
    pre = (torch.rand(2, cin, device=DEV) + 0.5, torch.randn(2, cin, device=DEV), 0.01)
    nw = g if g <= 4 else 1
    res = {}
    for uq in (7, 0, 1, 2, 4):
        lib.xh_set_option(28, uq)
        try:
            dws = [torch.zeros(cout // nw, cin // g, 3, 3, 3, device=DEV) for _ in range(nw)]
            dbs = [torch.zeros(cout // nw, device=DEV) for _ in range(nw)]
            X.ops.conv3d_wgrad(xa, xb, dy, dws, dbs, k=3, groups=g, pre=pre)
            assert "conv3_wgrad_q5_multi_kernel" in X.ops.last_conv_kernel()
            torch.cuda.synchronize()
            res[uq] = (torch.cat(dws, 0).cpu(), torch.cat(dbs, 0).cpu())
        finally:
            lib.xh_set_option(28, 7)
    for uq in (7, 1, 2, 4):
        assert l2_err(res[uq][0], res[0][0]) < 2e-6 and l2_err(res[uq][1], res[0][1]) < 2e-6, (uq, l2_err(res[uq][0], res[0][0]))


@pytest.mark.parametrize("cfg", [(12, 4, 1, (9, 16, 128), 4), (16, 16, 4, (6, 8, 128), None), (24, 8, 1, (7, 16, 64), 16), (20, 40, 5, (5, 8, 64), None),
                                 (48, 16, 1, (8, 16, 32), 32), (8, 8, 8, (10, 8, 64), None)],
                         ids=["12to4_w128", "16to16g4_w128", "24to8_w64", "20to40g5_w64", "48to16_w32", "depthwise8_w64"])
def test_wgrad_full_row_kernel_fp32_storage_equals_the_tile_kernel(cfg):
    """fp32 storage on the matrix cores (xh_conv_desc.arith XH_ARITH_F32_SPLIT): conv3_wgrad_q5_multi_kernel<2> (round 6: fp32 loads,
    operands rounded once to fp16 while staging, two rounds of loads in flight; OPT-IN, xh_set_option(28) bit 3 -- measured slower in
    the fp32_mfma step than the tile kernel, 695 against 592 us) against conv3_wgrad_q4_multi_kernel<2, ...> -- the same roundings
    and products, another summation order -- and against torch.nn.grad on the fp32 inputs (the fp16 operand rounding averages down
    over the voxels)."""
    lib = X._lib.load()
    torch.manual_seed(53)
    cin, cout, g, sp, split = cfg
    x = torch.randn((2, cin) + sp, device=DEV)
    dy = torch.randn((2, cout) + sp, device=DEV)
    xa, xb = (x, None) if split is None else (x[:, :split].contiguous(), x[:, split:].contiguous())
    pre = (torch.rand(2, cin, device=DEV) + 0.5, torch.randn(2, cin, device=DEV), 0.01)
    nw = g if g <= 4 else 1
    res = {}
    X.ops.set_fp32_mfma(True)
    try:
        for name, uq in (("full", 15), ("tile", 7)):
            lib.xh_set_option(28, uq)
            dws = [torch.zeros(cout // nw, cin // g, 3, 3, 3, device=DEV) for _ in range(nw)]
            dbs = [torch.zeros(cout // nw, device=DEV) for _ in range(nw)]
            X.ops.conv3d_wgrad(xa, xb, dy, dws, dbs, k=3, groups=g, pre=pre)
            kn = X.ops.last_conv_kernel()
            assert ("conv3_wgrad_q5_multi_kernel<2>" if uq == 15 else "conv3_wgrad_q4_multi_kernel<2") in kn, kn
            torch.cuda.synchronize()
            res[name] = (torch.cat(dws, 0).cpu(), torch.cat(dbs, 0).cpu())
    finally:
        lib.xh_set_option(28, 7)
        X.ops.set_fp32_mfma(False)
    xf = torch.nn.functional.leaky_relu(x * pre[0][:, :, None, None, None] + pre[1][:, :, None, None, None], 0.01)
    ref = torch.nn.grad.conv3d_weight(xf, (cout, cin // g, 3, 3, 3), dy, padding=1, groups=g).cpu()
    assert l2_err(res["full"][0], res["tile"][0]) < 2e-6 and l2_err(res["full"][1], res["tile"][1]) < 2e-6
    assert l2_err(res["full"][0], ref) < 5e-4 and l2_err(res["full"][1], dy.sum((0, 2, 3, 4)).cpu()) < 5e-4


@pytest.mark.parametrize("dtype", DTYPES, ids=["bf16", "f16"])
def test_wgrad_full_row_kernel_batched_128_cubed(dtype):
    """The end-of-backward batch at the benchmark's shapes: xh_conv3d_wgrad_batch with the 128^3 / 64^3 problems of one step (16 -> 16 g4,
    12 -> 4, 4 -> 4, 8 -> 8 g2 at 128^3; 24 -> 8, 8 -> 8 at 64^3) in ONE call -- several problems per launch, workgroups dealt by
    volume -- against the same problems through the tile kernel."""
    lib = X._lib.load()
    torch.manual_seed(43)
    probs = [(16, 16, 4, 128), (12, 4, 1, 128), (4, 4, 1, 128), (8, 8, 2, 128), (24, 8, 1, 64), (8, 8, 1, 64), (32, 32, 4, 64),
             (16, 16, 4, 128), (12, 4, 1, 128), (48, 16, 1, 32), (16, 16, 1, 32), (40, 80, 5, 32), (20, 40, 5, 64), (8, 8, 8, 64)]
    data = []
    for cin, cout, g, s in probs:
        x = torch.randn(1, cin, s, s, s, device=DEV).to(dtype)
        dy = (torch.randn(1, cout, s, s, s, device=DEV) * 0.1).to(dtype)
        pre = (torch.rand(1, cin, device=DEV) + 0.5, torch.randn(1, cin, device=DEV), 0.01)
        data.append((x, dy, pre, g, cin, cout))
    res = {}
    for name, on in (("full", 1), ("tile", 0)):
        lib.xh_set_option(21, on)
        X.ops.set_wgrad_defer(True)
        try:
            outs = []
            for x, dy, pre, g, cin, cout in data:
                nw = g if g <= 4 else 1
                dws = [torch.zeros(cout // nw, cin // g, 3, 3, 3, device=DEV) for _ in range(nw)]
                dbs = [torch.zeros(cout // nw, device=DEV) for _ in range(nw)]
                X.ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=g, pre=pre, side=True)
                outs.append((dws, dbs))
            X.ops.join_wgrad_stream()
            torch.cuda.synchronize()
            res[name] = [(torch.cat(a, 0).cpu(), torch.cat(b, 0).cpu()) for a, b in outs]
        finally:
            X.ops.set_wgrad_defer(False)
            lib.xh_set_option(21, 1)
    for (a, b), (c, d), pr in zip(res["full"], res["tile"], probs):
        assert l2_err(a, c) < 5e-6 and l2_err(b, d) < 5e-6, (pr, l2_err(a, c), l2_err(b, d))
        assert a.abs().max() > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("sp,ca,cb,cout", [((16, 16, 64), 8, 16, 8), ((8, 16, 128), 4, 8, 4), ((16, 16, 32), 16, 32, 16), ((12, 24, 64), 4, 0, 4)])
def test_two_convs_of_one_shape_in_one_launch_equal_two_launches(dtype, sp, ca, cb, cout):
    """xh_conv3d_fwd_pair (ops.conv_pair_scope): two independent 'ilc' convs of one shape -- the recon stream's on a virtual concat of
    two tensors, the seg stream's on one tensor, as in the decoders (buildingblocks.py:732-735) -- issued as ONE launch of the
    full-row kernel (rows of 64 / 128 voxels) or the tile kernel (rows of 32): the workgroups run the single launch's body at the
    single launch's block coordinates, so outputs are bit-identical and the epilogue sums equal to their summation order."""
    torch.manual_seed(13)
    n = 1
    cin = ca + cb
    xa = torch.randn((n, ca) + sp, device=DEV).to(dtype)
    xb = torch.randn((n, cb) + sp, device=DEV).to(dtype) if cb else None
    xs = torch.randn((n, cin) + sp, device=DEV).to(dtype)
    w1 = [torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.1]
    w2 = [torch.randn(cout, cin, 3, 3, 3, device=DEV) * 0.1]
    cnt = sp[0] * sp[1] * sp[2]

    def stats_of(t, u=None):
        red = torch.zeros((n, cin, 2), dtype=torch.float64, device=DEV)
        if u is not None:
            X.ops.moments2(t, u, red)
        else:
            X.ops.moments(t, red, 0)
        return red

    def run(pair):
        X.ops.set_conv_pairs(pair)
        try:
            X.ops.prepack_all()
            y = torch.zeros((n, 2 * cout) + sp, dtype=dtype, device=DEV)
            red = torch.zeros((n, 2 * cout, 2), dtype=torch.float64, device=DEV)
            r1, r2 = stats_of(xa, xb), stats_of(xs)
            with X.ops.conv_pair_scope():
                o1 = X.ops.conv3d(xa, xb, w1, None, k=3, cout=cout, in_stats=(r1, cnt, 0.01), epi=2, red=red[:, :cout], out=y[:, :cout])
                o2 = X.ops.conv3d(xs, None, w2, None, k=3, cout=cout, in_stats=(r2, cnt, 0.01), epi=2, red=red[:, cout:], out=y[:, cout:])
            torch.cuda.synchronize()
            return y.clone(), red.clone(), [t.clone() for t in o1[1:]], [t.clone() for t in o2[1:]], X.ops.last_conv_kernel()
        finally:
            X.ops.set_conv_pairs(True)
    run(True)                                                  # (registers the two convs: the pair needs prepacked fragments)
    y0, red0, s10, s20, _ = run(False)
    y1, red1, s11, s21, kern = run(True)
    assert "pair" in kern, kern
    assert torch.equal(y0, y1)
    assert (red0 - red1).abs().max().item() <= 1e-9 * red0.abs().max().item()
    for a_, b_ in zip(s10 + s20, s11 + s21):
        assert torch.equal(a_, b_)                             # (scale, shift, mean, rstd written by the fused finalisation)

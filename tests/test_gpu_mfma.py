"""-m gpu: the bf16-MFMA implicit-GEMM conv kernels against the vector kernels (same bf16 inputs) and stock ops."""
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
]


@pytest.mark.parametrize("cfg", CASES)
def test_mfma_conv_forward_backward(cfg):
    torch.manual_seed(11)
    n, cin, cout, g = 2, cfg["cin"], cfg["cout"], cfg["groups"]
    x = (torch.randn((n, cin) + cfg["sp"]) * 1.5 + 0.3).bfloat16()
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
    print(cfg, {k: f"{v:.2e}" for k, v in e.items()})
    assert e["y_vs_stock"] < 8e-3 and e["y_vs_vector"] < 8e-3, e
    assert e["dx_vs_stock"] < 5e-2 and e["dx_vs_vector"] < 5e-2, e
    gmax = max(w.grad.abs().max() for w in wo)
    for a, b in zip(dw1 + db1, [w.grad for w in wo] + [b.grad for b in bo]):
        assert l2_err(a, b) < 3e-2 or (a - b).abs().max() < 2e-2 * gmax


@pytest.mark.parametrize("shape", [(1, 4, 12, 20, 32), (2, 4, 9, 16, 64), (1, 4, 5, 7, 32)])
def test_k7_gate_conv_mfma_vs_vector_vs_stock(shape):
    """AttenModule2's composed 7^3 conv (4 pooled channels -> 2 sigmoid gates) and its data gradient (2 -> 4) on the
    Toeplitz-in-H MFMA kernel, against the vector kernel and stock fp32 ops on the same bf16-representable input."""
    torch.manual_seed(3)
    x = torch.randn(shape).bfloat16()
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
    y1, dx1, dw1, db1, name1 = run(True)
    y0, dx0, dw0, db0, name0 = run(False)
    assert "conv7_mfma_kernel" in name1 and "conv7_mfma_kernel" not in name0
    xo, wo, bo = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = torch.sigmoid(torch.nn.functional.conv3d(xo, wo, bo, padding=3))
    (yo * g).sum().backward()
    e = dict(y_vs_stock=l2_err(y1, yo), y_vs_vector=l2_err(y1, y0), dx_vs_stock=l2_err(dx1, xo.grad), dx_vs_vector=l2_err(dx1, dx0),
             dw_vs_stock=l2_err(dw1, wo.grad), db_vs_stock=l2_err(db1, bo.grad))
    print(shape, {k: f"{v:.2e}" for k, v in e.items()})
    assert e["y_vs_stock"] < 8e-3 and e["y_vs_vector"] < 8e-3, e
    assert e["dx_vs_stock"] < 2e-2 and e["dx_vs_vector"] < 2e-2, e
    assert e["dw_vs_stock"] < 2e-2 and e["db_vs_stock"] < 2e-2, e

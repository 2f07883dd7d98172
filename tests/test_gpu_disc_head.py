"""-m gpu: the discriminator's head as reduction kernels (xh_dlast_fwd / _dgrad / _wgrad, RA_HVED.py:223 `last` = Conv3d(512, 1, ks,
stride 1, padding 1, bias=False)) against stock fp32 conv3d on the same 16-bit values, and the whole Discriminator with the head
kernels on and off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402
from xlstm_hved_amd import disc as D  # noqa: E402
from xlstm_hved_amd import _lib as L  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("ks,sp,n", [(4, (15, 15, 15), 1), (3, (16, 16, 16), 2), (4, (6, 9, 5), 2)])
def test_head_kernels_vs_stock_conv3d(dtype, ks, sp, n):
    torch.manual_seed(3)
    C = 512
    x = torch.randn((n,) + sp + (C,), device=DEV).to(dtype)                       # channels-last
    w = (torch.randn(1, C, ks, ks, ks, device=DEV) * 0.02)
    so = tuple(s + 2 - ks + 1 for s in sp)
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    wp = D._pack(w, 0, 1, C, dtype)
    w16 = w.to(dtype).float()                                                     # the operand image rounds the weights once
    xr = x.float().permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    wr = w16.clone().requires_grad_(True)
    yr = F.conv3d(xr, wr, padding=1)
    y = torch.empty((n,) + so, dtype=dtype, device=DEV)
    L.check(lib.xh_dlast_fwd(st, X.ops._dt(x), ks, x.data_ptr(), wp.data_ptr(), y.data_ptr(), n, *sp, *so, C), "xh_dlast_fwd")
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert (y.float() - yr[:, 0]).abs().max().item() <= tol * yr.abs().max().item()
    dy = torch.randn((n,) + so, device=DEV).to(dtype)
    yr.backward(dy.float()[:, None])
    dx = torch.empty_like(x)
    L.check(lib.xh_dlast_dgrad(st, X.ops._dt(x), ks, dy.data_ptr(), wp.data_ptr(), dx.data_ptr(), n, *sp, *so, C), "xh_dlast_dgrad")
    dxr = xr.grad.permute(0, 2, 3, 4, 1)
    assert (dx.float() - dxr).abs().max().item() <= tol * dxr.abs().max().item()
    dw = torch.full_like(w, 0.25)                                                 # accumulated into
    L.check(lib.xh_dlast_wgrad(st, X.ops._dt(x), ks, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0.5, n, *sp, *so, C), "xh_dlast_wgrad")
    torch.cuda.synchronize()
    e = (dw - 0.25 - 0.5 * wr.grad).abs().max().item() / wr.grad.abs().max().item()
    assert e <= 1e-5, e                                                           # exact fp32 products, sums in another order


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_discriminator_with_head_kernels_equals_the_gemm_head(dtype):
    """Whole Discriminator forward + backward (ks = 4, 64^3 input): the head through xh_dlast_* against the 16-column GEMM tiles it
    replaces (disc.set_head_kernels(False)): same operands (the mode-0 / mode-1 images round the weights the same way), fp32
    accumulation in another order."""
    torch.manual_seed(5)
    m = X.Discriminator(7, ks=4).to(DEV)
    x = torch.randn(2, 7, 64, 64, 64, device=DEV).to(dtype)

    def run(on):
        D.set_head_kernels(on)
        try:
            for p in m.parameters():
                p.grad = None
            xin = x.clone().requires_grad_(True)
            out = m(xin)
            (out.float() * torch.linspace(-1, 1, out.numel(), device=DEV).view(out.shape)).sum().backward()
            torch.cuda.synchronize()
            return out.detach().float(), xin.grad.float(), {k: p.grad.clone() for k, p in m.named_parameters()}
        finally:
            D.set_head_kernels(True)
    o0, dx0, g0 = run(False)
    o1, dx1, g1 = run(True)
    tol = 2.0 ** -6 if dtype == torch.bfloat16 else 2.0 ** -9
    assert (o0 - o1).abs().max().item() <= tol * o0.abs().max().item()
    assert ((dx0 - dx1).norm() / dx0.norm()).item() <= 4 * tol
    for k in g0:
        assert ((g0[k] - g1[k]).norm() / g0[k].norm().clamp_min(1e-20)).item() <= 4 * tol, k

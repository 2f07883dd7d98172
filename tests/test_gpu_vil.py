"""-m gpu: the chunk-recurrent mLSTM (matrix-core contractions, carried C/n states) against
 (a) the recurrent CPU oracle in fp64 (oracle.mlstm_recurrent inside oracle.vil_layer) at S = 4096 and S = 32 768 tokens,
 (b) the tiled O(S^2) kernels it replaces (xh_set_option(2, 8) selects them) -- same inputs, forward and backward,
 (c) ragged sequence lengths (S not a multiple of the 64-token chunk, S < 64),
and U_HVEDConvXLSTMNet3D(f_maps=8, 'gcr') at 128^3, whose DoubleConv_ViL decoder puts ViL on 32 768 tokens
(buildingblocks.py:509-555; SURVEY f4: the dense form cannot run there)."""
import pytest
import torch

from gpu_common import l2_err, rel_err

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402
import xlstm_hved_oracle as O  # noqa: E402

DEV = "cuda"


def _layer(seed=0):
    torch.manual_seed(seed)
    m = X.ViLLayer(32)
    m.apply(X.init_weights)
    with torch.no_grad():                       # make the gates non-trivial: forget gates around sigmoid(+-2), varied inputs
        cell = m.vil.layer.mlstm_cell
        cell.fgate.bias.copy_(torch.linspace(-1.0, 3.0, 4))
        cell.igate.bias.copy_(torch.linspace(-2.0, 1.0, 4))
    return m


def _oracle(m, x, dtype=torch.float64, grad_w=None):
    sd = {"mViL." + k: v.detach().to(dtype).clone() for k, v in m.state_dict().items()}
    xo = x.to(dtype).clone().requires_grad_(grad_w is not None)
    y = O.vil_layer(O.P(sd).sub("mViL"), xo, recurrent=True)
    g = None
    if grad_w is not None:
        (y * grad_w.to(dtype)).sum().backward()
        g = xo.grad
    return y.detach(), g


def _run(m, x, grad_w=None, tiled=False):
    lib = X._lib.load()
    lib.xh_set_option(2, 8 if tiled else 0)
    try:
        mg = m.to(DEV)
        for p in mg.parameters():
            p.grad = None
        xg = x.to(DEV).requires_grad_(grad_w is not None)
        y = mg(xg)
        gx, gp = None, None
        if grad_w is not None:
            (y * grad_w.to(DEV)).sum().backward()
            gx = xg.grad.cpu()
            gp = {k: p.grad.cpu().clone() for k, p in mg.named_parameters() if p.grad is not None}
        torch.cuda.synchronize()
        return y.detach().cpu(), gx, gp
    finally:
        lib.xh_set_option(2, 0)


@pytest.mark.parametrize("shape", [(1, 32, 16, 16, 16), (2, 32, 8, 8, 12), (1, 32, 2, 3, 5), (1, 32, 4, 5, 7)],
                         ids=["S4096", "S768_B2", "S30", "S140"])
def test_chunk_mlstm_vs_recurrent_oracle_and_tiled_kernels(shape):
    m = _layer()
    torch.manual_seed(1)
    x = torch.randn(shape)
    wgt = torch.randn(shape)
    y_c, gx_c, gp_c = _run(m, x, wgt)
    y_t, gx_t, gp_t = _run(m, x, wgt, tiled=True)
    y_o, gx_o = _oracle(m.cpu(), x, grad_w=wgt)
    e = dict(y_vs_oracle=rel_err(y_c, y_o), y_vs_tiled=rel_err(y_c, y_t), dx_vs_oracle=rel_err(gx_c, gx_o), dx_vs_tiled=rel_err(gx_c, gx_t))
    gmax = max(v.abs().max().item() for v in gp_t.values())
    e["dparam_vs_tiled"] = max((gp_c[k] - gp_t[k]).abs().max().item() for k in gp_t) / gmax
    print(shape, {k: f"{v:.2e}" for k, v in e.items()})
    assert e["y_vs_oracle"] < 1e-4 and e["y_vs_tiled"] < 2e-5
    assert e["dx_vs_oracle"] < 5e-4 and e["dx_vs_tiled"] < 1e-4 and e["dparam_vs_tiled"] < 1e-4


def test_chunk_mlstm_32768_tokens_vs_recurrent_oracle():
    """S = 32 768 (the DoubleConv_ViL sequence of a 128^3 patch): forward against the fp64 recurrence, backward against
    the tiled kernels (the 32 768-step autograd graph of the oracle is not worth its minutes)."""
    m = _layer(3)
    torch.manual_seed(2)
    shape = (1, 32, 32, 32, 32)
    x = torch.randn(shape)
    wgt = torch.randn(shape)
    y_c, gx_c, gp_c = _run(m, x, wgt)
    y_t, gx_t, gp_t = _run(m, x, wgt, tiled=True)
    with torch.no_grad():
        y_o, _ = _oracle(m.cpu(), x)
    e_o, e_t, e_g = rel_err(y_c, y_o), rel_err(y_c, y_t), rel_err(gx_c, gx_t)
    gmax = max(v.abs().max().item() for v in gp_t.values())
    e_p = max((gp_c[k] - gp_t[k]).abs().max().item() for k in gp_t) / gmax
    print(f"S=32768: y vs fp64 recurrent oracle {e_o:.2e}, vs tiled {e_t:.2e}; dx vs tiled {e_g:.2e}; dparams vs tiled {e_p:.2e}")
    assert e_o < 1e-4 and e_t < 5e-5 and e_g < 2e-4 and e_p < 2e-4


def test_convxlstm_gcr_network_runs_at_128_with_32768_token_vil():
    torch.manual_seed(0)
    kw = dict(X.TRAIN_KWARGS)
    kw.update(layer_order="gcr", f_maps=8)
    m = X.U_HVEDConvXLSTMNet3D(1, 3, **kw)
    m.apply(X.init_weights)
    m = m.to(DEV).train()
    x = torch.rand(1, 4, 128, 128, 128, device=DEV)
    eps = [torch.randn(1, 2 * 2 ** l, 64 >> l, 64 >> l, 64 >> l, device=DEV) for l in range(4)]
    lib = X._lib.load()
    outs = []
    for tiled in (False, True):
        lib.xh_set_option(2, 8 if tiled else 0)
        try:
            for p in m.parameters():
                p.grad = None
            seg, (mu, lv), rec = m(x, [14], recon=True, eps_list=eps)
            rec = rec[0] if isinstance(rec, (list, tuple)) else rec
            (seg.mean() + rec.mean()).backward()
            torch.cuda.synchronize()
            vil = m.decoders[0].basic_module.ViL.vil.layer
            outs.append((seg.detach().cpu(), rec.detach().cpu(), vil.proj_up.weight.grad.cpu().clone()))
        finally:
            lib.xh_set_option(2, 0)
    (s0, r0, g0), (s1, r1, g1) = outs
    assert torch.isfinite(s0).all() and torch.isfinite(r0).all() and torch.isfinite(g0).all()
    e = (l2_err(s0, s1), l2_err(r0, r1), l2_err(g0, g1))
    print(f"U_HVEDConvXLSTMNet3D gcr 128^3 (ViL on 32768 tokens): chunk vs tiled seg/rec/dproj_up L2 {e[0]:.2e}/{e[1]:.2e}/{e[2]:.2e}")
    assert max(e) < 1e-3

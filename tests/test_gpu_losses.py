"""-m gpu: the loss / metric epilogues (xlstm_hved_amd.losses, csrc/loss.hip) against the fixture generated from the
reference's own loss.py / metrics.py objects (tests/golden/stage_losses.npz): values, input gradients, and the same
step composed through autograd with device-resident upstream gradients."""
import pytest
import torch

from gpu_common import check, l2_err, load

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_losses_vs_reference_fixture(dtype):
    g = load("stage_losses")
    f32 = dtype == torch.float32
    tol_v = 1e-5 if f32 else (2e-3 if dtype == torch.float16 else 1.5e-2)        # values
    tol_g = 2e-5 if f32 else (3e-3 if dtype == torch.float16 else 2e-2)          # gradients, relative L2
    prob, rec, disc, mu, lv = (g[k].to(DEV, dtype).requires_grad_(True) for k in ("prob", "rec", "disc", "mu", "lv"))
    tgt, xin = g["tgt"].to(DEV), g["xin"].to(DEV)                               # targets stay fp32 like mask_batch.float()
    dice = X.DiceLoss()(prob, tgt)
    mse = X.MSELoss()(rec, xin)
    gan = X.GANLoss()
    gan_t, gan_f = gan(disc, True), gan(disc, False)
    kld7 = X.compute_KLD(mu, lv, [7])
    kldm = X.compute_KLD(mu, lv, [2, 12])
    for v, k in ((dice, "dice"), (mse, "mse"), (gan_t, "gan_t"), (gan_f, "gan_f"), (kld7, "kld7"), (kldm, "kld_multi")):
        assert v.dim() == 0 and v.dtype == torch.float32
        assert abs(v.item() - g[k].item()) <= tol_v * max(1.0, abs(g[k].item())), (k, v.item(), g[k].item())
    (1.3 * dice + 0.2 * mse + 0.1 * gan_t + 0.05 * gan_f + 0.2 * kld7 + 0.3 * kldm).backward()
    torch.cuda.synchronize()
    for t, k in ((prob, "dprob"), (rec, "drec"), (disc, "ddisc"), (mu, "dmu"), (lv, "dlv")):
        assert t.grad.dtype == dtype
        e = l2_err(t.grad, g[k])
        assert e < tol_g, (k, e)
    if f32:
        check(X.ops.nested_weight(prob.detach()), g["nested"], 0, "nested weight map")
        syn = rec.detach().clone().requires_grad_(True)
        att = X.nested_attention(prob, syn)
        check(att, g["atten"], 1e-6, "nested attention")
        att.sum().backward()
        check(syn.grad, 1 + g["nested"].expand_as(g["atten"]), 1e-6, "d atten / d syn")
        assert abs(X.DiceCoefficient()(prob, tgt).item() - g["dice_coefficient"].item()) < 1e-6
        for i, r in enumerate(("WT", "TC", "EC")):
            assert abs(X.DiceRegion()(prob, tgt, r).item() - g["dice_region"][i].item()) < 1e-6


def test_losses_full_size_one_pass_properties():
    """128^3 (BASELINE config 2 size): the loss of a prediction against itself / its complement has a closed form."""
    torch.manual_seed(0)
    t = (torch.rand(1, 3, 128, 128, 128, device=DEV) > 0.6).float()
    p = t.bfloat16()
    assert abs(X.DiceLoss()(p, t).item()) < 1e-6                                 # dice(t, t) = 1
    assert abs(X.DiceCoefficient()(p, t).item() - 1.0) < 1e-6
    assert abs(X.mse_loss(p, t).item()) < 1e-12
    q = (1 - t).bfloat16()
    assert abs(X.DiceLoss()(q, t).item() - 1.0) < 1e-6                           # disjoint masks: dice 0
    assert abs(X.mse_loss(q, t).item() - 1.0) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mean_of_matches_torch_mean(dtype):
    torch.manual_seed(1)
    for shape in ((2, 3, 8, 12, 16), (1, 5, 2, 4, 4, 8)):
        x = torch.randn(shape, device=DEV).to(dtype).requires_grad_(True)
        m = X.losses.mean_of(x)
        (m * 3.0).backward()
        ref = x.detach().float().mean()
        assert abs(m.item() - ref.item()) < 1e-6
        assert torch.allclose(x.grad.float(), torch.full(shape, 3.0 / x.numel(), device=DEV).to(dtype).float())


def test_sum_of_means_matches_torch():
    torch.manual_seed(2)
    # the 3 x 64 x 128 x 128 tensor (>= 2^20 elements) is reduced as 16 rows that share its weight
    ts = [torch.randn(s, device=DEV).bfloat16().requires_grad_(True)
          for s in ((1, 3, 8, 8, 8), (1, 5, 1, 4, 4, 4), (2, 4, 6, 6, 10), (1, 3, 64, 128, 128))]
    m = X.losses.sum_of_means(ts)
    (m * 2.0).backward()
    ref = sum(t.detach().float().mean() for t in ts)
    assert abs(m.item() - ref.item()) < 1e-6
    for t in ts:
        assert torch.allclose(t.grad.float(), torch.full(t.shape, 2.0 / t.numel(), device=DEV).bfloat16().float())

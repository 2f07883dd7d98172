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


def test_loss_terms_combined_in_one_node_match_the_term_by_term_sum():
    """train.py:262 as losses.combine (ops.scalar_lincomb / scalar_fanout): value and every term's gradient equal the chain of scalar
    ATen operations; fp32 and fp64 terms mix (the KL reductions are fp64)."""
    from xlstm_hved_amd import losses
    torch.manual_seed(3)
    a = [torch.rand((), device=DEV, dtype=torch.float32, requires_grad=True) for _ in range(3)]
    b = [torch.rand((), device=DEV, dtype=torch.float64, requires_grad=True) for _ in range(2)]
    coefs = [1.0, 1.0, 0.2, 0.2, 0.1]
    tot = losses.combine(a + b, coefs)
    assert tot.dim() == 0 and tot.dtype == torch.float32
    seed = torch.full((), 3.0, device=DEV)
    tot.backward(seed)
    ref = sum(c * t.detach().double() for c, t in zip(coefs, a + b))
    assert abs(tot.item() - ref.item()) < 1e-6
    for c, t in zip(coefs, a + b):
        assert t.grad.dtype == t.dtype and abs(t.grad.item() - 3.0 * c) < 1e-6
    with pytest.raises(Exception):
        X.ops.scalar_lincomb([torch.zeros(2, device=DEV)], [1.0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_gan_pair_loss_equals_the_two_losses_on_the_halves(dtype):
    """alpha * 0.5 * (GANLoss(D(fake), False) + GANLoss(D(real), True)) (train.py:272-280) as one reduction against a resident
    label tensor: value and gradient of the two-loss form."""
    from xlstm_hved_amd import losses
    torch.manual_seed(4)
    out = torch.randn(4, 1, 6, 6, 6, device=DEV).to(dtype).requires_grad_(True)
    one = losses.gan_pair_loss(out, 2, 0.1)
    one.backward()
    g1 = out.grad.clone()
    out.grad = None
    gan = X.GANLoss()
    two = 0.1 * (gan(out[:2].float(), False) + gan(out[2:].float(), True)) * 0.5
    two.backward()
    assert abs(one.item() - two.item()) <= 1e-6 * max(1.0, abs(two.item()))
    assert g1.dtype == dtype
    assert l2_err(g1, out.grad) < (1e-6 if dtype == torch.float32 else 4e-3)


def test_kld_over_the_levels_as_one_node_equals_the_per_level_mean():
    """train.py:235-239: sum_l compute_KLD(mu[l], lv[l], subset) / 4 -- one node (losses.compute_KLD_levels) against the loop."""
    from xlstm_hved_amd import losses
    torch.manual_seed(5)
    shapes = [(1, 5, 8, 2, 2, 2), (1, 5, 4, 4, 4, 4), (1, 5, 2, 8, 8, 8), (1, 5, 1, 16, 16, 16)]
    mu = [(0.3 * torch.randn(s, device=DEV)).requires_grad_(True) for s in shapes]
    lv = [(0.2 * torch.randn(s, device=DEV)).requires_grad_(True) for s in shapes]
    for subset in ([7], [2, 12], torch.tensor([[1.0, 0.0, 1.0, 1.0]], device=DEV)):
        for t in mu + lv:
            t.grad = None
        one = losses.compute_KLD_levels(mu, lv, subset)
        one.backward(torch.full((), 2.0, device=DEV))
        g1 = [t.grad.clone() for t in mu + lv]
        for t in mu + lv:
            t.grad = None
        ref = None
        for l in range(4):
            k = X.compute_KLD(mu[l], lv[l], subset)
            ref = k if ref is None else ref + k
        ref = ref / 4
        ref.backward(torch.full((), 2.0, device=DEV))
        assert abs(one.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item())), (one.item(), ref.item())
        for a_, t in zip(g1, mu + lv):
            assert l2_err(a_, t.grad) < 1e-5

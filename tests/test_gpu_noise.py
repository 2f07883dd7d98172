"""-m gpu: the reparameterisation noise drawn inside the PoE kernel (xh_poe_multi with a generator state; RA_HVED.py:741-747 draws
eps ~ N(0,1) in fp32 with normal_()): Philox4x32-10 bit for bit against the numpy reference (oracle/philox_ref.py, pinned by the
published known-answer vectors), its statistics, z = mu + eps * std with exactly that eps, a backward pass that regenerates it, and a
captured graph that draws fresh noise on every replay."""
import numpy as np
import pytest
import torch

from gpu_common import load

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402
import philox_ref as P  # noqa: E402

DEV = "cuda"


def test_philox_words_and_normals_equal_the_reference():
    seed, ctr = 0x299f31d0a4093822, 0x0000000513198a2e
    for stream, n in ((0, 4096), (3, 1000), (255, 257)):
        w = X.ops.philox_normal(seed, ctr, stream, n, DEV, raw=True).cpu().numpy().view(np.uint32)
        assert (w == P.poe_noise_words(seed, ctr, stream, n)).all()
        z = X.ops.philox_normal(seed, ctr, stream, n, DEV).cpu().numpy()
        want = P.poe_noise(seed, ctr, stream, n)
        assert np.abs(z - want).max() < 2e-5                      # fp32 log / cos / sqrt against float64
    a = X.ops.philox_normal(seed, ctr, 1, 512, DEV)
    assert not torch.equal(a, X.ops.philox_normal(seed, ctr + 1, 1, 512, DEV))      # another draw
    assert not torch.equal(a, X.ops.philox_normal(seed, ctr, 2, 512, DEV))          # another level
    assert not torch.equal(a, X.ops.philox_normal(seed + 1, ctr, 1, 512, DEV))      # another seed


def test_noise_statistics_over_2_24_draws():
    n = 1 << 24
    z = X.ops.philox_normal(987654321, 7, 0, n, DEV).double()
    mean, var = z.mean().item(), z.var().item()
    lag1 = ((z[1:] - mean) * (z[:-1] - mean)).mean().item() / var
    kurt = ((z - mean) ** 4).mean().item() / var ** 2
    frac2 = (z.abs() > 2).double().mean().item()
    print(f"2^24 draws: mean {mean:.2e}, var {var:.5f}, lag-1 correlation {lag1:.2e}, kurtosis {kurt:.4f}, P(|z|>2) {frac2:.5f}, max {z.abs().max().item():.2f}")
    s = n ** -0.5                                                   # standard error of a mean of n unit-variance values
    assert abs(mean) < 5 * s and abs(var - 1) < 5 * (2 ** 0.5) * s and abs(lag1) < 5 * s
    assert abs(kurt - 3) < 0.01 and abs(frac2 - 0.0455003) < 2e-4
    # two levels of one draw are uncorrelated
    z1 = X.ops.philox_normal(987654321, 7, 1, n, DEV).double()
    assert abs((z * z1).mean().item()) < 5 * s


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["fp32", "bf16", "fp16"])
def test_poe_draws_exactly_that_noise_and_backward_regenerates_it(dtype):
    """ops.poe_fwd_multi(rng=...) == the same call with the noise handed in (eps in fp32 for fp32 storage: bit for bit; 16-bit storage:
    the in-kernel eps is fp32, the handed-in one is rounded to the storage type first -- that difference, 2^-9 / 2^-12 relative in eps,
    is the point of drawing in the kernel); the generator advances by one per launch and the backward call needs only the two words the
    forward recorded."""
    torch.manual_seed(5)
    n = 2
    shapes = [(1, (8, 8, 8)), (2, (4, 6, 8)), (4, (4, 4, 4)), (8, (2, 2, 2))]
    feats = [torch.randn((n, 8 * L_) + sp, device=DEV).to(dtype) for L_, sp in shapes]
    keep = torch.tensor([[1.0, 0.0, 1.0, 1.0], [1.0, 1.0, 1.0, 0.0]], device=DEV)
    Ls = [L_ for L_, _ in shapes]
    seed = 0x1234567887654321
    state = X.ops.rng_state(seed, DEV)
    state[1] = 41
    used = torch.zeros(2, dtype=torch.int64, device=DEV)
    got = X.ops.poe_fwd_multi(feats, keep, [None] * 4, Ls, False, rng=(state, used))
    torch.cuda.synchronize()
    assert state[:2].tolist() == [seed, 42] and int(state[2:].abs().sum()) == 0 and used.tolist() == [41, seed]
    epss = [X.ops.philox_normal(seed, 41, l, n * L_ * int(np.prod(sp)), DEV).view((n, L_) + sp) for l, (L_, sp) in enumerate(shapes)]
    want = X.ops.poe_fwd_multi(feats, keep, [e.to(dtype) for e in epss], Ls, False)
    tol = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -10}[dtype]
    for l in range(4):
        zg, zw = got[l][0].float(), want[l][0].float()
        assert (zg - zw).abs().max().item() <= tol * max(1.0, zw.abs().max().item()) * 8, l
        assert torch.equal(got[l][1], want[l][1]) and torch.equal(got[l][2], want[l][2])
    if dtype == torch.float32:
        assert all(torch.equal(got[l][0], want[l][0]) for l in range(4))
    dzs = [torch.randn_like(o[0]) for o in got]
    dmus = [torch.randn_like(o[1]) for o in got]
    dlvs = [torch.randn_like(o[2]) for o in got]
    state[1] = 1000                                                 # the generator has moved on: backward must not care
    dg = X.ops.poe_bwd_multi(feats, keep, [None] * 4, dzs, dmus, dlvs, Ls, False, rng_used=used)
    dw = X.ops.poe_bwd_multi(feats, keep, [e.to(dtype) for e in epss], dzs, dmus, dlvs, Ls, False)
    for l in range(4):
        a, b = dg[l].float(), dw[l].float()
        assert (a - b).abs().max().item() <= 8 * tol * max(1.0, b.abs().max().item()), l
        if dtype == torch.float32:
            assert torch.equal(dg[l], dw[l])
    # a second launch on the same state draws other noise
    again = X.ops.poe_fwd_multi(feats, keep, [None] * 4, Ls, False, rng=(state, used))
    assert not torch.equal(again[0][0], got[0][0])


def test_model_noise_is_seeded_advances_per_forward_and_per_graph_replay():
    """The network's default noise (no eps_list, valid=False): fixed by model.seed_noise, one draw per forward, fp32 eps with 16-bit
    storage; the forward equals the eps_list path fed with the same values; a captured forward draws fresh noise on every replay; the
    backward pass of a captured step sees the noise of ITS forward (gradients equal the eps_list run's)."""
    m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    m.load_state_dict(load("weights_seed1"), strict=True)
    m = m.to(DEV).train()
    torch.manual_seed(8)
    x = torch.rand(1, 4, 32, 32, 32).to(DEV)
    st = m.noise_state(torch.device(DEV, torch.cuda.current_device()))
    m.seed_noise(77)
    torch.cuda.synchronize()
    assert st[:2].tolist() == [77, 0] and int(st[2:].abs().sum()) == 0
    seg0, (mu0, lv0), rec0 = m(x, [14], recon=True)
    seg1, _, _ = m(x, [14], recon=True)
    torch.cuda.synchronize()
    assert st[:2].tolist() == [77, 2] and int(st[2:].abs().sum()) == 0 and (seg0 - seg1).abs().max().item() > 1e-4
    m.seed_noise(77)
    seg0b, _, _ = m(x, [14], recon=True)
    assert (seg0 - seg0b).abs().max().item() <= 1e-6               # same seed, same draw: the same forward (to the order of the fp64 statistics atomics)
    eps = [X.ops.philox_normal(77, 0, l, int(np.prod(s_)), DEV).view(s_) for l, s_ in
           enumerate([(1, 1, 16, 16, 16), (1, 2, 8, 8, 8), (1, 4, 4, 4, 4), (1, 8, 2, 2, 2)])]
    seg_e, (mu_e, lv_e), rec_e = m(x, [14], recon=True, eps_list=eps)
    assert (seg0 - seg_e).abs().max().item() <= 1e-6 and (rec0[0] - rec_e[0]).abs().max().item() <= 1e-5 * rec_e[0].abs().max().item()

    def grads(fn):
        for p in m.parameters():
            p.grad = None
        seg, (mu, lv), rec = fn()
        (seg.float().mean() + rec[0].float().mean() + sum(a.float().mean() + b.float().mean() for a, b in zip(mu, lv))).backward()
        X.ops.join_wgrad_stream()
        torch.cuda.synchronize()
        return torch.cat([p.grad.flatten() for p in m.parameters() if p.grad is not None])
    m.seed_noise(77)
    g_in = grads(lambda: m(x, [14], recon=True))
    g_eps = grads(lambda: m(x, [14], recon=True, eps_list=eps))
    assert (g_in - g_eps).abs().max().item() <= 2e-5 * g_eps.abs().max().item()
    # captured forward: every replay a new draw
    m.seed_noise(5)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        m(x, [14], recon=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    c0 = st.tolist()[1]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        out = m(x, [14], recon=True)[0]
    outs = []
    for _ in range(3):
        g.replay()
        outs.append(out.clone())
    torch.cuda.synchronize()
    assert st.tolist()[1] == c0 + 3 and int(st[2:].abs().sum()) == 0
    assert (outs[0] - outs[1]).abs().max().item() > 1e-4 and (outs[1] - outs[2]).abs().max().item() > 1e-4
    with torch.no_grad():
        eps_r = [X.ops.philox_normal(5, c0 + 2, l, e.numel(), DEV).view(e.shape) for l, e in enumerate(eps)]
        assert (outs[2] - m(x, [14], recon=True, eps_list=eps_r)[0]).abs().max().item() <= 1e-6

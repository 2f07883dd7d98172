"""Pins oracle/xlstm_hved_oracle.py to the golden vectors generated from the real reference
(tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest
import torch

import xlstm_hved_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def sd_of(g, prefix="sd."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def close(a, b, tol, what=""):
    err = (a.double() - b.double()).abs().max().item()
    scale = max(b.double().abs().max().item(), 1.0)
    assert err <= tol * scale, f"{what}: {err:.3e} > {tol:.1e}*{scale:.3e}"


STAGES = {
    "stage_singleconv_ilc": lambda p, x: O.single_conv(p, x, "ilc"),
    "stage_singleconv_ilc_s2": lambda p, x: O.single_conv(p, x, "ilc", stride=2),
    "stage_singleconv_gcr": lambda p, x: O.single_conv(p, x, "gcr"),
    "stage_singleconv_gcr_g1": lambda p, x: O.single_conv(p, x, "gcr"),
    "stage_encoder_pool": lambda p, x: O.encoder(p, x, pool=True),
    "stage_encoder_nopool": lambda p, x: O.encoder(p, x, pool=False),
    "stage_basicconv_1x1": lambda p, x: O.basic_conv(p, x),
    "stage_basicconv_dw": lambda p, x: O.basic_conv(p, x, groups=8),
    "stage_decoder_recon": lambda p, skip, x: O.double_conv(
        p.sub("basic_module"), torch.cat([skip, O.upsample_to(x, skip.shape[2:])], 1)),
    "stage_decoder_seg": lambda p, skip, x: O.double_conv(
        p.sub("basic_module"), O.atten_module2(p.sub("atten_module"), O.upsample_to(x, skip.shape[2:]), skip)),
    "stage_duse_train": lambda p, r, s: O.duse_attention(p, r, s, True),
    "stage_duse_eval": lambda p, r, s: O.duse_attention(p, r, s, False),
    "stage_skr_att_train": lambda p, x: O.skip_return_attention(p, x, True, momentum_steps=1),
    "stage_skr_att_eval": lambda p, x: O.skip_return_attention(p, x, False),
    "stage_vil_s64": lambda p, x: O.vil_layer(p, x),
    "stage_vil_s512": lambda p, x: O.vil_layer(p, x),
}


@pytest.mark.parametrize("name", sorted(STAGES))
def test_stage_matches_reference(name):
    g = load(name)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd_of(g).items()}
    ins = []
    i = 0
    while f"in{i}" in g:
        ins.append(g[f"in{i}"].clone().requires_grad_(True))
        i += 1
    out = STAGES[name](O.P(sd), *ins)
    outs = list(out) if isinstance(out, (tuple, list)) else [out]
    tol = 1e-4 if "vil" in name else 2e-5
    loss = 0
    for j, o in enumerate(outs):
        close(o, g[f"out{j}"], tol, f"{name}.out{j}")
        loss = loss + (o * g[f"w{j}"]).sum()
    loss.backward()
    for j, t in enumerate(ins):
        close(t.grad, g[f"gin{j}"], tol * 10, f"{name}.gin{j}")
    for k, v in g.items():
        if k.startswith("g."):
            close(sd[k[2:]].grad, v, tol * 10, f"{name}.{k}")
    for k, v in sd_of(g, "sd_after.").items():
        close(sd[k].detach(), v, 1e-5, f"{name}.buffer.{k}")


def test_mlstm_cell_dense_and_recurrent():
    g = load("stage_mlstm_cell")
    a = [g[k].double() for k in ("q", "k", "v", "ig", "fg")]
    close(O.mlstm_parallel(*a), g["h"], 1e-12, "dense")
    close(O.mlstm_recurrent(*a), g["h"], 1e-10, "recurrent")


def test_product_of_experts_all_subsets():
    g = load("stage_poe")
    for idx in range(15):
        m, l = O.product_of_experts(g["mu"], g["logvar"], O.SUBSETS_MODALITIES[idx])
        close(m, g[f"mu_{idx}"], 1e-6), close(l, g[f"lv_{idx}"], 1e-6)
    m, l, mu_after = O.product_of_experts_drop(g["mu"], g["logvar"], g["drop"])
    close(m, g["mu_drop"], 1e-6), close(l, g["lv_drop"], 1e-6), close(mu_after, g["mu_after"], 0)
    close(O.reparametrize(m, l, g["eps"]), g["z"], 1e-6)


def test_subset_table_order():
    # RA_HVED.py:733-738: singles, pairs (0,1)(0,2)(0,3)(1,2)(1,3)(2,3), triples, all
    s = O.SUBSETS_MODALITIES
    assert s[:4] == [(0,), (1,), (2,), (3,)]
    assert s[4:10] == [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    assert s[14] == (0, 1, 2, 3) and len(s) == 15


def _weights(dtype):
    w = load("weights_seed1")
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in w.items()}


def test_network_train_forward_backward_fp32():
    g = load("net32_train_subset14")
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in _weights(torch.float32).items()}
    eps = [g[f"eps{i}"] for i in range(4)]
    prob, logits, mu, lv, rec = O.xlstm_hved_forward(sd, g["x"], 14, eps_list=eps, training=True)
    close(prob, g["seg"], 2e-4, "seg"), close(rec, g["rec"], 2e-4, "rec")
    # fp64 tie-breaker: fp32 oracle error vs fp64 reference is of the order of the fp32 reference's own
    e_or = (prob.double() - g["f64.seg"]).abs().max().item()
    e_ref = (g["seg"].double() - g["f64.seg"]).abs().max().item()
    assert e_or <= 2 * e_ref + 1e-5
    for i in range(4):
        close(mu[i], g[f"mu{i}"], 1e-4), close(lv[i], g[f"lv{i}"], 1e-4)

    def rnd(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed))
    loss = (prob * rnd(prob.shape, 200)).sum() + 0.1 * (rec * rnd(rec.shape, 201)).sum()
    for i, (a, b) in enumerate(zip(mu, lv)):
        loss = loss + 0.05 * ((a * rnd(a.shape, 210 + i)).sum() + (b * rnd(b.shape, 220 + i)).sum())
    loss.backward()
    gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith("g."))
    n = 0
    for k, v in g.items():
        if k.startswith("g."):
            err = (sd[k[2:]].grad - v).abs().max().item() / gscale
            assert err < 5e-4, (k, err)
            n += 1
    assert n > 250
    # parameters the reference never reaches keep no gradient (DDP note)
    for k in ("rdecoder.finals.0.weight", "mViL.norm.weight", "skr_att.0.0.conv1.dwconv.weight",
              "srdecoder.dusfe_decoders.0.conv_fuse_ch1.weight"):
        assert sd[k].grad is None and ("g." + k) not in g
    for k, v in g.items():
        if k.startswith("after."):
            close(sd[k[6:]].detach(), v, 1e-5, k)


def test_network_all_subsets_and_instance_missing_fp64():
    g = load("net32_subsets_eval")
    w = _weights(torch.float64)
    x2 = g["x2"].double()
    for k in range(15):
        sd = {n: v.clone() for n, v in w.items()}
        prob, _, mu, lv, rec = O.xlstm_hved_forward(sd, x2[:1], k, eps_list=None, training=False)
        close(prob.flatten()[g["idx_seg"]], g[f"seg_{k}"], 1e-9), close(rec.flatten()[g["idx_rec"]], g[f"rec_{k}"], 1e-9)
        close(mu[3].flatten(), g[f"mu3_{k}"], 1e-9)
    xm = x2.clone()
    for i, mk in enumerate([(1, 3), (0,)]):
        for c in range(4):
            if c not in mk:
                xm[i, c] = 0
    S3 = 32 ** 3
    for train in (True, False):
        sd = {n: v.clone() for n, v in w.items()}
        prob, _, mu, lv, rec = O.xlstm_hved_forward(sd, xm, 14, instance_missing=True, eps_list=None, training=train)
        t = "train" if train else "eval"
        close(prob.flatten()[torch.cat([g["idx_seg"], g["idx_seg"] + 3 * S3])], g[f"im_{t}_seg"], 1e-9)
        close(rec.flatten()[torch.cat([g["idx_rec"], g["idx_rec"] + 4 * S3])], g[f"im_{t}_rec"], 1e-9)
        close(mu[0].flatten()[:8192], g[f"im_{t}_mu0"], 1e-9)
        if train:
            for k, v in g.items():
                if k.startswith("im_train_after."):
                    close(sd[k[len("im_train_after."):]].double(), v.double(), 1e-9, k)


def test_dice_region_metric():
    prob = torch.tensor([0.6, 0.4, 0.7, 0.2]).view(1, 1, 1, 2, 2).repeat(1, 3, 1, 1, 1)
    tgt = torch.tensor([1.0, 1.0, 0.0, 0.0]).view(1, 1, 1, 2, 2).repeat(1, 3, 1, 1, 1)
    d = O.dice_region(prob, tgt)
    assert torch.allclose(d, torch.full((3,), (2 * 1 + 1e-6) / (4 + 1e-6)))


# Secondary configurations (SURVEY.md section 2 variant classes that the reference can actually run):
VARIANT_FLAGS = {
    "uhved_conv_gcr": dict(order="gcr", mid_vil=False, skip_return=False, seg_recon_decoder=False),
    "uhved_convxlstm_gcr": dict(order="gcr", mid_vil=False, skip_return=False, seg_recon_decoder=False),
    "xlstm_hved_wodusfe": dict(order="ilc", mid_vil=True, skip_return=True, seg_recon_decoder=False),
    "xlstm_hved_noshared": dict(order="ilc", mid_vil=True, skip_return=True),       # shared_recon=False (Pretrain.py:142)
}


def variant_state(g, dtype):
    from seeded_weights import seeded_state
    ents = [(str(n), tuple(int(d) for d in str(s).split(",") if d), bool(f))
            for n, s, f in zip(g["names"], g["shapes"], g["isfloat"])]
    sd = seeded_state(ents, seed=5)
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def load_np(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (z[k] if z[k].dtype.kind in "US" else torch.from_numpy(z[k])) for k in z.files}


@pytest.mark.parametrize("tag", sorted(VARIANT_FLAGS))
def test_variant_forward_backward_fp64(tag):
    """gcr order / DoubleConv_ViL decoder / separate recon+seg decoders (RA_HVED.py:651-687) against the reference's fp64
    outputs and per-parameter gradient sums."""
    g = load_np("variant_" + tag)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in variant_state(g, torch.float64).items()}
    eps = [g[f"eps{i}"].double() for i in range(4)]
    prob, _, mu, lv, rec = O.xlstm_hved_forward(sd, g["x"].double(), 14, eps_list=eps, training=True, **VARIANT_FLAGS[tag])
    close(prob.flatten()[g["idx_seg"]], g["seg"], 1e-9, "seg")
    close(rec.flatten()[g["idx_rec"]], g["rec"], 1e-9, "rec")
    close(mu[3].flatten(), g["mu3"], 1e-9, "mu3"), close(lv[3].flatten(), g["lv3"], 1e-9, "lv3")

    def rnd(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).double()
    loss = (prob * rnd(prob.shape, 300)).sum() + 0.1 * (rec * rnd(rec.shape, 301)).sum()
    for i, (a, b) in enumerate(zip(mu, lv)):
        loss = loss + 0.05 * ((a * rnd(a.shape, 310 + i)).sum() + (b * rnd(b.shape, 320 + i)).sum())
    close(loss.detach(), g["loss"], 1e-10, "loss")
    loss.backward()
    scale = g["gabs"].max().item()
    for name, gsum, gabs in zip(g["gnames"], g["gsum"], g["gabs"]):
        gr = sd[str(name)].grad
        assert gr is not None, name
        assert abs(gr.sum().item() - gsum.item()) <= 1e-9 * scale, name
        assert abs(gr.abs().sum().item() - gabs.item()) <= 1e-9 * scale, name


def test_loss_and_metric_restatements_vs_reference_fixture():
    """DiceLoss / MSE / GANLoss / compute_KLD / nested weights / DiceCoefficient / DiceRegion (loss.py, metrics.py,
    train.py:232-262): values and input gradients of the reference's own objects, stored by make_golden.loss_cases."""
    g = load_np("stage_losses")
    po, ro, do_, muo, lvo = (g[k].clone().requires_grad_(True) for k in ("prob", "rec", "disc", "mu", "lv"))
    tgt, xin = g["tgt"], g["xin"]
    o_dice, o_mse = O.dice_loss(po, tgt), ((ro - xin) ** 2).mean()
    o_gt, o_gf = ((do_ - 1.0) ** 2).mean(), (do_ ** 2).mean()
    o_k7, o_km = O.compute_kld(muo, lvo, [7]), O.compute_kld(muo, lvo, [2, 12])
    (1.3 * o_dice + 0.2 * o_mse + 0.1 * o_gt + 0.05 * o_gf + 0.2 * o_k7 + 0.3 * o_km).backward()
    for a, k in ((o_dice, "dice"), (o_mse, "mse"), (o_gt, "gan_t"), (o_gf, "gan_f"), (o_k7, "kld7"), (o_km, "kld_multi"),
                 (po.grad, "dprob"), (ro.grad, "drec"), (do_.grad, "ddisc"), (muo.grad, "dmu"), (lvo.grad, "dlv")):
        close(a.detach(), g[k], 2e-6, k)
    close(O.nested_weight(g["prob"]), g["nested"], 0, "nested")
    close(O.dice_coefficient(g["prob"], tgt).mean(), g["dice_coefficient"], 1e-6, "dice coefficient")
    close(O.dice_region(g["prob"], tgt), g["dice_region"], 1e-6, "dice region")


@pytest.mark.parametrize("ks", [4, 3])
@pytest.mark.parametrize("tag,dt,tol", [("f32", torch.float32, 2e-5), ("f64", torch.float64, 1e-11)])
def test_discriminator_matches_reference(ks, tag, dt, tol):
    """oracle.discriminator against the REAL reference class (tests/golden/stage_disc_ks*.npz: train.py:146's
    Discriminator(in_channels=7, ks=4, strides=[1,2,2,2]) and the ks=3 default): output, input gradient, all 9 parameter
    gradients (strided samples + sums)."""
    import disc_common as DC
    g = DC.load_fixture(ks)
    sd = {k: v.to(dt).requires_grad_(True) for k, v in DC.seeded_disc_state(ks, g).items()}
    x = DC.seeded_input(g).to(dt).requires_grad_(True)
    y = O.discriminator(O.P(sd), x)
    assert tuple(y.shape) == tuple(g[f"{tag}.y"].shape)
    close(y, g[f"{tag}.y"], tol, "y")
    (y * g["gy"].to(dt)).sum().backward()
    gx = x.grad.flatten()
    close(gx[DC.sample_index(gx.numel(), 65536)], g[f"{tag}.dx"], tol * 10, "dx")
    for k, v in sd.items():
        gr = v.grad.flatten()
        ref = g[f"{tag}.g.{k}"]
        scale = max(ref.double().abs().max().item(), 1e-30)
        assert (gr[DC.sample_index(gr.numel())].double() - ref.double()).abs().max().item() <= tol * 10 * max(scale, 1.0), k
        assert abs(gr.double().sum().item() - float(g[f"{tag}.gsum.{k}"])) <= tol * 10 * max(float(g[f"{tag}.gabs.{k}"]), 1.0), k


def test_reference_cost_mode_is_the_same_function():
    """reference_cost=True re-evaluates the skip-return attention once per modality stream like RA_HVED.py:548-552 (what
    bench.py's cpu_baseline leg times): same outputs, same BatchNorm buffers (four momentum steps either way)."""
    g = load("net32_train_subset14")
    eps = [g[f"eps{i}"] for i in range(4)]
    outs, bufs = [], []
    for rc in (False, True):
        sd = {k: v.clone() for k, v in _weights(torch.float32).items()}
        with torch.no_grad():
            prob, _, mu, lv, rec = O.xlstm_hved_forward(sd, g["x"], 14, eps_list=eps, training=True, reference_cost=rc)
        outs.append((prob, rec, mu[3]))
        bufs.append({k: v for k, v in sd.items() if "running" in k or "num_batches" in k})
    # (four stock fused batch_norm calls vs one explicit form with a 4-step momentum: fp32 round-off, amplified by the network)
    for a, b in zip(outs[0], outs[1]):
        close(b, a, 2e-4, "reference_cost output")
    for k, v in bufs[0].items():
        close(bufs[1][k].double(), v.double(), 1e-5, k)


def test_oracle_reproduces_the_reference_mask_at_128():
    """tests/golden/mask_trained_like_128.npz is the REAL reference's thresholded segmentation of the benchmark-size parity
    case (make_mask_128.py); bench.py measures every storage mode against it.  The oracle must give the same mask."""
    import synth_blobs as SB
    z = np.load(os.path.join(GOLDEN, "mask_trained_like_128.npz"))
    shape = tuple(int(v) for v in z["shape"])
    ref = np.unpackbits(z["bits"])[: int(np.prod(shape))].reshape(shape).astype(bool)
    w = np.load(os.path.join(GOLDEN, "weights_trained_like.npz"))
    sd = {k: torch.from_numpy(w[k]) for k in w.files}
    x, _ = SB.blob_case(8, 1, 128)
    with torch.no_grad():
        prob = O.xlstm_hved_forward(sd, x, 14, eps_list=None, training=False)[0]
    got = (prob > 0.5).numpy()
    assert [int(got[0, c].sum()) for c in range(3)] == [int(v) for v in z["pos"]]
    assert int((got != ref).sum()) == 0


def test_philox_reference_known_answers():
    """oracle/philox_ref.py (the checker of the in-kernel reparameterisation noise) against the published known-answer vectors of
    Philox4x32-10 (Random123 kat_vectors), and the mapping of (seed, draw, level, element) onto counter / key words."""
    import philox_ref as P
    for c, k, want in P.KAT:
        got = P.philox4x32_10(np.array(c, dtype=np.uint32)[None], k)[0]
        assert tuple(int(v) for v in got) == want
    w = P.poe_noise_words(0xa4093822 | (0x299f31d0 << 32), 0x13198a2e | (0x03707344 << 32), 0x85, 1)
    # element 0 of level 0x85: counter (0, 0x85 << 24, draw lo, draw hi), key (seed lo, seed hi)
    ref = P.philox4x32_10(np.array([[0, 0x85000000, 0x13198a2e, 0x03707344]], dtype=np.uint32), (0xa4093822, 0x299f31d0))
    assert (w == ref).all()
    z = P.poe_noise(1234, 5, 2, 1 << 16)
    assert abs(z.mean()) < 0.02 and abs(z.var() - 1) < 0.02

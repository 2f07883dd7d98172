"""-m gpu: every fused HIP stage against the golden vectors produced by the REAL reference
(tests/golden/stage_*.npz: inputs, weights, outputs, input- and parameter-gradients) and against the oracle at
other shapes.  fp32 storage is held to round-off; bf16 storage to bf16 resolution."""
import pytest
import torch

from gpu_common import check, check_grads, l2_err, load, rel_err, rnd, sd_of

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402
import xlstm_hved_oracle as O  # noqa: E402

DEV = "cuda"


def _stage_modules():
    return {
        "stage_singleconv_ilc": lambda: X.SingleConv(4, 8, 3, 1, "ilc", 8, padding=1),
        "stage_singleconv_ilc_s2": lambda: X.SingleConv(8, 4, 3, 2, "ilc", 8, padding=1),
        "stage_singleconv_gcr": lambda: X.SingleConv(16, 8, 3, 1, "gcr", 8, padding=1),
        "stage_singleconv_gcr_g1": lambda: X.SingleConv(4, 8, 3, 1, "gcr", 8, padding=1),
        "stage_encoder_pool": lambda: X.Encoder(4, 8, conv_layer_order="ilc"),
        "stage_encoder_nopool": lambda: X.Encoder(4, 4, apply_pooling=False, conv_layer_order="ilc"),
        "stage_basicconv_1x1": lambda: X.BasicConv(2, 8, 1),
        "stage_basicconv_dw": lambda: X.BasicConv(8, 8, 3, padding=1, groups=8),
        "stage_decoder_recon": lambda: X.Decoder(24, 8, conv_layer_order="ilc"),
        "stage_decoder_seg": lambda: X.Decoder(12, 4, conv_layer_order="ilc", RSM=True, MVAE=True),
        "stage_duse_train": lambda: X.DuSEAttention(8),
        "stage_duse_eval": lambda: X.DuSEAttention(4),
        "stage_skr_att_train": lambda: X.SkipReturnAttention(8),
        "stage_skr_att_eval": lambda: X.SkipReturnAttention(4),
        "stage_vil_s64": lambda: X.ViLLayer(32),
        "stage_vil_s512": lambda: X.ViLLayer(32),
    }


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("name", sorted(_stage_modules()))
def test_stage_against_reference_golden(name, dtype):
    g = load(name)
    mod = _stage_modules()[name]()
    mod.load_state_dict(sd_of(g), strict=True)
    mod = mod.to(DEV)
    mod.train("eval" not in name)
    ins = []
    i = 0
    while f"in{i}" in g:
        ins.append(g[f"in{i}"].to(DEV, dtype).requires_grad_(True))
        i += 1
    out = mod(*ins)
    outs = list(out) if isinstance(out, (tuple, list)) else [out]
    f32 = dtype == torch.float32
    # fp32 storage: round-off.  bf16 storage: pointwise values to bf16 resolution; gradients in relative L2 (a sign
    # flip of a near-zero pre-activation moves a LeakyReLU gradient by 100x at that voxel, so max-norm is meaningless;
    # the skip-return block stacks two ReLUs, a channel arg-max and a third ReLU, hence its wider band)
    # fp16 storage (the reference's AMP dtype): 11 significant bits instead of 8 -> 8x tighter bands than bf16
    h16 = dtype == torch.float16
    tol_o = (2e-4 if "vil" in name else 5e-5) if f32 else (6e-3 if h16 else 4e-2)
    loss = 0
    for j, o in enumerate(outs):
        assert o.dtype == dtype
        check(o, g[f"out{j}"], tol_o, f"{name}.out{j}")
        loss = loss + (o.float() * g[f"w{j}"].to(DEV)).sum()
    loss.backward()
    torch.cuda.synchronize()
    for j, t in enumerate(ins):
        if f32:
            check(t.grad, g[f"gin{j}"], 2e-3, f"{name}.gin{j}")
        else:
            e = l2_err(t.grad, g[f"gin{j}"])
            assert e < (0.4 if "skr" in name else 0.12) * (0.2 if h16 else 1.0), f"{name}.gin{j}: relative L2 error {e:.3e}"
    params = {k: p.grad for k, p in mod.named_parameters()}
    ref = {k[2:]: v for k, v in g.items() if k.startswith("g.")}
    check_grads(params, ref, 2e-3 if f32 else (0.4 if "skr" in name else 0.12) * (0.2 if h16 else 1.0), name, l2=not f32)
    if f32:
        sd = mod.state_dict()
        for k, v in sd_of(g, "sd_after.").items():
            check(sd[k].float(), v.float(), 1e-5, f"{name}.buffer.{k}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_poe_all_subsets_and_drop(dtype):
    g = load("stage_poe")
    mu, lv = g["mu"], g["logvar"]                       # (5,N,L,d,h,w), index 0 = prior
    n, L = mu.shape[1], mu.shape[2]
    feat = torch.cat([torch.cat([mu[m + 1], lv[m + 1]], 1) for m in range(4)], 1).to(DEV, dtype).contiguous()
    tol = 1e-5 if dtype == torch.float32 else (4e-3 if dtype == torch.float16 else 3e-2)
    # the fixture's logvar is not clipped; the stage clips at +-50 like RA_HVED.py:580 (no value exceeds it here)
    assert lv.abs().max() < 50
    for idx, subset in enumerate(X.SUBSETS_MODALITIES):
        keep = torch.tensor([[1.0 if k in subset else 0.0 for k in range(4)]] * n, device=DEV)
        z, ms, ls = X.functional.PoE.apply(feat, keep, None, L, False)
        check(z, g[f"mu_{idx}"], tol, f"poe mu subset {idx}")
        check(ms.transpose(0, 1), mu, tol, "mu stack")
        check(ls.transpose(0, 1), lv, tol, "logvar stack")
    keep = (~g["drop"]).float().to(DEV)
    eps = g["eps"].to(DEV, dtype)
    z, ms, ls = X.functional.PoE.apply(feat, keep, eps, L, True)
    check(z, g["z"], tol if dtype == torch.float32 else (8e-3 if dtype == torch.float16 else 6e-2), "z with eps (instance missing)")
    check(ms.transpose(0, 1), g["mu_after"], tol, "masked mu stack")


def test_poe_backward_matches_oracle():
    torch.manual_seed(3)
    n, L, s = 2, 2, 4
    feat = (torch.randn(n, 4 * 2 * L, s, s, s) * 2).double()
    feat[0, 3] = 60.0            # exercise the clip mask
    eps = torch.randn(n, L, s, s, s).double()
    drop = torch.tensor([[False, True, False, False], [True, False, False, True]])
    wz, wm, wl = rnd((n, L, s, s, s), 1).double(), rnd((n, 5, L, s, s, s), 2).double(), rnd((n, 5, L, s, s, s), 3).double()

    def oracle(f):
        f5 = f.view(n, 4, 2 * L, s, s, s)
        mu = torch.stack([torch.zeros_like(f5[:, 0, :L])] + [f5[:, m, :L] for m in range(4)], 0)
        lv = torch.stack([torch.zeros_like(f5[:, 0, :L])] + [O.clip_logvar(f5[:, m, L:]) for m in range(4)], 0)
        pm, pl, mum = O.product_of_experts_drop(mu, lv, drop)
        return O.reparametrize(pm, pl, eps), mum.transpose(0, 1), lv.transpose(0, 1)
    f = feat.clone().requires_grad_(True)
    z, m, l = oracle(f)
    ((z * wz).sum() + (m * wm).sum() + (l * wl).sum()).backward()
    fg = feat.float().to(DEV).requires_grad_(True)
    z2, m2, l2 = X.functional.PoE.apply(fg, (~drop).float().to(DEV), eps.float().to(DEV), L, True)
    ((z2 * wz.float().to(DEV)).sum() + (m2 * wm.float().to(DEV)).sum() + (l2 * wl.float().to(DEV)).sum()).backward()
    check(z2, z, 1e-5, "z"), check(m2, m, 1e-6, "mu"), check(l2, l, 1e-6, "lv")
    check(fg.grad, f.grad, 1e-4, "dfeat")


@pytest.mark.parametrize("shape", [(1, 4, 8, 8, 16), (2, 3, 6, 10, 14), (1, 2, 2, 2, 2), (1, 2, 40, 36, 72), (1, 1, 8, 6, 160),
                                   (2, 2, 5, 70, 24)])
def test_maxpool_upsample_roundtrip(shape):
    torch.manual_seed(0)
    x = torch.randn(shape)
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)):
        xs = x.to(dtype).float()          # identical stored values on both sides
        xg = xs.to(DEV, dtype).requires_grad_(True)
        xo = xs.clone().requires_grad_(True)
        size = tuple(2 * s for s in shape[2:])
        up = X.functional.Upsample.apply(xg, size)
        upo = O.upsample_to(xo, size)
        check(up, upo, tol, "upsample")
        w = torch.randn(upo.shape)
        (up.float() * w.to(DEV)).sum().backward()
        (upo * w).sum().backward()
        check(xg.grad, xo.grad, tol * 4, "upsample grad")
        if all(s % 2 == 0 for s in shape[2:]):
            xg.grad = None
            xo.grad = None
            mp = X.functional.MaxPool2.apply(xg)
            mpo = torch.nn.functional.max_pool3d(xo, 2)
            check(mp, mpo, 0 if dtype == torch.float32 else 1e-9, "maxpool")
            w = torch.randn(mpo.shape)
            (mp.float() * w.to(DEV)).sum().backward()
            (mpo * w).sum().backward()
            if dtype == torch.float32:     # bf16 ties may pick a different (equal-valued) voxel
                check(xg.grad, xo.grad, 1e-6, "maxpool grad")


def test_upsample_arbitrary_size_is_adjoint_consistent():
    torch.manual_seed(1)
    x = torch.randn(1, 2, 5, 6, 7)
    size = (9, 13, 10)
    xg = x.to(DEV).requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    up, upo = X.functional.Upsample.apply(xg, size), O.upsample_to(xo, size)
    check(up, upo, 1e-6, "upsample")
    w = torch.randn(upo.shape)
    (up * w.to(DEV)).sum().backward()
    (upo * w).sum().backward()
    check(xg.grad, xo.grad, 1e-5, "upsample grad")


@pytest.mark.parametrize("cfg", [dict(cin=4, cout=4, sp=(12, 20, 36), groups=1), dict(cin=16, cout=32, sp=(8, 8, 8), groups=4),
                                 dict(cin=48, cout=16, sp=(4, 6, 10), groups=1, split=16), dict(cin=8, cout=16, sp=(16, 16, 16), groups=4, stride=2),
                                 dict(cin=12, cout=4, sp=(9, 7, 5), groups=1, split=4),
                                 dict(cin=16, cout=8, sp=(8, 12, 32), groups=4, stride=2),     # vectorised stride-2 fwd / dgrad
                                 dict(cin=4, cout=6, sp=(6, 4, 64), groups=1, stride=2),
                                 dict(cin=4, cout=4, sp=(7, 9, 10), groups=1, stride=2)])      # odd sizes: gather kernels
def test_in_lrelu_conv_shapes_vs_oracle(cfg):
    """Ragged / grouped / virtual-concat / strided shapes of the fused IN->LeakyReLU->conv stage vs stock ops."""
    torch.manual_seed(7)
    n, cin, cout, g, stride = 2, cfg["cin"], cfg["cout"], cfg["groups"], cfg.get("stride", 1)
    x = torch.randn((n, cin) + cfg["sp"]) * 2 + 0.5
    ws = [torch.randn(cout // g, cin // g, 3, 3, 3) * 0.2 for _ in range(g)]
    bs = [torch.randn(cout // g) for _ in range(g)]
    xo = x.clone().requires_grad_(True)
    wo = [w.clone().requires_grad_(True) for w in ws]
    bo = [b.clone().requires_grad_(True) for b in bs]
    h = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(xo, eps=1e-5), 0.01)
    yo = torch.nn.functional.conv3d(h, torch.cat(wo, 0), torch.cat(bo, 0), stride=stride, padding=1, groups=g)
    wgt = torch.randn(yo.shape)
    (yo * wgt).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    wg = [w.to(DEV).requires_grad_(True) for w in ws]
    bg = [b.to(DEV).requires_grad_(True) for b in bs]
    if "split" in cfg:
        xa, xb = xg[:, :cfg["split"]], xg[:, cfg["split"]:]
    else:
        xa, xb = xg, None
    y = X.functional.in_lrelu_conv(xa, xb, wg, bg, stride, g)
    (y * wgt.to(DEV)).sum().backward()
    check(y, yo, 5e-5, "y"), check(xg.grad, xo.grad, 1e-3, "dx")
    for a, b in zip(wg + bg, wo + bo):
        check(a.grad, b.grad, 1e-3, "dparam")


@pytest.mark.parametrize("shape", [(1, 4, 32, 32, 32), (2, 3, 16, 40, 48), (1, 2, 9, 70, 160), (1, 5, 8, 33, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_depthwise_k3_sliding_window_kernel_vs_stock(shape, dtype):
    """Depthwise 3^3 conv (BasicConv conv_blocks, ResBlock dwconvs) on volumes large enough for the sliding-window kernel:
    BasicConv forward (conv -> InstanceNorm -> LeakyReLU) and backward against stock fp32 ops."""
    torch.manual_seed(5)
    n, c = shape[:2]
    x = (torch.randn(shape) * 1.3 + 0.2).to(dtype).float()
    w = torch.randn(c, 1, 3, 3, 3) * 0.3
    # 16-bit storage: where the volume allows it the conv runs on the matrix cores (quad-channel kernel, diagonal 4 x 4 weight
    # blocks) with the weights rounded to the storage type, as for every other MFMA conv -- the comparator gets the same weights
    w = w.to(dtype).float()
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yo = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(
        torch.nn.functional.conv3d(xo, wo, None, padding=1, groups=c), eps=1e-5), 0.01)
    g = torch.randn(yo.shape)
    (yo * g).sum().backward()
    m = X.blocks.BasicConv(c, c, 3, padding=1, groups=c)
    with torch.no_grad():
        m.conv.weight.copy_(w)
    m = m.to(DEV)
    xg = x.to(DEV, dtype).requires_grad_(True)
    y = m(xg)
    (y.float() * g.to(DEV)).sum().backward()
    if dtype == torch.float32:
        check(y, yo, 2e-5, "dw fwd"), check(xg.grad, xo.grad, 2e-4, "dw dx"), check(m.conv.weight.grad, wo.grad, 2e-4, "dw dw")
    else:
        assert l2_err(y, yo) < 8e-3 and l2_err(xg.grad, xo.grad) < 3e-2 and l2_err(m.conv.weight.grad, wo.grad) < 3e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_in_affine_act_fused_finalisation_matches_two_launches(dtype):
    """xh_in_affine_act (InstanceNorm finalisation inside the activation pass) against xh_norm_finalize + xh_affine_act and
    against InstanceNorm3d + LeakyReLU in fp64; odd volume, two samples."""
    from xlstm_hved_amd import ops
    torch.manual_seed(4)
    n, c, sp = 2, 5, (7, 9, 24)
    x = (torch.randn(n, c, *sp) * 3.0 + 1.5).to(dtype).to(DEV)
    red = torch.zeros(n, c, 2, dtype=torch.float64, device=DEV)
    ops.moments(x, red)
    y, sc, sh, mean, rstd = ops.in_affine_act(x, red, ops.ACT_LRELU, 0.01)
    sc2, sh2, mean2, rstd2 = ops.norm_finalize(ops.MODE_IN, red, n, c, sp[0] * sp[1] * sp[2])
    y2 = ops.affine_act(x, sc2, sh2, ops.ACT_LRELU, 0.01)
    torch.cuda.synchronize()
    for a, b in ((sc, sc2), (sh, sh2), (mean, mean2), (rstd, rstd2)):
        assert ((a - b).abs() <= 4e-7 * b.abs() + 1e-7).all()
    xd = x.double().cpu()
    want = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(xd, eps=1e-5), 0.01)
    tol = 2e-6 if dtype == torch.float32 else (1e-2 if dtype == torch.bfloat16 else 1.5e-3)
    assert l2_err(y.double().cpu(), want) < tol and l2_err(y2.double().cpu(), want) < tol
    assert (y.float() - y2.float()).abs().max().item() <= (2e-6 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stride2_conv_channel_split_matches_unsplit(dtype):
    """Deep-level stride-2 convs (few output voxels, many channels) split the input channels over the thread groups of a block;
    xh_set_option(2, 256) switches the split off.  Same sums in a different order; also against stock conv3d."""
    from xlstm_hved_amd import ops
    torch.manual_seed(6)
    lib = X._lib.load()
    n, cin, cout, sp = 1, 32, 64, (16, 16, 16)
    x = torch.randn(n, cin, *sp, device=DEV).to(dtype)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV) * (2.0 / (27 * cin)) ** 0.5
    b = torch.randn(cout, device=DEV)
    outs = []
    for opt in (0, 256):
        lib.xh_set_option(2, opt)
        try:
            red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
            outs.append((ops.conv3d(x, None, [w], [b], k=3, cout=cout, stride=2, epi=2, red=red).float(), red))
        finally:
            lib.xh_set_option(2, 0)
    (y, r), (y0, r0) = outs
    ref = torch.nn.functional.conv3d(x.float(), w, b, stride=2, padding=1)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert l2_err(y, ref) < tol and l2_err(y0, ref) < tol and l2_err(y, y0) < tol
    assert ((r - r0).abs() <= 1e-3 * r0.abs() + 1e-2).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 8, 8, 12, 16), (1, 16, 16, 16, 32)], ids=["n2c8", "n1c16"])
def test_fused_gate_maxpool_equals_gate_then_maxpool_then_moments(shape, dtype):
    """ops.gate_maxpool / gate_maxpool_bwd (the skip-return gate, the next encoder's MaxPool3d(2) and the pooled tensor's channel
    sums in one pass; RA_HVED.py:552, buildingblocks.py:655-657) against the three separate launches: bit-identical pooled
    values, sums to fp64 round-off, identical gradients (the arg-max is recomputed from the same rounded products)."""
    from xlstm_hved_amd import ops
    torch.manual_seed(4)
    n, c, d, h, w = shape
    x = torch.randn(shape, device="cuda").to(dtype)
    a = torch.sigmoid(torch.randn(n, 1, d, h, w, device="cuda")).to(dtype)
    x[0, 0, :2, :2, :8] = 0.25                            # ties inside windows: the FIRST maximum in scan order must win in both
    assert ops.gate_maxpool_ok(x, a)
    red = torch.zeros(n, c, 2, dtype=torch.float64, device="cuda")
    y = ops.gate_maxpool(x, a, red)
    g = ops.gate(x, a)
    y_ref = ops.maxpool2(g)
    assert torch.equal(y, y_ref)
    red_ref = torch.zeros_like(red)
    ops.moments(y_ref, red_ref, 0)
    assert ((red - red_ref).abs() / red_ref.abs().clamp_min(1.0)).max().item() < 1e-7    # fp32 over a lane's 4 values, fp64 from there
    dy = torch.randn_like(y)
    dx, ds = ops.gate_maxpool_bwd(x, a, dy)
    dg = ops.maxpool2_bwd(g, dy)
    dx_ref, ds_ref = ops.gate_bwd(x, a, dg)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    # ds sums the routed contributions over channels: the fused kernel adds four per-wave partial sums, the separate one walks
    # the channels in order -- fp32 round-off before the rounding to storage
    assert (ds.float() - ds_ref.float()).abs().max().item() <= 1e-2 * ds_ref.float().abs().max().item() * (1e-4 if dtype == torch.float32 else 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 5, 7, 32), (1, 16, 16, 16), (1, 4, 6, 128), (1, 3, 5, 8)])
def test_one_two_channel_k3_conv_stencil_kernels_vs_stock(shape, dtype):
    """DuSEAttention's spatial conv (buildingblocks.py:313-327: 1 -> 2 channels + sigmoid), its data gradient (2 -> 1) and its
    weight gradient run on the streaming stencil kernels of csrc/conv3_tiny.hip; against torch fp32 on the stored values."""
    import torch.nn.functional as F
    ops = X.ops
    n, d, h, w = shape
    torch.manual_seed(5)
    x = torch.randn(n, 1, d, h, w, device=DEV).to(dtype)
    wt = torch.randn(2, 1, 3, 3, 3, device=DEV) * 0.3
    b = torch.randn(2, device=DEV)
    tol = 1e-5 if dtype == torch.float32 else (1.2e-2 if dtype == torch.bfloat16 else 2e-3)
    y = ops.conv3d(x, None, [wt], [b], k=3, cout=2, act=X.ops.ACT_SIGMOID)
    vw = 4 if dtype == torch.float32 else 8
    if w % vw == 0 and (w // vw) <= 64 and 64 % (w // vw) == 0:
        assert ops.last_conv_kernel().startswith("conv3_tiny_kernel<1 -> 2"), ops.last_conv_kernel()
    ref = torch.sigmoid(F.conv3d(x.float(), wt, b, padding=1))
    assert l2_err(y.float(), ref) < tol
    dy = torch.randn(n, 2, d, h, w, device=DEV).to(dtype)
    dx = ops.conv3d(dy, None, [wt], None, k=3, cout=1, transposed=True)
    ref_dx = F.conv_transpose3d(dy.float(), wt, padding=1)
    assert l2_err(dx.float(), ref_dx) < tol
    dw, db = torch.zeros_like(wt), torch.zeros_like(b)
    ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3)
    xr = x.float().requires_grad_(False)
    wr = wt.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    (F.conv3d(xr, wr, br, padding=1) * dy.float()).sum().backward()
    assert l2_err(dw, wr.grad) < 1e-4 and l2_err(db, br.grad) < 1e-4
    ops.conv3d_wgrad(x, None, dy, [dw], [db], k=3)                      # accumulates
    assert l2_err(dw, 2 * wr.grad) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_producers_that_leave_the_next_instance_norms_sums_match_a_moments_pass(dtype):
    """Three passes that also accumulate the channel sums (sum y, sum y^2) of what they store, so that the InstanceNorm of the conv
    behind them needs no moments launch: the max-pool of the skip-path encoders (xh_gate_maxpool_fwd without a gate), AttenModule2's
    gate + concat (xh_gate2_fwd) -- both bit-identical in their outputs to the plain passes, sums equal to xh_moments of the output --
    and the stride-2 conv that finalises its input's InstanceNorm itself (xh_conv_ptrs.fin_red) against the two-launch form."""
    ops = X.ops
    torch.manual_seed(7)
    n, c, d, h, w = 2, 8, 8, 12, 32
    x = torch.randn(n, c, d, h, w, device=DEV).to(dtype)
    # --- max-pool + sums
    assert ops.gate_maxpool_ok(x, None)
    red = torch.zeros(n, c, 2, dtype=torch.float64, device=DEV)
    y = ops.gate_maxpool(x, None, red)
    y_ref = ops.maxpool2(x)
    assert torch.equal(y, y_ref)
    red_ref = torch.zeros_like(red)
    ops.moments(y_ref, red_ref, 0)
    assert ((red - red_ref).abs() / red_ref.abs().clamp_min(1.0)).max().item() < 1e-7
    # --- gate + concat + sums
    b = torch.randn(n, 4, d, h, w, device=DEV).to(dtype)
    E = torch.sigmoid(torch.randn(n, 2, d, h, w, device=DEV)).to(dtype)
    red2 = torch.zeros(n, c + 4, 2, dtype=torch.float64, device=DEV)
    g = ops.gate2(x, b, E, red2)
    g_ref = ops.gate2(x, b, E)
    assert torch.equal(g, g_ref)
    red2_ref = torch.zeros_like(red2)
    ops.moments(g_ref, red2_ref, 0)
    assert ((red2 - red2_ref).abs() / red2_ref.abs().clamp_min(1.0)).max().item() < 1e-7
    # --- stride-2 conv with the finalisation inside (16-bit storage; fp32 storage: the library runs the finalisation launch itself)
    wts = [torch.randn(2, 2, 3, 3, 3, device=DEV) * 0.2 for _ in range(4)]
    bs = [torch.randn(2, device=DEV) for _ in range(4)]
    st = torch.zeros(n, c, 2, dtype=torch.float64, device=DEV)
    ops.moments(x, st, 0)
    cnt = d * h * w
    out = ops.conv3d(x, None, wts, bs, k=3, cout=8, stride=2, groups=4, in_stats=(st, cnt, 0.01))
    yf, sc, sh, mean, rstd = out
    sc_r, sh_r, mean_r, rstd_r = ops.norm_finalize(0, st, n, c, cnt)
    y2 = ops.conv3d(x, None, wts, bs, k=3, cout=8, stride=2, groups=4, pre=(sc_r, sh_r, 0.01))
    tol = 0 if dtype == torch.float32 else 2e-6            # in-kernel rsqrt + Newton step: <= 1 ulp of the fp32 scale
    assert (sc - sc_r).abs().max().item() <= tol * sc_r.abs().max().item() + 0.0
    assert (mean - mean_r).abs().max().item() <= 1e-6 and (rstd - rstd_r).abs().max().item() <= tol * rstd_r.abs().max().item()
    assert l2_err(yf.float(), y2.float()) < (1e-6 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(1, 2, 8, 16, 16, 16), (2, 4, 16, 8, 12, 32), (1, 8, 4, 32, 32, 32)], ids=["16", "ragged", "32"])
def test_vu_block_with_the_upsampling_in_one_launch_vs_stock(shape, dtype):
    """BasicConv(k=1)(x, up2x=True) = F.interpolate(LeakyReLU(InstanceNorm3d(Conv3d(x))), scale_factor=2, mode='trilinear')
    (RA_HVED.py:599-601): norm + activation applied inside the upsampling launch (xh_upsample2x_in_act_fwd), its adjoint with the
    activation-masked norm-backward sums (xh_upsample2x_bwd_act_reduce).  Against stock fp32 modules on the same values, and
    against the unfused launches (xh_set_option(2, 2): the exact-2x kernels off)."""
    import torch.nn.functional as F
    torch.manual_seed(21)
    n, cin, cout, d, h, w = shape
    m = X.blocks.BasicConv(cin, cout, 1).to(DEV)
    x = torch.randn(n, cin, d, h, w, device=DEV).to(dtype)
    wgt = torch.randn(n, cout, 2 * d, 2 * h, 2 * w, device=DEV)

    def run():
        m.zero_grad()
        xg = x.clone().requires_grad_(True)
        y = m(xg, up2x=True)
        (y.float() * wgt).sum().backward()
        return y.float(), xg.grad.float(), m.conv.weight.grad.clone()

    y1, dx1, dw1 = run()
    lib = X._lib.load()
    lib.xh_set_option(2, 2)
    try:
        y0, dx0, dw0 = run()
    finally:
        lib.xh_set_option(2, 0)
    xr = x.float().requires_grad_(True)
    wr = m.conv.weight.detach().clone().requires_grad_(True)
    yr = F.interpolate(F.leaky_relu(F.instance_norm(F.conv3d(xr, wr), eps=1e-5), 0.01), scale_factor=2, mode="trilinear")
    (yr * wgt).sum().backward()
    ty, tg = {torch.float32: (2e-5, 2e-4), torch.bfloat16: (8e-3, 3e-2), torch.float16: (1.5e-3, 6e-3)}[dtype]
    assert l2_err(y1, yr) < ty and l2_err(y1, y0) < ty, (l2_err(y1, yr), l2_err(y1, y0))
    assert l2_err(dx1, xr.grad) < tg and l2_err(dx1, dx0) < tg, (l2_err(dx1, xr.grad), l2_err(dx1, dx0))
    assert l2_err(dw1, wr.grad) < tg and l2_err(dw1, dw0) < tg, (l2_err(dw1, wr.grad), l2_err(dw1, dw0))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("c,sp_shape", [(4, (8, 16, 32)), (8, (5, 7, 9)), (16, (8, 8, 16)), (6, (4, 8, 8))], ids=["c4", "c8_ragged", "c16", "c6_two_pass"])
def test_duse_gate_backward_in_one_pass_vs_formulas(c, sp_shape, dtype):
    """xh_duse_gate_bwd for 4 / 8 / 16 channels is one voxel-major pass that also takes dsp through the sigmoid's backward
    (duse_gate_bwd_fused_kernel); other channel counts (and xh_set_option(2, 1024)) keep the row-major + voxel-major pair and a
    separate act_bwd.  Both against the formulas dx = du (1 + ch + sp), dsp = sum_c du x, dch = sum_p du x in fp64."""
    torch.manual_seed(13)
    n = 2
    x = torch.randn((n, c) + sp_shape, device=DEV).to(dtype)
    du = torch.randn((n, c) + sp_shape, device=DEV).to(dtype)
    sp = torch.sigmoid(torch.randn((n, 1) + sp_shape, device=DEV)).to(dtype)
    ch = torch.rand(n, c, device=DEV)
    lib = X._lib.load()

    def run():
        fused = X.ops.duse_gate_bwd_fuses(c)
        dsp = torch.empty_like(sp)
        dx, dch = X.ops.duse_gate_bwd(x, ch, sp, du, dsp, sigmoid_bwd=fused)
        dpre = dsp if fused else X.ops.act_bwd(dsp, sp, 3)
        return fused, dx.float(), dpre.float(), dch.clone()

    f1, dx1, dp1, dch1 = run()
    assert f1 == (c in (4, 8, 16))
    lib.xh_set_option(2, 1024)
    try:
        f0, dx0, dp0, dch0 = run()
    finally:
        lib.xh_set_option(2, 0)
    assert not f0
    xd, dud, spd = x.double(), du.double(), sp.double()
    dx_r = dud * (1 + ch.double()[:, :, None, None, None] + spd)
    dp_r = (dud * xd).sum(1, keepdim=True) * spd * (1 - spd)
    dch_r = (dud * xd).sum((2, 3, 4))
    tol = 1e-5 if dtype == torch.float32 else 1.2e-2
    for got in ((dx1, dp1, dch1), (dx0, dp0, dch0)):
        assert l2_err(got[0], dx_r.float()) < tol and l2_err(got[1], dp_r.float()) < tol
        assert (got[2] - dch_r).abs().max().item() <= 1e-4 * dch_r.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_poe_of_all_levels_in_one_launch_equals_the_per_level_launches(dtype):
    """xh_poe_multi (functional.PoEAll): the PoE / reparameterisation of several latent levels and its backward, bit for bit what
    xh_poe_fwd / xh_poe_bwd give level by level -- with and without noise, with a dropped modality, batch 2."""
    torch.manual_seed(17)
    n = 2
    shapes = [(1, (8, 8, 8)), (2, (4, 6, 8)), (4, (4, 4, 4)), (8, (2, 2, 2))]
    feats = [torch.randn((n, 8 * L_) + sp, device=DEV).to(dtype) for L_, sp in shapes]
    keep = torch.tensor([[1.0, 0.0, 1.0, 1.0], [1.0, 1.0, 1.0, 0.0]], device=DEV)
    Ls = [L_ for L_, _ in shapes]
    for with_eps in (True, False):
        epss = [torch.randn((n, L_) + sp, device=DEV).to(dtype) if with_eps else None for L_, sp in shapes]
        for mask_mu in (False, True):
            multi = X.ops.poe_fwd_multi(feats, keep, epss, Ls, mask_mu)
            dzs = [torch.randn_like(o[0]) for o in multi]
            dmus = [torch.randn_like(o[1]) for o in multi]
            dlvs = [torch.randn_like(o[2]) for o in multi]
            dmulti = X.ops.poe_bwd_multi(feats, keep, epss, dzs, dmus, dlvs, Ls, mask_mu)
            for l in range(len(shapes)):
                single = X.ops.poe_fwd(feats[l], keep, epss[l], Ls[l], mask_mu)
                for a, b in zip(multi[l], single):
                    assert torch.equal(a, b)
                assert torch.equal(dmulti[l], X.ops.poe_bwd(feats[l], keep, epss[l], dzs[l], dmus[l], dlvs[l], Ls[l], mask_mu))

"""-m gpu: the HIP Discriminator (csrc/dconv.hip: channels-last implicit-GEMM MFMA convolutions, forward / data gradient /
weight gradient, fused InstanceNorm statistics) for ks = 4 (train.py:146) and ks = 3 (class default):
  * whole network against the fixture generated from the REAL reference class (tests/golden/stage_disc_ks*.npz), fp32 reference
    values, on the same 16-bit-rounded input;
  * single convolution entry points against F.conv3d on ragged extents;
  * every layer of the full 128^3 chain (ks = 4: 127 -> 63 -> 31 -> 15 -> 14) against stock fp32 conv3d on the same 16-bit inputs."""
import pytest
import torch
import torch.nn.functional as F

import disc_common as DC
from gpu_common import l2_err

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402
from xlstm_hved_amd import disc as D  # noqa: E402

DEV = "cuda"
DT = [torch.bfloat16, torch.float16]


def _cl(t):            # NCDHW -> channels-last contiguous
    return t.permute(0, 2, 3, 4, 1).contiguous()


def _nc(t):
    return t.permute(0, 4, 1, 2, 3).contiguous()


def _conv_case(n, cin, cout, s, sp, ks, dtype, seed=5, check_halo_variants=True):
    """One convolution through the three entry points (forward + statistics, data gradient, weight gradient) vs F.conv3d in
    fp32 on the same 16-bit-rounded operands.  Returns the relative L2 errors."""
    torch.manual_seed(seed)
    cpad = 8 if cin == 7 else cin
    x = torch.randn(n, cin, *sp).to(dtype)
    w = (torch.randn(cout, cin, ks, ks, ks) * (2.0 / (ks ** 3 * cin)) ** 0.5)
    b = torch.randn(cout)
    xo, wo, bo = x.float().requires_grad_(True), w.to(dtype).float().requires_grad_(True), b.clone().requires_grad_(True)
    yo = F.conv3d(xo, wo, bo, stride=s, padding=1)
    gy = torch.randn_like(yo).to(dtype)
    (yo * gy.float()).sum().backward()
    spo = tuple(yo.shape[2:])
    assert spo == D.conv_out(sp, ks, s)
    xp = torch.zeros(n, cpad, *sp, dtype=dtype)
    xp[:, :cin] = x
    xcl = _cl(xp).to(DEV)
    wd = w.to(DEV)
    red = torch.zeros(n, cout, 2, dtype=torch.float64, device=DEV)
    y = D._conv(xcl, D._pack(wd, 2 if cin == 7 else 0, cout, cpad, dtype), b.to(DEV), 0, s, n, sp, spo, cpad, cout, red=red, ks=ks)
    torch.cuda.synchronize()
    e = dict(y=l2_err(_nc(y.cpu()), yo))
    ys = _nc(y.cpu()).double()
    e["s0"] = ((red[..., 0].cpu() - ys.sum((2, 3, 4))).abs() / ys.abs().sum((2, 3, 4))).max().item()
    e["s1"] = ((red[..., 1].cpu() - (ys * ys).sum((2, 3, 4))).abs() / (ys * ys).sum((2, 3, 4))).max().item()
    k16 = 1.0 if dtype == torch.bfloat16 else 0.2
    if cin == 7 and check_halo_variants:
        # without statistics the first conv takes the LDS-halo kernel (bias + LeakyReLU fused); ragged block edges here
        y2 = D._conv(xcl, D._pack(wd, 2, cout, cpad, dtype), b.to(DEV), 0, s, n, sp, spo, cpad, cout, act=D.L.ACT_LRELU, ks=ks)
        e_y2 = l2_err(_nc(y2.cpu()), F.leaky_relu(yo, D.SLOPE))
        assert e_y2 < 6e-3 * k16, e_y2
        D.L.load().xh_set_option(14, 2048)                # ... and agrees with the generic kernel
        y3 = D._conv(xcl, D._pack(wd, 2, cout, cpad, dtype), b.to(DEV), 0, s, n, sp, spo, cpad, cout, act=D.L.ACT_LRELU, ks=ks)
        D.L.load().xh_set_option(14, 0)
        assert l2_err(y2.float().cpu(), y3.float().cpu()) < 4e-3 * k16
    # data gradient: dY padded to a multiple of 32 channels
    cop = (cout + 31) // 32 * 32
    gyp = torch.zeros(n, cop, *spo, dtype=dtype)
    gyp[:, :cout] = gy
    gcl = _cl(gyp).to(DEV)
    dx = D._conv(gcl, D._pack(wd, 1, cop, cpad, dtype), None, 1, s, n, spo, sp, cop, cpad, ks=ks)
    e["dx"] = l2_err(_nc(dx.cpu())[:, :cin], xo.grad)
    if s == 2 and cpad % 64 == 0:                         # source-block data gradient (forced: small volumes) vs the gather kernel
        D.L.load().xh_set_option(14, 32768)
        dx2 = D._conv(gcl, D._pack(wd, 1, cop, cpad, dtype), None, 1, s, n, spo, sp, cop, cpad, ks=ks)
        D.L.load().xh_set_option(14, 0)
        assert l2_err(dx.float().cpu(), dx2.float().cpu()) < 4e-3 * k16
        assert l2_err(_nc(dx2.cpu())[:, :cin], xo.grad) < 8e-3 * k16
    if cin == 7 and cout == 64 and check_halo_variants:   # LDS-halo data gradient vs the generic kernel
        D.L.load().xh_set_option(14, 1024)
        dx2 = D._conv(gcl, D._pack(wd, 1, cop, cpad, dtype), None, 1, s, n, spo, sp, cop, cpad, ks=ks)
        D.L.load().xh_set_option(14, 0)
        assert l2_err(dx.float().cpu(), dx2.float().cpu()) < 4e-3 * k16
    # weight gradient (dY channels padded to a multiple of 8)
    co8 = (cout + 7) // 8 * 8
    g8 = torch.zeros(n, co8, *spo, dtype=dtype)
    g8[:, :cout] = gy
    g8cl = _cl(g8).to(DEV)
    dwp = D._wgrad(xcl, g8cl, s, n, sp, spo, cpad, co8, ks=ks)
    dw = torch.zeros_like(wd)
    D._unpack(dwp, dw, co8, cpad)
    torch.cuda.synchronize()
    e["dw"] = l2_err(dw.cpu(), wo.grad)
    if s == 2 and ks == 4 and cpad % 64 == 0 and co8 % 128 == 0:   # source-block weight gradient (default) vs the gather kernel
        assert e["dw"] < 8e-3 * k16
        D.L.load().xh_set_option(14, 131072)
        dw2 = torch.zeros_like(wd)
        D._unpack(D._wgrad(xcl, g8cl, s, n, sp, spo, cpad, co8, ks=ks), dw2, co8, cpad)
        D.L.load().xh_set_option(14, 0)
        assert l2_err(dw2.cpu(), dw.cpu()) < 1e-4         # the same products in another fp32 order
    if cin == 7 and cout == 64 and check_halo_variants:   # LDS-halo weight gradient vs the generic kernel
        D.L.load().xh_set_option(14, 8192)
        dw2 = torch.zeros_like(wd)
        D._unpack(D._wgrad(xcl, g8cl, s, n, sp, spo, cpad, co8, ks=ks), dw2, co8, cpad)
        D.L.load().xh_set_option(14, 0)
        assert l2_err(dw2.cpu(), dw.cpu()) < 1e-3
    return e


CONV_CASES = [dict(cin=32, cout=64, stride=2, sp=(12, 10, 16)), dict(cin=64, cout=128, stride=2, sp=(9, 11, 14)),
              dict(cin=64, cout=32, stride=1, sp=(6, 7, 9)), dict(cin=128, cout=1, stride=1, sp=(5, 6, 7)),
              dict(cin=7, cout=64, stride=1, sp=(6, 9, 20)), dict(cin=128, cout=256, stride=2, sp=(7, 15, 13))]


@pytest.mark.parametrize("ks", [4, 3], ids=["k4", "k3"])
@pytest.mark.parametrize("dtype", DT, ids=["bf16", "f16"])
@pytest.mark.parametrize("cfg", CONV_CASES, ids=lambda c: f"{c['cin']}to{c['cout']}s{c['stride']}")
def test_dconv_forward_dgrad_wgrad_vs_stock(cfg, dtype, ks):
    e = _conv_case(2, cfg["cin"], cfg["cout"], cfg["stride"], cfg["sp"], ks, dtype)
    print(cfg, dtype, ks, {k: f"{v:.2e}" for k, v in e.items()})
    k = 1.0 if dtype == torch.bfloat16 else 0.2
    assert e["y"] < 6e-3 * k and e["dx"] < 8e-3 * k and e["dw"] < 8e-3 * k
    # channel sums: fp32 over the 64 voxels of a wave's tile, fp64 across tiles (relative to the sum of magnitudes)
    assert e["s0"] < 2e-6 and e["s1"] < 2e-6


def _hip_disc(ks, g=None):
    hip = X.Discriminator(in_channels=7, ks=ks, strides=[1, 2, 2, 2])            # the call of train.py:146
    sd = DC.seeded_disc_state(ks, g)
    assert list(hip.state_dict().keys()) == list(sd.keys())
    hip.load_state_dict(sd, strict=True)
    return hip.to(DEV)


@pytest.mark.parametrize("ks", [4, 3], ids=["k4", "k3"])
@pytest.mark.parametrize("dtype", DT + [torch.float32], ids=["bf16", "f16", "f32in"])
def test_discriminator_vs_reference_fixture(ks, dtype):
    """Whole Discriminator forward + backward against the fixture the REAL reference class produced in fp32 (make_golden.py
    disc_cases): output, input gradient, all 9 parameter gradients (strided samples + sums)."""
    g = DC.load_fixture(ks)
    hip = _hip_disc(ks, g)
    x = DC.seeded_input(g)
    xg = x.to(dtype).to(DEV).requires_grad_(True)
    y = hip(xg)
    assert tuple(y.shape) == tuple(g["f32.y"].shape) and y.dtype == dtype
    (y.float() * g["gy"].to(DEV)).sum().backward()
    torch.cuda.synchronize()
    gx = xg.grad.float().flatten().cpu()
    e = dict(y=l2_err(y, g["f32.y"]), dx=l2_err(gx[DC.sample_index(gx.numel(), 65536)], g["f32.dx"]))
    for kname, p in hip.named_parameters():
        if kname.endswith(".bias") and not kname.startswith("disc.0."):
            assert float(p.grad.abs().max()) == 0.0          # bias in front of an InstanceNorm: exactly zero here
            continue
        gr = p.grad.flatten().cpu()
        e[kname] = l2_err(gr[DC.sample_index(gr.numel())], g["f32.g." + kname])
        assert abs(gr.double().sum().item() - float(g["f32.gsum." + kname])) <= 0.05 * float(g["f32.gabs." + kname]), kname
    # The yardstick is the REFERENCE class itself under torch.autocast of the same dtype on the same weights and input
    # (make_golden.py disc_cases: amp_bf16.* / amp_f16.*; ks = 4: bf16 y 1.0e-2, dx 9.8e-2, parameter gradients 0.13, fp16 1.2e-3 /
    # 3.8e-2 / 5.1e-2): five 16-bit conv layers whose LeakyReLU(0.2) masks flip wherever the 16-bit forward changes a sign bound
    # ANY 16-bit backward from below, so the bar is "deviate no more than the reference's own mixed precision does" (measured
    # here: bf16 7.4e-3 / 8.5e-2 / <= 0.12, fp16 9e-4 / 3e-2 / <= 3.9e-2).  An fp32 input runs in fp16 inside (the exact fp32
    # route is test_discriminator_exact_fp32_route_vs_reference_fixture).
    tag = "amp_bf16" if dtype == torch.bfloat16 else "amp_f16"
    ys = {k_: float(g[f"{tag}.{k_}"]) for k_ in ("y", "dx", "g")}
    gworst = max(v for kk, v in e.items() if kk not in ("y", "dx"))
    print(ks, dtype, {k_: f"{v:.2e}" for k_, v in e.items()}, "reference under autocast:", {k_: f"{v:.2e}" for k_, v in ys.items()})
    assert e["y"] <= 1.25 * ys["y"] and e["dx"] <= 1.25 * ys["dx"] and gworst <= 1.25 * ys["g"], (e, ys)       # (two 16-bit roundings of the same net: within a quarter of each other)


def test_discriminator_rejects_what_the_reference_cannot_build():
    with pytest.raises(NotImplementedError):
        X.Discriminator(in_channels=7, ks=5)
    with pytest.raises(NotImplementedError):
        X.Discriminator(in_channels=9, ks=4)
    d = X.Discriminator(in_channels=7, ks=4, strides=[1, 2, 2, 2]).to(DEV)
    with pytest.raises(ValueError):
        d(torch.zeros(1, 7, 16, 16, 16, device=DEV, dtype=torch.bfloat16))     # 16 -> 15 -> 7 -> 3 -> 1 -> 0


FULL = [dict(cin=7, cout=64, stride=1, sp=128), dict(cin=64, cout=128, stride=2, sp=127), dict(cin=128, cout=256, stride=2, sp=63),
        dict(cin=256, cout=512, stride=2, sp=31), dict(cin=512, cout=1, stride=1, sp=15)]


@pytest.mark.parametrize("cfg", FULL, ids=lambda c: f"{c['cin']}to{c['cout']}@{c['sp']}")
def test_k4_full_size_chain_layer_vs_stock(cfg):
    """Every layer of the ks = 4 chain at the extents of a 128^3 patch (127 -> 63 -> 31 -> 15 -> 14: odd extents, ragged tiles,
    the tile plans the training step really takes) against stock fp32 conv3d on the same bf16 operands."""
    sp = (cfg["sp"],) * 3
    e = _conv_case(1, cfg["cin"], cfg["cout"], cfg["stride"], sp, 4, torch.bfloat16, seed=13, check_halo_variants=False)
    print(cfg, {k: f"{v:.2e}" for k, v in e.items()})
    assert e["y"] < 6e-3 and e["dx"] < 8e-3 and e["dw"] < 1e-2
    assert e["s0"] < 2e-6 and e["s1"] < 2e-6


@pytest.mark.parametrize("mode", ["forward", "dgrad_s2"])
@pytest.mark.parametrize("ks", [4, 3], ids=["k4", "k3"])
def test_dconv_256x128_tiles_match_128x128_tiles(mode, ks):
    """Launches of >= 1024 row tiles take 256 x 128 workgroup tiles (128 x 64 per wave); same numbers as the 128 x 128 plan,
    and the forward agrees with stock conv3d on a sample of output planes."""
    torch.manual_seed(9)
    dtype = torch.bfloat16
    lib = D.L.load()
    if mode == "forward":
        cs, cn, s, sp = 64, 128, 1, (32, 64, 64)
        spo = D.conv_out(sp, ks, s)
        x = torch.randn(1, *sp, cs, device=DEV).to(dtype)
        w = torch.randn(cn, cs, ks, ks, ks, device=DEV) * (2.0 / (ks ** 3 * cs)) ** 0.5
        wp = D._pack(w, 0, cn, cs, dtype)
        outs = []
        for opt in (0, 256):
            lib.xh_set_option(14, opt)
            red = torch.zeros(1, cn, 2, dtype=torch.float64, device=DEV)
            outs.append((D._conv(x, wp, None, 0, s, 1, sp, spo, cs, cn, red=red, ks=ks), red))
        lib.xh_set_option(14, 0)
        (y_a, r_a), (y_b, r_b) = outs
        assert torch.equal(y_a, y_b)
        assert ((r_a - r_b).abs() / r_b.abs().clamp_min(1.0)).max().item() < 1e-5
        ref = F.conv3d(_nc(x[:, 10:10 + ks + 1].float().cpu()), w.to(dtype).float().cpu(), padding=(0, 1, 1))     # output planes 11, 12
        assert l2_err(_nc(y_a[:, 11:13].float().cpu()), ref) < 6e-3
    else:
        cs, cn, s = 256, 128, 2
        sp = (32, 64, 64) if ks == 3 else (31, 63, 63)
        spo = D.conv_out(sp, ks, s)
        gy = torch.randn(1, *spo, cs, device=DEV).to(dtype)
        w = torch.randn(cs, cn, ks, ks, ks, device=DEV) * (2.0 / (ks ** 3 * cn)) ** 0.5
        wpt = D._pack(w, 1, cs, cn, dtype)
        outs = []
        for opt in (16384, 16384 | 256, 0):                 # bit 16384: the gather kernel instead of the source-block kernel (round 5)
            lib.xh_set_option(14, opt)
            outs.append(D._conv(gy, wpt, None, 1, s, 1, spo, sp, cs, cn, ks=ks))
        lib.xh_set_option(14, 0)
        assert torch.equal(outs[0], outs[1])
        # the source-block kernel (default) sums the same products slice by slice instead of tap by tap: fp32 round-off before the
        # 16-bit rounding of the result
        assert l2_err(outs[2].float().cpu(), outs[0].float().cpu()) < 2e-3
        ref = F.conv_transpose3d(_nc(gy.float().cpu()), w.to(dtype).float().cpu(), stride=2, padding=1,
                                 output_padding=tuple(sp[i] - ((spo[i] - 1) * 2 - 2 + ks) for i in range(3)))
        assert l2_err(_nc(outs[2].float().cpu()), ref) < 8e-3


@pytest.mark.parametrize("ks", [4, 3], ids=["k4", "k3"])
def test_discriminator_exact_fp32_route_vs_reference_fixture(ks):
    """Discriminator.fp32_exact = True: fp32 activations, direct fp32 convolutions (csrc/dconv.hip: dconv_exact_*), the generic fp32
    norm / activation passes -- no 16-bit operand anywhere.  Against the fp32 run of the REAL reference class (the same fixture):
    output, input gradient and every parameter gradient at fp32 round-off."""
    g = DC.load_fixture(ks)
    hip = _hip_disc(ks, g)
    hip.fp32_exact = True
    x = DC.seeded_input(g)
    xg = x.float().to(DEV).requires_grad_(True)
    y = hip(xg)
    assert tuple(y.shape) == tuple(g["f32.y"].shape) and y.dtype == torch.float32
    (y * g["gy"].to(DEV)).sum().backward()
    torch.cuda.synchronize()
    gx = xg.grad.flatten().cpu()
    e = dict(y=l2_err(y, g["f32.y"]), dx=l2_err(gx[DC.sample_index(gx.numel(), 65536)], g["f32.dx"]))
    gabs = max(float(g["f32.gabs." + kname]) / p.numel() for kname, p in hip.named_parameters())
    for kname, p in hip.named_parameters():
        gr = p.grad.flatten().cpu()
        if kname.endswith(".bias") and not kname.startswith("disc.0."):
            assert float(gr.abs().max()) <= 1e-3 * max(gabs, 1e-12) * 1e3        # round-off around an exact zero, like the reference
            continue
        e[kname] = l2_err(gr[DC.sample_index(gr.numel())], g["f32.g." + kname])
        assert abs(gr.double().sum().item() - float(g["f32.gsum." + kname])) <= 1e-3 * float(g["f32.gabs." + kname]), kname
    print(ks, "exact fp32", {k_: f"{v:.2e}" for k_, v in e.items()})
    # fp32 against fp32 (different summation orders; a LeakyReLU mask flips where a pre-activation is within round-off of zero)
    assert e["y"] < 1e-4 and e["dx"] < 2e-3 and all(v < 2e-3 for kk, v in e.items() if kk not in ("y", "dx")), e


@pytest.mark.parametrize("dtype", DT, ids=["bf16", "f16"])
@pytest.mark.parametrize("ks", [3, 4])
def test_weight_images_through_lds_tiles_equal_the_element_pack(dtype, ks):
    """xh_dconv_pack modes 0 / 1 on multiples of 64 channels go through dpack_tile_kernel: the same bytes as the element-per-thread
    kernel (xh_set_option(14, 4096)), for the layer shapes of the network and the one-output head."""
    torch.manual_seed(11)
    for cout, cin in ((128, 64), (256, 128), (1, 512), (64, 192)):
        w = torch.randn(cout, cin, ks, ks, ks, device=DEV)
        for mode in (0, 1):
            if mode == 1 and cout % 64:
                continue
            a = D._pack(w, mode, cout, cin, dtype)
            try:
                D.L.load().xh_set_option(14, 4096)
                b = D._pack(w, mode, cout, cin, dtype)
            finally:
                D.L.load().xh_set_option(14, 0)
            torch.cuda.synchronize()
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (cout, cin, mode)
    # ... and the way back: packed fp32 weight gradient -> dw[co][ci][tap] +=, tile kernel against the element kernel
    for cout, cin in ((128, 64), (512, 256)):
        dwp = torch.randn(ks ** 3 * cout * cin, device=DEV)
        base = torch.randn(cout, cin, ks, ks, ks, device=DEV)
        a, b = base.clone(), base.clone()
        D._unpack(dwp, a, cout, cin)
        try:
            D.L.load().xh_set_option(14, 4096)
            D._unpack(dwp, b, cout, cin)
        finally:
            D.L.load().xh_set_option(14, 0)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        assert torch.allclose(a - base, dwp.view(ks ** 3, cout, cin).permute(1, 2, 0).reshape(cout, cin, ks, ks, ks), atol=1e-6)
    # the reducing activation backward with its capped grid (<= 256 workgroups per sample) against the sums taken by torch
    n, c, v = 2, 64, 40 * 40 * 40
    dy = torch.randn(n, v, c, device=DEV).to(dtype)
    x = torch.randn(n, v, c, device=DEV).to(dtype)
    sc, sh = torch.rand(n, c, device=DEV) + 0.5, torch.randn(n, c, device=DEV) * 0.1
    red = torch.zeros(n, c, 2, dtype=torch.float64, device=DEV)
    D.L.check(D.L.load().xh_cl_act_bwd(D._s(), X.ops._dt(dy), 0, dy.data_ptr(), x.data_ptr(), None, sc.data_ptr(), sh.data_ptr(), D.SLOPE,
                                       None, None, None, red.data_ptr(), n, c, v), "xh_cl_act_bwd")
    g = dy.double() * torch.where(x.double() * sc.double()[:, None] + sh.double()[:, None] > 0, 1.0, D.SLOPE)
    ref = torch.stack([g.sum(1), (g * x.double()).sum(1)], -1)
    torch.cuda.synchronize()
    assert ((red - ref).abs().max() / ref.abs().max()).item() < 1e-5

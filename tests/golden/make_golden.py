"""Generates the golden vectors under tests/golden/ by running the REAL reference.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
It (1) imports the reference through tools/ref_shim.py, (2) runs reference modules / the whole
XLSTM_HVED network on seeded inputs, (3) stores inputs + expected outputs + expected gradients as
.npz, and (4) replays every fixture through oracle/xlstm_hved_oracle.py and fails if the oracle
disagrees, so a committed fixture set always certifies the committed oracle.

Fixtures are data only (inputs, weights drawn by the reference's own initialisers, outputs).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402
import xlstm_hved_oracle as O  # noqa: E402
from seeded_weights import seeded_state, entries_of  # noqa: E402

torch.set_num_threads(8)
ns = ref_shim.load_reference()
R = ns.RA_HVED
import buildingblocks as BB  # noqa: E402  (reference module, on sys.path via the shim)
from modules.DuSFE import DuSEAttention  # noqa: E402
from sa_modules.sa_module import ResBlock  # noqa: E402
from sa_modules.attention_blocks import SpacialAttention3D  # noqa: E402
from UxLSTM.nnunetv2.nets.UxLSTMEnc_3d import ViLLayer  # noqa: E402


def rnd(shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np_(v) if torch.is_tensor(v) else np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def check(tag, a, b, tol):
    err = (a.double() - b.double()).abs().max().item()
    ref = max(b.double().abs().max().item(), 1e-30)
    assert err <= tol * max(ref, 1.0), f"{tag}: oracle deviates from reference by {err:.3e} (ref absmax {ref:.3e})"
    return err


def module_case(name, mod, inputs, oracle_fn, init_seed=3, train=True, extra=None, tol=2e-5):
    """Runs a reference nn.Module fwd+bwd with loss = sum(out_i * w_i), stores everything, and replays it
    through the oracle function (oracle_fn(P(sd), *inputs) -> tensor or tuple)."""
    torch.manual_seed(init_seed)
    mod.apply(ns.utils.init_weights)
    for p_ in mod.parameters():     # move BatchNorm/affine parameters off their symmetric defaults
        if p_.dim() == 1:
            with torch.no_grad():
                p_.add_(0.1 * rnd(p_.shape, 17 + p_.numel()))
    mod.train(train)
    sd0 = {k: v.clone() for k, v in mod.state_dict().items()}
    ins = [i.clone().requires_grad_(True) for i in inputs]
    out = mod(*ins)
    outs = list(out) if isinstance(out, (tuple, list)) else [out]
    ws = [rnd(o.shape, 100 + i) for i, o in enumerate(outs)]
    loss = sum((o * w).sum() for o, w in zip(outs, ws))
    loss.backward()
    arrs = {}
    for k, v in sd0.items():
        arrs["sd." + k] = v
    for k, v in mod.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrs["sd_after." + k] = v
    for i, t in enumerate(ins):
        arrs[f"in{i}"] = t
        arrs[f"gin{i}"] = t.grad
    for i, (o, w) in enumerate(zip(outs, ws)):
        arrs[f"out{i}"] = o
        arrs[f"w{i}"] = w
    for k, p_ in mod.named_parameters():
        if p_.grad is not None:
            arrs["g." + k] = p_.grad
    if extra:
        arrs.update(extra)
    save(name, **arrs)
    # replay through the oracle
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd0.items()}
    ins2 = [i.clone().requires_grad_(True) for i in inputs]
    out2 = oracle_fn(O.P(sd), *ins2)
    outs2 = list(out2) if isinstance(out2, (tuple, list)) else [out2]
    sum((o * w).sum() for o, w in zip(outs2, ws)).backward()
    for i, (a, b) in enumerate(zip(outs2, outs)):
        check(f"{name}.out{i}", a, b, tol)
    for a, b in zip(ins2, ins):
        check(f"{name}.gin", a.grad, b.grad, tol * 10)
    for k, p_ in mod.named_parameters():
        if p_.grad is not None:
            check(f"{name}.g.{k}", sd[k].grad, p_.grad, tol * 10)
    print(f"  oracle replay of {name}: ok")


# ---------------------------------------------------------------------------------- per-stage cases
def stage_cases():
    module_case("stage_singleconv_ilc", BB.SingleConv(4, 8, 3, 1, "ilc", 8, padding=1),
                [rnd((2, 4, 8, 12, 16), 1)], lambda p, x: O.single_conv(p, x, "ilc"))
    module_case("stage_singleconv_ilc_s2", BB.SingleConv(8, 4, 3, 2, "ilc", 8, padding=1),
                [rnd((1, 8, 8, 8, 16), 2)], lambda p, x: O.single_conv(p, x, "ilc", stride=2))
    module_case("stage_singleconv_gcr", BB.SingleConv(16, 8, 3, 1, "gcr", 8, padding=1),
                [rnd((2, 16, 8, 8, 8), 3)], lambda p, x: O.single_conv(p, x, "gcr"))
    module_case("stage_singleconv_gcr_g1", BB.SingleConv(4, 8, 3, 1, "gcr", 8, padding=1),
                [rnd((1, 4, 8, 8, 8), 4)], lambda p, x: O.single_conv(p, x, "gcr"))
    module_case("stage_encoder_pool", BB.Encoder(4, 8, conv_layer_order="ilc"),
                [rnd((1, 4, 16, 16, 16), 5)], lambda p, x: O.encoder(p, x, pool=True))
    module_case("stage_encoder_nopool", BB.Encoder(4, 4, apply_pooling=False, conv_layer_order="ilc"),
                [rnd((1, 4, 8, 8, 16), 6)], lambda p, x: O.encoder(p, x, pool=False))
    module_case("stage_basicconv_1x1", BB.BasicConv(2, 8, 1), [rnd((2, 2, 8, 8, 8), 7)],
                lambda p, x: O.basic_conv(p, x))
    module_case("stage_basicconv_dw", BB.BasicConv(8, 8, 3, padding=1, groups=8), [rnd((1, 8, 8, 8, 16), 8)],
                lambda p, x: O.basic_conv(p, x, groups=8))

    # recon decoder: Decoder(RSM=False): upsample x to skip size, cat(skip, x), DoubleConv
    def rdec(p, skip, x):
        return O.double_conv(p.sub("basic_module"), torch.cat([skip, O.upsample_to(x, skip.shape[2:])], 1))
    module_case("stage_decoder_recon", BB.Decoder(24, 8, conv_layer_order="ilc"),
                [rnd((1, 8, 8, 16, 16), 9), rnd((1, 16, 4, 8, 8), 10)], rdec)

    # seg decoder: Decoder(RSM=True, MVAE=True): upsample, AttenModule2, DoubleConv
    def sdec(p, skip, x):
        return O.double_conv(p.sub("basic_module"),
                             O.atten_module2(p.sub("atten_module"), O.upsample_to(x, skip.shape[2:]), skip))
    module_case("stage_decoder_seg", BB.Decoder(12, 4, conv_layer_order="ilc", RSM=True, MVAE=True),
                [rnd((1, 4, 16, 16, 16), 11), rnd((1, 8, 8, 8, 8), 12)], sdec)

    module_case("stage_duse_train", DuSEAttention(8), [rnd((2, 8, 8, 8, 8), 13), rnd((2, 8, 8, 8, 8), 14)],
                lambda p, r, s: O.duse_attention(p, r, s, True))
    module_case("stage_duse_eval", DuSEAttention(4), [rnd((1, 4, 8, 8, 16), 15), rnd((1, 4, 8, 8, 16), 16)],
                lambda p, r, s: O.duse_attention(p, r, s, False), train=False)
    skr = torch.nn.Sequential(ResBlock(8, 8, lkdw=True), SpacialAttention3D(kernel_size=1))
    module_case("stage_skr_att_train", skr, [rnd((2, 8, 8, 8, 8), 17)],
                lambda p, x: O.skip_return_attention(p, x, True, momentum_steps=1))
    skr = torch.nn.Sequential(ResBlock(4, 4, lkdw=True), SpacialAttention3D(kernel_size=1))
    module_case("stage_skr_att_eval", skr, [rnd((1, 4, 8, 8, 16), 18)],
                lambda p, x: O.skip_return_attention(p, x, False), train=False)

    for S3, tag in [((4, 4, 4), "s64"), ((8, 8, 8), "s512")]:
        vil = ViLLayer(dim=32)
        module_case(f"stage_vil_{tag}", vil, [rnd((2 if tag == "s64" else 1, 32) + S3, 19)],
                    lambda p, x: O.vil_layer(p, x), tol=1e-4)
    # recurrent form of the cell == dense form
    q, k, v = (rnd((1, 4, 96, 16), s) for s in (20, 21, 22))
    ig, fg = rnd((1, 4, 96, 1), 23), rnd((1, 4, 96, 1), 24) + 2
    from UxLSTM.nnunetv2.nets.vision_lstm import parallel_stabilized_simple
    href = parallel_stabilized_simple(q.double(), k.double(), v.double(), ig.double(), fg.double())
    check("mlstm dense", O.mlstm_parallel(q.double(), k.double(), v.double(), ig.double(), fg.double()), href, 1e-12)
    check("mlstm recurrent", O.mlstm_recurrent(q.double(), k.double(), v.double(), ig.double(), fg.double()), href, 1e-10)
    save("stage_mlstm_cell", q=q, k=k, v=v, ig=ig, fg=fg, h=href)


def poe_cases():
    """ProductOfExperts for all 15 subsets + ProductOfExperts2 with per-sample masks + reparametrize."""
    mu = rnd((5, 2, 2, 4, 4, 4), 30)
    lv = rnd((5, 2, 2, 4, 4, 4), 31, scale=3.0)
    mu[0] = 0
    lv[0] = 0
    arrs = dict(mu=mu, logvar=lv)
    poe, poe2 = BB.ProductOfExperts(), BB.ProductOfExperts2()
    for idx, subset in enumerate(R.SUBSETS_MODALITIES):
        m, l = poe(mu, lv, subset)
        arrs[f"mu_{idx}"], arrs[f"lv_{idx}"] = m, l
        m2, l2 = O.product_of_experts(mu, lv, O.SUBSETS_MODALITIES[idx])
        check(f"poe{idx}", m2, m, 1e-6), check(f"poe{idx}", l2, l, 1e-6)
    drop = torch.tensor([[False, True, False, True], [True, True, False, False]])
    mu_c = mu.clone()
    m, l = poe2(mu_c, lv, drop)
    arrs.update(drop=drop, mu_drop=m, lv_drop=l, mu_after=mu_c)
    m2, l2, mum = O.product_of_experts_drop(mu, lv, drop)
    check("poe2", m2, m, 1e-6), check("poe2", l2, l, 1e-6), check("poe2 mu", mum, mu_c, 0)
    torch.manual_seed(5)
    z = R.reparametrize(m, l, False)
    torch.manual_seed(5)
    eps = torch.empty(m.shape).normal_()
    check("reparam", O.reparametrize(m, l, eps), z, 1e-6)
    arrs.update(eps=eps, z=z)
    save("stage_poe", **arrs)


# ---------------------------------------------------------------------------------- whole network
def _inject_eps(eps_list):
    """Patches the reference's module-level reparametrize (RA_HVED.py:741-747; the name the forward
    resolves is the LAST definition in the module) to consume a provided eps list."""
    it = iter(eps_list)

    def rep(mu, logvar, valid=False):
        if valid:
            return mu
        return next(it).to(mu.dtype) * torch.exp(0.5 * logvar) + mu
    R.reparametrize = rep


def network_cases():
    S = 32
    model = ref_shim.build_reference_model(ns, seed=1)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    save("weights_seed1", **{k: v for k, v in sd0.items()})
    gx = torch.Generator().manual_seed(1)
    x = torch.rand(1, 4, S, S, S, generator=gx)
    shapes = [(1, 2 ** l, S // 2 ** (l + 1), S // 2 ** (l + 1), S // 2 ** (l + 1)) for l in range(4)]
    torch.manual_seed(123)
    eps = [torch.empty(s).normal_() for s in shapes]
    orig_rep = R.reparametrize

    def ref_run(dtype, subset, inst, valid, train, xin, want_grads):
        m = ref_shim.build_reference_model(ns, seed=1).to(dtype)
        m.train(train)
        _inject_eps(eps)
        seg, (mu, lv), rec = m(xin.to(dtype).clone(), [subset], instance_missing=inst, recon=True, valid=valid)
        R.reparametrize = orig_rep
        out = dict(seg=seg, rec=rec[0], **{f"mu{i}": a for i, a in enumerate(mu)}, **{f"lv{i}": a for i, a in enumerate(lv)})
        if want_grads:
            ws, wr = rnd(seg.shape, 200).to(dtype), rnd(rec[0].shape, 201).to(dtype)
            loss = (seg * ws).sum() + 0.1 * (rec[0] * wr).sum()
            for i, (a, b) in enumerate(zip(mu, lv)):
                loss = loss + 0.05 * ((a * rnd(a.shape, 210 + i).to(dtype)).sum() + (b * rnd(b.shape, 220 + i).to(dtype)).sum())
            loss.backward()
            grads = {}
            for k, p_ in m.named_parameters():
                if p_.grad is not None:
                    grads["g." + k.replace("decoders.", "srdecoder.sdecoders.", 1) if k.startswith("decoders.") else "g." + k] = p_.grad
            out.update(grads)
            out["loss"] = loss.detach()
        out.update({"after." + k: v for k, v in m.state_dict().items() if "running" in k or "num_batches" in k})
        return out

    def oracle_run(dtype, subset, inst, valid, train, xin, want_grads):
        sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        prob, logits, mu, lv, rec = O.xlstm_hved_forward(
            sd, xin.to(dtype).clone(), subset, instance_missing=inst,
            eps_list=None if valid else [e.to(dtype) for e in eps], training=train)
        out = dict(seg=prob, rec=rec, **{f"mu{i}": a for i, a in enumerate(mu)}, **{f"lv{i}": a for i, a in enumerate(lv)})
        if want_grads:
            ws, wr = rnd(prob.shape, 200).to(dtype), rnd(rec.shape, 201).to(dtype)
            loss = (prob * ws).sum() + 0.1 * (rec * wr).sum()
            for i, (a, b) in enumerate(zip(mu, lv)):
                loss = loss + 0.05 * ((a * rnd(a.shape, 210 + i).to(dtype)).sum() + (b * rnd(b.shape, 220 + i).to(dtype)).sum())
            loss.backward()
            for k, v in sd.items():
                if v.requires_grad and v.grad is not None:
                    out["g." + k] = v.grad
        out.update({"after." + k: v for k, v in sd.items() if "running" in k or "num_batches" in k})
        return out

    def compare(tag, a, b, tol):
        worst = 0.0
        gscale = max([v.double().abs().max().item() for k, v in b.items() if k.startswith("g.")] + [1e-30])
        for k, v in b.items():
            if k == "loss":
                continue
            if k not in a:
                raise AssertionError(f"{tag}: oracle has no '{k}'")
            if k.startswith("g."):
                err = (a[k].double() - v.double()).abs().max().item() / gscale
            else:
                err = (a[k].double() - v.double()).abs().max().item() / max(v.double().abs().max().item(), 1.0)
            worst = max(worst, err)
            assert err <= tol, f"{tag}: {k} deviates by {err:.3e}"
        print(f"  oracle vs reference [{tag}]: worst scaled deviation {worst:.2e}")

    # main case: train mode, all modalities, sampled latent; fp32 fixture + fp64 tie-breaker
    r32 = ref_run(torch.float32, 14, False, False, True, x, True)
    r64 = ref_run(torch.float64, 14, False, False, True, x, True)
    compare("fp64 train subset14", oracle_run(torch.float64, 14, False, False, True, x, True), r64, 1e-9)
    compare("fp32 train subset14", oracle_run(torch.float32, 14, False, False, True, x, True), r32, 2e-4)
    main = dict(x=x, **{f"eps{i}": e for i, e in enumerate(eps)})
    main.update({k: v for k, v in r32.items()})
    main.update({"f64." + k: v for k, v in r64.items() if not k.startswith("g.") and not k.startswith("after.")})
    save("net32_train_subset14", **main)

    # every subset + two instance-missing masks, valid=True, eval BN: sampled outputs (fp64 reference)
    gi = torch.Generator().manual_seed(7)
    idx_seg = torch.randint(0, 3 * S ** 3, (4096,), generator=gi)
    idx_rec = torch.randint(0, 4 * S ** 3, (4096,), generator=gi)
    arrs = dict(idx_seg=idx_seg, idx_rec=idx_rec)
    x2 = torch.rand(2, 4, S, S, S, generator=gx)
    for k in range(15):
        r = ref_run(torch.float64, k, False, True, False, x2[:1], False)
        o = oracle_run(torch.float64, k, False, True, False, x2[:1], False)
        compare(f"fp64 eval subset{k}", o, r, 1e-9)
        arrs[f"seg_{k}"] = r["seg"].flatten()[idx_seg]
        arrs[f"rec_{k}"] = r["rec"].flatten()[idx_rec]
        arrs[f"mu3_{k}"] = r["mu3"].flatten()
    xm = x2.clone()
    masks = [(1, 3), (0,)]
    for i, mk in enumerate(masks):
        for c in range(4):
            if c not in mk:
                xm[i, c] = 0
    for train in (True, False):
        r = ref_run(torch.float64, 14, True, True, train, xm, False)
        o = oracle_run(torch.float64, 14, True, True, train, xm, False)
        compare(f"fp64 instance-missing train={train}", o, r, 1e-9)
        t = "train" if train else "eval"
        arrs[f"im_{t}_seg"] = r["seg"].flatten()[torch.cat([idx_seg, idx_seg + 3 * S ** 3])]
        arrs[f"im_{t}_rec"] = r["rec"].flatten()[torch.cat([idx_rec, idx_rec + 4 * S ** 3])]
        arrs[f"im_{t}_mu0"] = r["mu0"].flatten()[:8192]
        if train:
            arrs.update({"im_train_" + k: v for k, v in r.items() if k.startswith("after.")})
    arrs["x2"] = x2
    save("net32_subsets_eval", **arrs)

    # state_dict manifest (names, shapes) + who receives a gradient
    names = list(sd0.keys())
    with open(os.path.join(HERE, "state_dict_manifest.txt"), "w") as f:
        for k in names:
            f.write(f"{k}\t{','.join(str(d) for d in sd0[k].shape)}\t{str(sd0[k].dtype).replace('torch.', '')}\n")
    print("wrote state_dict_manifest.txt", len(names))


VARIANTS = {
    # tag: (reference class, ctor overrides, oracle flags)
    # 'gcr' needs channel counts divisible by num_groups=8 (f_maps=8) and skip_return needs f_maps == 4 modalities
    # (x0_init, RA_HVED.py:621) as does seg_recon_decoder (sfinals emits 4 channels into final_conv, RA_HVED.py:640), so
    # 'gcr' is only reachable in the classes without either: U_HVEDConvNet3D and U_HVEDConvXLSTMNet3D.
    "uhved_conv_gcr": ("U_HVEDConvNet3D", dict(layer_order="gcr", f_maps=8),
                       dict(order="gcr", mid_vil=False, skip_return=False, seg_recon_decoder=False)),
    "uhved_convxlstm_gcr": ("U_HVEDConvXLSTMNet3D", dict(layer_order="gcr", f_maps=8),
                            dict(order="gcr", mid_vil=False, skip_return=False, seg_recon_decoder=False)),
    "xlstm_hved_wodusfe": ("XLSTM_HVED_woDuSFE", dict(),
                           dict(order="ilc", mid_vil=True, skip_return=True, seg_recon_decoder=False)),
    # shared_recon=False (Pretrain.py:142): four recon decoder streams behind shared seg decoders / DuSE blocks
    "xlstm_hved_noshared": ("XLSTM_HVED", dict(shared_recon=False), dict(order="ilc", mid_vil=True, skip_return=True)),
}


def variant_cases():
    """Secondary configurations of the same forward (RA_HVED.py:651-687 non-DuSFE decoder path, 'gcr' SingleConv order,
    DoubleConv_ViL decoder): fp64 reference run pins the oracle; the fixture keeps weights + sampled outputs + per-parameter
    gradient sums so the CPU suite can replay it."""
    S = 32
    gx = torch.Generator().manual_seed(11)
    x = torch.rand(1, 4, S, S, S, generator=gx)
    gi = torch.Generator().manual_seed(8)
    idx_seg = torch.randint(0, 3 * S ** 3, (4096,), generator=gi)
    idx_rec = torch.randint(0, 4 * S ** 3, (4096,), generator=gi)
    orig_rep = R.reparametrize
    for tag, (cls, over, flags) in VARIANTS.items():
        m = ref_shim.build_reference_model(ns, seed=2, cls=cls, **over)
        ents = entries_of(m.state_dict())
        sd0 = seeded_state(ents, seed=5)
        m.load_state_dict(sd0, strict=True)
        m = m.double().train()
        fm = over.get("f_maps", 4)
        shapes = [(1, fm // 4 * 2 ** l, S // 2 ** (l + 1), S // 2 ** (l + 1), S // 2 ** (l + 1)) for l in range(4)]
        torch.manual_seed(321)
        eps = [torch.empty(s).normal_() for s in shapes]
        _inject_eps(eps)
        seg, (mu, lv), rec = m(x.double().clone(), [14], recon=True, valid=False)
        R.reparametrize = orig_rep
        if isinstance(rec, (list, tuple)):
            rec = rec[0] if len(rec) == 1 else torch.cat(list(rec), 1)
        ws, wr = rnd(seg.shape, 300).double(), rnd(rec.shape, 301).double()
        loss = (seg * ws).sum() + 0.1 * (rec * wr).sum()
        for i, (a, b) in enumerate(zip(mu, lv)):
            loss = loss + 0.05 * ((a * rnd(a.shape, 310 + i).double()).sum() + (b * rnd(b.shape, 320 + i).double()).sum())
        loss.backward()
        has_sr = hasattr(m, "srdecoder")
        rg = {}
        for k, p_ in m.named_parameters():
            if p_.grad is not None:
                kk = k.replace("decoders.", "srdecoder.sdecoders.", 1) if (has_sr and k.startswith("decoders.")) else k
                rg[kk] = p_.grad
        # oracle, same weights/inputs
        sd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        prob, _, omu, olv, orec = O.xlstm_hved_forward(sd, x.double().clone(), 14, eps_list=[e.double() for e in eps],
                                                       training=True, **flags)
        oloss = (prob * ws).sum() + 0.1 * (orec * wr).sum()
        for i, (a, b) in enumerate(zip(omu, olv)):
            oloss = oloss + 0.05 * ((a * rnd(a.shape, 310 + i).double()).sum() + (b * rnd(b.shape, 320 + i).double()).sum())
        oloss.backward()
        check(f"{tag} seg", prob, seg, 1e-9)
        check(f"{tag} rec", orec, rec, 1e-9)
        for i in range(4):
            check(f"{tag} mu{i}", omu[i], mu[i], 1e-9)
            check(f"{tag} lv{i}", olv[i], lv[i], 1e-9)
        gscale = max(g.abs().max().item() for g in rg.values())
        n_checked = 0
        for k, g in rg.items():
            og = sd[k].grad
            assert og is not None, f"{tag}: oracle left {k} without gradient"
            err = (og - g).abs().max().item() / gscale
            assert err < 1e-9, f"{tag}: grad {k} deviates by {err:.2e}"
            n_checked += 1
        print(f"  oracle vs reference [{tag}]: outputs + {n_checked} parameter gradients agree (fp64)")
        arrs = dict(names=np.array([e[0] for e in ents]), shapes=np.array([",".join(map(str, e[1])) for e in ents]),
                    isfloat=np.array([e[2] for e in ents]))
        arrs.update(x=x, idx_seg=idx_seg, idx_rec=idx_rec, seg=seg.flatten()[idx_seg], rec=rec.flatten()[idx_rec],
                    mu3=mu[3].flatten(), lv3=lv[3].flatten(), loss=loss.detach(),
                    gnames=np.array(list(rg.keys())), gsum=torch.stack([g.sum() for g in rg.values()]),
                    gabs=torch.stack([g.abs().sum() for g in rg.values()]))
        arrs.update({f"eps{i}": e for i, e in enumerate(eps)})
        save("variant_" + tag, **arrs)


def loss_cases():
    """Training-step epilogues from the reference's own loss.py / metrics.py (train.py:171-175,232-262): inputs, values
    and input gradients; the oracle restatements are certified against them."""
    import contextlib
    import io
    import loss as RL            # reference module (on sys.path via the shim)
    import metrics as RM
    n, S = 2, (6, 10, 12)
    prob = torch.sigmoid(rnd((n, 3) + S, 401) * 2).requires_grad_(True)
    tgt = (rnd((n, 3) + S, 402) > 0.3).float()
    rec = rnd((n, 4) + S, 403).requires_grad_(True)
    xin = rnd((n, 4) + S, 404)
    disc = rnd((n, 1, 3, 5, 6), 405).requires_grad_(True)
    L_ = 2
    mu = (rnd((n, 5, L_) + S, 406)).clone()
    lv = (rnd((n, 5, L_) + S, 407) * 0.7).clone()
    mu[:, 0] = 0
    lv[:, 0] = 0
    mu.requires_grad_(True)
    lv.requires_grad_(True)
    with contextlib.redirect_stdout(io.StringIO()):      # compute_per_channel_dice prints its shapes (loss.py:269-270)
        dice = RL.DiceLoss()(prob, tgt)
    mse = torch.nn.MSELoss()(rec, xin)
    gan_t, gan_f = RL.GANLoss()(disc, True), RL.GANLoss()(disc, False)
    kld7 = RL.compute_KLD(mu, lv, [7])
    kld_multi = RL.compute_KLD(mu, lv, [2, 12])
    tot = 1.3 * dice + 0.2 * mse + 0.1 * gan_t + 0.05 * gan_f + 0.2 * kld7 + 0.3 * kld_multi
    tot.backward()
    # nested weights exactly as train.py:244-250
    f_weight = prob.detach()
    f_weight = torch.where(f_weight > 0.5, f_weight, torch.zeros_like(f_weight))
    f_nested_w = f_weight[:, 0]
    f_nested_w[f_weight[:, 1] > 0.5] = f_weight[:, 1][f_weight[:, 1] > 0.5]
    f_nested_w[f_weight[:, 2] > 0.5] = f_weight[:, 2][f_weight[:, 2] > 0.5]
    atten = rec.detach() * (1 + f_nested_w.unsqueeze(1))
    dc = RM.DiceCoefficient()(prob.detach(), tgt)
    dcr = torch.stack([RM.DiceRegion()(prob.detach(), tgt, r) for r in ("WT", "TC", "EC")])
    # oracle
    po, ro, do_, muo, lvo = (t.detach().clone().requires_grad_(True) for t in (prob, rec, disc, mu, lv))
    o_dice, o_mse = O.dice_loss(po, tgt), ((ro - xin) ** 2).mean()
    o_gt, o_gf = ((do_ - 1.0) ** 2).mean(), (do_ ** 2).mean()
    o_k7, o_km = O.compute_kld(muo, lvo, [7]), O.compute_kld(muo, lvo, [2, 12])
    (1.3 * o_dice + 0.2 * o_mse + 0.1 * o_gt + 0.05 * o_gf + 0.2 * o_k7 + 0.3 * o_km).backward()
    for tag, a, b in (("dice", o_dice, dice), ("mse", o_mse, mse), ("gan_t", o_gt, gan_t), ("gan_f", o_gf, gan_f), ("kld7", o_k7, kld7),
                      ("kld_multi", o_km, kld_multi), ("dprob", po.grad, prob.grad), ("drec", ro.grad, rec.grad),
                      ("ddisc", do_.grad, disc.grad), ("dmu", muo.grad, mu.grad), ("dlv", lvo.grad, lv.grad),
                      ("nested", O.nested_weight(prob.detach()), f_nested_w.unsqueeze(1)),
                      ("dice_coefficient", O.dice_coefficient(prob.detach(), tgt).mean(), dc),
                      ("dice_region", O.dice_region(prob.detach(), tgt), dcr)):
        check("loss " + tag, a, b, 2e-6)
    save("stage_losses", prob=prob, tgt=tgt, rec=rec, xin=xin, disc=disc, mu=mu, lv=lv, dice=dice, mse=mse, gan_t=gan_t,
         gan_f=gan_f, kld7=kld7, kld_multi=kld_multi, dprob=prob.grad, drec=rec.grad, ddisc=disc.grad, dmu=mu.grad, dlv=lv.grad,
         nested=f_nested_w.unsqueeze(1), atten=atten, dice_coefficient=dc, dice_region=dcr)


DISC_X_SHAPE, DISC_X_SEED, DISC_W_SEED = (2, 7, 40, 36, 44), 501, 7


def disc_sample_index(numel, cap=4096):
    """Positions of a parameter gradient the fixture keeps (every step-th element; whole tensor when small)."""
    step = max(1, numel // cap)
    return torch.arange(0, numel, step)


def disc_cases():
    """The adversarial step's Discriminator exactly as train.py:146-147 builds it: Discriminator(in_channels=7, ks=4,
    strides=[1,2,2,2]) + init_weights, and the class-default ks=3 instance.  11 M parameters: the fixture carries no weights
    (tests rebuild them with the same seed through xlstm_hved_amd.init_weights and check the stored per-tensor checksums) and
    keeps a strided sample + sum + abs-sum of every parameter gradient.  fp32 and fp64 reference runs; the oracle must
    reproduce both."""
    for ks in (4, 3):
        torch.manual_seed(DISC_W_SEED)
        ref = R.Discriminator(in_channels=7, ks=ks, strides=[1, 2, 2, 2])
        ref.apply(ns.utils.init_weights)
        sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
        x = rnd(DISC_X_SHAPE, DISC_X_SEED)       # regenerated by the tests from the same seed; checksums below
        arrs = dict(xsum=x.double().sum(), xabs=x.double().abs().sum(), ks=np.array(ks), names=np.array(list(sd0.keys())))
        for k, v in sd0.items():
            arrs["wsum." + k] = v.double().sum()
            arrs["wabs." + k] = v.double().abs().sum()
        gy = None
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            m = ref.to(dt)
            m.zero_grad()
            xi = x.to(dt).clone().requires_grad_(True)
            y = m(xi)
            if gy is None:
                gy = rnd(y.shape, 502)
                arrs["gy"] = gy
            (y * gy.to(dt)).sum().backward()
            # oracle on the same weights
            sd = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd0.items()}
            xo = x.to(dt).clone().requires_grad_(True)
            yo = O.discriminator(O.P(sd), xo)
            (yo * gy.to(dt)).sum().backward()
            tol = 2e-5 if dt == torch.float32 else 1e-12
            check(f"disc ks{ks} {tag} y", yo, y, tol)
            check(f"disc ks{ks} {tag} dx", xo.grad, xi.grad, tol * 10)
            arrs[f"{tag}.y"] = y
            gx = xi.grad.flatten()
            arrs[f"{tag}.dx"] = gx[disc_sample_index(gx.numel(), 65536)]
            arrs[f"{tag}.dxsum"] = gx.double().sum()
            arrs[f"{tag}.dxabs"] = gx.double().abs().sum()
            for k, p_ in m.named_parameters():
                check(f"disc ks{ks} {tag} g.{k}", sd[k].grad, p_.grad, tol * 10)
                g = p_.grad.flatten()
                arrs[f"{tag}.g.{k}"] = g[disc_sample_index(g.numel())]
                arrs[f"{tag}.gsum.{k}"] = g.double().sum()
                arrs[f"{tag}.gabs.{k}"] = g.double().abs().sum()
            ref = m.float()
        # the reference's OWN mixed precision on the same weights / input (train.py:218 runs the step under autocast): how far its
        # output and gradients move from its fp32 run -- the yardstick for the HIP path's 16-bit deviations (LeakyReLU(0.2) masks
        # flip wherever a 16-bit forward changes a sign, which bounds any 16-bit gradient from below)
        f32 = {k: torch.from_numpy(np.asarray(arrs[k])) if not torch.is_tensor(arrs[k]) else arrs[k] for k in arrs if k.startswith("f32.")}
        for tag, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            try:
                m = ref.float()
                m.zero_grad()
                xi = x.clone().requires_grad_(True)
                with torch.autocast("cpu", dtype=dt):
                    y = m(xi)
                (y.float() * gy).sum().backward()
                l2 = lambda a, b: float(((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)))
                arrs[f"amp_{tag}.y"] = np.array(l2(y.detach().float(), f32["f32.y"]))
                gx = xi.grad.flatten()
                arrs[f"amp_{tag}.dx"] = np.array(l2(gx[disc_sample_index(gx.numel(), 65536)], f32["f32.dx"]))
                worst = 0.0
                for k, p_ in m.named_parameters():
                    g = p_.grad.flatten()
                    if k.endswith(".bias") and not k.startswith("disc.0."):
                        continue                           # a bias in front of an InstanceNorm: its gradient is round-off around zero
                    worst = max(worst, l2(g[disc_sample_index(g.numel())], f32[f"f32.g.{k}"]))
                arrs[f"amp_{tag}.g"] = np.array(worst)
                print(f"  reference Discriminator(ks={ks}) under {tag} autocast vs its fp32 run: y {float(arrs[f'amp_{tag}.y']):.2e}, "
                      f"dx {float(arrs[f'amp_{tag}.dx']):.2e}, worst parameter gradient {worst:.2e} (relative L2)")
            except Exception as e:                         # an autocast dtype this torch build cannot run on CPU
                print(f"  ({tag} autocast yardstick skipped: {e!r})")
        save(f"stage_disc_ks{ks}", **arrs)
        print(f"  oracle vs reference Discriminator(ks={ks}): output, input gradient and 9 parameter gradients agree (fp32, fp64)")


if __name__ == "__main__":
    if "--losses-only" in sys.argv:
        loss_cases()
        sys.exit(0)
    if "--disc-only" in sys.argv:
        disc_cases()
        sys.exit(0)
    stage_cases()
    poe_cases()
    network_cases()
    variant_cases()
    loss_cases()
    disc_cases()
    print("all fixtures written and certified against the oracle")

"""CPU oracle for the XLSTM-HVED volumetric forward path.  TEST INFRASTRUCTURE ONLY.

This file is a functional restatement (stock torch ops on CPU, fp32 or fp64) of the reference
algorithm.  It is the checker the HIP path is compared with; it is never the thing shipped or
measured.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity pinning: the restatement is checked against the real reference (imported from
/root/reference through tools/ref_shim.py in the build container) by tests/golden/make_golden.py,
which also writes the golden vectors under tests/golden/ that `-m "not gpu"` tests replay
(tests/test_oracle_golden.py).  The reference itself holds no golden vectors or tests (SURVEY.md
section 4), so reference-generated fixtures are the pin.

Parameters are addressed by the reference's own state_dict keys (e.g.
"encoders.0.1.basic_module.0.SingleConv2.conv.weight") so a reference checkpoint drives the oracle
unchanged.  Every function cites the reference lines it restates.
"""
import math
from itertools import chain, combinations

import torch
import torch.nn.functional as F

# RA_HVED.py:733-738: all non-empty subsets of the 4 modalities, singles first.
MODALITIES = (0, 1, 2, 3)
SUBSETS_MODALITIES = list(chain(*[combinations(MODALITIES, r) for r in range(1, 5)]))

LEAK = 0.01          # buildingblocks.py:416  nn.LeakyReLU(negative_slope=1e-2)
NORM_EPS = 1e-5      # torch default for InstanceNorm3d/BatchNorm3d/GroupNorm/LayerNorm
BN_MOMENTUM = 0.1    # torch default for BatchNorm3d


class P:
    """Prefix view over a flat {state_dict key: tensor} mapping."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def sub(self, name):
        return P(self.sd, f"{self.prefix}{name}.")

    def get(self, name, default=None):
        return self.sd.get(self.prefix + name, default)

    def __getitem__(self, name):
        return self.sd[self.prefix + name]

    def __setitem__(self, name, value):
        self.sd[self.prefix + name] = value


# ----------------------------------------------------------------------------------------------
# conv stages
# ----------------------------------------------------------------------------------------------
def single_conv(p, x, order="ilc", stride=1, num_groups=8):
    """buildingblocks.py:381-461 create_conv/SingleConv.

    'ilc': InstanceNorm3d(no affine) -> LeakyReLU(0.01) -> Conv3d(k3,p1,bias)
    'gcr': GroupNorm(num_groups or 1) -> Conv3d(k3,p1,no bias) -> ReLU
    """
    w = p["conv.weight"]
    pad = w.shape[-1] // 2
    if order == "ilc":
        h = F.leaky_relu(F.instance_norm(x, eps=NORM_EPS), LEAK, inplace=True)      # nn.LeakyReLU(inplace=True), buildingblocks.py:416
        return F.conv3d(h, w, p["conv.bias"], stride=stride, padding=pad)
    if order == "gcr":
        c = x.shape[1]
        g = num_groups if c >= num_groups else 1          # buildingblocks.py:425-426
        h = F.group_norm(x, g, p["groupnorm.weight"], p["groupnorm.bias"], eps=NORM_EPS)
        return F.relu(F.conv3d(h, w, None, stride=stride, padding=pad), inplace=True)    # buildingblocks.py:414
    raise ValueError(order)


def double_conv(p, x, order="ilc"):
    """buildingblocks.py:464-507 DoubleConv: two SingleConvs (channel plan lives in the weights).
    DoubleConv_ViL (buildingblocks.py:509-555) appends LeakyReLU() and a ViLLayer when its parameters are present."""
    y = single_conv(p.sub("SingleConv2"), single_conv(p.sub("SingleConv1"), x, order), order)
    if p.get("ViL.vil.norm.weight") is not None:
        y = vil_layer(p.sub("ViL"), F.leaky_relu(y, LEAK))
    return y


def encoder(p, x, pool, order="ilc"):
    """buildingblocks.py:607-659 Encoder: [MaxPool3d(2)] -> DoubleConv (num_block=1)."""
    if pool:
        x = F.max_pool3d(x, 2)
    return double_conv(p.sub("basic_module.0"), x, order)


def basic_conv(p, x, groups=1):
    """buildingblocks.py:13-31 BasicConv: Conv3d(no bias) -> InstanceNorm3d -> LeakyReLU(0.01)."""
    w = p["conv.weight"]
    y = F.conv3d(x, w, None, padding=w.shape[-1] // 2, groups=groups)
    return F.leaky_relu(F.instance_norm(y, eps=NORM_EPS), LEAK, inplace=True)       # buildingblocks.py:21


def upsample_to(x, size):
    """buildingblocks.py:785-787 F.interpolate(mode='trilinear') (align_corners=False)."""
    return F.interpolate(x, size=tuple(size), mode="trilinear", align_corners=False)


def channel_pool(x):
    """buildingblocks.py:136-138 ChannelPool: cat[max over C, mean over C]."""
    return torch.cat([x.max(1, keepdim=True)[0], x.mean(1, keepdim=True)], 1)


def atten_module2(p, seg_x, enc_x):
    """buildingblocks.py:259-301 AttenModule2 (the MVAE 'ROI attentive skip connection')."""
    spa = channel_pool(seg_x)
    enc_spa = torch.cat([spa, channel_pool(enc_x)], 1)
    e = F.conv3d(enc_spa, p["enc_spatial.weight"], p["enc_spatial.bias"], padding=3, groups=4)
    e = torch.sigmoid(F.conv3d(e, p["enc_spatial2.weight"], p["enc_spatial2.bias"]))
    s = F.conv3d(spa, p["seg_spatial.weight"], p["seg_spatial.bias"], padding=3, groups=2)
    s = torch.sigmoid(F.conv3d(s, p["seg_spatial2.weight"], p["seg_spatial2.bias"]))
    return torch.cat([seg_x * (1 + s), enc_x + enc_x * e], 1)


def batch_norm(p, x, training, momentum_steps=1):
    """nn.BatchNorm3d semantics.  In training mode the running buffers in `p` are advanced
    `momentum_steps` times with the same batch statistics (the skip-return attention is
    evaluated 4x per forward on identical input, RA_HVED.py:548-552)."""
    rm, rv = p["running_mean"], p["running_var"]
    if momentum_steps == 1 and rm.dtype == x.dtype:
        # the stock fused op (what the reference's nn.BatchNorm3d calls): same function as the explicit form below, and the
        # same cost as the reference, which matters where this file is timed as the CPU baseline (bench.py)
        rm2, rv2 = rm.detach().clone(), rv.detach().clone()
        y = F.batch_norm(x, rm2, rv2, p["weight"], p["bias"], training, BN_MOMENTUM, NORM_EPS)
        if training:
            p["running_mean"], p["running_var"] = rm2, rv2
            if p.get("num_batches_tracked") is not None:
                p["num_batches_tracked"] = p["num_batches_tracked"] + 1
        return y
    if training:
        dims = (0, 2, 3, 4)
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            keep = (1 - BN_MOMENTUM) ** momentum_steps
            p["running_mean"] = keep * rm + (1 - keep) * mean.detach().to(rm.dtype)
            p["running_var"] = keep * rv + (1 - keep) * (var.detach() * n / max(n - 1, 1)).to(rv.dtype)
            if p.get("num_batches_tracked") is not None:
                p["num_batches_tracked"] = p["num_batches_tracked"] + momentum_steps
    else:
        mean, var = rm.to(x.dtype), rv.to(x.dtype)
    sh = (1, -1, 1, 1, 1)
    xh = (x - mean.view(sh)) / torch.sqrt(var.view(sh) + NORM_EPS)
    return xh * p["weight"].view(sh) + p["bias"].view(sh)


def dw_conv_norm(p, x, training, momentum_steps):
    """sa_modules/sa_module.py:56-85 DWConvNorm(norm='BATCH', leaky=False):
    depthwise 3^3 (no bias) -> pointwise 1x1 (bias) -> BatchNorm3d -> ReLU."""
    c = x.shape[1]
    y = F.conv3d(x, p["dwconv.weight"], None, padding=1, groups=c)
    y = F.conv3d(y, p["pwconv.weight"], p["pwconv.bias"])
    return F.relu(batch_norm(p.sub("norm"), y, training, momentum_steps), inplace=True)       # sa_module.py:68


def skip_return_attention(p, x, training, momentum_steps=4):
    """RA_HVED.py:371-384: nn.Sequential(ResBlock(c, c, lkdw=True), SpacialAttention3D(kernel_size=1)).

    ResBlock (sa_module.py:99-137): conv1 -> conv2 -> + identity -> ReLU (need_map is False).
    SpacialAttention3D (attention_blocks.py:112-126): sigmoid(Conv1x1_{2->1}([max_c, mean_c])).
    Returns the (N,1,D,H,W) attention map."""
    r = p.sub("0")
    y = dw_conv_norm(r.sub("conv2"), dw_conv_norm(r.sub("conv1"), x, training, momentum_steps),
                     training, momentum_steps)
    y = F.relu(y + x, inplace=True)                            # sa_module.py:106,135
    pooled = torch.cat([y.max(1, keepdim=True)[0], y.mean(1, keepdim=True)], 1)
    return torch.sigmoid(F.conv3d(pooled, p["1.conv.weight"], None))


def duse_attention(p, r, s, training):
    """modules/DuSFE.py:113-155 DuSEAttention.forward (conv_fuse_ch{1,2} are dead parameters)."""
    n, c = r.shape[:2]
    g = F.linear(torch.cat([r.mean((2, 3, 4)), s.mean((2, 3, 4))], 1), p["fc_comb.weight"], p["fc_comb.bias"])
    ch1 = torch.sigmoid(F.linear(g, p["fc_ch1.weight"], p["fc_ch1.bias"])).view(n, c, 1, 1, 1)
    ch2 = torch.sigmoid(F.linear(g, p["fc_ch2.weight"], p["fc_ch2.bias"])).view(n, c, 1, 1, 1)
    sq = torch.cat([F.conv3d(r, p["conv_squeeze_ch1.weight"], p["conv_squeeze_ch1.bias"]),
                    F.conv3d(s, p["conv_squeeze_ch2.weight"], p["conv_squeeze_ch2.bias"])], 1)
    comb = F.conv3d(sq, p["conv_comb.weight"], p["conv_comb.bias"])
    sp1 = torch.sigmoid(F.conv3d(comb, p["conv_adjust_ch1.weight"], p["conv_adjust_ch1.bias"], padding=1))
    sp2 = torch.sigmoid(F.conv3d(comb, p["conv_adjust_ch2.weight"], p["conv_adjust_ch2.bias"], padding=1))
    out_r = batch_norm(p.sub("bn_fuse_ch1"), r + r * ch1 + r * sp1, training)
    out_s = batch_norm(p.sub("bn_fuse_ch2"), s + s * ch2 + s * sp2, training)
    return out_r, out_s


# ----------------------------------------------------------------------------------------------
# S-MVAE: product of experts + reparameterisation
# ----------------------------------------------------------------------------------------------
def clip_logvar(v):
    """RA_HVED.py:749-753."""
    return torch.clamp(v, -50.0, 50.0)


def product_of_experts(mu, logvar, subset, eps=1e-8):
    """buildingblocks.py:853-866.  mu/logvar: (5,N,L,d,h,w), index 0 = prior; subset = modality ids."""
    idx = [m + 1 for m in subset] + [0]
    t = 1.0 / (torch.exp(logvar[idx]) + eps)
    tsum = t.sum(0)
    return (mu[idx] * t).sum(0) / tsum, torch.log(1.0 / tsum)


def product_of_experts_drop(mu, logvar, drop, eps=1e-8):
    """buildingblocks.py:875-886 ProductOfExperts2: all 5 experts, dropped modalities have mu and T
    zeroed per sample (ZeroLayerF, buildingblocks.py:308-323).  Returns the masked mu as well because
    the reference zeroes `mu` in place and `mu_list` aliases it (RA_HVED.py:582)."""
    t = 1.0 / (torch.exp(logvar) + eps)
    keep = torch.ones_like(mu[:, :, :1, :1, :1, :1])
    keep[1:] = (~drop).t().to(mu.dtype).view(drop.shape[1], drop.shape[0], 1, 1, 1, 1)
    mu_m, t_m = mu * keep, t * keep
    tsum = t_m.sum(0)
    return (mu_m * t_m).sum(0) / tsum, torch.log(1.0 / tsum), mu_m


def reparametrize(mu, logvar, eps_noise):
    """RA_HVED.py:741-747.  eps_noise=None is valid=True (return the mean)."""
    if eps_noise is None:
        return mu
    return eps_noise * torch.exp(0.5 * logvar) + mu


# ----------------------------------------------------------------------------------------------
# ViL (mLSTM) layer
# ----------------------------------------------------------------------------------------------
def mlstm_parallel(q, k, v, igate, fgate, eps=1e-6):
    """vision_lstm.py:48-130 parallel_stabilized_simple.  q,k,v: (B,NH,S,DH); gates: (B,NH,S,1)."""
    B, NH, S, DH = q.shape
    logf = F.logsigmoid(fgate)
    csum = torch.cat([logf.new_zeros(B, NH, 1, 1), torch.cumsum(logf, -2)], -2)       # (B,NH,S+1,1)
    fmat = (csum - csum.transpose(-2, -1))[:, :, 1:, 1:]                                # F_t - F_s
    tril = torch.tril(torch.ones(S, S, dtype=torch.bool, device=q.device))
    logd = torch.where(tril, fmat, fmat.new_full((), -float("inf"))) + igate.transpose(-2, -1)
    m = logd.max(-1, keepdim=True)[0]
    d = torch.exp(logd - m)
    c = (q @ (k / math.sqrt(DH)).transpose(-2, -1)) * d
    norm = torch.maximum(c.sum(-1, keepdim=True).abs(), torch.exp(-m))
    return (c / (norm + eps)) @ v


def mlstm_recurrent(q, k, v, igate, fgate, eps=1e-6):
    """The same cell as a token-by-token recurrence; O(S*DH^2), used to cross-check the
    dense form and as the oracle for sequences the dense form cannot hold."""
    B, NH, S, DH = q.shape
    ks = k / math.sqrt(DH)
    C = q.new_zeros(B, NH, DH, DH)
    nvec = q.new_zeros(B, NH, DH)
    m = q.new_full((B, NH), -float("inf"))
    out = []
    logf = F.logsigmoid(fgate)[..., 0]
    ig = igate[..., 0]
    for t in range(S):
        m_new = torch.maximum(logf[:, :, t] + m, ig[:, :, t])
        fa = torch.exp(logf[:, :, t] + m - m_new)
        ia = torch.exp(ig[:, :, t] - m_new)
        C = fa[..., None, None] * C + ia[..., None, None] * (v[:, :, t, :, None] * ks[:, :, t, None, :])
        nvec = fa[..., None] * nvec + ia[..., None] * ks[:, :, t]
        m = m_new
        num = (C * q[:, :, t, None, :]).sum(-1)
        den = torch.maximum((nvec * q[:, :, t]).sum(-1).abs(), torch.exp(-m)) + eps
        out.append(num / den[..., None])
    return torch.stack(out, 2)


def headwise_linear(x, w):
    """vision_lstm.py:158-168 LinearHeadwiseExpand (no bias): block-diagonal, w: (nh, out_d, d)."""
    nh, od, d = w.shape
    xb = x.reshape(*x.shape[:-1], nh, d)
    return torch.einsum("...hd,hod->...ho", xb, w).reshape(*x.shape[:-1], nh * od)


def vil_tokens(p, x, recurrent=False):
    """vision_lstm.py:494-502 ViLBlock (DropPath p=0 -> x + layer(norm(x))) with the inner ViLLayer
    (vision_lstm.py:415-453) and MatrixLSTMCell (vision_lstm.py:302-339).  x: (B,S,C)."""
    B, S, C = x.shape
    h = F.layer_norm(x, (C,), 1.0 + p["norm.weight"], None, NORM_EPS)       # vision_lstm.py:224-259
    L = p.sub("layer")
    inner = F.linear(h, L["proj_up.weight"])
    xm, z = inner.chunk(2, -1)
    # CausalConv1d (vision_lstm.py:213-221): depthwise k=4 along S with left padding 3
    cw = L["conv1d.conv.weight"]
    kk = cw.shape[-1]
    xc = F.conv1d(F.pad(xm.transpose(1, 2), (kk - 1, 0)), cw, L["conv1d.conv.bias"], groups=cw.shape[0])
    xa = F.silu(xc.transpose(1, 2))
    q = headwise_linear(xa, L["q_proj.weight"])
    k = headwise_linear(xa, L["k_proj.weight"])
    v = headwise_linear(xm, L["v_proj.weight"])
    cell = L.sub("mlstm_cell")
    nh = cell["igate.weight"].shape[0]
    gate_in = torch.cat([q, k, v], -1)
    ig = F.linear(gate_in, cell["igate.weight"], cell["igate.bias"]).transpose(1, 2).unsqueeze(-1)
    fg = F.linear(gate_in, cell["fgate.weight"], cell["fgate.bias"]).transpose(1, 2).unsqueeze(-1)
    split = lambda t: t.view(B, S, nh, -1).transpose(1, 2)
    cellfn = mlstm_recurrent if recurrent else mlstm_parallel
    hs = cellfn(split(q), split(k), split(v), ig, fg)                          # (B,NH,S,DH)
    # MultiHeadLayerNorm (vision_lstm.py:271-287): per-head normalisation over DH, weight 1+w
    hs = hs.transpose(1, 2)                                                     # (B,S,NH,DH)
    mu = hs.mean(-1, keepdim=True)
    var = hs.var(-1, unbiased=False, keepdim=True)
    hn = ((hs - mu) / torch.sqrt(var + NORM_EPS)).reshape(B, S, -1) * (1.0 + cell["outnorm.weight"])
    hg = (hn + L["learnable_skip"] * xa) * F.silu(z)
    return x + F.linear(hg, L["proj_down.weight"])


def vil_layer(p, x, recurrent=False):
    """UxLSTMEnc_3d.py:54-63,77-87 outer ViLLayer.forward_patch_token: NCDHW -> (B,S,C) with W
    fastest -> ViLBlock -> NCDHW.  (The outer `norm` LayerNorm is never applied.)"""
    B, C = x.shape[:2]
    tok = x.reshape(B, C, -1).transpose(1, 2)
    out = vil_tokens(p.sub("vil"), tok, recurrent)
    return out.transpose(1, 2).reshape(x.shape)


# ----------------------------------------------------------------------------------------------
# the network
# ----------------------------------------------------------------------------------------------
def xlstm_hved_forward(sd, x, subset_idx=14, instance_missing=False, drop=None, seg=True, recon=True,
                       eps_list=None, training=True, mid_vil=True, skip_return=True, levels=4,
                       recurrent_mlstm=False, taps=None, order="ilc", seg_recon_decoder=True, reference_cost=False):
    """RA_HVED.py:510-648 AbstractFusion3DUNet.forward for the XLSTM_HVED flag set
    (RA_HVED.py:945-958: skip_return, mid_ViL, seg_recon_decoder, MVAE, MVAE_reduction, 'ilc').

    sd        flat state_dict-style mapping (tensors already in the dtype to compute in).  BatchNorm
              buffers are replaced in `sd` when training=True (running-stat side effects).
    eps_list  per-level N(0,1) draws for the reparameterisation (RA_HVED.py:744); None == valid=True.
    training  BatchNorm mode (model.train()/eval()); independent of eps_list/valid like the reference.
    taps      optional dict that receives named intermediates (for per-stage golden vectors).
    reference_cost  True: evaluate the skip-return attention once PER MODALITY STREAM on the identical input, exactly as the
              reference does (RA_HVED.py:548-552: 4 x 15 convolutions per level instead of 15) -- the same numbers and the
              same BatchNorm buffer updates, the reference's arithmetic cost.  bench.py's cpu_baseline leg times this form
              (the restatement's wall time must match the reference's, tools/calibrate_cpu_baseline.py).
    Returns (seg_prob, seg_logits, mu_list, logvar_list, recon).
    """
    p = P(sd)
    N = x.shape[0]
    tap = (lambda k, v: taps.__setitem__(k, v)) if taps is not None else (lambda k, v: None)
    if instance_missing:
        if drop is None:
            drop = x.sum((2, 3, 4)) == 0                                           # RA_HVED.py:515
    x_list = [F.conv3d(x[:, i:i + 1], p[f"init_blocks.{i}.0.weight"], p[f"init_blocks.{i}.0.bias"])
              for i in range(4)]                                                    # RA_HVED.py:534-536
    mu_list, logvar_list, feats = [], [], []
    skip = None
    for level in range(levels):
        if skip_return and skip is not None and reference_cost:
            x_list = [skip_return_attention(p.sub(f"skr_att.{levels - level}"), skip, training, momentum_steps=1) * xi + xi
                      for xi in x_list]                                             # RA_HVED.py:548-552, stream by stream
        elif skip_return and skip is not None:
            a = skip_return_attention(p.sub(f"skr_att.{levels - level}"), skip, training)   # skr_att[-level]
            tap(f"skr_att.{level}", a)
            x_list = [a * xi + xi for xi in x_list]                                 # RA_HVED.py:552
        x_list = [encoder(p.sub(f"encoders.{level}.{i}"), x_list[i], pool=level > 0, order=order) for i in range(4)]
        tap(f"enc.{level}.0", x_list[0])
        mod_mu, mod_lv = [], []
        for j in range(4):
            f = single_conv(p.sub(f"DRBs.{level}.{j}.0"), x_list[j], order, stride=2)      # RA_HVED.py:569
            L = f.shape[1] // 2
            mod_mu.append(f[:, :L])
            mod_lv.append(clip_logvar(f[:, L:]))
        mu = torch.stack([torch.zeros_like(mod_mu[0])] + mod_mu, 0)                 # RA_HVED.py:576-580
        lv = torch.stack([torch.zeros_like(mod_lv[0])] + mod_lv, 0)
        if instance_missing:
            sub_mu, sub_lv, mu = product_of_experts_drop(mu, lv, drop)
        else:
            sub_mu, sub_lv = product_of_experts(mu, lv, SUBSETS_MODALITIES[subset_idx])
        mu_list.append(mu.transpose(0, 1))
        logvar_list.append(lv.transpose(0, 1))
        z = reparametrize(sub_mu, sub_lv, None if eps_list is None else eps_list[level])
        tap(f"z.{level}", z)
        z = basic_conv(p.sub(f"VU_blocks.{level}.0"), z)                            # RA_HVED.py:599
        z = upsample_to(z, [2 * s for s in z.shape[2:]])                            # RA_HVED.py:600-601
        z = basic_conv(p.sub(f"conv_blocks.{level}"), z, groups=z.shape[1])         # RA_HVED.py:603
        tap(f"feat.{level}", z)
        feats.insert(0, z)
        if skip_return:                                                             # RA_HVED.py:617-621
            if skip is None:
                skip = F.conv3d(x, p["x0_init.0.weight"], p["x0_init.0.bias"])
            else:
                skip = encoder(p.sub(f"skr_encoders.{levels - 1 - level}"), skip, pool=True, order=order)
    if mid_vil and skip_return:                                                     # RA_HVED.py:623-626
        v = vil_layer(p.sub("mViL"), feats[0] + skip, recurrent_mlstm)
        tap("vil", v)
        feats[0] = feats[0] + v
    if not seg_recon_decoder:
        # RA_HVED.py:651-687: separate ReconDecoder (RA_HVED.py:68-95) then the seg decoders (AttenModule2 + DoubleConv)
        rout = feats[0]
        for j in range(levels - 1):
            skipf = feats[j + 1]
            rout = double_conv(p.sub(f"rdecoder.multi_decoders.0.{j}.basic_module"),
                               torch.cat([skipf, upsample_to(rout, skipf.shape[2:])], 1), order)
        rec = F.conv3d(rout, p["rdecoder.finals.0.weight"], p["rdecoder.finals.0.bias"])
        logits = prob = None
        if seg:
            sout = feats[0]
            for j in range(levels - 1):
                skipf = feats[j + 1]
                dp = p.sub(f"decoders.{j}")
                sout = double_conv(dp.sub("basic_module"),
                                   atten_module2(dp.sub("atten_module"), upsample_to(sout, skipf.shape[2:]), skipf), order)
                tap(f"dec.{j}", sout)
            logits = F.conv3d(sout, p["final_conv.weight"], p["final_conv.bias"])
            prob = torch.sigmoid(logits)
        return prob, logits, mu_list, logvar_list, rec
    # Seg_Recon_DuSFEDecoder.forward, RA_HVED.py:158-201.  shared_recon=True: one stream; shared_recon=False (Pretrain.py:142):
    # four recon decoder streams, each restarting from feats[0]; the seg decoders and the first three DuSE blocks are shared
    # by every stream (the zip at RA_HVED.py:171 stops after three of the twelve registered DuSE blocks).
    n_streams = 1 if "srdecoder.multi_decoders.1.0.basic_module.SingleConv1.conv.weight" not in sd else 4
    recs, sfins = [], []
    for i in range(n_streams):
        rout = sout = feats[0]
        for j in range(levels - 1):
            skipf = feats[j + 1]
            size = skipf.shape[2:]
            rd = p.sub(f"srdecoder.multi_decoders.{i}.{j}")
            rout = double_conv(rd.sub("basic_module"), torch.cat([skipf, upsample_to(rout, size)], 1), order)
            if seg:
                sdp = p.sub(f"srdecoder.sdecoders.{j}")
                sout = double_conv(sdp.sub("basic_module"),
                                   atten_module2(sdp.sub("atten_module"), upsample_to(sout, size), skipf), order)
                tap(f"dec.{j}.pre_duse", (rout, sout))
                rout, sout = duse_attention(p.sub(f"srdecoder.dusfe_decoders.{j}"), rout, sout, training)
            tap(f"dec.{j}", (rout, sout))
        recs.append(F.conv3d(rout, p[f"srdecoder.rfinals.{i}.weight"], p[f"srdecoder.rfinals.{i}.bias"]))
        if seg:
            sfins.append(F.conv3d(sout, p[f"srdecoder.sfinals.{i}.weight"], p[f"srdecoder.sfinals.{i}.bias"]))
    rec = recs[0] if n_streams == 1 else torch.cat(recs, 1)          # the reference returns the list (RA_HVED.py:644)
    logits = prob = None
    if seg:
        logits = F.conv3d(torch.cat(sfins, 1), p["final_conv.weight"], p["final_conv.bias"])      # RA_HVED.py:199,640
        prob = torch.sigmoid(logits)                                               # RA_HVED.py:641
    return prob, logits, mu_list, logvar_list, rec


# ----------------------------------------------------------------------------------------------
# metric restatement (parity metric)
# ----------------------------------------------------------------------------------------------
def dice_region(prob, target, eps=1e-6):
    """metrics.py:85-107 DiceRegion(mode='sigmoid') for WT/TC/ET at once: returns (3,) tensor."""
    pred = (prob > 0.5).to(target.dtype)
    inter = (pred * target).sum((2, 3, 4))
    den = (pred + target).sum((2, 3, 4))
    return ((2 * inter + eps) / (den + eps)).mean(0)


# ----------------------------------------------------------------------------------------------
# loss / metric restatements (training-step epilogues)
# ----------------------------------------------------------------------------------------------
def dice_loss(prob, target, eps=1e-6):
    """loss.py:188-209 DiceLoss + compute_per_channel_dice (loss.py:257-285): channel-first flatten over batch and space,
    2 sum(p t) / clamp(sum p^2 + sum t^2, eps), loss = 1 - mean over channels."""
    C = prob.shape[1]
    p = prob.transpose(0, 1).reshape(C, -1)
    t = target.to(prob.dtype).transpose(0, 1).reshape(C, -1)
    inter = (p * t).sum(-1)
    den = (p * p).sum(-1) + (t * t).sum(-1)
    return 1.0 - (2 * (inter / den.clamp(min=eps))).mean()


def kl_divergence(mu1, lv1, mu2, lv2, eps=1e-8):
    """loss.py:29-40 with an explicit second distribution."""
    return 0.5 * torch.mean(-1 + lv2 - lv1 + (lv1.exp() + (mu1 - mu2) ** 2) / (lv2.exp() + eps))


def compute_kld(mu_stack, lv_stack, subset_index_list):
    """loss.py:85-115: stacks (B,5,L,...) -> transpose to (5,B,...); PoE over prior + subset; KL against the prior."""
    mu, lv = mu_stack.transpose(0, 1), lv_stack.transpose(0, 1)
    tot = 0.0
    for idx in subset_index_list:
        sm, sl = product_of_experts(mu, lv, SUBSETS_MODALITIES[idx])
        tot = tot + kl_divergence(sm, sl, mu[0], lv[0])
    return tot / len(subset_index_list)


def nested_weight(seg):
    """train.py:244-248: where(p > .5, p, 0); channel 0 overridden by channel 1, then channel 2, where those exceed .5."""
    w = torch.where(seg > 0.5, seg, torch.zeros_like(seg))
    out = w[:, 0].clone()
    out[w[:, 1] > 0.5] = w[:, 1][w[:, 1] > 0.5]
    out[w[:, 2] > 0.5] = w[:, 2][w[:, 2] > 0.5]
    return out.unsqueeze(1)


def dice_coefficient(prob, target, eps=1e-6):
    """metrics.py:27-48 (the metrics.py variant thresholds at 0.5): per-channel (2 I + eps)/(sum + eps), mean over batch."""
    pred = (prob > 0.5).to(target.dtype)
    inter = (pred * target).sum((2, 3, 4))
    den = (pred + target).sum((2, 3, 4))
    return ((2 * inter + eps) / (den + eps)).mean(0)


def discriminator(p, x, strides=(1, 2, 2, 2), slope=0.2):
    """RA_HVED.py:204-236 Discriminator.forward with buildingblocks.py:342-358 discriminator_block (double=False):
    disc.0 = Conv3d(k, stride, padding=1) -> LeakyReLU(0.2)  (normalization=False), disc.1..3 = Conv3d -> InstanceNorm3d (no
    affine, eps 1e-5) -> LeakyReLU(0.2), last = Conv3d(512, 1, k, padding=1, bias=False).  The kernel size is that of the
    weights (train.py:146 builds ks=4, the class default is 3)."""
    for i, st in enumerate(strides):
        x = F.conv3d(x, p[f"disc.{i}.0.weight"], p[f"disc.{i}.0.bias"], stride=st, padding=1)
        if i > 0:
            x = F.instance_norm(x, eps=NORM_EPS)
        x = F.leaky_relu(x, slope, inplace=True)             # buildingblocks.py:350,356
    return F.conv3d(x, p["last.weight"], None, padding=1)


def bench_loss(prob, mu_list, logvar_list, rec):
    """loss that reaches every used parameter."""
    loss = prob.float().mean() + rec.float().mean()
    for m, l in zip(mu_list, logvar_list):
        loss = loss + m.float().mean() + l.float().mean()
    return loss

"""Network assembly: the reference's AbstractFusion3DUNet family (RA_HVED.py:239-1116) on the HIP stages.

Constructor kwargs, forward signature, return tuple and state_dict keys follow the reference so the classes drop
into train.py / test.py style callers (`model(x, subset_idx_list=[k], instance_missing=..., drop=..., seg=...,
recon=..., valid=...)`).  Only the configurations the reference can actually run are built:
MVAE + MVAE_reduction with DoubleConv blocks.

MI355X-first differences that do not change results:
  * the 4 modality streams are carried as ONE (N, 4*C, D, H, W) tensor and run through grouped kernels with one
    weight pointer per stream (4x fewer, 4x larger launches);
  * the skip-return attention is evaluated once per level instead of once per stream (identical input), its
    BatchNorm running statistics are advanced 4 steps;
  * linear stages with no non-linearity between them are composed (7^3 grouped conv + 1x1, squeeze + comb 1x1,
    sfinals + final_conv);
  * no hard-coded .cuda() (RA_HVED.py:520): masks are built on x.device.
"""
from itertools import chain, combinations

import contextlib

import torch
from torch import nn

from . import functional as Fn
from .blocks import (BasicConv, Decoder, DoubleConv, DoubleConv_ViL, DuSEAttention, Encoder, ProductOfExperts,
                     ProductOfExperts2, SingleConv, SkipReturnAttention, ViLLayer, number_of_features_per_level)
from . import ops
from .ops import ACT_SIGMOID

MODALITIES = [0, 1, 2, 3]
SUBSETS_MODALITIES = list(chain(*[combinations(MODALITIES, r) for r in range(1, 5)]))   # RA_HVED.py:733-738


class ReconDecoder(nn.Module):
    """RA_HVED.py:16-95."""

    def __init__(self, basic_module=DoubleConv, multi_stream=4, f_maps=64, shared_recon=True, MVAE=False, MVAE_reduction=False,
                 ks=3, num_groups=8, num_levels=4, layer_order="gcr", conv_kernel_size=3, conv_padding=1):
        super().__init__()
        streams, last_output = (1, 4) if shared_recon else (multi_stream, 1)
        f = number_of_features_per_level(f_maps, num_levels)
        rev = list(reversed(f))
        multi, finals = [], []
        for _ in range(streams):
            multi.append(nn.ModuleList([Decoder(rev[i] + rev[i + 1], rev[i + 1], basic_module=basic_module,
                                                conv_layer_order=layer_order, conv_kernel_size=conv_kernel_size,
                                                num_groups=num_groups, padding=conv_padding) for i in range(len(rev) - 1)]))
            finals.append(nn.Conv3d(f[0], last_output, 1))
        self.finals = nn.ModuleList(finals)
        self.multi_decoders = nn.ModuleList(multi)

    def forward(self, encoders_features, x, size_list=None):
        level_outputs = [[] for _ in encoders_features]
        finals = []
        for i, decs in enumerate(self.multi_decoders):
            out = x
            for j, (dec, enc) in enumerate(zip(decs, encoders_features)):
                out = dec(enc, out)
                level_outputs[j].append(out)
            finals.append(Fn.conv(out, [self.finals[i].weight], [self.finals[i].bias]))
        return level_outputs, finals


class Seg_Recon_DuSFEDecoder(nn.Module):
    """RA_HVED.py:97-201: recon decoder + seg decoder interleaved with DuSEAttention per level."""

    def __init__(self, sdecoders, basic_module=DoubleConv, multi_stream=4, f_maps=64, shared_recon=True, MVAE=False,
                 MVAE_reduction=False, ks=3, num_groups=8, seg=True, num_levels=4, layer_order="gcr", conv_kernel_size=3,
                 conv_padding=1):
        super().__init__()
        self.sdecoders = sdecoders
        streams, last_output = (1, 4) if shared_recon else (multi_stream, 1)
        f = number_of_features_per_level(f_maps, num_levels)
        rev = list(reversed(f))
        self.rfinals, self.sfinals = nn.ModuleList(), nn.ModuleList()
        self.multi_decoders = nn.ModuleList()
        dusfe = []
        for _ in range(streams):
            decs = nn.ModuleList()
            for i in range(len(rev) - 1):
                decs.append(Decoder(rev[i] + rev[i + 1], rev[i + 1], basic_module=basic_module, conv_layer_order=layer_order,
                                    conv_kernel_size=conv_kernel_size, num_groups=num_groups, padding=conv_padding))
                dusfe.append(DuSEAttention(rev[i + 1]))
            self.rfinals.append(nn.Conv3d(f[0], last_output, 1))
            self.sfinals.append(nn.Conv3d(f[0], last_output, 1))
            self.multi_decoders.append(decs)
        self.dusfe_decoders = nn.ModuleList(dusfe)

    def forward(self, encoders_features, x, size_list=None, seg=True):
        """Returns (level_outputs, [recon output per stream], [seg feature per stream] or None).  Every stream restarts from
        `x` (RA_HVED.py:171-183); the seg decoders and the FIRST three DuSE blocks are shared by all streams, exactly like the
        reference's zip over `self.dusfe_decoders` (RA_HVED.py:171)."""
        level_outputs = [[] for _ in encoders_features]
        rfinal, souts = [], []
        if self._pair_ok(x, seg):
            return self._forward_pair(encoders_features, x, level_outputs)
        for i, rdecs in enumerate(self.multi_decoders):
            rout = sout = x
            for j, (rdec, feat, sdec, dusfe) in enumerate(zip(rdecs, encoders_features, self.sdecoders, self.dusfe_decoders)):
                fused = seg and type(rdec.basic_module) is DoubleConv and type(sdec.basic_module) is DoubleConv \
                    and rdec.basic_module.SingleConv1.order == "ilc"
                if fused:                  # the decoders' last convs hand their output sums to DuSE's channel squeeze
                    # the level's feature has three consumers (recon decoder's concat, seg decoder's pooling and gating):
                    # aliases that share one gradient buffer (Fn.fanout) instead of two element-wise adds by autograd
                    f_r, f_p, f_g = Fn.fanout(feat, 3) if sdec.RSM else (feat, feat, feat)
                    rout, st_r = rdec(f_r, rout, out_stats=True)
                    sout, st_s = sdec((f_p, f_g) if sdec.RSM else feat, sout, out_stats=True)
                    rout, sout = dusfe(rout, sout, st_r, st_s)
                else:
                    rout = rdec(feat, rout)
                    if seg:
                        sout = sdec(feat, sout)
                        rout, sout = dusfe(rout, sout)
                level_outputs[j].append(rout)
            rfinal.append(Fn.conv(rout, [self.rfinals[i].weight], [self.rfinals[i].bias]))
            souts.append(sout)
        return level_outputs, rfinal, (souts if seg else None)


def _pair_methods():
    def _pair_ok(self, x, seg):
        """The recon | seg pair path (functional: "The recon | seg PAIR"): one recon stream, one sample, every level a fused 'ilc'
        DoubleConv pair with AttenModule2 in front of the seg stream."""
        if not (seg and ops.PAIR[0] and x.is_cuda and x.shape[0] == 1 and len(self.multi_decoders) == 1):
            return False
        for rdec, sdec in zip(self.multi_decoders[0], self.sdecoders):
            if not (type(rdec.basic_module) is DoubleConv and type(sdec.basic_module) is DoubleConv and sdec.RSM and not rdec.RSM
                    and rdec.basic_module.SingleConv1.order == "ilc" and sdec.basic_module.SingleConv1.order == "ilc"):
                return False
        return True

    def _forward_pair(self, encoders_features, x, level_outputs):
        pair = None                                            # [recon | seg] (1, 2C, ...) after each level's DuSE block
        for j, (rdec, feat, sdec, dusfe) in enumerate(zip(self.multi_decoders[0], encoders_features, self.sdecoders, self.dusfe_decoders)):
            f_r, f_p, f_g = Fn.fanout(feat, 3)
            size = tuple(feat.shape[2:])
            if pair is None:                                   # both streams start from x: ONE upsampling, two consumers
                u_r, u_s = Fn.fanout(Fn.Upsample.apply(x, size), 2)
            else:
                u_r, u_s = Fn.upsample2(pair, size)
            rb, sb = rdec.basic_module, sdec.basic_module
            c = rb.SingleConv1.conv.out_channels
            y1 = torch.empty((1, 2 * c) + size, dtype=x.dtype, device=x.device)          # the first convs' outputs, side by side
            red1 = ops.zeros_red(x, 1, 2 * c)
            g_s, st_s = sdec.atten_module(u_s, (f_p, f_g), stats=True)
            # the two first convs have the same shape and different inputs (recon: the virtual concat [feature | upsampled]; seg: the
            # gated concat): one launch for both where the full-row kernel takes them (ops.conv_pair_scope)
            with ops.conv_pair_scope():
                y1_r, _ = Fn.in_lrelu_conv(f_r, u_r, [rb.SingleConv1.conv.weight], [rb.SingleConv1.conv.bias], out_stats=True,
                                           drop_bias=True, into=(y1[:, :c], red1[:, :c]))
                y1_s, _ = Fn.in_lrelu_conv(g_s, None, [sb.SingleConv1.conv.weight], [sb.SingleConv1.conv.bias], in_stats=st_s,
                                           out_stats=True, drop_bias=True, into=(y1[:, c:], red1[:, c:]))
            y2, st2 = Fn.InLreluConv2.apply(y1_r, y1_s, y1, red1, rb.SingleConv2.conv.weight, sb.SingleConv2.conv.weight,
                                            rb.SingleConv2.conv.bias, sb.SingleConv2.conv.bias)
            pair = dusfe.forward_pair(y2, st2)
            level_outputs[j].append(pair[:, :c])
        rout, sout = Fn.split2(pair)
        rfinal = [Fn.conv(rout, [self.rfinals[0].weight], [self.rfinals[0].bias])]
        return level_outputs, rfinal, [sout]
    return _pair_ok, _forward_pair


Seg_Recon_DuSFEDecoder._pair_ok, Seg_Recon_DuSFEDecoder._forward_pair = _pair_methods()


def x_is_shared_decoder(model):
    """One recon stream (shared_recon=True: the training configuration) with a single composed seg head."""
    sr = getattr(model, "srdecoder", None)
    return sr is not None and len(sr.multi_decoders) == 1 and len(sr.sfinals) == 1 and all(d.RSM for d in sr.sdecoders)


class AbstractFusion3DUNet(nn.Module):
    """RA_HVED.py:239-687."""

    def __init__(self, in_channels, out_channels, final_sigmoid, basic_module, f_maps=64, layer_order="gcr", multi_stream=4,
                 fusion_level=4, recon_decoder=False, seg_recon_decoder=False, skip_return=False, ori_vae_fusion=False,
                 shared_recon=True, recon_skip=False, fusion=False, MVAE=False, MVAE_reduction=False, num_groups=8,
                 num_levels=4, num_block=(1, 1, 1, 1), conv_kernel_size=3, pool_kernel_size=2, conv_padding=1, ViL=False,
                 mid_ViL=False, **kwargs):
        super().__init__()
        if not (MVAE and MVAE_reduction and basic_module is DoubleConv and multi_stream == 4 and fusion_level == 4
                and not fusion and not ori_vae_fusion and recon_skip and final_sigmoid and num_levels == 4):
            raise NotImplementedError(
                "only the configurations the reference can run are built: MVAE=True, MVAE_reduction=True, DoubleConv, "
                "multi_stream=4, fusion_level=4, recon_skip=True, final_sigmoid=True (SURVEY.md F10)")
        if isinstance(f_maps, int):
            enc = number_of_features_per_level(f_maps, num_levels)
        else:
            enc = list(f_maps)
        dec = list(enc)
        self.multi_stream, self.fusion_level = multi_stream, fusion_level
        self.recon_decoder, self.recon_skip = recon_decoder, recon_skip
        self.seg_recon_decoder, self.skip_return = seg_recon_decoder, skip_return
        self.MVAE, self.MVAE_reduction, self.mid_ViL = MVAE, MVAE_reduction, mid_ViL
        self.layer_order = layer_order
        self.latent_dims = 128
        self.experts, self.experts_drop = ProductOfExperts(), ProductOfExperts2()
        self.MVAE_latents = number_of_features_per_level(enc[0] // 4, num_levels)
        self.reduction_latents = list(self.MVAE_latents)
        if mid_ViL:
            self.mViL = ViLLayer(dim=dec[-1])
        if skip_return:
            self.x0_init = nn.Sequential(nn.Conv3d(enc[0], enc[0], 1))
        mk_enc = lambda cin, cout, pool: Encoder(cin, cout, num_block=1, apply_pooling=pool, basic_module=basic_module,
                                                 conv_layer_order=layer_order, conv_kernel_size=conv_kernel_size,
                                                 num_groups=num_groups, pool_kernel_size=pool_kernel_size, padding=conv_padding)
        init_blocks, encoders, DRBs, VU, conv_blocks, skr_enc, skr_att = [], [], [], [], [], [], []
        for i, c in enumerate(enc):
            if i == 0:
                init_blocks = [nn.Sequential(nn.Conv3d(in_channels, c, 1)) for _ in range(4)]
            encoders.append(nn.ModuleList([mk_enc(c if i == 0 else enc[i - 1], c, i > 0) for _ in range(4)]))
            if i > 0 and skip_return:
                if i == 1:
                    skr_att.insert(0, SkipReturnAttention(enc[0]))
                skr_enc.insert(0, mk_enc(enc[i - 1], c, True))
                skr_att.insert(0, SkipReturnAttention(c))
            DRBs.append(nn.ModuleList([nn.Sequential(SingleConv(c, self.MVAE_latents[i] * 2, conv_kernel_size, 2, layer_order,
                                                                num_groups, padding=conv_padding)) for _ in range(4)]))
            VU.append(nn.Sequential(BasicConv(self.MVAE_latents[i], dec[i], 1)))
            conv_blocks.append(BasicConv(dec[i], dec[i], 3, padding=1, groups=dec[i]))
        self.DRBs = nn.ModuleList(DRBs)
        self.VU_blocks = nn.ModuleList(VU)
        self.init_blocks = nn.ModuleList(init_blocks)
        self.encoders = nn.ModuleList(encoders)
        self.conv_blocks = nn.ModuleList(conv_blocks)
        if skip_return:
            self.skr_encoders = nn.ModuleList(skr_enc)
            self.skr_att = nn.ModuleList(skr_att)
        rev = list(reversed(dec))
        self.decoders = nn.ModuleList([
            Decoder(rev[i] + rev[i + 1], rev[i + 1], basic_module=DoubleConv_ViL if (ViL and i < 1) else basic_module,
                    conv_layer_order=layer_order, conv_kernel_size=conv_kernel_size, num_groups=num_groups,
                    padding=conv_padding, RSM=True, MVAE=MVAE) for i in range(len(rev) - 1)])
        self.final_conv = nn.Conv3d(dec[0], out_channels, 1)
        self.final_activation = nn.Sigmoid()
        if recon_decoder:
            self.rdecoder = ReconDecoder(basic_module=basic_module, multi_stream=multi_stream, f_maps=dec[0],
                                         shared_recon=shared_recon, MVAE=MVAE, MVAE_reduction=MVAE_reduction,
                                         layer_order=layer_order)
        if seg_recon_decoder:
            self.srdecoder = Seg_Recon_DuSFEDecoder(sdecoders=self.decoders, basic_module=basic_module,
                                                    multi_stream=multi_stream, f_maps=dec[0], shared_recon=shared_recon,
                                                    MVAE=MVAE, MVAE_reduction=MVAE_reduction, layer_order=layer_order)

    # ------------------------------------------------------------------------------------------------
    def noise_state(self, device):
        """The reparameterisation-noise generator of this model on `device`: ops.rng_state = {seed, draw counter, ticket words} (xh_poe_multi).
        Created on first use with a seed drawn from torch's default CPU generator (so torch.manual_seed governs it, as it governs
        RA_HVED.py:744's normal_()); seed_noise() sets it explicitly (data parallel: base seed + rank)."""
        states = self.__dict__.setdefault("_xh_rng", {})
        key = str(device)
        if key not in states:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the model's noise state must exist before a stream capture: run one eager forward first, or call "
                                   "model.noise_state(device)")
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            states[key] = ops.rng_state(seed, device)
        return states[key]

    def seed_noise(self, seed, device=None):
        """Restart the reparameterisation noise from (seed, draw 0).  In place: a captured graph keeps drawing from the same words."""
        devs = [str(device)] if device is not None else list(self.__dict__.get("_xh_rng", {}))
        for key in devs:
            st = self.noise_state(torch.device(key))
            st.copy_(ops.rng_state(seed, "cpu"))

    def _stream_weights(self, level, which):
        mods = [getattr(e.basic_module[0], which).conv for e in self.encoders[level]]
        return [m.weight for m in mods], [m.bias for m in mods]

    def _keep_mask(self, x, subset_idx_list, instance_missing, drop):
        n = x.shape[0]
        if torch.is_tensor(subset_idx_list):
            # a device-resident (N, 4) keep mask instead of a subset index (not in the reference signature): the form a captured
            # hipGraph needs, since train.py:222-223 draws a new subset every step and a mask baked at capture would freeze it
            if instance_missing or tuple(subset_idx_list.shape) != (n, 4) or subset_idx_list.dtype != torch.float32:
                raise ValueError("a subset mask is an (N, 4) float32 tensor and excludes instance_missing")
            return subset_idx_list.to(x.device).contiguous()
        if instance_missing:
            if drop is None:
                drop = x.float().sum((2, 3, 4)) == 0                                       # RA_HVED.py:515
            keep = (~drop.to(torch.bool)).to(torch.float32)
        else:
            key = (subset_idx_list[0], n, x.device)                                         # RA_HVED.py:517-520
            cache = self.__dict__.setdefault("_keep_cache", {})
            if key not in cache:
                subset = SUBSETS_MODALITIES[subset_idx_list[0]]
                row = torch.tensor([[1.0 if k in subset else 0.0 for k in range(4)]], dtype=torch.float32)
                cache[key] = row.repeat(n, 1).to(x.device).contiguous()
            return cache[key]
        return keep.to(x.device).contiguous()

    def forward(self, x, subset_idx_list=[14], instance_missing=False, drop=None, seg=True, recon=False, valid=False,
                eps_list=None):
        """RA_HVED.py:510-648.  `eps_list` (optional, not in the reference signature) injects the reparameterisation
        noise per level for parity tests; by default it is drawn with torch.randn like RA_HVED.py:744."""
        # `self.fp32_arith` (optional attribute: "split" | "vector" | None = the process default, ops.set_fp32_mfma): the arithmetic of
        # fp32 storage for THIS model's calls -- forward here, backward through the Functions' saved mode (functional.Function)
        with ops.arith_scope(getattr(self, "fp32_arith", None)), self._bn_counters():
            # composed first: ComposeAll is then the LAST node of the backward pass, so the weight gradients of the composed
            # tensors' convs can wait for the end-of-backward batch (functional._direct)
            pre = self._precompose(seg, x)
            try:
                enc = self._encode(x, bn_steps=4)
                return self._decode(enc, subset_idx_list, instance_missing, drop, seg, recon, valid, eps_list)
            finally:
                self._drop_precomposed(pre)

    def forward_shared(self, x, calls, seg=True, recon=False):
        """Several forwards of the SAME input that differ only in the modality subset / sampling (train.py:224-225 runs
        `model(x, [14])` and `model(x, subset)` back to back): init blocks, encoders, DRBs and the skip-return path are
        functions of the input alone (mu/logvar stacks are bit-identical between such calls), so they run
        once and only PoE -> reparameterisation -> decoders run per call.  `calls` is a list of dicts with the per-call
        keyword arguments of forward() (subset_idx_list, instance_missing, drop, valid, eps_list).  Returns the list of
        forward() results.  BatchNorm buffers of the skip-return attention advance 4 steps per call, as they would."""
        with ops.arith_scope(getattr(self, "fp32_arith", None)), self._bn_counters():
            pre = self._precompose(seg, x)                    # the composed weights are the same for every call
            outs = []
            try:
                enc = self._encode(x, bn_steps=4 * len(calls))
                for kw in calls:
                    outs.append(self._decode(enc, kw.get("subset_idx_list", [14]), kw.get("instance_missing", False), kw.get("drop"),
                                             seg, recon, kw.get("valid", False), kw.get("eps_list")))
            finally:
                self._drop_precomposed(pre)
        return outs

    def _precompose(self, seg, x=None):
        """All parameter compositions of a forward in ONE launch (Fn.ComposeAll): the AttenModule2 gates, the DuSE blocks and
        the segmentation head, each of which otherwise composes its weights with a launch (or a handful of ATen ops) of its own
        right before it is used -- and scatters the gradients back with another; with 16-bit storage also the skip-return
        ResBlocks' depthwise o pointwise pairs (one dense 3^3 conv each instead of two launches and a tensor in between; fp32
        storage keeps the two-conv form: its vector kernels pay for the denser weights).  The composed tensors are parked on the
        modules for the duration of this forward; a module called on its own composes for itself as before."""
        dec = self.seg_recon_decoder and seg and x_is_shared_decoder(self)
        attens, duses = [], []
        if dec:
            sr = self.srdecoder
            attens = [d.atten_module for d in sr.sdecoders if d.RSM]
            duses = list(sr.dusfe_decoders)[:len(sr.multi_decoders[0])]
            if not attens or len(attens) > 4 or len(duses) > 4:
                dec, attens, duses = False, [], []
        skrs = []
        # (fp32 storage on the matrix cores -- arith "split" -- takes the dense form too: the two-term split kernel serves a dense
        # 4 -> 4 conv, while the separable form falls to the generic fp32 stencils: 0.28 ms of the 6.07 ms step in round 5)
        if self.skip_return and x is not None and ops.SEP_COMPOSE[0] and (x.dtype != torch.float32 or (ops.current_arith() & 1)):
            levels = len(self.encoders)
            skrs = [self.skr_att[levels - level] for level in range(1, levels)]         # the ones forward() calls (RA_HVED.py:552)
            if len(skrs) > 4:
                skrs = []
        if not dec and not skrs:
            return []
        params = []
        for a in attens:
            params += a.compose_params()
        for d in duses:
            params += d.compose_params()
        if dec:
            params += [self.final_conv.weight, self.final_conv.bias, sr.sfinals[0].weight, sr.sfinals[0].bias]
        s_plan = []
        for k in skrs:
            params += k.compose_params()
            s_plan += [k[0].conv1.dwconv.weight.shape[0]] * 2
        plan = ([(2, 4, a.expan) for a in attens], [d.conv_squeeze_ch1.in_channels for d in duses], dec, tuple(s_plan))
        outs = Fn.ComposeAll.apply(plan, *params)
        if outs[0].requires_grad and any(getattr(o, "_xh_gbuf", None) is None for o in outs):
            raise RuntimeError("ComposeAll: the composed tensors lost their gradient buffers")     # (Function.apply returns forward's own tensors)
        mods, oi = [], 0
        for a in attens:
            a.__dict__["_pre"] = (outs[oi], outs[oi + 1])
            oi += 2
            mods.append(a)
        for d in duses:
            d.__dict__["_pre"] = tuple(outs[oi:oi + 4])
            oi += 4
            mods.append(d)
        if dec:
            self.__dict__["_head_pre"] = (outs[oi], outs[oi + 1])
            oi += 2
        for k in skrs:
            k.__dict__["_pre"] = (outs[oi], outs[oi + 1])
            oi += 2
            mods.append(k)
        return mods

    def _drop_precomposed(self, mods):
        for m_ in mods:
            m_.__dict__.pop("_pre", None)
        self.__dict__.pop("_head_pre", None)

    def _bn_counters(self):
        """One flat update of all BatchNorm `num_batches_tracked` counters per forward (blocks.BNCounters)."""
        from .blocks import BNCounters
        c = self.__dict__.get("_bnc")
        if c is None:
            c = self.__dict__["_bnc"] = BNCounters()
        return c.collect(self)

    def _encode5(self, x, bn_steps):
        """_encode for one sample per launch with the SKIP stream as a fifth group of the modality streams' launches.  From level 1
        on the skip-return encoder (RA_HVED.py:374-381,617-621: MaxPool3d(2) -> DoubleConv on the skip feature) has, stage by
        stage, the shapes of ONE modality encoder of the same level and no dependency on them: its pooling rides in the gate +
        max-pool launch (ungated channels, Fn.GateMaxPool5), its two convs are group 5 of the grouped conv launches (one weight
        pointer per group), and so do their data gradients, norm-backward passes and pooling backward.  [X (4C) | S (C)] live in one
        (1, 5C, ...) tensor; with N == 1 its channel ranges are plain contiguous views (Fn.slice_view) for the consumers that want
        one part: the DRB convs (X) and the skip-return attention (S)."""
        ops.red_arena_reset(x.device)
        Fn.nb_pending_clear()
        ops.prepack_all()
        x = x.contiguous()
        levels = len(self.encoders)
        feat_list = []
        # level 0: the four streams' DoubleConv; its second conv and x0_init write the two parts of one buffer
        w, b = self._stream_weights(0, "SingleConv1")
        iw, ib = [blk[0].weight for blk in self.init_blocks], [blk[0].bias for blk in self.init_blocks]
        if Fn.init_fold_ok(x, iw, w):
            Y, st = Fn.init_in_lrelu_conv(x, iw, ib, w, b)      # the init blocks' output is never stored (Fn.InitInLreluConv)
        else:
            X, st0 = Fn.conv(x, iw, ib, groups=4, out_stats=True, drop_bias=True)
            Y, st = Fn.in_lrelu_conv(X, None, w, b, 1, 4, in_stats=st0, out_stats=True, drop_bias=True)
        w, b = self._stream_weights(0, "SingleConv2")
        c4 = sum(t.shape[0] for t in w)
        c = c4 // 4
        xs_buf = torch.empty((1, c4 + c) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)
        red_buf = ops.zeros_red(x, 1, c4)
        X, st = Fn.in_lrelu_conv(Y, None, w, b, 1, 4, in_stats=st, out_stats=True, sole_consumer=True, into=(xs_buf[:, :c4], red_buf))
        S = Fn.conv(x, [self.x0_init[0].weight], [self.x0_init[0].bias], into=xs_buf[:, c4:])
        X_drb, X_main = Fn.fanout(X, 2)
        drb = [m[0].conv for m in self.DRBs[0]]
        feat_list.append(Fn.in_lrelu_conv(X_drb, None, [m.weight for m in drb], [m.bias for m in drb], 2, 4, in_stats=st))
        parts, base = (X_main, S), xs_buf
        skip = None
        for level in range(1, levels):
            if base is not None:                           # level 1: X and S are the two parts of level 0's buffer
                s_att, s_pool = Fn.fanout(parts[1], 2)
                a = self.skr_att[levels - level](s_att, steps=bn_steps)
                P, stp = Fn.GateMaxPool5.apply(a, base, parts[0], s_pool)
            else:
                a = self.skr_att[levels - level](parts[1], steps=bn_steps)
                P, stp = Fn.GateMaxPool5.apply(a, None, parts[0])
            enc_s = self.skr_encoders[levels - 1 - level].basic_module[0]
            w, b = self._stream_weights(level, "SingleConv1")
            w, b = w + [enc_s.SingleConv1.conv.weight], b + [enc_s.SingleConv1.conv.bias]
            Y, st = Fn.in_lrelu_conv(P, None, w, b, 1, 5, in_stats=stp, out_stats=True, drop_bias=True)
            w, b = self._stream_weights(level, "SingleConv2")
            w, b = w + [enc_s.SingleConv2.conv.weight], b + [enc_s.SingleConv2.conv.bias]
            XS, st = Fn.in_lrelu_conv(Y, None, w, b, 1, 5, in_stats=st, out_stats=True, sole_consumer=True)
            c5 = XS.shape[1]
            c4 = c5 // 5 * 4
            if level + 1 < levels:
                xs_d, xs_g, xs_a = Fn.fanout(XS, 3)
                parts, base = (xs_g, Fn.slice_view(xs_a, c4, c5)), None
                x_drb = Fn.slice_view(xs_d, 0, c4)
            else:
                x_drb, skip = Fn.split_at(XS, c4)         # last level: the DRB convs take X, the mid ViL block takes S
            drb = [m[0].conv for m in self.DRBs[level]]
            feat_list.append(Fn.in_lrelu_conv(x_drb, None, [m.weight for m in drb], [m.bias for m in drb], 2, 4, in_stats=st[:, :c4]))
        return x, feat_list, skip

    def _encode5_ok(self, x):
        if not (ops.STREAM5[0] and self.layer_order == "ilc" and self.skip_return and x.is_cuda and x.shape[0] == 1 and len(self.encoders) >= 2):
            return False
        for level in range(1, len(self.encoders)):
            e = self.skr_encoders[len(self.encoders) - 1 - level]
            if not (e.pooling is not None and len(e.basic_module) == 1 and type(e.basic_module[0]) is DoubleConv
                    and e.basic_module[0].SingleConv1.order == "ilc"):
                return False
            m0 = self.encoders[level][0].basic_module[0]
            for which in ("SingleConv1", "SingleConv2"):
                if getattr(m0, which).conv.weight.shape != getattr(e.basic_module[0], which).conv.weight.shape:
                    return False
        d, h, w_ = x.shape[2:]
        nl = len(self.encoders)                             # every pooled level: even D / H, rows of a multiple of 8 voxels
        return d % (1 << nl) == 0 and h % (1 << nl) == 0 and w_ % (8 << (nl - 2)) == 0 and x.dtype in (torch.float32, torch.bfloat16, torch.float16)

    def _encode(self, x, bn_steps):
        """The input-only part: per-level DRB outputs (4 streams x [mu | logvar] before PoE) and the skip-return feature."""
        if self._encode5_ok(x):
            return self._encode5(x, bn_steps)
        batched = self.layer_order == "ilc"     # 'ilc' runs the 4 modality streams as one grouped launch per stage
        ops.red_arena_reset(x.device)
        Fn.nb_pending_clear()                   # (entries an aborted backward left behind)
        ops.prepack_all()                       # the MFMA weight fragments of every k=3 conv of the step: one launch per 24
        x = x.contiguous()
        st0 = None
        iw, ib = [blk[0].weight for blk in self.init_blocks], [blk[0].bias for blk in self.init_blocks]
        fold = batched and Fn.init_fold_ok(x, iw, self._stream_weights(0, "SingleConv1")[0])
        if fold:
            X = x                                   # the init blocks are folded into level 0's first conv (Fn.InitInLreluConv)
        elif batched:
            # level 0's first InstanceNorm takes its sums from here; it is the init blocks' only consumer, so their bias add
            # is an identity (drop_bias: a channel w*x + b with a small w would otherwise spend its 16-bit mantissa on b)
            X, st0 = Fn.conv(x, iw, ib, groups=4, out_stats=True, drop_bias=True)
        else:
            X = [Fn.conv(x[:, i:i + 1].contiguous(), [b[0].weight], [b[0].bias]) for i, b in enumerate(self.init_blocks)]
        feat_list = []
        skip = None
        levels = len(self.encoders)
        for level in range(levels):
            stp = None
            if self.skip_return and skip is not None:
                # the skip feature feeds this level's attention and the next skip encoder: two consumers, one gradient buffer
                skip_att, skip = Fn.fanout(skip, 2)
                a = self.skr_att[levels - level](skip_att, steps=bn_steps)                  # skr_att[-level], RA_HVED.py:552
                if batched and level > 0 and ops.gate_maxpool_ok(X, a):
                    X, stp = Fn.GateMaxPool.apply(X, a)        # gate, this level's pooling and its first norm's sums: one pass
                else:
                    X = Fn.Gate.apply(X, a) if batched else [Fn.Gate.apply(xi, a) for xi in X]
            if batched:
                if level > 0 and stp is None:
                    X = Fn.MaxPool2.apply(X)
                # each conv's epilogue accumulates the channel sums the next InstanceNorm needs (no separate pass)
                w, b = self._stream_weights(level, "SingleConv1")
                if level == 0 and fold:
                    X, st = Fn.init_in_lrelu_conv(x, iw, ib, w, b)
                else:
                    X, st = Fn.in_lrelu_conv(X, None, w, b, 1, 4, in_stats=st0 if level == 0 else stp, out_stats=True,
                                             drop_bias=True)              # consumed by SingleConv2's InstanceNorm only
                w, b = self._stream_weights(level, "SingleConv2")
                X, st = Fn.in_lrelu_conv(X, None, w, b, 1, 4, in_stats=st, out_stats=True, sole_consumer=True)
                drb = [m[0].conv for m in self.DRBs[level]]
                # this level's output feeds its DRB and (gated, pooled) the next level: two consumers, one gradient buffer
                X_drb, X = Fn.fanout(X, 2) if (level + 1 < levels and self.skip_return) else (X, X)
                feat = Fn.in_lrelu_conv(X_drb, None, [m.weight for m in drb], [m.bias for m in drb], 2, 4, in_stats=st)   # RA_HVED.py:569
            else:
                # 'gcr' (U_HVEDConvNet3D / U_HVEDConvXLSTMNet3D defaults): one stream at a time through the same HIP stages
                X = [enc(xi) for enc, xi in zip(self.encoders[level], X)]
                feat = torch.cat([m(xi) for m, xi in zip(self.DRBs[level], X)], 1)
            feat_list.append(feat)
            if self.skip_return:                                                            # RA_HVED.py:617-621
                if skip is None:
                    skip = Fn.conv(x, [self.x0_init[0].weight], [self.x0_init[0].bias])
                else:
                    skip = self.skr_encoders[levels - 1 - level](skip)
        return x, feat_list, skip

    def _decode(self, enc, subset_idx_list, instance_missing, drop, seg, recon, valid, eps_list):
        """PoE over the chosen experts -> reparameterisation -> VU blocks -> (mid ViL) -> decoders -> heads."""
        x, feat_list, skip = enc
        n = x.shape[0]
        keep = self._keep_mask(x, subset_idx_list, instance_missing, drop)
        sdt = x.dtype
        if ops.MIXED[0] is not None and x.dtype == torch.float32 and x.is_cuda:
            # mixed storage (ops.set_mixed_storage): the encoder half ran in fp32 storage; everything the decoder half sees of it
            # passes through the DRB outputs and the coarsest skip feature (RA_HVED.py:569-626) -- small tensors, cast here
            sdt = ops.MIXED[0]
            feat_list = [f.to(sdt) for f in feat_list]
            skip = skip.to(sdt) if skip is not None else None
        mu_list, logvar_list, feats = [], [], []
        noise = None
        nlev = len(feat_list)
        # RA_HVED.py:744 draws N(0,1) noise per level (fp32, `std.data.new(std.size()).normal_()`).  On the device the PoE kernel
        # draws it itself: Philox4x32-10 keyed by (this model's seed, a device-resident draw counter the launch advances, level,
        # element) -- fp32 whatever the storage type, no noise tensor, no generator launch, and a replayed hipGraph draws fresh
        # noise every replay; the backward pass regenerates the same values (functional.PoEAll)
        in_kernel = not valid and eps_list is None and x.is_cuda and nlev <= ops.POE_MAX
        if not valid and eps_list is None and not in_kernel:
            # (more levels than one launch takes / host tensors: ONE torch draw in the storage type serves the levels)
            shapes = [(n, self.MVAE_latents[l]) + tuple(f.shape[2:]) for l, f in enumerate(feat_list)]
            sizes = [int(torch.Size(s_).numel()) for s_ in shapes]
            flat = torch.randn(sum(sizes), device=x.device, dtype=sdt)
            noise, o = [], 0
            for s_, k in zip(shapes, sizes):
                noise.append(flat[o:o + k].view(s_))
                o += k
        # The four per-level chains (PoE -> VU block -> upsample -> conv block) are independent of each other.  Running the three
        # coarse ones on side streams inside the captured graph was measured SLOWER (5.96 -> 6.93 ms per step, DESIGN.md
        # section 7) and the switch has been removed; they run back to back on the caller's stream.
        outs = [None] * len(feat_list)
        epss = [None] * nlev
        if not valid and not in_kernel:
            epss = [noise[l] if noise is not None else eps_list[l].to(device=x.device, dtype=sdt).contiguous() for l in range(nlev)]
        # the PoE of all levels in one launch (they depend on the encoder outputs only)
        if nlev <= ops.POE_MAX:
            if in_kernel:
                Fn.PoEAll.rng_state = self.noise_state(x.device)
            zml = Fn.PoEAll.apply(keep, tuple(self.MVAE_latents[:nlev]), bool(instance_missing), nlev, *feat_list, *epss)
        else:
            zml = [t for l in range(nlev) for t in Fn.PoE.apply(feat_list[l], keep, epss[l], self.MVAE_latents[l], bool(instance_missing))]
        batch = (Fn.LATENT_BATCH[0] and x.is_cuda and 1 < nlev <= 4 and not (ops.LEVEL_STREAMS[0] and not torch.cuda.is_current_stream_capturing())
                 and all(type(self.VU_blocks[l][0]).__name__ == "BasicConv" and self.VU_blocks[l][0].conv.kernel_size[0] == 1
                         and self.VU_blocks[l][0].groups == 1 and type(self.conv_blocks[l]).__name__ == "BasicConv" for l in range(nlev))
                 and all(zml[3 * l].shape[-1] % 4 == 0 for l in range(nlev)))
        if batch:
            # all levels as ONE autograd node with multi-problem launches for the element-wise passes (Fn.LatentPath)
            zs = [zml[3 * l] for l in range(nlev)]
            try:
                feats_l = Fn.LatentPath.apply(nlev, tuple(self.conv_blocks[l].groups for l in range(nlev)), *zs,
                                              *[self.VU_blocks[l][0].conv.weight for l in range(nlev)],
                                              *[self.conv_blocks[l].conv.weight for l in range(nlev)])
                for level in range(nlev):
                    outs[level] = (feats_l[level], zml[3 * level + 1], zml[3 * level + 2])
            except Fn.LatentFallback:
                batch = False
        side = None
        # (A/B switch, EAGER only.  Under stream capture it is inert: as parallel branches of a graph the chains really run
        # concurrently, and tensors that are allocated on one stream and last read on another -- PoE outputs saved by a side-stream
        # node, gradients handed between streams -- can be recycled by the allocator while the other branch still reads them;
        # round 6 saw gradients off by 1e-3..1e-2 in one capture out of three.  Measured slower than one stream anyway)
        if ops.LEVEL_STREAMS[0] and x.is_cuda and nlev > 1 and not torch.cuda.is_current_stream_capturing():
            pool = self.__dict__.setdefault("_level_streams", {}).setdefault(x.device, [])
            while len(pool) < nlev - 1:
                pool.append(torch.cuda.Stream(x.device))
            side, cur = pool[:nlev - 1], torch.cuda.current_stream(x.device)
        for level in range(0 if not batch else nlev, nlev):
            z, mu, lv = zml[3 * level:3 * level + 3]
            st = side[level - 1] if side is not None and level >= 1 else None
            if st is not None:
                st.wait_stream(cur)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                z = self.VU_blocks[level][0](z, up2x=True)                                  # RA_HVED.py:599-601 (conv block + 2x upsampling)
                z = self.conv_blocks[level](z)                                              # RA_HVED.py:603
            if st is not None:
                z.record_stream(cur)
            outs[level] = (z, mu, lv)
        if side is not None:
            for st in side:
                cur.wait_stream(st)
        for z, mu, lv in outs:
            mu_list.append(mu)
            logvar_list.append(lv)
            feats.insert(0, z)
        if self.mid_ViL and self.skip_return:                                               # RA_HVED.py:623-626
            feats[0] = self.mViL(feats[0], skip, residual_input=True)
        if not self.recon_skip:
            raise NotImplementedError
        recon_x, recon_features = feats[0], feats[1:]
        if self.seg_recon_decoder:                                                          # RA_HVED.py:637-648
            _, recon_outputs, sout = self.srdecoder(recon_features, recon_x, seg=seg)
            seg_outputs = self._seg_head(sout, list(self.srdecoder.sfinals)) if seg else None
            if recon and self.recon_decoder:
                return seg_outputs, (mu_list, logvar_list), recon_outputs
            return seg_outputs, []
        _, recon_outputs = self.rdecoder(recon_features, recon_x)                           # RA_HVED.py:651-652
        recon_outputs = recon_outputs[0] if len(recon_outputs) == 1 else torch.cat(recon_outputs, 1)
        out = None
        if seg:                                                                             # RA_HVED.py:666-678
            out = recon_x
            for dec, f in zip(self.decoders, recon_features):
                out = dec(f, out)
            out = self._seg_head(out, None)
        else:
            out = recon_x
        if recon and self.recon_decoder:
            return out, (mu_list, logvar_list), recon_outputs
        return out, []

    def _seg_head(self, souts, sfinals):
        """sigmoid(final_conv(cat_i sfinals[i](sout_i))) (RA_HVED.py:192-199,640-641).  With the shared decoder (one stream)
        the two 1x1 convs compose into ONE 1x1 conv.  The composition is parameter-sized fp32 arithmetic and must stay fp32
        under a caller's `with autocast():` (train.py:218), hence the explicit autocast-off region."""
        pre = self.__dict__.get("_head_pre")
        if pre is not None and sfinals is not None and len(sfinals) == 1:      # composed in the step's batched launch
            return Fn.conv(souts[0], [pre[0]], [pre[1]], act=ACT_SIGMOID)
        with torch.autocast(device_type=self.final_conv.weight.device.type, enabled=False):
            wf = self.final_conv.weight.float().view(self.final_conv.out_channels, -1)
            bf = self.final_conv.bias.float()
            if sfinals is None:
                sout = souts
            elif len(sfinals) == 1:
                sout = souts[0]
                ws = sfinals[0].weight.float().view(sfinals[0].out_channels, -1)
                bf = wf @ sfinals[0].bias.float() + bf
                wf = wf @ ws
            else:
                # shared_recon=False (Pretrain.py:142): one 1-channel sfinals conv per stream, concatenated (RA_HVED.py:199)
                sout = torch.cat([Fn.conv(so, [sf.weight], [sf.bias]) for so, sf in zip(souts, sfinals)], 1)
            wf = wf.reshape(wf.shape[0], wf.shape[1], 1, 1, 1).contiguous()
        return Fn.conv(sout, [wf], [bf], act=ACT_SIGMOID)

    def seg_parameters(self):
        """RA_HVED.py:496-500 (the reference lists a non-existent atten_blocks; omitted)."""
        mods = [self.init_blocks, self.encoders, self.DRBs, self.VU_blocks, self.decoders, self.final_conv]
        return [p for m in mods for p in m.parameters()]

    def rd_parameters(self):
        return self.rdecoder.parameters()


def _variant(name, doc, **flags):
    def __init__(self, in_channels, out_channels, multi_stream=4, fusion_level=4, final_sigmoid=True, f_maps=8, layer_order="gcr",
                 num_groups=8, num_levels=4, recon_decoder=True, MVAE=True, is_segmentation=True, conv_padding=1, **kwargs):
        merged = dict(flags)
        merged.update(kwargs)
        AbstractFusion3DUNet.__init__(self, in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid,
                                      basic_module=DoubleConv, f_maps=f_maps, layer_order=layer_order, multi_stream=multi_stream,
                                      fusion_level=fusion_level, num_groups=num_groups, num_levels=num_levels,
                                      conv_padding=conv_padding, recon_decoder=recon_decoder, MVAE=MVAE, **merged)
    return type(name, (AbstractFusion3DUNet,), {"__init__": __init__, "__doc__": doc})


XLSTM_HVED = _variant("XLSTM_HVED", "RA_HVED.py:945-958", seg_recon_decoder=True, skip_return=True, mid_ViL=True)
U_HVEDConvDuSFEmViLSkrNet3D = _variant("U_HVEDConvDuSFEmViLSkrNet3D", "RA_HVED.py:907-920", seg_recon_decoder=True,
                                       skip_return=True, mid_ViL=True)
XLSTM_HVED_woViL = _variant("XLSTM_HVED_woViL", "RA_HVED.py:1103-1116", seg_recon_decoder=True, skip_return=True, mid_ViL=False)
XLSTM_HVED_woSMVAE = _variant("XLSTM_HVED_woSMVAE", "RA_HVED.py:983-996", seg_recon_decoder=True, skip_return=False, mid_ViL=True)
U_HVEDConvDuSFEmViLNet3D = _variant("U_HVEDConvDuSFEmViLNet3D", "RA_HVED.py:869-882", seg_recon_decoder=True, skip_return=False,
                                    mid_ViL=True)
U_HVEDConvDuSFESkrNet3D = _variant("U_HVEDConvDuSFESkrNet3D", "RA_HVED.py:831-844", seg_recon_decoder=True, skip_return=True)
U_HVEDConvDuSFENet3D = _variant("U_HVEDConvDuSFENet3D", "RA_HVED.py:793-806", seg_recon_decoder=True)
XLSTM_HVED_woDuSFE = _variant("XLSTM_HVED_woDuSFE", "RA_HVED.py:1062-1075", seg_recon_decoder=False, skip_return=True, mid_ViL=True)
U_HVEDConvNet3D = _variant("U_HVEDConvNet3D", "RA_HVED.py:717-730")
U_HVEDConvXLSTMNet3D = _variant("U_HVEDConvXLSTMNet3D", "RA_HVED.py:755-768", ViL=True)

MODELS = {c.__name__: c for c in (XLSTM_HVED, U_HVEDConvDuSFEmViLSkrNet3D, XLSTM_HVED_woViL, XLSTM_HVED_woSMVAE,
                                  U_HVEDConvDuSFEmViLNet3D, U_HVEDConvDuSFESkrNet3D, U_HVEDConvDuSFENet3D, XLSTM_HVED_woDuSFE,
                                  U_HVEDConvNet3D, U_HVEDConvXLSTMNet3D)}


def find_model_using_name(model_name):
    """classic_models/__init__.py:16-29 (the reference's registry raises NameError on import)."""
    return MODELS[model_name]

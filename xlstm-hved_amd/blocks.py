"""nn.Module mirror of the reference building blocks on the hot path.

Same class names, constructor signatures, sub-module names and therefore state_dict keys as
buildingblocks.py / modules/DuSFE.py / sa_modules/* / UxLSTM vision_lstm.py, so reference checkpoints load and
`model.apply(init_weights)` finds stock nn.Conv3d / nn.Linear / nn.BatchNorm3d parameter holders.  The holders
never run: every forward() routes to the HIP stages in functional.py."""
import math

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .ops import ACT_SIGMOID


def number_of_features_per_level(init_channel_number, num_levels):
    """utils.py:494-495."""
    return [init_channel_number * 2 ** k for k in range(num_levels)]


# --------------------------------------------------------------------------------------------- conv blocks
class BasicConv(nn.Module):
    """buildingblocks.py:13-31: Conv3d(no bias) -> InstanceNorm3d -> LeakyReLU(0.01)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1, groups=1, relu=True, norm=True,
                 bias=False):
        super().__init__()
        if not (relu and norm and not bias and stride == 1 and dilation == 1 and padding == kernel_size // 2):
            raise NotImplementedError("BasicConv: only the conv->InstanceNorm->LeakyReLU form used by XLSTM_HVED is built")
        self.out_channels = out_planes
        self.groups = groups
        self.conv = nn.Conv3d(in_planes, out_planes, kernel_size, stride, padding, dilation, groups, bias)
        self.norm = nn.InstanceNorm3d(out_planes)
        self.relu = nn.LeakyReLU(negative_slope=1e-2, inplace=True)

    def forward(self, x, up2x=False):
        """up2x (not in the reference signature): also F.interpolate(scale 2, trilinear) of the result, in the launch that
        applies the norm (the VU block + upsampling pair of RA_HVED.py:599-601)."""
        return Fn.ConvInLrelu.apply(x, self.conv.weight, self.groups, up2x)


# BatchNorm `num_batches_tracked` counters: a network forward touches a dozen of them; updated one by one that is a dozen
# 4-microsecond launches.  Inside `BNCounters.collect()` (the model's forward) the increments are only recorded, and applied
# at the end as ONE add on a flat int64 buffer the counters are views of.  Outside it (a block used on its own) the counter
# is incremented directly.
class BNCounters:
    active = None

    def __init__(self):
        self.flat, self.mods, self.inc_cache = None, [], {}

    def bind(self, root):
        """Points every BatchNorm3d counter of `root` at a slot of one flat buffer (re-done when the device changed or a
        load_state_dict / .to() replaced the buffers)."""
        mods = [m for m in root.modules() if isinstance(m, nn.BatchNorm3d) and m.num_batches_tracked is not None]
        if not mods:
            self.flat, self.mods = None, []
            return
        dev = mods[0].num_batches_tracked.device
        ok = (self.flat is not None and self.flat.device == dev and len(mods) == len(self.mods) and
              all(m.num_batches_tracked.data_ptr() == self.flat.data_ptr() + 8 * i for i, m in enumerate(mods)))
        if not ok:
            flat = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in mods]).to(torch.int64).contiguous()
            for i, m in enumerate(mods):
                m._buffers["num_batches_tracked"] = flat[i]
            self.flat, self.mods, self.inc_cache = flat, mods, {}
        self.index = {id(m): i for i, m in enumerate(self.mods)}

    def collect(self, root):
        self.bind(root)
        self.pending = {}
        return self

    def __enter__(self):
        BNCounters.active = self
        return self

    def __exit__(self, *exc):
        BNCounters.active = None
        if self.flat is not None and self.pending and exc[0] is None:
            key = tuple(sorted(self.pending.items()))
            inc = self.inc_cache.get(key)
            if inc is None:                      # built once per pattern of increments (no host-to-device copy per step)
                v = [0] * len(self.mods)
                for i, s_ in key:
                    v[i] = s_
                inc = self.inc_cache[key] = torch.tensor(v, dtype=torch.int64, device=self.flat.device)
            self.flat.add_(inc)
        return False


def bn_tick(bn, steps):
    c = BNCounters.active
    if c is not None and c.flat is not None and id(bn) in c.index:
        i = c.index[id(bn)]
        c.pending[i] = c.pending.get(i, 0) + steps
    else:
        bn.num_batches_tracked += steps


class SingleConv(nn.Module):
    """buildingblocks.py:440-461 (+ create_conv :381-437) for the two layer orders XLSTM-HVED can reach
    (SURVEY F4/F10): 'ilc' and 'gcr'.  Sub-module names follow create_conv."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, order="gcr", num_groups=8, padding=1):
        super().__init__()
        self.order, self.stride = order, stride
        if order == "ilc":
            self.add_module("instancenorm", nn.InstanceNorm3d(in_channels))
            self.add_module("LeakyReLU", nn.LeakyReLU(negative_slope=1e-2, inplace=True))
            self.add_module("conv", nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding=padding, bias=True))
        elif order == "gcr":
            if in_channels < num_groups:
                num_groups = 1
            assert in_channels % num_groups == 0, (
                f"Expected number of channels in input to be divisible by num_groups. num_channels={in_channels}, "
                f"num_groups={num_groups}")
            self.num_groups = num_groups
            self.add_module("groupnorm", nn.GroupNorm(num_groups=num_groups, num_channels=in_channels))
            self.add_module("conv", nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding=padding, bias=False))
            self.add_module("ReLU", nn.ReLU(inplace=True))
        else:
            raise NotImplementedError(f"layer order '{order}' is not on the XLSTM-HVED path (supported: 'ilc', 'gcr')")
        if padding != kernel_size // 2:
            raise NotImplementedError("only 'same' padding is supported")

    def forward(self, x, x2=None, in_stats=None, out_stats=False, drop_bias=False, sole_consumer=False):
        """in_stats / out_stats ('ilc' only): take the input's channel sums from the producer's epilogue / also return
        (y, sums of y) accumulated by this conv's epilogue, so chained stages skip their statistics pass.
        drop_bias ('ilc' only): the only consumer is an InstanceNorm (Fn.in_lrelu_conv).  sole_consumer ('ilc' only): x is the
        output of another 'ilc' SingleConv that nothing else reads (Fn.InLreluConv: norm-backward fold)."""
        if self.order == "ilc":
            return Fn.in_lrelu_conv(x, x2, [self.conv.weight], [self.conv.bias], self.stride, in_stats=in_stats,
                                    out_stats=out_stats, drop_bias=drop_bias, sole_consumer=sole_consumer)
        if out_stats:
            raise NotImplementedError("out_stats is an 'ilc' feature")
        if x2 is not None:
            x = torch.cat([x, x2], 1)
        return Fn.GnConvRelu.apply(x, self.conv.weight, self.groupnorm.weight, self.groupnorm.bias, self.num_groups, self.stride)


class DoubleConv(nn.Module):
    """buildingblocks.py:464-507."""

    def __init__(self, in_channels, out_channels, encoder=False, kernel_size=3, pool_stride=1, order="gcr", num_groups=8,
                 padding=1):
        super().__init__()
        if encoder:
            c1_in, c1_out = in_channels, max(out_channels // 2, in_channels)
            c2_in, c2_out = c1_out, out_channels
        else:
            c1_in, c1_out, c2_in, c2_out = in_channels, out_channels, out_channels, out_channels
        self.add_module("SingleConv1", SingleConv(c1_in, c1_out, kernel_size, 1, order, num_groups, padding=padding))
        self.add_module("SingleConv2", SingleConv(c2_in, c2_out, kernel_size, pool_stride, order, num_groups, padding=padding))

    def forward(self, x, x2=None, out_stats=False, in_stats=None):
        if self.SingleConv1.order != "ilc":
            return self.SingleConv2(self.SingleConv1(x, x2))
        # conv1's epilogue feeds conv2's InstanceNorm -- its only consumer, so conv1's bias add is an identity (drop_bias)
        y1, st1 = self.SingleConv1(x, x2, in_stats=in_stats, out_stats=True, drop_bias=True)
        return self.SingleConv2(y1, in_stats=st1, out_stats=out_stats, sole_consumer=True)      # y1 never leaves this block


class DoubleConv_ViL(DoubleConv):
    """buildingblocks.py:509-555: DoubleConv -> LeakyReLU() -> ViLLayer(dim)."""

    def __init__(self, in_channels, out_channels, encoder=False, kernel_size=3, pool_stride=1, order="gcr", num_groups=8,
                 padding=1):
        super().__init__(in_channels, out_channels, encoder, kernel_size, pool_stride, order, num_groups, padding)
        self.add_module("leakyRelu", nn.LeakyReLU())
        self.add_module("ViL", ViLLayer(dim=out_channels if not encoder else out_channels))

    def forward(self, x, x2=None):
        from . import ops
        y = super().forward(x, x2)
        y = ops_affine_lrelu(y)
        return self.ViL(y)


def ops_affine_lrelu(y):
    return _LeakyRelu.apply(y)


class _LeakyRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from . import ops
        y = ops.affine_act(x, None, None, ops.ACT_LRELU, 0.01)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        (x,) = ctx.saved_tensors
        n, c = x.shape[:2]
        one = torch.ones((n, c), dtype=torch.float32, device=x.device)
        zero = torch.zeros((n, c), dtype=torch.float32, device=x.device)
        return ops.norm_bwd_apply(Fn._blk(dy), x, (one, zero, zero), have_g=False, sc=one, sh=zero, slope=0.01)


class Encoder(nn.Module):
    """buildingblocks.py:607-659: [MaxPool3d(2)] -> DoubleConv."""

    def __init__(self, in_channels, out_channels, num_block=1, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=2,
                 pool_type="max", basic_module=DoubleConv, conv_layer_order="gcr", num_groups=8, padding=1):
        super().__init__()
        if apply_pooling:
            if pool_type != "max" or pool_kernel_size != 2:
                raise NotImplementedError("only MaxPool3d(2) pooling is on the XLSTM-HVED path")
            self.pooling = nn.MaxPool3d(kernel_size=pool_kernel_size)
        else:
            self.pooling = None
        layers = []
        for _ in range(num_block):
            layers.append(basic_module(in_channels, out_channels, encoder=True, kernel_size=conv_kernel_size,
                                       order=conv_layer_order, num_groups=num_groups, padding=padding))
            in_channels = out_channels
        self.basic_module = nn.Sequential(*layers)

    def forward(self, x):
        if self.pooling is not None:
            first = self.basic_module[0]
            if type(first) is DoubleConv and first.SingleConv1.order == "ilc" and x.is_cuda and ops.gate_maxpool_ok(x, None):
                x, st = Fn.MaxPool2Stats.apply(x)            # the pooling leaves the first InstanceNorm's sums
                x = first(x, in_stats=st)
                for m in list(self.basic_module)[1:]:
                    x = m(x)
                return x
            x = Fn.MaxPool2.apply(x)
        return self.basic_module(x)


class ChannelPool(nn.Module):
    """buildingblocks.py:136-138 (kept for the module tree; the fused path pools inside AttenModule2)."""

    def forward(self, x):
        return Fn.ChannelPool2.apply(x, x)[:, :2]


class AttenModule2(nn.Module):
    """buildingblocks.py:259-301.  The 7^3 grouped conv and the 1x1 conv behind it have no non-linearity between
    them, so they are applied as ONE 7^3 conv with composed weights (4x fewer FLOPs; same function)."""

    def __init__(self, cat_channels, in_channels, reduction_ratio=4, pool_types=("avg", "max")):
        super().__init__()
        k, in_cha = 7, 2
        self.compress = ChannelPool()
        self.expan = 4
        self.enc_spatial = nn.Conv3d(in_cha * 2, self.expan * in_cha * 2, k, stride=1, padding=3, groups=in_cha * 2)
        self.enc_spatial2 = nn.Conv3d(self.expan * in_cha * 2, 1, 1, stride=1)
        self.seg_spatial = nn.Conv3d(in_cha, self.expan * in_cha, k, stride=1, padding=3, groups=in_cha)
        self.seg_spatial2 = nn.Conv3d(self.expan * in_cha, 1, 1, stride=1)

    def compose_params(self):
        return (self.seg_spatial.weight, self.seg_spatial.bias, self.seg_spatial2.weight, self.seg_spatial2.bias,
                self.enc_spatial.weight, self.enc_spatial.bias, self.enc_spatial2.weight, self.enc_spatial2.bias)

    def composed(self):
        """w (2,4,7,7,7): row 0 = seg gate over pooled channels 0,1 (rows 2,3 zero), row 1 = enc gate; b (2,).  Taken from the
        step's batched composition (model._precompose -> Fn.ComposeAll) when there is one."""
        pre = self.__dict__.get("_pre")
        if pre is not None:
            return pre
        return Fn.ComposeAtten.apply(2, 4, self.expan, self.seg_spatial.weight, self.seg_spatial.bias, self.seg_spatial2.weight,
                                     self.seg_spatial2.bias, self.enc_spatial.weight, self.enc_spatial.bias,
                                     self.enc_spatial2.weight, self.enc_spatial2.bias)

    def forward(self, seg_x, enc_x, recon_x=None, stats=False):
        """enc_x: the encoder feature, or a pair of aliases of it from Fn.fanout (one per consumer here: pooling, gating) when
        the caller shares the feature with other consumers (one gradient buffer instead of autograd's adds)."""
        enc_p, enc_g = enc_x if isinstance(enc_x, tuple) else Fn.fanout(enc_x, 2)
        seg_p, seg_g = Fn.fanout(seg_x, 2)
        pooled = Fn.ChannelPool2.apply(seg_p, enc_p)
        w, b = self.composed()
        # the gates' only consumer is the gating pass, whose backward also takes the gradient through the sigmoid (pre_act_grad)
        gates = Fn.conv(pooled, [w], [b], act=ACT_SIGMOID, pre_act_grad=True)        # [:,0] seg scale, [:,1] enc scale
        return Fn.GateCat.apply(seg_g, enc_g, gates, stats, True)      # stats: (output, its channel sums for the next InstanceNorm)


class Upsampling(nn.Module):
    """buildingblocks.py:737-787 (interpolation form; no parameters)."""

    def __init__(self, transposed_conv=False, in_channels=None, out_channels=None, kernel_size=3, scale_factor=(2, 2, 2),
                 mode="trilinear"):
        super().__init__()
        if transposed_conv or mode != "trilinear":
            raise NotImplementedError("only trilinear interpolation upsampling is on the XLSTM-HVED path")
        self.conv1 = None

    def forward(self, encoder_features, x, up_size=None):
        size = encoder_features.shape[2:] if encoder_features is not None else up_size
        return Fn.Upsample.apply(x, tuple(size))


class Decoder(nn.Module):
    """buildingblocks.py:662-734."""

    def __init__(self, in_channels, out_channels, conv_kernel_size=3, scale_factor=(2, 2, 2), basic_module=DoubleConv,
                 conv_layer_order="gcr", num_groups=8, mode="trilinear", padding=1, RSM=False, MVAE=False):
        super().__init__()
        if basic_module not in (DoubleConv, DoubleConv_ViL):
            raise NotImplementedError("only DoubleConv decoders are on the XLSTM-HVED path")
        self.upsampling = Upsampling(False, in_channels, out_channels, conv_kernel_size, scale_factor, mode)
        self.RSM = RSM
        if RSM:
            if not MVAE:
                raise NotImplementedError("AttenModule (non-MVAE RSM) is not on the XLSTM-HVED path")
            self.atten_module = AttenModule2(in_channels, out_channels)
        self.basic_module = basic_module(in_channels, out_channels, encoder=False, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups, padding=padding)

    def forward(self, encoder_features, x, up_size=None, recon_features=None, out_stats=False):
        kw = dict(out_stats=True) if out_stats else {}
        x = self.upsampling(encoder_features[0] if isinstance(encoder_features, tuple) else encoder_features, x, up_size)
        if self.RSM:
            if type(self.basic_module) is DoubleConv and self.basic_module.SingleConv1.order == "ilc":
                y, st = self.atten_module(x, encoder_features, stats=True)   # the gate pass leaves the first InstanceNorm's sums
                return self.basic_module(y, in_stats=st, **kw)
            return self.basic_module(self.atten_module(x, encoder_features), **kw)
        if encoder_features is not None:
            return self.basic_module(encoder_features, x, **kw)      # virtual torch.cat((enc, x), 1)
        return self.basic_module(x, **kw)


class ProductOfExperts(nn.Module):
    """buildingblocks.py:846-866.  Kept for the module tree / API; XLSTM_HVED.forward uses the fused PoE stage."""

    def forward(self, mu_list, logvar_list, mod_list, eps=1e-8):
        idx = [m + 1 for m in mod_list] + [0]
        t = 1.0 / (torch.exp(logvar_list[idx]) + eps)
        return (mu_list[idx] * t).sum(0) / t.sum(0), torch.log(1.0 / t.sum(0))


class ProductOfExperts2(nn.Module):
    """buildingblocks.py:868-886 (per-sample drop mask)."""

    def forward(self, mu, logvar, drop, eps=1e-8):
        keep = torch.ones((5, drop.shape[0]), dtype=mu.dtype, device=mu.device)
        keep[1:] = (~drop).t().to(mu.dtype)
        keep = keep.view(5, drop.shape[0], 1, 1, 1, 1)
        t = keep / (torch.exp(logvar) + eps)
        return (mu * keep * t).sum(0) / t.sum(0), torch.log(1.0 / t.sum(0))


# --------------------------------------------------------------------------------------------- DuSFE / SFECA
class DuSEAttention(nn.Module):
    """modules/DuSFE.py:89-155."""

    def __init__(self, n_channels_extract=32):
        super().__init__()
        c = n_channels_extract
        self.avg_pool_ch1 = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.avg_pool_ch2 = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc_comb = nn.Linear(c * 2, c, bias=True)
        self.fc_ch1 = nn.Linear(c, c, bias=True)
        self.fc_ch2 = nn.Linear(c, c, bias=True)
        self.conv_squeeze_ch1 = nn.Conv3d(c, 1, kernel_size=1, bias=True)
        self.conv_squeeze_ch2 = nn.Conv3d(c, 1, kernel_size=1, bias=True)
        self.conv_comb = nn.Conv3d(2, 1, kernel_size=1, bias=True)
        self.conv_adjust_ch1 = nn.Conv3d(1, 1, kernel_size=3, padding=1, bias=True)
        self.conv_adjust_ch2 = nn.Conv3d(1, 1, kernel_size=3, padding=1, bias=True)
        self.conv_fuse_ch1 = nn.Conv3d(c * 3, c, kernel_size=3, padding=1, bias=True)     # dead in the reference
        self.bn_fuse_ch1 = nn.BatchNorm3d(c)
        self.conv_fuse_ch2 = nn.Conv3d(c * 3, c, kernel_size=3, padding=1, bias=True)     # dead in the reference
        self.bn_fuse_ch2 = nn.BatchNorm3d(c)

    def compose_params(self):
        return (self.conv_comb.weight, self.conv_comb.bias, self.conv_squeeze_ch1.weight, self.conv_squeeze_ch1.bias,
                self.conv_squeeze_ch2.weight, self.conv_squeeze_ch2.bias, self.conv_adjust_ch1.weight, self.conv_adjust_ch1.bias,
                self.conv_adjust_ch2.weight, self.conv_adjust_ch2.bias)

    def forward(self, inp_ch1, inp_ch2, stats1=None, stats2=None):
        """stats1/stats2 (optional): per-(n,c) fp64 [sum, sum of squares] of the inputs from their producers' epilogues."""
        c = inp_ch1.shape[1]
        pre = self.__dict__.get("_pre")                   # the step's batched composition (model._precompose), if any
        sqw, sqb, adjw, adjb = pre if pre is not None else Fn.ComposeDuSE.apply(c, *self.compose_params())
        b1, b2 = self.bn_fuse_ch1, self.bn_fuse_ch2
        out = Fn.DuSE.apply(inp_ch1, inp_ch2, stats1, stats2, self.training, b1.running_mean, b1.running_var, b2.running_mean,
                            b2.running_var, self.fc_comb.weight, self.fc_comb.bias, self.fc_ch1.weight, self.fc_ch1.bias,
                            self.fc_ch2.weight, self.fc_ch2.bias, sqw, sqb, adjw, adjb, b1.weight, b1.bias, b2.weight, b2.bias)
        if self.training:
            bn_tick(b1, 1)
            bn_tick(b2, 1)
        return out

    def forward_pair(self, pair, stats):
        """forward() on the recon | seg pair [inp_ch1 | inp_ch2] (1, 2C, ...) with its channel sums (1, 2C, 2); returns the pair
        of outputs (Fn.DuSE2: one launch per gate / BatchNorm pass for both streams)."""
        c = pair.shape[1] // 2
        pre = self.__dict__.get("_pre")
        sqw, sqb, adjw, adjb = pre if pre is not None else Fn.ComposeDuSE.apply(c, *self.compose_params())
        b1, b2 = self.bn_fuse_ch1, self.bn_fuse_ch2
        out = Fn.DuSE2.apply(pair, stats, self.training, b1.running_mean, b1.running_var, b2.running_mean, b2.running_var,
                             self.fc_comb.weight, self.fc_comb.bias, self.fc_ch1.weight, self.fc_ch1.bias, self.fc_ch2.weight,
                             self.fc_ch2.bias, sqw, sqb, adjw, adjb, b1.weight, b1.bias, b2.weight, b2.bias)
        if self.training:
            bn_tick(b1, 1)
            bn_tick(b2, 1)
        return out


# --------------------------------------------------------------------------------------------- skip-return attention
class ConvNorm(nn.Module):
    """sa_modules/sa_module.py:10-54 (only as the parameter holder of ResBlock.identity_mapping)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, leaky=True, norm="BATCH", activation=True, deform=False):
        super().__init__()
        if norm != "BATCH" or deform:
            raise NotImplementedError
        self.act = nn.PReLU() if leaky else nn.ReLU(inplace=True)
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, (kernel_size - 1) // 2, bias=False)
        self.norm = nn.BatchNorm3d(out_channels)


class DWConvNorm(nn.Module):
    """sa_modules/sa_module.py:56-97."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, leaky=True, norm="BATCH", activation=True):
        super().__init__()
        if norm != "BATCH" or leaky:
            raise NotImplementedError
        self.act = nn.ReLU(inplace=True)
        self.dwconv = nn.Conv3d(in_channels, in_channels, kernel_size, stride, padding, bias=False, groups=in_channels)
        self.pwconv = nn.Conv3d(in_channels, out_channels, kernel_size=1)
        self.norm = nn.BatchNorm3d(out_channels)


class ResBlock(nn.Module):
    """sa_modules/sa_module.py:99-137 with lkdw=True, in==out, stride 1 (RA_HVED.py:371,382)."""

    def __init__(self, in_channels, out_channels, stride=1, leaky=False, lkdw=False, norm="BATCH", deform=False):
        super().__init__()
        if not lkdw or leaky or norm != "BATCH" or in_channels != out_channels or stride != 1:
            raise NotImplementedError("only ResBlock(c, c, lkdw=True) is on the XLSTM-HVED path")
        self.act = nn.ReLU(inplace=True)
        self.conv1 = DWConvNorm(in_channels, out_channels, 3, stride, 1, leaky, norm, True)
        self.conv2 = DWConvNorm(out_channels, out_channels, 3, 1, 1, leaky, norm, True)
        self.identity_mapping = ConvNorm(in_channels, out_channels, 1, stride, leaky, norm, False)


class SpacialAttention3D(nn.Module):
    """sa_modules/attention_blocks.py:112-126."""

    def __init__(self, kernel_size=7):
        super().__init__()
        if kernel_size != 1:
            raise NotImplementedError("only SpacialAttention3D(kernel_size=1) is on the XLSTM-HVED path")
        self.conv = nn.Conv3d(2, 1, kernel_size, 1, kernel_size // 2, bias=False)
        self.sigmoid = nn.Sigmoid()


class SkipReturnAttention(nn.Sequential):
    """nn.Sequential(ResBlock(c, c, lkdw=True), SpacialAttention3D(kernel_size=1)) as built at RA_HVED.py:371-384;
    forward returns the (N,1,D,H,W) attention map.  `steps` = how many times the reference would have evaluated it
    this forward (4: once per modality stream, RA_HVED.py:548-552)."""

    def __init__(self, channels):
        super().__init__(ResBlock(channels, channels, lkdw=True), SpacialAttention3D(kernel_size=1))

    def compose_params(self):
        """(dwconv.weight, pwconv.weight) of the two DWConvNorm blocks: what Fn.ComposeAll turns into two dense 3^3 weights."""
        c1, c2 = self[0].conv1, self[0].conv2
        return (c1.dwconv.weight, c1.pwconv.weight, c2.dwconv.weight, c2.pwconv.weight)

    def forward(self, x, steps=1):
        rb, sa = self[0], self[1]
        c1, c2 = rb.conv1, rb.conv2
        pre = self.__dict__.get("_pre")                   # (wc1, wc2) from the step's batched composition (model._precompose)
        wcs = pre if (pre is not None and x.dtype != torch.float32) else (None, None)
        a = Fn.SkipReturnAttention.apply(
            x, self.training, steps, c1.norm.running_mean, c1.norm.running_var, c2.norm.running_mean, c2.norm.running_var,
            c1.dwconv.weight, c1.pwconv.weight, c1.pwconv.bias, c1.norm.weight, c1.norm.bias,
            c2.dwconv.weight, c2.pwconv.weight, c2.pwconv.bias, c2.norm.weight, c2.norm.bias, sa.conv.weight, *wcs)
        if self.training:
            bn_tick(c1.norm, steps)
            bn_tick(c2.norm, steps)
        return a


# --------------------------------------------------------------------------------------------- ViL
class LinearHeadwiseExpand(nn.Module):
    """vision_lstm.py:133-175 (parameter holder; block-diagonal projection)."""

    def __init__(self, dim, num_heads, bias=False):
        super().__init__()
        assert dim % num_heads == 0 and not bias
        self.dim, self.num_heads = dim, num_heads
        d = dim // num_heads
        self.weight = nn.Parameter(torch.empty(num_heads, d, d))
        self.bias = None
        nn.init.normal_(self.weight.data, mean=0.0, std=math.sqrt(2 / 5 / d))


class CausalConv1d(nn.Module):
    """vision_lstm.py:178-221 (parameter holder)."""

    def __init__(self, dim, kernel_size=4, bias=True):
        super().__init__()
        self.dim, self.kernel_size, self.pad = dim, kernel_size, kernel_size - 1
        self.conv = nn.Conv1d(dim, dim, kernel_size, padding=self.pad, groups=dim, bias=bias)


class LayerNorm(nn.Module):
    """vision_lstm.py:224-268 (weight stored as a residual around 1, no bias)."""

    def __init__(self, ndim=-1, weight=True, bias=False, eps=1e-5, residual_weight=True):
        super().__init__()
        assert weight and not bias and residual_weight
        self.weight = nn.Parameter(torch.zeros(ndim))
        self.bias = None
        self.eps, self.ndim = eps, ndim


class MultiHeadLayerNorm(LayerNorm):
    """vision_lstm.py:271-287."""


class MatrixLSTMCell(nn.Module):
    """vision_lstm.py:290-348 (parameter holder)."""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.igate = nn.Linear(3 * dim, num_heads)
        self.fgate = nn.Linear(3 * dim, num_heads)
        self.outnorm = MultiHeadLayerNorm(ndim=dim, weight=True, bias=False)
        with torch.no_grad():      # vision_lstm.py:341-348
            self.fgate.weight.zero_()
            self.fgate.bias.copy_(torch.linspace(3.0, 6.0, num_heads))
            self.igate.weight.zero_()
            nn.init.normal_(self.igate.bias, mean=0.0, std=0.1)


class _InnerViLLayer(nn.Module):
    """vision_lstm.py:351-477 (parameter holder for the fused ViL stage)."""

    def __init__(self, dim, expansion=2, qkv_block_size=4, kernel_size=4):
        super().__init__()
        if dim % qkv_block_size != 0:
            raise NotImplementedError("model dim must be a multiple of 4")
        self.dim = dim
        inner = expansion * dim
        self.proj_up = nn.Linear(dim, 2 * inner, bias=False)
        self.q_proj = LinearHeadwiseExpand(inner, inner // qkv_block_size)
        self.k_proj = LinearHeadwiseExpand(inner, inner // qkv_block_size)
        self.v_proj = LinearHeadwiseExpand(inner, inner // qkv_block_size)
        self.conv1d = CausalConv1d(inner, kernel_size, bias=True)
        self.mlstm_cell = MatrixLSTMCell(inner, qkv_block_size)
        self.learnable_skip = nn.Parameter(torch.ones(inner))
        self.proj_down = nn.Linear(inner, dim, bias=False)
        with torch.no_grad():      # vision_lstm.py:455-477
            std = math.sqrt(2 / (5 * dim))
            nn.init.normal_(self.proj_up.weight, 0.0, std)
            nn.init.normal_(self.proj_down.weight, 0.0, 2 / math.sqrt(dim))
            for p in (self.q_proj, self.k_proj, self.v_proj):
                nn.init.normal_(p.weight, 0.0, std)


class ViLBlock(nn.Module):
    """vision_lstm.py:480-506 (DropPath with p=0 is the plain residual)."""

    def __init__(self, dim, direction=None, drop_path=0.0, norm_bias=False):
        super().__init__()
        if drop_path != 0.0 or norm_bias:
            raise NotImplementedError
        self.dim = dim
        self.drop_path = nn.Identity()
        self.norm = LayerNorm(ndim=dim, weight=True, bias=False)
        self.layer = _InnerViLLayer(dim)


class ViLLayer(nn.Module):
    """UxLSTMEnc_3d.py:42-87: NCDHW -> tokens -> ViLBlock -> NCDHW, fp32 arithmetic.  `self.norm` exists in the
    reference but is never applied."""

    def __init__(self, dim, d_state=16, d_conv=4, expand=2, channel_token=False):
        super().__init__()
        if channel_token:
            raise NotImplementedError
        self.dim = dim
        self.norm = nn.LayerNorm(dim)
        self.vil = ViLBlock(dim=dim)
        self.channel_token = channel_token

    def stage_params(self):
        L = self.vil.layer
        cell = L.mlstm_cell
        return (self.vil.norm.weight, L.proj_up.weight, L.conv1d.conv.weight, L.conv1d.conv.bias, L.q_proj.weight,
                L.k_proj.weight, L.v_proj.weight, cell.igate.weight, cell.igate.bias, cell.fgate.weight, cell.fgate.bias,
                cell.outnorm.weight, L.learnable_skip, L.proj_down.weight)

    def forward(self, x, skip=None, residual_input=False):
        """out = ViLBlock(x + skip) (+ x when residual_input: the fused form of RA_HVED.py:626)."""
        return Fn.ViL.apply(x, skip, residual_input, *self.stage_params())

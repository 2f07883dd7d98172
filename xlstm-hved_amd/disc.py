"""Discriminator of the adversarial training step (RA_HVED.py:204-236, buildingblocks.py:342-358) on the HIP path.

    disc.0: Conv3d(7, 64, ks, s1, p1) -> LeakyReLU(0.2)                      (normalization=False)
    disc.k: Conv3d(c, 2c, ks, s2, p1) -> InstanceNorm3d -> LeakyReLU(0.2)    k = 1..3  (64 -> 128 -> 256 -> 512)
    last:   Conv3d(512, 1, ks, s1, p1, bias=False)

ks = 4 is what the reference trains with (train.py:146, Pretrain.py:150: a 128^3 patch shrinks 127 -> 63 -> 31 -> 15 -> 14,
64 taps per convolution), ks = 3 the class default; both run on the same kernels.

Same constructor, attribute names and state_dict keys as the reference class (`disc.{k}.0.weight/bias`, `last.weight`);
parameters live in stock nn.Conv3d holders, all compute is csrc/dconv.hip: channels-last 16-bit activations inside,
implicit-GEMM convolutions on the matrix cores (forward, data gradient, weight gradient), InstanceNorm statistics in the
conv epilogue.  Activations are 16-bit (bf16 / fp16) like in the reference's `with autocast():` step (train.py:218,260); an
fp32 input is taken in fp16 -- the reference's autocast dtype -- with the backward pass scaled on the device so that the
incoming gradient peaks at 2^10 (what GradScaler does for the reference, train.py:207,265: fp16 gradients keep their
range whatever the caller's loss magnitude), and the input gradient comes back in fp32.
The biases of disc.1..3 sit in front of an InstanceNorm: their add is an identity and is skipped, their gradient is the
exact zero the reference approximates with round-off."""
import ctypes as C

import torch
from torch import nn
from torch.autograd import Function

from . import _lib as L
from . import ops
from .blocks import number_of_features_per_level

SLOPE = 0.2
GRAD_PEAK = 1024.0                 # fp32 caller: the backward pass runs with the incoming gradient scaled to this peak


def _s():
    return torch.cuda.current_stream().cuda_stream


def _pack(w, mode, cout_pad, cin_pad, dtype):
    cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    k3 = ks ** 3
    n = ks * ks * cout * 32 if mode == 2 else (k3 * cout * cin_pad if mode == 0 else k3 * cin_pad * cout_pad)
    out = torch.empty(n, dtype=dtype, device=w.device)
    L.check(L.load().xh_dconv_pack(_s(), ops._dt(out), mode, ks, w.data_ptr(), out.data_ptr(), cout, cin, cout_pad, cin_pad), "xh_dconv_pack")
    return out


# The packed 16-bit weight images depend on the weights alone.  A training step runs the discriminator several times on the
# same weights (train.py:260,272,276), so inside a `pack_scope()` (TrainStep.compute opens one) an image is built once per
# (weight tensor, version, layout) and reused by the later passes; outside a scope every call packs afresh, which is what a
# caller who updates the weights between calls -- or replays a captured forward -- needs.
_SCOPE = {"depth": 0, "cache": {}}


class pack_scope:
    def __enter__(self):
        _SCOPE["depth"] += 1
        return self

    def __exit__(self, *exc):
        _SCOPE["depth"] -= 1
        if _SCOPE["depth"] == 0:
            _SCOPE["cache"].clear()
        return False


def _pack_cached(w, mode, cout_pad, cin_pad, dtype):
    if _SCOPE["depth"] == 0:
        return _pack(w, mode, cout_pad, cin_pad, dtype)
    key = (w.data_ptr(), w._version, mode, cout_pad, cin_pad, dtype)
    hit = _SCOPE["cache"].get(key)
    if hit is None:
        hit = _SCOPE["cache"][key] = _pack(w, mode, cout_pad, cin_pad, dtype)
    return hit


def _conv(x, wp, bias, mode, stride, n, sp_in, sp_out, cs, cn, red=None, act=L.ACT_NONE, ks=3, mask=None):
    y = torch.empty((n,) + tuple(sp_out) + (cn,), dtype=x.dtype, device=x.device)
    L.check(L.load().xh_dconv_cl(_s(), ops._dt(x), mode, stride, ks, x.data_ptr(), wp.data_ptr(), ops._p(bias), y.data_ptr(), ops._p(red), n,
                                 *sp_in, *sp_out, cs, cn, act, SLOPE, ops._p(mask)), "xh_dconv_cl")
    return y


def _wgrad(x, dy, stride, n, sp_in, sp_out, cs, cn, ks=3, gs=None):
    dwp = torch.zeros(ks ** 3 * cn * cs, dtype=torch.float32, device=x.device)
    L.check(L.load().xh_dconv_wgrad_cl(_s(), ops._dt(x), stride, ks, x.data_ptr(), dy.data_ptr(), dwp.data_ptr(), n, *sp_in, *sp_out, cs, cn),
            "xh_dconv_wgrad_cl")
    return dwp.div_(gs) if gs is not None else dwp


def _unpack(dwp, target, cout_pad, cin_pad):
    cout, cin, ks = target.shape[0], target.shape[1], target.shape[2]
    L.check(L.load().xh_dconv_unpack_grad(_s(), ks, dwp.data_ptr(), target.data_ptr(), cout, cin, cout_pad, cin_pad), "xh_dconv_unpack_grad")


def conv_out(sp, ks, stride):
    """Extents of a padding-1 convolution (nn.Conv3d(..., ks, stride, padding=1), buildingblocks.py:350,354)."""
    return tuple((s + 2 - ks) // stride + 1 for s in sp)


STRIDES = (1, 2, 2, 2, 1)          # disc.0 .. disc.3, last (RA_HVED.py:206,223)


class DiscFn(Function):
    """The whole discriminator as one autograd node: x (N, Cin<=8, D, H, W) 16-bit NCDHW -> (N, 1, D', H', W');
    ks = 4: 128 -> 127 -> 63 -> 31 -> 15 -> 14 per axis, ks = 3: 128 -> 128 -> 64 -> 32 -> 16 -> 16."""

    @staticmethod
    def forward(ctx, x, *params):
        w0, b0, w1, b1, w2, b2, w3, b3, wl = params
        lib = L.load()
        n, cin, d, h, w = x.shape
        ks = w0.shape[2]
        if cin > 8 or x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
            raise TypeError("the HIP discriminator takes <= 8 channels of bf16 / fp16 / fp32 input (train.py:218 runs it under autocast)")
        in_dt = x.dtype
        if in_dt == torch.float32:
            x = x.to(torch.float16)
        dt, dev = x.dtype, x.device
        x = x.contiguous()
        V = d * h * w
        sp = [(d, h, w)]                                      # sp[k] = input extents of layer k, sp[k + 1] its output
        for st in STRIDES:
            sp.append(conv_out(sp[-1], ks, st))
        if min(sp[-1]) < 1:
            raise ValueError(f"input {d}x{h}x{w} is too small for the discriminator (ks={ks})")
        xin = torch.empty((n, d, h, w, 8), dtype=dt, device=dev)
        L.check(lib.xh_cl_from_ncdhw(_s(), ops._dt(x), x.data_ptr(), cin * V, cin, None, 0, 0, xin.data_ptr(), 8, n, V), "xh_cl_from_ncdhw")
        y0 = _conv(xin, _pack_cached(w0, 2, 64, 8, dt), b0, 0, 1, n, sp[0], sp[1], 8, w0.shape[0], act=L.ACT_LRELU, ks=ks)
        acts, raws, stats = [y0], [], []
        for k, wk in enumerate((w1, w2, w3), 1):
            cs, cn = wk.shape[1], wk.shape[0]
            red = torch.zeros((n, cn, 2), dtype=torch.float64, device=dev)
            c = _conv(acts[-1], _pack_cached(wk, 0, cn, cs, dt), None, 0, 2, n, sp[k], sp[k + 1], cs, cn, red=red, ks=ks)
            cnt = sp[k + 1][0] * sp[k + 1][1] * sp[k + 1][2]
            sc, sh, mean, rstd = ops.norm_finalize(ops.MODE_IN, red, n, cn, cnt)
            a = torch.empty_like(c)
            L.check(lib.xh_cl_affine_act(_s(), ops._dt(c), c.data_ptr(), a.data_ptr(), sc.data_ptr(), sh.data_ptr(), SLOPE, n, cn, cnt),
                    "xh_cl_affine_act")
            raws.append(c)
            acts.append(a)
            stats.append((sc, sh, mean, rstd))
        out = _conv(acts[-1], _pack_cached(wl, 0, 1, wl.shape[1], dt), None, 0, 1, n, sp[4], sp[5], wl.shape[1], 1, ks=ks)
        ctx.save_for_backward(xin, *acts, *raws, *[t for st in stats for t in st], w0, w1, w2, w3, wl)
        ctx.meta = (n, cin, sp, dt, ks, in_dt)
        ctx.params = params
        out = out.view(n, 1, *sp[5])                  # one channel: channels-last == NCDHW
        return out.float() if in_dt == torch.float32 else out

    @staticmethod
    def backward(ctx, dout):
        from .functional import _targets
        lib = L.load()
        sv = ctx.saved_tensors
        xin, acts, raws = sv[0], sv[1:5], sv[5:8]
        stats = [sv[8 + 4 * i:12 + 4 * i] for i in range(3)]
        w0, w1, w2, w3, wl = sv[20:25]
        n, cin, sp, dt, ks, in_dt = ctx.meta
        dev = dout.device
        # fp32 caller: no outer loss scale protects the fp16 gradients -> a device-side scale (no host synchronisation)
        gs = (GRAD_PEAK / dout.abs().amax().clamp_min(1e-30).float()) if in_dt == torch.float32 else None
        # parameter gradients are skipped when no parameter asks for one (TrainStep freezes the discriminator for the
        # generator's pass: train.py:265 computes them there too, and optimizer_d.zero_grad() at :282 throws them away)
        need_w = any(ctx.needs_input_grad[1:])
        if need_w:
            bufs, rets = _targets(ctx.params)
        else:
            bufs, rets = [None] * 9, [None] * 9
        g_w0, g_b0, g_w1, g_b1, g_w2, g_b2, g_w3, g_b3, g_wl = bufs
        # last conv (512 -> 1): its single gradient channel is padded to 32 so that it is a K step of the GEMMs
        dy = torch.zeros((n,) + sp[5] + (32,), dtype=dt, device=dev)
        dy[..., 0] = (dout.reshape((n,) + sp[5]) * gs if gs is not None else dout.reshape((n,) + sp[5])).to(dt)
        c3 = wl.shape[1]
        if need_w:
            _unpack(_wgrad(acts[3], dy, 1, n, sp[4], sp[5], c3, 32, ks=ks, gs=gs), g_wl, 32, c3)
        da = _conv(dy, _pack_cached(wl, 1, 32, c3, dt), None, 1, 1, n, sp[5], sp[4], 32, c3, ks=ks)
        for k, wk, g_w in ((3, w3, g_w3), (2, w2, g_w2), (1, w1, g_w1)):
            cs, cn = wk.shape[1], wk.shape[0]
            sc, sh, mean, rstd = stats[k - 1]
            c = raws[k - 1]
            cnt = sp[k + 1][0] * sp[k + 1][1] * sp[k + 1][2]
            red = torch.zeros((n, cn, 2), dtype=torch.float64, device=dev)
            args = (ops._dt(c), da.data_ptr(), c.data_ptr())
            L.check(lib.xh_cl_act_bwd(_s(), args[0], 0, args[1], args[2], None, sc.data_ptr(), sh.data_ptr(), SLOPE, None, None, None,
                                      red.data_ptr(), n, cn, cnt), "xh_cl_act_bwd")
            A, B, Cc = ops.norm_bwd_coef(ops.MODE_IN, red, cnt, mean, rstd)
            dc = torch.empty_like(c)
            L.check(lib.xh_cl_act_bwd(_s(), args[0], 1, args[1], args[2], dc.data_ptr(), sc.data_ptr(), sh.data_ptr(), SLOPE, A.data_ptr(),
                                      B.data_ptr(), Cc.data_ptr(), None, n, cn, cnt), "xh_cl_act_bwd")
            if need_w:
                _unpack(_wgrad(acts[k - 1], dc, 2, n, sp[k], sp[k + 1], cs, cn, ks=ks, gs=gs), g_w, cn, cs)
            if k > 1:
                da = _conv(dc, _pack_cached(wk, 1, cn, cs, dt), None, 1, 2, n, sp[k + 1], sp[k], cn, cs, ks=ks)
        # block 0: conv + bias -> LeakyReLU (no norm): g0 = da * leaky'(y0), bias gradient = sum g0.  Both ride in the epilogue of
        # disc.1's data gradient (mask = the stored activation y0): no separate pass over the 64-channel full-resolution tensor
        c0 = w0.shape[0]
        V0 = sp[0][0] * sp[0][1] * sp[0][2]
        red0 = torch.zeros((n, c0, 2), dtype=torch.float64, device=dev)
        g0 = _conv(dc, _pack_cached(w1, 1, w1.shape[0], c0, dt), None, 1, 2, n, sp[2], sp[1], w1.shape[0], c0, ks=ks, red=red0, mask=acts[0])
        if need_w:
            g_b0 += (red0[:, :, 0].sum(0) / gs if gs is not None else red0[:, :, 0].sum(0)).float()
            _unpack(_wgrad(xin, g0, 1, n, sp[0], sp[1], 8, c0, ks=ks, gs=gs), g_w0, c0, 8)
        dx = None
        if ctx.needs_input_grad[0]:
            dxin = _conv(g0, _pack_cached(w0, 1, c0, 8, dt), None, 1, 1, n, sp[1], sp[0], c0, 8, ks=ks)
            dx = torch.empty((n, cin) + sp[0], dtype=dt, device=dev)
            L.check(lib.xh_cl_to_ncdhw(_s(), ops._dt(dx), dxin.data_ptr(), 8, dx.data_ptr(), cin * V0, cin, None, 0, 0, n, V0), "xh_cl_to_ncdhw")
            if in_dt == torch.float32:
                dx = dx.float() / gs
        return (dx, *rets)


class Discriminator(nn.Module):
    """RA_HVED.py:204-236 on the HIP path (module docstring).  `Discriminator(in_channels=7, ks=4, strides=[1,2,2,2])` is the
    call of train.py:146 / Pretrain.py:150; ks = 3 is the class default."""

    def __init__(self, in_channels=3, f_maps=64, ks=3, num_levels=4, strides=(1, 2, 2, 2)):
        super().__init__()
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels)
        if ks not in (3, 4) or tuple(strides) != (1, 2, 2, 2) or list(f_maps) != [64, 128, 256, 512] or in_channels > 8:
            raise NotImplementedError("the HIP discriminator is built for the reference's configurations: ks 4 (train.py:146) or 3 (the "
                                      "class default), strides (1,2,2,2), f_maps 64..512, <= 8 input channels")
        blocks = []
        for i, (out_f, st) in enumerate(zip(f_maps, strides)):
            layers = [nn.Conv3d(in_channels, out_f, ks, stride=st, padding=1)]
            if i > 0:
                layers.append(nn.InstanceNorm3d(out_f))
            layers.append(nn.LeakyReLU(0.2, inplace=True))
            blocks.append(nn.Sequential(*layers))
            in_channels = out_f
        self.disc = nn.ModuleList(blocks)
        self.last = nn.Conv3d(512, 1, ks, padding=1, bias=False)

    def forward(self, x, input_level=0):
        if input_level != 0:
            raise NotImplementedError("input_level > 0 is not used by train.py")
        ps = []
        for blk in self.disc:
            ps += [blk[0].weight, blk[0].bias]
        return DiscFn.apply(x, *ps, self.last.weight)

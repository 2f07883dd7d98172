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


def _wgrad(x, dy, stride, n, sp_in, sp_out, cs, cn, ks=3, gs=None, dwp=None):
    """Packed fp32 weight gradient; `dwp`: a ZEROED destination of ks^3 * cn * cs floats (default: a fresh one)."""
    if dwp is None:
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


class DiscShare:
    """Activations of one discriminator pass kept so that a second pass can join it in ONE batched backward.

    train.py:260 runs D on the fake sample for the generator's loss and train.py:272 runs D on the SAME values again
    (`fake.detach()`, the discriminator's weights have not changed in between) for the discriminator's loss: the second forward
    recomputes the first bit for bit.  With a DiscShare the first pass (Discriminator.forward(x, share=s)) writes its activations
    into the first half of buffers sized for two batches; Discriminator.forward_pair(real, s) then runs the forward of `real`
    only, into the second half, and returns the outputs of both -- [D(fake); D(real)] -- as one autograd node whose backward is
    the batched weight-gradient pass over both halves."""

    def __init__(self):
        self.bufs = None
        self.meta = None
        self.weights = None


def _extents(x_shape, ks):
    n, cin, d, h, w = x_shape
    sp = [(d, h, w)]                                          # sp[k] = input extents of layer k, sp[k + 1] its output
    for st in STRIDES:
        sp.append(conv_out(sp[-1], ks, st))
    if min(sp[-1]) < 1:
        raise ValueError(f"input {d}x{h}x{w} is too small for the discriminator (ks={ks})")
    return sp


def _alloc(nb, sp, chans, dt, dev):
    """Channels-last activation buffers for a batch of nb: xin, acts[0..3], raws[0..2], per-layer statistics, out."""
    b = {"xin": torch.empty((nb,) + sp[0] + (8,), dtype=dt, device=dev),
         "acts": [torch.empty((nb,) + sp[k + 1] + (chans[k],), dtype=dt, device=dev) for k in range(4)],
         "raws": [torch.empty((nb,) + sp[k + 1] + (chans[k],), dtype=dt, device=dev) for k in range(1, 4)],
         "red": [torch.zeros((nb, chans[k], 2), dtype=torch.float64, device=dev) for k in range(1, 4)],
         "stats": [[torch.empty((nb, chans[k]), dtype=torch.float32, device=dev) for _ in range(4)] for k in range(1, 4)],
         "out": torch.empty((nb,) + sp[5] + (1,), dtype=dt, device=dev)}
    return b


HEAD = [True]                      # A/B switch: the head (C -> 1) through the reduction kernels xh_dlast_* instead of 16-column GEMM tiles


def set_head_kernels(enabled):
    HEAD[0] = bool(enabled)


def _head_ok(wl):
    return HEAD[0] and wl.shape[0] == 1 and wl.shape[1] % 512 == 0 and wl.is_contiguous()


def _conv_into(y, x, wp, bias, mode, stride, n, sp_in, sp_out, cs, cn, red=None, act=L.ACT_NONE, ks=3, mask=None):
    L.check(L.load().xh_dconv_cl(_s(), ops._dt(x), mode, stride, ks, x.data_ptr(), wp.data_ptr(), ops._p(bias), y.data_ptr(), ops._p(red), n,
                                 *sp_in, *sp_out, cs, cn, act, SLOPE, ops._p(mask)), "xh_dconv_cl")
    return y


def _forward(x, params, bufs, lo, sp, ks):
    """Forward of samples x (n, cin, D, H, W; 16-bit NCDHW) into rows [lo, lo + n) of the activation buffers."""
    w0, b0, w1, b1, w2, b2, w3, b3, wl = params
    lib = L.load()
    n, cin = x.shape[:2]
    dt = x.dtype
    V = sp[0][0] * sp[0][1] * sp[0][2]
    sl = slice(lo, lo + n)
    xin = bufs["xin"][sl]
    L.check(lib.xh_cl_from_ncdhw(_s(), ops._dt(x), x.data_ptr(), cin * V, cin, None, 0, 0, xin.data_ptr(), 8, n, V), "xh_cl_from_ncdhw")
    _conv_into(bufs["acts"][0][sl], xin, _pack_cached(w0, 2, 64, 8, dt), b0, 0, 1, n, sp[0], sp[1], 8, w0.shape[0], act=L.ACT_LRELU, ks=ks)
    for k, wk in enumerate((w1, w2, w3), 1):
        cs, cn = wk.shape[1], wk.shape[0]
        red = bufs["red"][k - 1][sl]
        c = _conv_into(bufs["raws"][k - 1][sl], bufs["acts"][k - 1][sl], _pack_cached(wk, 0, cn, cs, dt), None, 0, 2, n, sp[k], sp[k + 1],
                       cs, cn, red=red, ks=ks)
        cnt = sp[k + 1][0] * sp[k + 1][1] * sp[k + 1][2]
        sc, sh, mean, rstd = (t[sl] for t in bufs["stats"][k - 1])
        L.check(lib.xh_norm_finalize(_s(), ops.MODE_IN, red.data_ptr(), n, cn, cnt, 1, ops.NORM_EPS, None, None, None, None, 1,
                                     sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr()), "xh_norm_finalize")
        a = bufs["acts"][k][sl]
        L.check(lib.xh_cl_affine_act(_s(), ops._dt(c), c.data_ptr(), a.data_ptr(), sc.data_ptr(), sh.data_ptr(), SLOPE, n, cn, cnt),
                "xh_cl_affine_act")
    if _head_ok(wl):
        # the head (one output channel): a dot product per voxel, not a GEMM (xh_dlast_fwd)
        L.check(lib.xh_dlast_fwd(_s(), ops._dt(x), ks, bufs["acts"][3][sl].data_ptr(), _pack_cached(wl, 0, 1, wl.shape[1], dt).data_ptr(),
                                 bufs["out"][sl].data_ptr(), n, *sp[4], *sp[5], wl.shape[1]), "xh_dlast_fwd")
    else:
        _conv_into(bufs["out"][sl], bufs["acts"][3][sl], _pack_cached(wl, 0, 1, wl.shape[1], dt), None, 0, 1, n, sp[4], sp[5], wl.shape[1], 1, ks=ks)
    return bufs["out"][sl]


def _backward(bufs, lo, n, sp, ks, dt, cin, weights, params, dout, need_w, need_dx, gs):
    """Backward over rows [lo, lo + n) of the activation buffers.  Returns (dx or None, parameter-gradient returns)."""
    from .functional import _targets
    lib = L.load()
    w0, w1, w2, w3, wl = weights
    dev = dout.device
    sl = slice(lo, lo + n)
    xin, acts, raws = bufs["xin"][sl], [a[sl] for a in bufs["acts"]], [r[sl] for r in bufs["raws"]]
    stats = [[t[sl] for t in st] for st in bufs["stats"]]
    if need_w:
        gbufs, rets = _targets(params)
    else:
        gbufs, rets = [None] * 9, [None] * 9
    g_w0, g_b0, g_w1, g_b1, g_w2, g_b2, g_w3, g_b3, g_wl = gbufs
    c3 = wl.shape[1]
    # every zeroed scratch of the pass in TWO fills: the fp64 sums of the three norm layers + block 0's bias sums, and (with parameter
    # gradients) the packed fp32 weight gradients of the four convolutions -- instead of one small fill in front of each kernel
    ws_ = (w1, w2, w3)
    red_n = [n * wk.shape[0] * 2 for wk in ws_] + [n * w0.shape[0] * 2]
    red_all = torch.zeros(sum(red_n), dtype=torch.float64, device=dev)
    red_of, o = [], 0
    for r in red_n:
        red_of.append(red_all[o:o + r]); o += r
    dwp_of = [None] * 4
    if need_w:
        dw_n = [ks ** 3 * wk.shape[0] * (8 if wk is w0 else wk.shape[1]) for wk in (w0, w1, w2, w3)]
        dw_all = torch.zeros(sum(dw_n), dtype=torch.float32, device=dev)
        o = 0
        for i, r in enumerate(dw_n):
            dwp_of[i] = dw_all[o:o + r]; o += r
    if _head_ok(wl):
        # the head's backward as two reduction kernels (xh_dlast_wgrad adds straight into the parameter gradient; xh_dlast_dgrad)
        dy1 = (dout.reshape((n,) + sp[5]) * gs if gs is not None else dout.reshape((n,) + sp[5])).to(dt).contiguous()
        if need_w:
            L.check(lib.xh_dlast_wgrad(_s(), ops._dt(dy1), ks, acts[3].data_ptr(), dy1.data_ptr(), g_wl.data_ptr(),
                                       1.0 / gs if gs is not None else 1.0, n, *sp[4], *sp[5], c3), "xh_dlast_wgrad")
        da = torch.empty((n,) + tuple(sp[4]) + (c3,), dtype=dt, device=dev)
        L.check(lib.xh_dlast_dgrad(_s(), ops._dt(dy1), ks, dy1.data_ptr(), _pack_cached(wl, 0, 1, c3, dt).data_ptr(), da.data_ptr(), n,
                                   *sp[4], *sp[5], c3), "xh_dlast_dgrad")
    else:
        # (other widths: the single gradient channel padded to 32 so that it is a K step of the GEMMs)
        dy = torch.zeros((n,) + sp[5] + (32,), dtype=dt, device=dev)
        dy[..., 0] = (dout.reshape((n,) + sp[5]) * gs if gs is not None else dout.reshape((n,) + sp[5])).to(dt)
        if need_w:
            _unpack(_wgrad(acts[3], dy, 1, n, sp[4], sp[5], c3, 32, ks=ks, gs=gs), g_wl, 32, c3)
        da = _conv(dy, _pack_cached(wl, 1, 32, c3, dt), None, 1, 1, n, sp[5], sp[4], 32, c3, ks=ks)
    dc = None
    for k, wk, g_w in ((3, w3, g_w3), (2, w2, g_w2), (1, w1, g_w1)):
        cs, cn = wk.shape[1], wk.shape[0]
        sc, sh, mean, rstd = stats[k - 1]
        c = raws[k - 1]
        cnt = sp[k + 1][0] * sp[k + 1][1] * sp[k + 1][2]
        red = red_of[k - 1].view(n, cn, 2)
        args = (ops._dt(c), da.data_ptr(), c.data_ptr())
        L.check(lib.xh_cl_act_bwd(_s(), args[0], 0, args[1], args[2], None, sc.data_ptr(), sh.data_ptr(), SLOPE, None, None, None,
                                  red.data_ptr(), n, cn, cnt), "xh_cl_act_bwd")
        A, B, Cc = ops.norm_bwd_coef(ops.MODE_IN, red, cnt, mean, rstd)
        dc = torch.empty_like(c)
        L.check(lib.xh_cl_act_bwd(_s(), args[0], 1, args[1], args[2], dc.data_ptr(), sc.data_ptr(), sh.data_ptr(), SLOPE, A.data_ptr(),
                                  B.data_ptr(), Cc.data_ptr(), None, n, cn, cnt), "xh_cl_act_bwd")
        if need_w:
            _unpack(_wgrad(acts[k - 1], dc, 2, n, sp[k], sp[k + 1], cs, cn, ks=ks, gs=gs, dwp=dwp_of[k]), g_w, cn, cs)
        if k > 1:
            da = _conv(dc, _pack_cached(wk, 1, cn, cs, dt), None, 1, 2, n, sp[k + 1], sp[k], cn, cs, ks=ks)
    if not need_w and not need_dx:
        return None, rets
    # block 0: conv + bias -> LeakyReLU (no norm): g0 = da * leaky'(y0), bias gradient = sum g0.  Both ride in the epilogue of
    # disc.1's data gradient (mask = the stored activation y0): no separate pass over the 64-channel full-resolution tensor
    c0 = w0.shape[0]
    V0 = sp[0][0] * sp[0][1] * sp[0][2]
    red0 = red_of[3].view(n, c0, 2)
    g0 = _conv(dc, _pack_cached(w1, 1, w1.shape[0], c0, dt), None, 1, 2, n, sp[2], sp[1], w1.shape[0], c0, ks=ks, red=red0, mask=acts[0])
    if need_w:
        g_b0 += (red0[:, :, 0].sum(0) / gs if gs is not None else red0[:, :, 0].sum(0)).float()
        _unpack(_wgrad(xin, g0, 1, n, sp[0], sp[1], 8, c0, ks=ks, gs=gs, dwp=dwp_of[0]), g_w0, c0, 8)
    dx = None
    if need_dx:
        dxin = _conv(g0, _pack_cached(w0, 1, c0, 8, dt), None, 1, 1, n, sp[1], sp[0], c0, 8, ks=ks)
        dx = torch.empty((n, cin) + sp[0], dtype=dt, device=dev)
        L.check(lib.xh_cl_to_ncdhw(_s(), ops._dt(dx), dxin.data_ptr(), 8, dx.data_ptr(), cin * V0, cin, None, 0, 0, n, V0), "xh_cl_to_ncdhw")
    return dx, rets


def _check_input(x, ks):
    if x.shape[1] > 8 or x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise TypeError("the HIP discriminator takes <= 8 channels of bf16 / fp16 / fp32 input (train.py:218 runs it under autocast)")


class DiscFn(Function):
    """The whole discriminator as one autograd node: x (N, Cin<=8, D, H, W) NCDHW -> (N, 1, D', H', W');
    ks = 4: 128 -> 127 -> 63 -> 31 -> 15 -> 14 per axis, ks = 3: 128 -> 128 -> 64 -> 32 -> 16 -> 16."""

    @staticmethod
    def forward(ctx, x, share, *params):
        w0, b0, w1, b1, w2, b2, w3, b3, wl = params
        n, cin = x.shape[:2]
        ks = w0.shape[2]
        _check_input(x, ks)
        in_dt = x.dtype
        if in_dt == torch.float32:
            x = x.to(torch.float16)
        dt, dev = x.dtype, x.device
        x = x.contiguous()
        sp = _extents(x.shape, ks)
        chans = [w0.shape[0], w1.shape[0], w2.shape[0], w3.shape[0]]
        bufs = _alloc(2 * n if share is not None else n, sp, chans, dt, dev)
        out = _forward(x, params, bufs, 0, sp, ks)
        if share is not None:                             # a second pass may join (DiscShare)
            share.bufs, share.meta = bufs, (n, cin, sp, dt, ks, chans)
            share.weights = tuple((p, p._version) for p in params)
        ctx.bufs = bufs
        ctx.save_for_backward(w0, w1, w2, w3, wl)
        ctx.meta = (n, cin, sp, dt, ks, in_dt)
        ctx.params = params
        out = out.view(n, 1, *sp[5])                      # one channel: channels-last == NCDHW
        return out.float() if in_dt == torch.float32 else out.clone() if share is not None else out

    @staticmethod
    def backward(ctx, dout):
        n, cin, sp, dt, ks, in_dt = ctx.meta
        # fp32 caller: no outer loss scale protects the fp16 gradients -> a device-side scale (no host synchronisation)
        gs = (GRAD_PEAK / dout.abs().amax().clamp_min(1e-30).float()) if in_dt == torch.float32 else None
        # parameter gradients are skipped when no parameter asks for one (TrainStep freezes the discriminator for the
        # generator's pass: train.py:265 computes them there too, and optimizer_d.zero_grad() at :282 throws them away)
        need_w = any(ctx.needs_input_grad[2:])
        dx, rets = _backward(ctx.bufs, 0, n, sp, ks, dt, cin, ctx.saved_tensors, ctx.params, dout, need_w, ctx.needs_input_grad[0], gs)
        if dx is not None and in_dt == torch.float32:
            dx = dx.float() / gs
        return (dx, None, *rets)


class DiscPairFn(Function):
    """[D(first); D(x)]: the forward of x only, into the second half of the buffers a DiscShare-d first pass left; backward =
    ONE batched pass over both halves (parameter gradients; x is a detached sample, train.py:272-277)."""

    @staticmethod
    def forward(ctx, x, share, *params):
        n, cin, sp, dt, ks, chans = share.meta
        if x.dtype == torch.float32:
            x = x.to(dt)
        if tuple(x.shape) != (n, cin) + sp[0] or x.dtype != dt:
            raise ValueError("the second pass must have the first pass's shape and storage type")
        if any(p is not q or p._version != v for p, (q, v) in zip(params, share.weights)):
            raise RuntimeError("the discriminator's weights changed since the shared first pass: its activations are stale")
        _forward(x.contiguous(), params, share.bufs, n, sp, ks)
        ctx.bufs, ctx.meta, ctx.params = share.bufs, share.meta, params
        ctx.save_for_backward(params[0], params[2], params[4], params[6], params[8])
        share.bufs = None                                 # one use
        return ctx.bufs["out"].view(2 * n, 1, *sp[5]).clone()

    @staticmethod
    def backward(ctx, dout):
        n, cin, sp, dt, ks, chans = ctx.meta
        _, rets = _backward(ctx.bufs, 0, 2 * n, sp, ks, dt, cin, ctx.saved_tensors, ctx.params, dout.contiguous(), True, False, None)
        return (None, None, *rets)


def _xconv(mode, x, w, b, y, n, cin, cout, sp_in, sp_out, ks, stride):
    L.check(L.load().xh_dconv_exact(_s(), mode, ops._p(x), ops._p(w), ops._p(b), ops._p(y), n, cin, cout, *sp_in, *sp_out, ks, stride),
            "xh_dconv_exact")


class DiscExactFn(Function):
    """The discriminator in EXACT fp32 (Discriminator.fp32_exact): NCDHW fp32 activations, direct fp32 convolutions
    (xh_dconv_exact), InstanceNorm / LeakyReLU(0.2) through the generic fp32 passes of the generator's path.  No 16-bit operand:
    a training step in fp32 storage then agrees with the fp32 CPU restatement under tests/ to fp32 round-off end to end.  A parity route (~1 TFLOP/s),
    not the product path."""

    @staticmethod
    def forward(ctx, x, *params):
        w0, b0, w1, b1, w2, b2, w3, b3, wl = params
        x = x.contiguous()
        n, cin = x.shape[:2]
        ks = w0.shape[2]
        sp = _extents(x.shape, ks)
        ws, bs = (w0, w1, w2, w3, wl), (b0, b1, b2, b3, None)
        new = lambda c, s_: torch.empty((n, c) + tuple(s_), dtype=torch.float32, device=x.device)
        ins, raws, stats = [x], [], []
        h = x
        for k in range(4):
            c = ws[k].shape[0]
            raw = new(c, sp[k + 1])
            _xconv(0, h, ws[k], bs[k], raw, n, ws[k].shape[1], c, sp[k], sp[k + 1], ks, STRIDES[k])
            raws.append(raw)
            if k == 0:
                h = ops.affine_act(raw, None, None, L.ACT_LRELU, SLOPE)
                stats.append(None)
            else:
                red = ops.zeros_red(raw, n, c)
                ops.moments(raw, red)
                h, sc, sh, mean, rstd = ops.in_affine_act(raw, red, L.ACT_LRELU, SLOPE)
                stats.append((sc, sh, mean, rstd))
            ins.append(h)
        out = new(1, sp[5])
        _xconv(0, h, wl, None, out, n, wl.shape[1], 1, sp[4], sp[5], ks, 1)
        ctx.save_for_backward(*ins, *raws, *[t for st in stats[1:] for t in st], w0, w1, w2, w3, wl)
        ctx.meta = (n, cin, sp, ks)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        from .functional import _targets
        n, cin, sp, ks = ctx.meta
        sv = ctx.saved_tensors
        ins, raws, st, ws = sv[:5], sv[5:9], sv[9:21], sv[21:26]
        stats = [None] + [st[4 * i:4 * i + 4] for i in range(3)]
        need_w = any(ctx.needs_input_grad[1:])
        gb, rets = _targets(ctx.params) if need_w else ([None] * 9, [None] * 9)
        g_w = (gb[0], gb[2], gb[4], gb[6], gb[8])
        g_b = (gb[1], gb[3], gb[5], gb[7], None)
        dy = dout.contiguous().float()
        new = lambda c, s_: torch.empty((n, c) + tuple(s_), dtype=torch.float32, device=dy.device)
        if need_w:
            _xconv(2, ins[4], g_w[4], None, dy, n, ws[4].shape[1], 1, sp[4], sp[5], ks, 1)
        da = new(ws[4].shape[1], sp[4])
        _xconv(1, dy, ws[4], None, da, n, ws[4].shape[1], 1, sp[4], sp[5], ks, 1)
        dx = None
        for k in (3, 2, 1, 0):
            c, ci = ws[k].shape[0], ws[k].shape[1]
            raw = raws[k]
            if k == 0:                                      # conv + bias -> LeakyReLU
                one = torch.ones((n, c), dtype=torch.float32, device=dy.device)
                zero = torch.zeros((n, c), dtype=torch.float32, device=dy.device)
                dc = ops.norm_bwd_apply(da, raw, (one, zero, zero), have_g=False, sc=one, sh=zero, slope=SLOPE)
            else:                                           # conv (+ identity bias) -> InstanceNorm -> LeakyReLU
                sc, sh, mean, rstd = stats[k]
                red = ops.act_bwd_reduce(da, raw, sc, sh, SLOPE)
                dc = ops.in_bwd_apply(da, raw, red, mean, rstd, have_g=False, sc=sc, sh=sh, slope=SLOPE)
            if need_w:
                _xconv(2, ins[k], g_w[k], g_b[k], dc, n, ci, c, sp[k], sp[k + 1], ks, STRIDES[k])
            if k > 0 or ctx.needs_input_grad[0]:
                nxt = new(ci, sp[k])
                _xconv(1, dc, ws[k], None, nxt, n, ci, c, sp[k], sp[k + 1], ks, STRIDES[k])
                if k == 0:
                    dx = nxt
                da = nxt
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

    def _params(self):
        ps = []
        for blk in self.disc:
            ps += [blk[0].weight, blk[0].bias]
        return ps + [self.last.weight]

    def forward(self, x, input_level=0, share=None):
        """`share` (a DiscShare, not in the reference signature): keep this pass's activations so that forward_pair() can
        batch a second pass with it."""
        if input_level != 0:
            raise NotImplementedError("input_level > 0 is not used by train.py")
        if getattr(self, "fp32_exact", False) and x.dtype == torch.float32:
            return DiscExactFn.apply(x, *self._params())   # (no shared second pass: forward_pair falls back to two passes)
        return DiscFn.apply(x, share, *self._params())

    def forward_pair(self, x, share):
        """[D(first pass's input); D(x)] (2N outputs): only x is run forward, the first pass's activations are reused -- the
        reference's `disc(fake.detach())` after `disc(fake)` (train.py:260,272) recomputes identical values."""
        return DiscPairFn.apply(x, share, *self._params())

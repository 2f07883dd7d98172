"""Fused stages of the XLSTM-HVED hot path as torch.autograd.Functions.

Every forward and every backward below is a sequence of C-ABI kernel launches (ops.py); autograd is only
the bookkeeping that connects stages.  Saved tensors are stage inputs plus per-(n,c) statistics, never
normalised activations.  Reference lines are cited at each stage."""
import math

import torch

from . import ops


class Function(torch.autograd.Function):
    """torch.autograd.Function whose backward issues its launches in the fp32-storage ARITHMETIC MODE its forward ran in
    (ops.ARITH: xh_conv_desc.arith of every conv call).  The mode belongs to the call, so a backward pass that autograd runs long
    after the forward -- outside the model's arith_scope, possibly interleaved with another model's passes -- must carry it along."""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        fwd, bwd = cls.__dict__.get("forward"), cls.__dict__.get("backward")
        if fwd is not None:
            f = fwd.__func__ if isinstance(fwd, staticmethod) else fwd

            def forward(ctx, *a, **k):
                ctx._xh_arith = ops.ARITH[0]
                return f(ctx, *a, **k)
            forward.__doc__ = f.__doc__
            cls.forward = staticmethod(forward)
        if bwd is not None:
            b = bwd.__func__ if isinstance(bwd, staticmethod) else bwd

            def backward(ctx, *g):
                prev = ops.ARITH[0]
                ops.ARITH[0] = getattr(ctx, "_xh_arith", None)
                try:
                    return b(ctx, *g)
                finally:
                    ops.ARITH[0] = prev
            backward.__doc__ = b.__doc__
            cls.backward = staticmethod(backward)

from .ops import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, LEAK, MODE_BN_EVAL, MODE_BN_TRAIN, MODE_GN, MODE_IN)


def _blk(t):
    """Make a gradient tensor sample-contiguous (autograd may hand us expanded / strided views)."""
    if t is None:
        return None
    if t.dim() >= 2 and t[0].is_contiguous() and (t.shape[0] == 1 or t.stride(0) >= t[0].numel()):
        return t
    return t.contiguous()


def _dhw(t):
    return t.shape[2] * t.shape[3] * t.shape[4]


def _direct(*rets):
    """True when every gradient target of a wgrad call is an existing .grad buffer (nothing is returned to autograd), so
    the launch may be deferred to the end of backward (ops.set_wgrad_defer) or run on the weight-gradient side stream
    (ops.set_wgrad_overlap).  That includes the tensors composed by ComposeAll: each carries a gradient buffer of its own
    (`_xh_gbuf`, _targets) that every use accumulates into; ComposeAll.backward joins the outstanding launches, then reads it."""
    return all(r is None for r in rets)


DIRECT_GRADS = [True]


def _targets(params):
    """Where the parameter-gradient kernels accumulate, and what autograd gets back.

    All weight/bias gradient kernels ACCUMULATE (+=).  If a leaf parameter already owns a contiguous fp32 .grad
    (e.g. a view of parallel.FlatGrads' bucket, or last step's gradient after zero_grad(set_to_none=False)) the kernel
    adds straight into it and autograd receives None: no zero-fill, no AccumulateGrad add, and the flat bucket is ready
    for the all-reduce.  Otherwise a zeroed tensor is allocated and returned the usual way."""
    bufs, rets, need = [], [], []
    for i, p in enumerate(params):
        g = p.grad if (p is not None and p.is_leaf and p.requires_grad) else None
        if g is None and p is not None:
            g = getattr(p, "_xh_gbuf", None)      # a tensor composed by ComposeAll: its step-long gradient buffer
        if DIRECT_GRADS[0] and g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape:
            bufs.append(g)
            rets.append(None)
        else:
            bufs.append(None)
            rets.append(None)
            need.append(i)
    if need:                                  # one zero buffer for all of them (16-float aligned views)
        offs, total = [], 0
        for i in need:
            offs.append(total)
            total += (params[i].numel() + 15) // 16 * 16
        # Gradients of NON-leaf weights (composed 7^3 / head weights) only travel through autograd to the compose kernels
        # within this backward: a slice of the per-step fp32 zero arena serves (no fill launch).  A LEAF parameter without a
        # .grad may have the returned tensor adopted as its .grad by AccumulateGrad -- that must be private storage.
        if all(not params[i].is_leaf for i in need):
            flat = ops.zeros_f32(params[need[0]].device, total)
        else:
            flat = torch.zeros(total, dtype=torch.float32, device=params[need[0]].device)
        for i, o in zip(need, offs):
            z = flat[o:o + params[i].numel()].view(params[i].shape)
            bufs[i] = rets[i] = z
    return bufs, rets


# ------------------------------------------------------------------------------------------------------
# Gradient slots.  A forward tensor with several consumers gets one gradient per consumer, and autograd sums them with an
# element-wise add launch each (16 per step, the largest over 16 channels x 128^3).  Where the consumers are this package's own
# Functions, they can add into ONE buffer instead: `fanout(x, n)` hands out n aliases of x that carry a GradSlot; the first
# slot-aware backward to run allocates the buffer and returns it, the later ones add into it in place (their kernels take an
# accumulate flag) and return None; Fanout.backward -- which autograd runs after ALL consumers -- returns the buffer and adds
# whatever a consumer that does not know about slots returned on its own.  Everything is ordered on one stream.
class GradSlot:
    __slots__ = ("buf", "dest")

    def __init__(self, dest=None):
        self.buf = None
        self.dest = dest          # (TwinDest, half) or None: where the FIRST consumer writes (instead of a new tensor)


class TwinDest:
    """Gradient buffer of a recon | seg pair (1, 2C, ...) whose two halves are produced by different consumers: the consumers
    that know the protocol WRITE their half here (`_out` / `_dst`), so the producer's backward (Upsample2, Split2) finds both
    halves side by side and runs ONE launch over the pair; anything else handed back is copied in (correct, one pass more).
    Allocated at the first request of a backward pass."""

    def __init__(self, shape, c, dtype, device):
        self.shape, self.c, self.dtype, self.device, self.buf = tuple(shape), c, dtype, device, None

    def half(self, i):
        """Part 0 = channels [0, c), part 1 = the rest (the pair: c = C of 2C; a general split point for Fn.split_at)."""
        if self.buf is None:
            self.buf = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        return self.buf[:, :self.c] if i == 0 else self.buf[:, self.c:]

    def joined(self, ga, gb):
        for i, g in enumerate((ga, gb)):
            h = self.half(i)
            if g is None:
                h.zero_()
            elif g.data_ptr() != h.data_ptr():
                ops.add(_blk(g), None, out=h)
        buf, self.buf = self.buf, None          # a retained graph's next backward pass gets a buffer of its own
        return buf


def _dst(d):
    """The tensor a (TwinDest, half) pair stands for, or None."""
    return d[0].half(d[1]) if d is not None else None


def _out(slot):
    """Where the FIRST consumer of a slot writes: the slot's destination half, or None (= a new tensor)."""
    if slot is None or isinstance(slot, SubSlot):
        return None
    return _dst(slot.dest) if slot.buf is None else None


class Fanout(Function):
    @staticmethod
    def forward(ctx, x, n, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        slot = ctx.slot
        total = slot.buf
        slot.buf = None
        for g in gs:
            if g is None or (total is not None and g.data_ptr() == total.data_ptr()):
                continue
            total = _blk(g) if total is None else ops.add(total, _blk(g), out=total)
        return total, None, None


_FANOUT = [True]


def fanout(x, n):
    """n aliases of x whose consumers share one gradient buffer (see above).  set_fanout(False): plain x, n times (A/B)."""
    if not _FANOUT[0] or not x.requires_grad:
        return (x,) * n
    slot = GradSlot(getattr(x, "_xh_dest", None))      # the aliases' shared buffer lands where x's gradient is wanted
    outs = Fanout.apply(x, n, slot)
    for o in outs:
        o._xh_slot = slot
    return outs


def set_fanout(enabled):
    _FANOUT[0] = bool(enabled)


class SubSlot:
    """A channel range [c0, c1) of a GradSlot's buffer: the gradient slot of a SliceView alias.  A consumer of the slice adds its
    share straight into that range of the parent's buffer once a consumer of the WHOLE tensor has created it; before that it
    hands its gradient to autograd like any other op (SliceView.backward embeds it)."""
    __slots__ = ("parent", "c0", "c1")

    def __init__(self, parent, c0, c1):
        self.parent, self.c0, self.c1 = parent, c0, c1


def _slot(t):
    return getattr(t, "_xh_slot", None) if t is not None else None


def _acc(slot):
    """The buffer a slot-aware backward adds into (None: it is the first -- or there is no slot -- and allocates)."""
    if isinstance(slot, SubSlot):
        return slot.parent.buf[:, slot.c0:slot.c1] if slot.parent.buf is not None else None
    return slot.buf if slot is not None else None


def _ret(slot, t):
    """What that backward returns for the tensor: the buffer if it has just created it, None if it added into an existing one."""
    if slot is None:
        return t
    if isinstance(slot, SubSlot):
        return None if slot.parent.buf is not None else t
    if slot.buf is None:
        slot.buf = t
        return t
    return None


class SliceView(Function):
    """x[:, c0:c1] of a fanout alias as a tensor of its own whose consumers add their gradient into that range of the alias'
    shared buffer (SubSlot).  Plain slicing would have autograd build a zero tensor of x's size per slice and add it."""

    @staticmethod
    def forward(ctx, x, c0, c1):
        ctx.meta = (tuple(x.shape), x.dtype, x.device, c0, c1)
        ctx.set_materialize_grads(False)
        return x[:, c0:c1]

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        shape, dtype, device, c0, c1 = ctx.meta
        full = torch.zeros(shape, dtype=dtype, device=device)          # (a consumer ran before the whole-tensor one: rare, small)
        ops.add(_blk(g), None, out=full[:, c0:c1])
        return full, None, None


def slice_view(x, c0, c1):
    """Channels [c0, c1) of x (one sample per launch: the slice is contiguous); x may be a fanout alias."""
    y = SliceView.apply(x, c0, c1)
    ps = _slot(x)
    if isinstance(ps, GradSlot):
        y._xh_slot = SubSlot(ps, c0, c1)
    return y


# ------------------------------------------------------------------------------------------------------
# Norm-backward fold.  Between the two convs of a DoubleConv the backward pass used to run: data gradient of conv 2 (leaves the
# masked gradient g and the sums) -> xh_in_bwd_apply (reads g and y1, writes dy1) -> data gradient of conv 1 (reads dy1).  When
# y1 has no other consumer, conv 2's backward now returns an UNWRITTEN tensor for dy1 and parks (g, y1, sums, statistics) here
# under its address; conv 1's backward takes the entry and lets its data-gradient launch apply the norm backward while staging
# (ops.conv3d nb=..., xh_conv_desc.pre == 2), which also writes dy1 for the weight gradient.  Whoever pops an entry fills
# the tensor -- with the fused launch or, where that does not apply, with the xh_in_bwd_apply pass.  The entry owns the tensor,
# so its address cannot be reused while it is parked; model.forward() drops leftovers of an aborted backward.
_NB_PENDING = ops.NB_PENDING


def nb_pending_clear():
    """Start of a forward pass: nothing may be parked.  Leftovers mean the previous backward handed a gradient over that nobody
    took (an aborted backward, or a consumer this package does not know): its gradients were not valid."""
    if _NB_PENDING:
        import warnings
        warnings.warn(f"xlstm_hved_amd: {len(_NB_PENDING)} norm-backward hand-over(s) of the previous backward pass were never taken; "
                      "that pass's gradients were incomplete (ops.set_norm_bwd_fold(False) disables the hand-over)")
        _NB_PENDING.clear()


NB_FOLD_MAX = 1 << 22       # elements of the handed-over tensor up to which the fold pays (measured on MI355X: 16 ch @64^3 gains 4 us,
                            # 4 ch @128^3 loses 5 us, 16 ch @128^3 loses 9 us -- the element-wise pass streams at 6 TB/s, the conv at 3.3)


class InLreluConv(Function):
    """SingleConv 'ilc' (buildingblocks.py:406-433,440-461): Conv3d(LeakyReLU(InstanceNorm3d(x))) + bias, k=3.

    Inputs may be a virtual concat (xa | xb) -- the decoder's torch.cat((enc, x), 1) at buildingblocks.py:732 --
    and may be `groups` independent streams with one weight tensor per group (the 4 modality encoders,
    RA_HVED.py:548-553, batched along channels)."""
    _into = None

    @staticmethod
    def forward(ctx, xa, xb, in_stats, out_stats, stride, groups, nw, drop_bias, sole_consumer, *wb):
        """in_stats: the (n, C, 2) fp64 sums [sum x, sum x^2] of xa if its producer already accumulated them in its
        epilogue (then no moments pass is run here); out_stats: also return the same sums of the output, accumulated by
        this conv's epilogue, for the next stage.  drop_bias: see in_lrelu_conv.  sole_consumer: xa is the output of another
        InLreluConv and nothing else reads it (the inside of a DoubleConv): its gradient may be handed over unwritten
        (_NB_PENDING)."""
        ctx.sole = bool(sole_consumer) and xb is None
        into = InLreluConv._into                  # (y destination, sums destination) set by in_lrelu_conv(into=...)
        InLreluConv._into = None
        ctx.dests = (getattr(xa, "_xh_dest", None), getattr(xb, "_xh_dest", None) if xb is not None else None)
        weights, biases = list(wb[:nw]), list(wb[nw:])
        n, ca = xa.shape[:2]
        cin = ca + (xb.shape[1] if xb is not None else 0)
        if in_stats is not None and xb is None:
            red = in_stats
        else:
            red = ops.zeros_red(xa, n, cin)
            if xb is not None:
                ops.moments2(xa, xb, red)
            else:
                ops.moments(xa, red, 0)
        cout = sum(w.shape[0] for w in weights)
        k = weights[0].shape[-1]
        red_y = (into[1] if into is not None else ops.zeros_red(xa, n, cout)) if out_stats else None
        y, sc, sh, mean, rstd = ops.conv3d(xa, xb, weights, None if drop_bias else biases, k=k, cout=cout, stride=stride,
                                           groups=groups, in_stats=(red, _dhw(xa), LEAK), epi=2 if out_stats else 0, red=red_y,
                                           out=into[0] if into is not None else None)
        ctx.save_for_backward(xa, xb, sc, sh, mean, rstd, *weights)
        ctx.cfg = (stride, groups, nw, k, cin, ca)
        ctx.params = (weights, biases)
        ctx.slots = (_slot(xa), _slot(xb))
        if out_stats:
            ctx.mark_non_differentiable(red_y)
            ctx.set_materialize_grads(False)          # no zero-filled "gradient" for the statistics output
            return y, red_y
        return y

    @staticmethod
    def backward(ctx, dy, _dred=None):
        xa, xb, sc, sh, mean, rstd, *weights = ctx.saved_tensors
        sa, sb = ctx.slots
        stride, groups, nw, k, cin, ca = ctx.cfg
        dy = _blk(dy)
        dws, rws = _targets(ctx.params[0])
        dbs, rbs = _targets(ctx.params[1])
        need_dx = ctx.needs_input_grad[0] or (xb is not None and ctx.needs_input_grad[1])
        nb = _NB_PENDING.pop(dy.data_ptr(), None)          # dy handed over unwritten by the next conv's backward (see above)
        if nb is not None and not (need_dx and stride == 1):
            ops.in_bwd_apply(nb[0], nb[1], nb[2], nb[3], nb[4], have_g=True, out=dy)       # nobody to fold it into: write it now
            nb = None
        if nb is None:
            ops.conv3d_wgrad(xa, xb, dy, dws, dbs, k=k, stride=stride, groups=groups, pre=(sc, sh, LEAK), side=_direct(*rws, *rbs))
        dxa = dxb = None
        if need_dx:
            n = xa.shape[0]
            red = ops.zeros_red(xa, n, cin)
            e = (xa, xb, sc, sh, LEAK)
            if nb is not None:
                # the data gradient applies the pending norm backward on load and writes dy; the weight gradient, which reads
                # dy, is issued behind it
                g = ops.conv3d(nb[0], None, weights, None, k=k, cout=cin, groups=groups, transposed=True, epi=1, e=e, red=red,
                               nb=(nb[1], nb[2], nb[3], nb[4], dy))
                ops.conv3d_wgrad(xa, xb, dy, dws, dbs, k=k, stride=stride, groups=groups, pre=(sc, sh, LEAK), side=_direct(*rws, *rbs))
            elif stride == 1:
                g = ops.conv3d(dy, None, weights, None, k=k, cout=cin, groups=groups, transposed=True, epi=1, e=e, red=red)
            else:
                g = ops.conv3d_dgrad_s2(dy, weights, cin=cin, in_spatial=tuple(xa.shape[2:]), groups=groups, e=e, red=red)
            if xb is not None:
                da_, db_ = ctx.dests
                dxa, dxb = ops.in_bwd_apply2(g, xa, xb, red, mean, rstd, acc_a=_acc(sa), acc_b=_acc(sb),
                                             out_a=_out(sa) if sa is not None else _dst(da_), out_b=_out(sb) if sb is not None else _dst(db_))
                dxa, dxb = _ret(sa, dxa), _ret(sb, dxb)
            elif ctx.sole and sa is None and ops._NB_FOLD[0] and xa.numel() <= NB_FOLD_MAX:
                dxa = torch.empty_like(xa, memory_format=torch.contiguous_format)          # written by whoever takes the entry
                _NB_PENDING[dxa.data_ptr()] = (g, xa, red, mean, rstd, dxa)
            else:
                dxa = _ret(sa, ops.in_bwd_apply(g, xa, red, mean, rstd, have_g=True, c0=0, acc=_acc(sa),
                                                out=_dst(ctx.dests[0]) if sa is None else None))
        return (dxa, dxb, None, None, None, None, None, None, None, *rws, *rbs)


def in_lrelu_conv(xa, xb, weights, biases, stride=1, groups=1, in_stats=None, out_stats=False, drop_bias=False, sole_consumer=False,
                  into=None):
    """drop_bias=True: the caller guarantees that every consumer of the output is an InstanceNorm (which subtracts the
    per-channel mean, so IN(conv + b) == IN(conv) exactly): the bias add is skipped and the tensor is stored without the
    offset.  With 16-bit storage that matters: the reference initialises biases N(0,1) (utils.py:199), and a channel
    stored as `offset + small signal` spends its significant bits on the offset.  The bias still gets its gradient (the sum
    of dY, mathematically zero behind an InstanceNorm -- the reference returns round-off there too).
    into: (y, sums) destinations -- channel slices of a pair buffer the caller owns (the decoder's recon | seg pair)."""
    InLreluConv._into = into
    try:
        return InLreluConv.apply(xa, xb, in_stats, bool(out_stats), stride, groups, len(weights), bool(drop_bias), bool(sole_consumer),
                                 *weights, *biases)
    finally:
        InLreluConv._into = None


class InitInLreluConv(Function):
    """The init blocks (1x1 convs from ONE modality to B channels, RA_HVED.py:345-349,548) and the first SingleConv 'ilc' of the
    level-0 encoders (buildingblocks.py:406-433) WITHOUT the init convs' output: their only consumer is that SingleConv's
    InstanceNorm, and IN(w_c x_m + b_c) = sc_c x_m + sh_c is an affine of the input modality itself (ops.init_fold_fwd).  The conv
    reads x as a broadcast operand (xh_conv_desc.bcast: one stored channel per group through four (sc, sh) pairs).  Removed from the
    step: the init conv launch and its 16-channel output (the second most flip-sensitive tensor of 16-bit storage,
    tools/precision_sweep.py), in backward the store of the first conv's data gradient, its InstanceNorm-backward pass and the init
    convs' weight-gradient problem -- the init weights' gradient is eps R^3 (S1 - mean S0) from the data gradient's epilogue sums
    (ops.init_fold_bwd; the init biases' gradient is exactly zero)."""

    @staticmethod
    def forward(ctx, x, nm, *params):
        init_w, init_b = list(params[:nm]), list(params[nm:2 * nm])
        conv_w, conv_b = list(params[2 * nm:3 * nm]), list(params[3 * nm:])
        n, cnt = x.shape[0], _dhw(x)
        red_x = ops.zeros_red(x, n, nm)
        ops.moments(x, red_x, 0)
        sc, sh, _, ctr = ops.init_fold_fwd(red_x, cnt, n, init_w)
        cout = sum(w.shape[0] for w in conv_w)
        red_y = ops.zeros_red(x, n, cout)
        y = ops.conv3d(x, None, conv_w, None, k=3, cout=cout, groups=nm, pre=(sc, sh, LEAK), epi=2, red=red_y, bcast=4)
        ctx.save_for_backward(x, red_x, sc, sh, ctr, *conv_w, *init_w)
        ctx.nm = nm
        ctx.params = (init_w, init_b, conv_w, conv_b)
        ctx.mark_non_differentiable(red_y)
        ctx.set_materialize_grads(False)
        return y, red_y

    @staticmethod
    def backward(ctx, dy, _dred=None):
        nm = ctx.nm
        x, red_x, sc, sh, ctr, *wts = ctx.saved_tensors
        conv_w, init_w = wts[:nm], wts[nm:]
        dy = _blk(dy)
        nb = _NB_PENDING.pop(dy.data_ptr(), None)           # dy handed over unwritten by the next conv's backward: write it now
        if nb is not None:
            ops.in_bwd_apply(nb[0], nb[1], nb[2], nb[3], nb[4], have_g=True, out=dy)
        diw, riw = _targets(ctx.params[0])
        _, rib = _targets(ctx.params[1])                     # exactly zero: existing buffers are left alone, new ones are zeros
        dws, rws = _targets(ctx.params[2])
        dbs, rbs = _targets(ctx.params[3])
        n, cnt, cin = x.shape[0], _dhw(x), 4 * nm
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=3, groups=nm, pre=(sc, sh, LEAK), side=_direct(*rws, *rbs), bcast=4)
        red = ops.zeros_red(x, n, cin)
        ops.conv3d(dy, None, conv_w, None, k=3, cout=cin, groups=nm, transposed=True, epi=1, e=(x, None, sc, sh, LEAK, ctr), red=red,
                   out=False, bcast=4)
        ops.init_fold_bwd(red_x, cnt, n, init_w, red, diw)
        return (None, None, *riw, *rib, *rws, *rbs)


INIT_FOLD = [True]


def set_init_fold(enabled):
    """A/B switch: the init blocks folded into the first encoder conv (InitInLreluConv); off = the init convs' output is stored."""
    INIT_FOLD[0] = bool(enabled)


def init_fold_ok(x, init_w, conv_w):
    """The fold applies: one input channel per init block and four output channels (a quad per modality), 16-bit storage on the
    full-row kernels, no gradient wanted for x."""
    return (INIT_FOLD[0] and x.is_cuda and not x.requires_grad and len(init_w) == x.shape[1] == len(conv_w)
            and all(tuple(w.shape) == (4, 1, 1, 1, 1) for w in init_w)
            and all(tuple(w.shape[:2]) == (4, 4) and w.shape[-1] == 3 for w in conv_w)
            and ops.conv3d_supports_bcast(x, 4 * len(conv_w), len(conv_w)))


def init_in_lrelu_conv(x, init_w, init_b, conv_w, conv_b):
    """(y, output sums) of conv1(lrelu(IN(init(x)))) with the init blocks folded (InitInLreluConv)."""
    return InitInLreluConv.apply(x, len(init_w), *init_w, *init_b, *conv_w, *conv_b)


class GnConvRelu(Function):
    """SingleConv 'gcr' (buildingblocks.py:421-429): ReLU(Conv3d(GroupNorm(x))), no conv bias."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, num_groups, stride):
        n, c = x.shape[:2]
        gs = c // num_groups
        red = ops.zeros_red(x, n, c)
        ops.moments(x, red, 0)
        sc, sh, mean, rstd = ops.norm_finalize(MODE_GN, red, n, c, _dhw(x), gs=gs, gamma=gamma, beta=beta)
        k = weight.shape[-1]
        y = ops.conv3d(x, None, [weight], None, k=k, cout=weight.shape[0], stride=stride, pre=(sc, sh, 1.0), act=ACT_RELU)
        ctx.save_for_backward(x, y, weight, gamma, sc, sh, mean, rstd)
        ctx.cfg = (gs, stride, k)
        ctx.params = (weight, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, gamma, sc, sh, mean, rstd = ctx.saved_tensors
        gs, stride, k = ctx.cfg
        n, c = x.shape[:2]
        dyr = ops.act_bwd(_blk(dy), y, ACT_RELU)
        (dw, dgamma, dbeta), (rw, rgamma, rbeta) = _targets(ctx.params)
        ops.conv3d_wgrad(x, None, dyr, [dw], None, k=k, stride=stride, pre=(sc, sh, 1.0), side=_direct(rw))
        red = ops.zeros_red(x, n, c)
        e = (x, None, sc, sh, 1.0)
        if stride == 1:
            g = ops.conv3d(dyr, None, [weight], None, k=k, cout=c, transposed=True, epi=1, e=e, red=red)
        else:
            g = ops.conv3d_dgrad_s2(dyr, [weight], cin=c, in_spatial=tuple(x.shape[2:]), e=e, red=red)
        dx = ops.norm_bwd_fused(MODE_GN, g, x, red, mean, rstd, gs=gs, gamma=gamma, dgamma=dgamma, dbeta=dbeta)
        return dx, rw, rgamma, rbeta, None, None


class ConvInLrelu(Function):
    """BasicConv (buildingblocks.py:13-31): LeakyReLU(InstanceNorm3d(Conv3d(x, no bias))); k=1 dense (VU_blocks,
    RA_HVED.py:401-403) or k=3 depthwise (conv_blocks, RA_HVED.py:406)."""

    @staticmethod
    def forward(ctx, x, weight, groups, up2x=False):
        """up2x: also the trilinear 2x upsampling that follows the VU blocks (RA_HVED.py:600-601) -- norm and activation are
        then applied inside the upsampling launch and the low-resolution activated tensor is never written."""
        n = x.shape[0]
        cout, k = weight.shape[0], weight.shape[-1]
        red = ops.zeros_red(x, n, cout)
        y0 = ops.conv3d(x, None, [weight], None, k=k, cout=cout, groups=groups, epi=2, red=red)
        r = ops.upsample2x_in_act(y0, red, LEAK) if up2x else None
        if r is not None:
            y, sc, sh, mean, rstd = r
        else:
            y, sc, sh, mean, rstd = ops.in_affine_act(y0, red, ACT_LRELU, LEAK)
            if up2x:
                y = ops.upsample(y, tuple(2 * s for s in y.shape[2:]))
        ctx.save_for_backward(x, y0, weight, sc, sh, mean, rstd)
        ctx.cfg = (groups, k, bool(up2x))
        ctx.params = (weight,)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y0, weight, sc, sh, mean, rstd = ctx.saved_tensors
        groups, k, up2x = ctx.cfg
        dy = _blk(dy)
        r = ops.upsample2x_bwd_act_reduce(dy, y0, sc, sh, LEAK) if up2x else None
        if r is not None:
            dy, red = r
        else:
            if up2x:
                dy = ops.upsample_bwd(dy, tuple(y0.shape[2:]))
            red = ops.act_bwd_reduce(dy, y0, sc, sh, LEAK)
        dy0 = ops.in_bwd_apply(dy, y0, red, mean, rstd, have_g=False, sc=sc, sh=sh, slope=LEAK)
        (dw,), (rw,) = _targets(ctx.params)
        ops.conv3d_wgrad(x, None, dy0, [dw], None, k=k, groups=groups, side=_direct(rw))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv3d(dy0, None, [weight], None, k=k, cout=x.shape[1], groups=groups, transposed=True)
        return dx, rw, None, None


LATENT_BATCH = [True]


class LatentFallback(RuntimeError):
    """LatentPath cannot take this configuration (the exact-2x upsampling kernels are switched off or do not take the layout):
    the caller runs the per-level nodes instead."""


def set_latent_batch(enabled):
    """A/B switch: the latent path of all fusion levels as one autograd node whose element-wise passes are multi-problem launches
    (LatentPath) instead of two ConvInLrelu nodes per level."""
    LATENT_BATCH[0] = bool(enabled)


class LatentPath(Function):
    """RA_HVED.py:599-603 for ALL fusion levels at once: per level z -> BasicConv(1x1) -> 2x trilinear upsampling -> BasicConv
    (depthwise 3^3), i.e. two ConvInLrelu nodes per level.  The levels do not depend on each other and all but the finest are a few
    dozen workgroups per launch, so their element-wise passes -- norm + activation inside the upsampling, norm + activation of the
    conv block, and backward the activation-masked sums, the two InstanceNorm backward passes and the upsampling adjoint -- run as
    ONE launch per pass for the four levels (ops.*_multi, include/xlstm_hved.h xh_*_multi) instead of one per level: 24 launches
    of ~5 us less per step.  The convolutions themselves stay one launch per level (different kernels per size).  Same kernel
    bodies, same per-problem grids: the same bits as the per-level path (tests/test_gpu_network.py)."""

    @staticmethod
    def forward(ctx, nlev, groups2, *tw):
        """tw = z_0 .. z_{L-1}, w1_0 .. w1_{L-1} (1x1 weights), w2_0 .. w2_{L-1} (depthwise 3^3 weights); groups2[l] = groups of
        the level's second conv."""
        zs, w1s, w2s = tw[:nlev], tw[nlev:2 * nlev], tw[2 * nlev:3 * nlev]
        y0a, reda = [], []
        ops.conv1x1_collect()                              # the four 1x1 convs (+ output moments): one launch
        try:
            for z, w in zip(zs, w1s):
                red = ops.zeros_red(z, z.shape[0], w.shape[0])
                y0a.append(ops.conv3d(z, None, [w], None, k=1, cout=w.shape[0], groups=1, epi=2, red=red))
                reda.append(red)
        finally:
            ops.conv1x1_flush()
        ups = ops.upsample2x_in_act_multi(y0a, reda, LEAK)
        if ups is None:
            raise LatentFallback("the exact-2x upsampling kernel does not take this layout")
        y0b, redb = [], []
        for (u, _, _, _, _), w, g in zip(ups, w2s, groups2):   # depthwise 3^3 (+ output moments)
            red = ops.zeros_red(u, u.shape[0], w.shape[0])
            y0b.append(ops.conv3d(u, None, [w], None, k=w.shape[-1], cout=w.shape[0], groups=g, epi=2, red=red))
            redb.append(red)
        fin = ops.in_affine_act_multi(y0b, redb, ACT_LRELU, LEAK)
        saved = []
        for l in range(nlev):
            saved += [zs[l], y0a[l], w1s[l], ups[l][1], ups[l][2], ups[l][3], ups[l][4], ups[l][0], y0b[l], w2s[l], fin[l][1], fin[l][2],
                      fin[l][3], fin[l][4]]
        ctx.save_for_backward(*saved)
        ctx.cfg = (nlev, tuple(groups2))
        ctx.params = (tuple(w1s), tuple(w2s))
        return tuple(f[0] for f in fin)

    @staticmethod
    def backward(ctx, *dys):
        nlev, groups2 = ctx.cfg
        t = ctx.saved_tensors
        lv = [t[14 * l:14 * l + 14] for l in range(nlev)]      # z, y0a, w1, sc1, sh1, mean1, rstd1, u, y0b, w2, sc2, sh2, mean2, rstd2
        dys = [_blk(dy) for dy in dys]
        # second BasicConv (depthwise 3^3): activation-masked sums, InstanceNorm backward -- one launch each for all levels
        red2 = ops.act_bwd_reduce_multi(dys, [v[8] for v in lv], [v[10] for v in lv], [v[11] for v in lv], LEAK)
        dy0b = ops.in_bwd_apply_multi(dys, [v[8] for v in lv], red2, [v[12] for v in lv], [v[13] for v in lv], have_g=False,
                                      scs=[v[10] for v in lv], shs=[v[11] for v in lv], slope=LEAK)
        w1s, w2s = ctx.params
        dws2, rws2 = _targets(w2s)
        du = []
        for l, v in enumerate(lv):
            k = v[9].shape[-1]
            ops.conv3d_wgrad(v[7], None, dy0b[l], [dws2[l]], None, k=k, groups=groups2[l], side=_direct(rws2[l]))
            du.append(ops.conv3d(dy0b[l], None, [v[9]], None, k=k, cout=v[7].shape[1], groups=groups2[l], transposed=True))
        # upsampling adjoint + the first BasicConv's activation-masked sums, then its InstanceNorm backward
        r = ops.upsample2x_bwd_act_reduce_multi(du, [v[1] for v in lv], [v[3] for v in lv], [v[4] for v in lv], LEAK)
        if r is None:
            raise RuntimeError("LatentPath.backward: the exact-2x upsampling adjoint does not take this layout")
        dy0a = ops.in_bwd_apply_multi([x[0] for x in r], [v[1] for v in lv], [x[1] for x in r], [v[5] for v in lv], [v[6] for v in lv],
                                      have_g=False, scs=[v[3] for v in lv], shs=[v[4] for v in lv], slope=LEAK)
        dws1, rws1 = _targets(w1s)
        dzs = []
        ops.conv1x1_collect()                              # the four data gradients of the 1x1 convs: one launch
        try:
            for l, v in enumerate(lv):
                ops.conv3d_wgrad(v[0], None, dy0a[l], [dws1[l]], None, k=1, groups=1, side=_direct(rws1[l]))
                dzs.append(ops.conv3d(dy0a[l], None, [v[2]], None, k=1, cout=v[0].shape[1], groups=1, transposed=True)
                           if ctx.needs_input_grad[2 + l] else None)
        finally:
            ops.conv1x1_flush()
        return (None, None, *dzs, *rws1, *rws2)


class Conv(Function):
    """Plain Conv3d (+bias) with optional sigmoid: init_blocks / x0_init / heads (RA_HVED.py:323,347,148-149,
    480,640-641) and AttenModule2's collapsed 7^3 convs (buildingblocks.py:283-296).  `groups` streams may
    carry one weight tensor each."""
    _into = None

    @staticmethod
    def forward(ctx, x, groups, act, nw, has_bias, out_stats, drop_bias, pre_act_grad, *wb):
        """pre_act_grad: the (single) consumer hands back the gradient of the PRE-activation (GateCat sig_bwd): backward skips the
        activation's own backward pass."""
        weights = list(wb[:nw])
        biases = list(wb[nw:]) if has_bias else None
        cout = sum(w.shape[0] for w in weights)
        k = weights[0].shape[-1]
        into, Conv._into = Conv._into, None               # destination (a channel slice of a buffer the caller owns), conv(into=...)
        red_y = ops.zeros_red(x, x.shape[0], cout) if out_stats else None     # output channel sums for the next norm
        y = ops.conv3d(x, None, weights, None if drop_bias else biases, k=k, cout=cout, groups=groups, act=act,
                       epi=2 if out_stats else 0, red=red_y, out=into)
        if pre_act_grad:
            act = ACT_NONE
        ctx.save_for_backward(x, y if act != ACT_NONE else None, *weights)
        ctx.cfg = (groups, act, nw, has_bias, k)
        ctx.dest = getattr(x, "_xh_dest", None)
        ctx.params = (weights, biases)
        if out_stats:
            ctx.mark_non_differentiable(red_y)
            ctx.set_materialize_grads(False)
            return y, red_y
        return y

    @staticmethod
    def backward(ctx, dy, _dred=None):
        x, y, *weights = ctx.saved_tensors
        groups, act, nw, has_bias, k = ctx.cfg
        dy = _blk(dy)
        if act != ACT_NONE:
            dy = ops.act_bwd(dy, y, act)
        dws, rws = _targets(ctx.params[0])
        dbs, rbs = _targets(ctx.params[1]) if has_bias else (None, [])
        ops.conv3d_wgrad(x, None, dy, dws, dbs, k=k, groups=groups, side=_direct(*rws, *rbs))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv3d(dy, None, weights, None, k=k, cout=x.shape[1], groups=groups, transposed=True, out=_dst(ctx.dest))
        return (dx, None, None, None, None, None, None, None, *rws, *rbs)


def conv(x, weights, biases=None, groups=1, act=ACT_NONE, out_stats=False, drop_bias=False, pre_act_grad=False, into=None):
    """drop_bias: as in in_lrelu_conv (output consumed only by InstanceNorm; needs act == ACT_NONE).  into: the tensor to write
    (a channel slice of a buffer the caller owns)."""
    if drop_bias and act != ACT_NONE:
        raise ValueError("drop_bias needs a linear output")
    Conv._into = into
    try:
        return Conv.apply(x, groups, act, len(weights), biases is not None, bool(out_stats), bool(drop_bias), bool(pre_act_grad),
                          *weights, *(biases or []))
    finally:
        Conv._into = None


class MaxPool2(Function):
    """nn.MaxPool3d(2) (buildingblocks.py:635-636)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        ctx.slot = _slot(x)
        return ops.maxpool2(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _ret(ctx.slot, ops.maxpool2_bwd(x, _blk(dy), acc=_acc(ctx.slot)))


class MaxPool2Stats(Function):
    """nn.MaxPool3d(2) that also returns the (n, C, 2) fp64 channel sums of its output: the pooling of an 'ilc' encoder and the
    moments pass of its first InstanceNorm in one launch (xh_gate_maxpool_fwd without a gate)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        ctx.slot = _slot(x)
        x = x.contiguous()
        red = ops.zeros_red(x, x.shape[0], x.shape[1])
        y = ops.gate_maxpool(x, None, red)
        ctx.mark_non_differentiable(red)
        ctx.set_materialize_grads(False)
        return y, red

    @staticmethod
    def backward(ctx, dy, _dred=None):
        (x,) = ctx.saved_tensors
        return _ret(ctx.slot, ops.maxpool2_bwd(x, _blk(dy), acc=_acc(ctx.slot)))


class Upsample(Function):
    """F.interpolate(mode='trilinear') to `size` (buildingblocks.py:785-787, RA_HVED.py:600-601)."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_size = tuple(x.shape[2:])
        return ops.upsample(x, tuple(size))

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample_bwd(_blk(dy), ctx.in_size), None


class PoE(Function):
    """Prior + modality experts -> product of experts -> reparameterisation (RA_HVED.py:573-597,741-753;
    buildingblocks.py:853-886).  feat: (N, 4*2L, d,h,w) = the 4 DRB outputs stacked along channels."""

    @staticmethod
    def forward(ctx, feat, keep, eps, L_, mask_mu):
        feat = feat.contiguous()
        z, mu, lv = ops.poe_fwd(feat, keep, eps, L_, mask_mu)
        ctx.save_for_backward(feat, keep, eps)
        ctx.cfg = (L_, mask_mu)
        ctx.set_materialize_grads(False)
        return z, mu, lv

    @staticmethod
    def backward(ctx, dz, dmu, dlv):
        feat, keep, eps = ctx.saved_tensors
        L_, mask_mu = ctx.cfg
        if dz is None:
            dz = torch.zeros((feat.shape[0], L_) + tuple(feat.shape[2:]), dtype=feat.dtype, device=feat.device)
        g = lambda t: None if t is None else t.contiguous()
        return ops.poe_bwd(feat, keep, eps, g(dz), g(dmu), g(dlv), L_, mask_mu), None, None, None, None


class PoEAll(Function):
    """PoE of every latent level of a forward pass in ONE launch, and their backward in one (ops.poe_*_multi): the levels are
    independent functions of the encoder outputs (RA_HVED.py:573-597 runs them in the level loop).  apply(keep, Ls, mask_mu,
    nlev, *feats, *epss) with epss entries None for the posterior mean; returns (z_0, mu_0, lv_0, z_1, ...).
    `PoEAll.rng_state` (set by the caller for one apply): the generator state of in-kernel noise -- levels with eps None then draw
    eps in the kernel (ops.poe_fwd_multi) and the backward pass regenerates it from the two words the forward recorded."""
    rng_state = None

    @staticmethod
    def forward(ctx, keep, Ls, mask_mu, nlev, *te):
        feats = [t.contiguous() for t in te[:nlev]]
        epss = list(te[nlev:])
        state, PoEAll.rng_state = PoEAll.rng_state, None
        used = torch.empty(2, dtype=torch.int64, device=keep.device) if state is not None else None
        outs = ops.poe_fwd_multi(feats, keep, epss, Ls, mask_mu, rng=(state, used) if state is not None else None)
        ctx.rng_used = used
        ctx.save_for_backward(keep, *feats, *[e for e in epss if e is not None])
        ctx.cfg = (tuple(Ls), mask_mu, nlev, [e is not None for e in epss])
        ctx.set_materialize_grads(False)
        return tuple(t for o in outs for t in o)

    @staticmethod
    def backward(ctx, *g):
        Ls, mask_mu, nlev, has_eps = ctx.cfg
        keep, *rest = ctx.saved_tensors
        feats, it = rest[:nlev], iter(rest[nlev:])
        epss = [next(it) if h else None for h in has_eps]
        c = lambda t: None if t is None else t.contiguous()
        dzs, dmus, dlvs = [], [], []
        for l in range(nlev):
            dz = g[3 * l]
            if dz is None:
                dz = torch.zeros((feats[l].shape[0], Ls[l]) + tuple(feats[l].shape[2:]), dtype=feats[l].dtype, device=feats[l].device)
            dzs.append(c(dz)); dmus.append(c(g[3 * l + 1])); dlvs.append(c(g[3 * l + 2]))
        dfeats = ops.poe_bwd_multi(feats, keep, epss, dzs, dmus, dlvs, Ls, mask_mu, rng_used=ctx.rng_used)
        return (None, None, None, None, *dfeats, *([None] * nlev))


class ChannelPool2(Function):
    """[ChannelPool(seg_x), ChannelPool(enc_x)] -> 4 channels (buildingblocks.py:279-282)."""

    @staticmethod
    def forward(ctx, a, b):
        out = ops.channel_pool2(a, b)                          # both tensors in one launch
        ctx.save_for_backward(a, b)
        ctx.slots = (_slot(a), _slot(b))
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        sa, sb = ctx.slots
        da, db = ops.channel_pool2_bwd(a, b, _blk(dout), acc_a=_acc(sa), acc_b=_acc(sb), out_a=_out(sa))
        return _ret(sa, da), _ret(sb, db)


class GateCat(Function):
    """cat[a*(1+E[:,0]), b*(1+E[:,1])] (buildingblocks.py:287,297-299)."""

    @staticmethod
    def forward(ctx, a, b, E, stats=False, sig_bwd=False):
        """stats: also return the (n, C, 2) fp64 channel sums of the output (the next conv's InstanceNorm: no moments pass).
        sig_bwd: E comes from conv(..., act=ACT_SIGMOID, pre_act_grad=True): backward returns the gradient of the sigmoid's
        PRE-activation for it (one pass instead of gate backward + sigmoid backward)."""
        ctx.sig_bwd = bool(sig_bwd)
        red = ops.zeros_red(a, a.shape[0], a.shape[1] + b.shape[1]) if stats else None
        out = ops.gate2(a, b, E, red)                          # both halves of the concat in one launch
        ctx.save_for_backward(a, b, E)
        ctx.slots = (_slot(a), _slot(b))
        if stats:
            ctx.mark_non_differentiable(red)
            ctx.set_materialize_grads(False)              # no zero-filled "gradient" of the sums in backward
            return out, red
        return out

    @staticmethod
    def backward(ctx, dout, *_):
        a, b, E = ctx.saved_tensors
        sa, sb = ctx.slots
        da, db, dE = ops.gate2_bwd(a, b, E, _blk(dout), acc_a=_acc(sa), acc_b=_acc(sb), sig_bwd=ctx.sig_bwd, out_a=_out(sa))
        return _ret(sa, da), _ret(sb, db), dE, None, None


class Gate(Function):
    """x*(1+a): the skip-return attention applied to every stream, x_i = a*x_i + x_i (RA_HVED.py:552)."""

    @staticmethod
    def forward(ctx, x, a):
        ctx.save_for_backward(x, a)
        return ops.gate(x, a)

    @staticmethod
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        return ops.gate_bwd(x, a, _blk(dy))


class GateMaxPool(Function):
    """MaxPool3d(2)(x * (1 + a)) + the channel sums of the result for the InstanceNorm that follows: the skip-return gate
    (RA_HVED.py:552), the next encoder's pooling (buildingblocks.py:655-657) and the moments pass of its first SingleConv in
    ONE launch; backward = arg-max routing and gate backward in one launch."""

    @staticmethod
    def forward(ctx, x, a):
        ctx.slot = _slot(x)
        x, a = x.contiguous(), a.contiguous()
        red = ops.zeros_red(x, x.shape[0], x.shape[1])
        y = ops.gate_maxpool(x, a, red)
        ctx.save_for_backward(x, a)
        ctx.mark_non_differentiable(red)
        ctx.set_materialize_grads(False)
        return y, red

    @staticmethod
    def backward(ctx, dy, _dred=None):
        x, a = ctx.saved_tensors
        dx, da = ops.gate_maxpool_bwd(x, a, _blk(dy), acc=_acc(ctx.slot))
        return _ret(ctx.slot, dx), da


class GateMaxPool5(Function):
    """GateMaxPool on the four modality streams with the SKIP stream riding along ungated: xs = [X (4C) | S (C)] (1, 5C, ...) ->
    MaxPool3d(2)([X * (1 + a) | S]) and the channel sums of the result -- the skip-return path's own pooling (the Encoder of
    RA_HVED.py:374-381 starts with MaxPool3d(2)) in the modality streams' launch.  apply(a, base, *parts): parts = (xs,), or the two
    halves (X, S) of the buffer `base` when different stages produced them (level 0: encoder conv and x0_init)."""

    @staticmethod
    def forward(ctx, a, base, *parts):
        xs = base if base is not None else parts[0]
        n, c5 = xs.shape[:2]
        cg = c5 // 5 * 4
        red = ops.zeros_red(xs, n, c5)
        y = ops.gate_maxpool(xs, a.contiguous(), red, gated=cg)
        ctx.save_for_backward(xs, a)
        ctx.cg = cg
        ctx.slots = tuple(_slot(p) for p in parts)
        ctx.mark_non_differentiable(red)
        ctx.set_materialize_grads(False)
        return y, red

    @staticmethod
    def backward(ctx, dy, _dred=None):
        xs, a = ctx.saved_tensors
        cg = ctx.cg
        dy = _blk(dy)
        if len(ctx.slots) == 1:
            slot = ctx.slots[0]
            dx, da = ops.gate_maxpool_bwd(xs, a, dy, acc=_acc(slot), gated=cg)
            return (da, None, _ret(slot, dx))
        sx, ss = ctx.slots
        dx, da = ops.gate_maxpool_bwd(xs, a, dy, gated=cg)
        outs = []
        for slot, half in ((sx, dx[:, :cg]), (ss, dx[:, cg:])):
            acc = _acc(slot)
            if acc is not None:                            # (another consumer of this half ran first)
                ops.add(acc, half, out=acc)
                outs.append(None)
            else:
                outs.append(_ret(slot, half))
        return (da, None, *outs)


class Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


# ------------------------------------------------------------------------------------------------------
class SkipReturnAttention(Function):
    """nn.Sequential(ResBlock(c, c, lkdw=True), SpacialAttention3D(1)) (RA_HVED.py:371-384; sa_module.py:56-137;
    attention_blocks.py:112-126) evaluated ONCE per level; BatchNorm running statistics are advanced `steps`
    times to reproduce the reference's 4 evaluations on identical input (RA_HVED.py:548-552)."""

    @staticmethod
    def forward(ctx, x, training, steps, rm1, rv1, rm2, rv2, dw1, pw1w, pw1b, g1, b1, dw2, pw2w, pw2b, g2, b2, saw, wc1=None, wc2=None):
        """wc1 / wc2 (from ComposeAll, 16-bit storage): the dense 3^3 weights pw o dw of the two DWConvNorm blocks -- each block is
        then ONE conv launch (the depthwise output never exists) and one data gradient in the backward pass."""
        n, c = x.shape[:2]
        cnt = _dhw(x)
        mode = MODE_BN_TRAIN if training else MODE_BN_EVAL
        comp = wc1 is not None
        red1 = ops.zeros_red(x, n, c) if training else None
        if comp:
            u1 = u2 = None
            t1 = ops.conv3d(x, None, [wc1], [pw1b], k=3, cout=c, epi=2 if training else 0, red=red1)
        else:
            u1 = ops.conv3d(x, None, [dw1], None, k=3, cout=c, groups=c)
            t1 = ops.conv3d(u1, None, [pw1w], [pw1b], k=1, cout=c, epi=2 if training else 0, red=red1)
        red2 = ops.zeros_red(x, n, c) if training else None
        # (one sample, training: the two BatchNorm finalisations ride inside their consumers -- the second dense conv and the tail
        # pass -- instead of two one-workgroup launches; a batch statistic of one sample is its instance statistic)
        bnf = comp and training and n == 1 and c <= 64 and ops.conv3d_fuses_bn(t1, c)
        if bnf:
            t2, sc1, sh1, m1, r1 = ops.conv3d(t1, None, [wc2], [pw2b], k=3, cout=c, in_stats=(red1, cnt, 0.0, (g1, b1, rm1, rv1, steps)),
                                              epi=2, red=red2)
        else:
            sc1, sh1, m1, r1 = ops.norm_finalize(mode, red1, n, c, cnt, gamma=g1, beta=b1, running_mean=rm1, running_var=rv1,
                                                 steps=steps, device=x.device)
            if comp:
                t2 = ops.conv3d(t1, None, [wc2], [pw2b], k=3, cout=c, pre=(sc1, sh1, 0.0), epi=2 if training else 0, red=red2)
            else:
                u2 = ops.conv3d(t1, None, [dw2], None, k=3, cout=c, groups=c, pre=(sc1, sh1, 0.0))
                t2 = ops.conv3d(u2, None, [pw2w], [pw2b], k=1, cout=c, epi=2 if training else 0, red=red2)
        w2 = saw.reshape(2).contiguous()
        if bnf:
            a, sc2, sh2, m2, r2 = ops.skr_tail_bn(t2, x, red2, g2, b2, rm2, rv2, steps, w2)
        else:
            sc2, sh2, m2, r2 = ops.norm_finalize(mode, red2, n, c, cnt, gamma=g2, beta=b2, running_mean=rm2, running_var=rv2,
                                                 steps=steps, device=x.device)
            a = ops.skr_tail(t2, x, sc2, sh2, w2)
        ctx.save_for_backward(x, u1, t1, u2, t2, a, sc1, sh1, m1, r1, sc2, sh2, m2, r2, dw1, pw1w, g1, dw2, pw2w, g2, w2, wc1, wc2)
        ctx.mode = mode
        ctx.params = (dw1, pw1w, pw1b, g1, b1, dw2, pw2w, pw2b, g2, b2)
        ctx.wcs = (wc1, wc2)
        ctx.saw = saw
        ctx.xslot = _slot(x)
        return a

    @staticmethod
    def backward(ctx, da):
        (x, u1, t1, u2, t2, a, sc1, sh1, m1, r1, sc2, sh2, m2, r2, dw1, pw1w, g1, dw2, pw2w, g2, w2, wc1, wc2) = ctx.saved_tensors
        mode = ctx.mode
        comp = wc1 is not None
        n, c = x.shape[:2]
        cnt = _dhw(x)
        (ddw1, dpw1w, dpw1b, dg1, db1, ddw2, dpw2w, dpw2b, dg2, db2), rets = _targets(ctx.params)
        sd = _direct(*rets)
        # the 1x1 attention conv's two weight gradients go straight into the parameter's gradient buffer
        (g_saw,), (r_saw,) = _targets((ctx.saw,))
        # x's gradient has two parts (the residual branch here, the first depthwise conv's data gradient at the end); when x
        # shares a gradient buffer with another consumer (fanout) the residual part is added straight into that buffer
        slot = ctx.xslot
        dtg, dx_res, _ = ops.skr_tail_bwd(t2, x, sc2, sh2, w2, a, _blk(da), dw2_out=g_saw.view(-1), dx_acc=_acc(slot))
        # BatchNorm 2
        red = ops.act_bwd_reduce(dtg, t2, sc2, sh2, 1.0)
        dt2 = ops.norm_bwd_fused(mode, dtg, t2, red, m2, r2, gamma=g2, dgamma=dg2, dbeta=db2)
        red = ops.zeros_red(x, n, c)
        if comp:
            # the composed tensors collect their weight gradients in their own buffers (ComposeAll scatters them to dw / pw at the
            # end of the backward pass); the pointwise biases are the dense convs' biases
            (gwc1, gwc2), rwc = _targets(ctx.wcs)
            sdc = sd and _direct(*rwc)
            ops.conv3d_wgrad(t1, None, dt2, [gwc2], [dpw2b], k=3, pre=(sc1, sh1, 0.0), side=sdc)
            gt1 = ops.conv3d(dt2, None, [wc2], None, k=3, cout=c, transposed=True, epi=1, e=(t1, None, sc1, sh1, 0.0), red=red)
            dt1 = ops.norm_bwd_fused(mode, gt1, t1, red, m1, r1, gamma=g1, dgamma=dg1, dbeta=db1)
            ops.conv3d_wgrad(x, None, dt1, [gwc1], [dpw1b], k=3, side=sdc)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = ops.conv3d(dt1, None, [wc1], None, k=3, cout=c, transposed=True)
                dx = _ret(slot, ops.add(dx_res, dx, out=dx_res))
            return (dx, None, None, None, None, None, None, *rets, r_saw, *rwc)
        # pointwise 2
        ops.conv3d_wgrad(u2, None, dt2, [dpw2w], [dpw2b], k=1, side=sd)
        du2 = ops.conv3d(dt2, None, [pw2w], None, k=1, cout=c, transposed=True)
        # depthwise 2 (input = relu(bn1(t1)))
        ops.conv3d_wgrad(t1, None, du2, [ddw2], None, k=3, groups=c, pre=(sc1, sh1, 0.0), side=sd)
        gt1 = ops.conv3d(du2, None, [dw2], None, k=3, cout=c, groups=c, transposed=True, epi=1, e=(t1, None, sc1, sh1, 0.0), red=red)
        dt1 = ops.norm_bwd_fused(mode, gt1, t1, red, m1, r1, gamma=g1, dgamma=dg1, dbeta=db1)
        # pointwise 1, depthwise 1
        ops.conv3d_wgrad(u1, None, dt1, [dpw1w], [dpw1b], k=1, side=sd)
        du1 = ops.conv3d(dt1, None, [pw1w], None, k=1, cout=c, transposed=True)
        ops.conv3d_wgrad(x, None, du1, [ddw1], None, k=3, groups=c, side=sd)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv3d(du1, None, [dw1], None, k=3, cout=c, groups=c, transposed=True)
            dx = _ret(slot, ops.add(dx_res, dx, out=dx_res))
        return (dx, None, None, None, None, None, None, *rets, r_saw, None, None)


class DuSE(Function):
    """DuSEAttention.forward (modules/DuSFE.py:113-155).  sqw/sqb are the collapsed squeeze+comb 1x1 conv
    (conv_comb o [conv_squeeze_ch1 | conv_squeeze_ch2], linear in [r; s]); adjw/adjb the two 3^3 adjust convs
    stacked on the output axis."""

    @staticmethod
    def forward(ctx, r, s, stats_r, stats_s, training, rm1, rv1, rm2, rv2, wc, bc, w1, b1, w2, b2, sqw, sqb, adjw, adjb, g1, be1,
                g2, be2):
        n, c = r.shape[:2]
        cnt = _dhw(r)
        mode = MODE_BN_TRAIN if training else MODE_BN_EVAL
        # the channel sums: a producer conv's epilogue sums when given, else a moments pass; the backward takes the pooled means
        # duse_fc_fwd leaves in storage of their own (the sums live in the per-forward scratch arena)
        red_r, red_s = stats_r, stats_s
        if red_r is None:
            red_r = ops.zeros_red(r, n, c)
            ops.moments(r, red_r)
        if red_s is None:
            red_s = ops.zeros_red(s, n, c)
            ops.moments(s, red_s)
        fc = dict(wc=wc, bc=bc, w1=w1, b1=b1, w2=w2, b2=b2)
        gvec, ch1, ch2, means = ops.duse_fc_fwd(red_r, red_s, cnt, n, c, fc)
        comb = ops.conv3d(r, s, [sqw], [sqb], k=1, cout=1)
        sp = ops.conv3d(comb, None, [adjw], [adjb], k=3, cout=2, act=ACT_SIGMOID)
        red_ur = ops.zeros_red(r, n, c) if training else None     # BatchNorm sums of the gated outputs, left by the gate pass
        red_us = ops.zeros_red(r, n, c) if training else None
        u_r = ops.duse_gate(r, ch1, sp[:, 0:1], red=red_ur)
        u_s = ops.duse_gate(s, ch2, sp[:, 1:2], red=red_us)
        outs, stats = [], []
        for u, red, gam, bet, rm, rv in ((u_r, red_ur, g1, be1, rm1, rv1), (u_s, red_us, g2, be2, rm2, rv2)):
            y, sc, sh, m, rs = ops.bn_affine_act(mode, u, red, ACT_NONE, gamma=gam, beta=bet, running_mean=rm, running_var=rv, steps=1)
            outs.append(y)
            stats += [sc, sh, m, rs]
        ctx.save_for_backward(r, s, means, gvec, ch1, ch2, comb, sp, u_r, u_s, *stats, wc, w1, w2, sqw, adjw, g1, g2)
        ctx.mode = mode
        ctx.params = (wc, bc, w1, b1, w2, b2, sqw, sqb, adjw, adjb, g1, be1, g2, be2)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, dor, dos):
        (r, s, means, gvec, ch1, ch2, comb, sp, u_r, u_s, sc1, sh1, m1, rs1, sc2, sh2, m2, rs2, wc, w1, w2, sqw, adjw,
         g1, g2) = ctx.saved_tensors
        mode = ctx.mode
        n, c = r.shape[:2]
        cnt = _dhw(r)
        (dwc, dbc, dw1, db1, dw2, db2, dsqw, dsqb, dadjw, dadjb, dg1, dbe1, dg2, dbe2), rets = _targets(ctx.params)
        dus = []
        for do, u, sc, sh, m, rs, gam, dg, db in ((dor, u_r, sc1, sh1, m1, rs1, g1, dg1, dbe1), (dos, u_s, sc2, sh2, m2, rs2, g2, dg2, dbe2)):
            do = _blk(do)
            red = ops.act_bwd_reduce(do, u, sc, sh, 1.0)
            dus.append(ops.norm_bwd_fused(mode, do, u, red, m, rs, gamma=gam, dgamma=dg, dbeta=db))
        dsp = torch.empty_like(sp)
        fused = ops.duse_gate_bwd_fuses(c)       # one pass per stream that also takes dsp through the sigmoid's backward
        dr, dch1 = ops.duse_gate_bwd(r, ch1, sp[:, 0:1], dus[0], dsp[:, 0:1], sigmoid_bwd=fused)
        ds, dch2 = ops.duse_gate_bwd(s, ch2, sp[:, 1:2], dus[1], dsp[:, 1:2], sigmoid_bwd=fused)
        dpre = dsp if fused else ops.act_bwd(dsp, sp, ACT_SIGMOID)
        ops.conv3d_wgrad(comb, None, dpre, [dadjw], [dadjb], k=3, side=_direct(rets[8], rets[9]))
        dcomb = ops.conv3d(dpre, None, [adjw], None, k=3, cout=1, transposed=True)
        ops.conv3d_wgrad(r, s, dcomb, [dsqw], [dsqb], k=1, side=_direct(rets[6], rets[7]))
        fc = dict(wc=wc, w1=w1, w2=w2)
        fcg = dict(wc=dwc, bc=dbc, w1=dw1, b1=db1, w2=dw2, b2=db2)
        dmr, dms = ops.duse_fc_bwd(means, cnt, n, c, fc, gvec, ch1, ch2, dch1, dch2, fcg)
        sq = sqw.reshape(-1)
        ops.rank1_add(dr, dcomb, sq[:c].contiguous(), dmr)
        ops.rank1_add(ds, dcomb, sq[c:].contiguous(), dms)
        return (dr, ds, None, None, None, None, None, None, None, *rets)


class ViL(Function):
    """out = xa + ViLBlock(xa + xb) on flattened patch tokens (RA_HVED.py:626; UxLSTMEnc_3d.py:54-87;
    vision_lstm.py:415-453,494-502).  Always fp32 arithmetic (UxLSTMEnc_3d.py:77-80)."""

    NAMES = ["norm_w", "proj_up", "conv_w", "conv_b", "q_w", "k_w", "v_w", "ig_w", "ig_b", "fg_w", "fg_b", "outnorm_w",
             "skip", "proj_down"]

    @staticmethod
    def forward(ctx, xa, xb, add_xa, *params):
        p = dict(zip(ViL.NAMES, [t.contiguous() for t in params]))
        out, ws = ops.vil_fwd(xa, xb, p, add_xa)
        ctx.save_for_backward(xa, xb, ws, *p.values())
        ctx.add_xa = add_xa
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, dout):
        xa, xb, ws, *params = ctx.saved_tensors
        p = dict(zip(ViL.NAMES, params))
        dout = dout.contiguous()
        bufs, rets = _targets(ctx.params)
        # the kernels index contiguous fp32 storage: a non-contiguous parameter got a zeroed contiguous buffer above
        dxin = ops.vil_bwd(xa, xb, dout, p, ws, dict(zip(ViL.NAMES, bufs)))
        dxa = ops.add(dout, dxin) if ctx.add_xa else dxin
        return (dxa, dxin if xb is not None else None, None, *rets)


class ComposeAtten(Function):
    """AttenModule2's grouped 7^3 conv o 1x1 conv as ONE 7^3 conv's weights (blocks.AttenModule2.composed)."""

    @staticmethod
    def forward(ctx, ns, ne, e, *params):
        params = tuple(t.contiguous() for t in params)
        ctx.cfg = (ns, ne, e)
        ctx.params = params
        ctx.save_for_backward(*params)
        w, b = ops.compose_atten_fwd(params, ns, ne, e)
        k = round(w.shape[-1] ** (1.0 / 3.0))
        return w.view(2, ne, k, k, k), b

    @staticmethod
    def backward(ctx, gw, gb):
        ns, ne, e = ctx.cfg
        grads, rets = _targets(ctx.params)
        ops.compose_atten_bwd(ctx.saved_tensors, ns, ne, e, gw.contiguous(), gb.contiguous(), grads)
        return (None, None, None, *rets)


class ComposeAll(Function):
    """Every parameter composition of a forward in ONE launch, and their backward in one (ops.compose_multi): the AttenModule2
    gates (ComposeAtten), the DuSE blocks (ComposeDuSE) and the segmentation head final_conv o sfinals.  `plan` = (list of
    (ns, ne, e) per AttenModule2, list of c per DuSE block, has_head); params = 8 per AttenModule2, 10 per DuSE block, then
    final_conv.weight (Co, Cm), final_conv.bias, sfinals.weight (Cm, Ci), sfinals.bias, then (dwconv.weight, pwconv.weight) per
    entry of the optional 4th plan element (channel counts of depthwise o pointwise pairs: the skip-return ResBlock's DWConvNorm,
    sa_modules/sa_module.py:79-85).  Returns the flat tuple of composed tensors: (w, b) per AttenModule2, (sqw, sqb, adjw, adjb)
    per DuSE block, (w, b) of the head, one dense (C, C, k, k, k) weight per pair."""
    _store = {}
    _gen = [0]

    @staticmethod
    def _jobs(plan, params, outs, bwd, grads=None, gouts=None):
        a_plan, d_plan, has_head = plan[:3]
        s_plan = plan[3] if len(plan) > 3 else ()
        atten, duse, head, pi, oi = [], [], None, 0, 0
        for ns, ne, e in a_plan:
            j = dict(params=params[pi:pi + 8], ns=ns, ne=ne, e=e)
            if bwd:
                j.update(grads=grads[pi:pi + 8], gw=gouts[oi], gb=gouts[oi + 1])
            else:
                j.update(w=outs[oi], b=outs[oi + 1])
            atten.append(j)
            pi, oi = pi + 8, oi + 2
        for c in d_plan:
            j = dict(params=params[pi:pi + 10], c=c)
            if bwd:
                j.update(grads=grads[pi:pi + 10], gout=gouts[oi:oi + 4])
            else:
                j.update(out=outs[oi:oi + 4])
            duse.append(j)
            pi, oi = pi + 10, oi + 4
        if has_head:
            wf, bf, ws, bs = params[pi:pi + 4]
            head = dict(wf=wf, bf=bf, ws=ws, bs=bs)
            if bwd:
                head.update(dwf=grads[pi], dbf=grads[pi + 1], dws=grads[pi + 2], dbs=grads[pi + 3], gw=gouts[oi], gb=gouts[oi + 1])
            else:
                head.update(w=outs[oi], b=outs[oi + 1])
            pi, oi = pi + 4, oi + 2
        sep = []
        for _c in s_plan:                                   # depthwise o pointwise (the skip-return ResBlock's DWConvNorm pairs)
            q = dict(dw=params[pi], pw=params[pi + 1])
            if bwd:
                q.update(gw=gouts[oi], g_dw=grads[pi], g_pw=grads[pi + 1])
            else:
                q.update(w=outs[oi])
            sep.append(q)
            pi, oi = pi + 2, oi + 1
        return atten, duse, head, sep

    @staticmethod
    def forward(ctx, plan, *params):
        params = tuple(t.contiguous() for t in params)
        a_plan, d_plan, has_head = plan[:3]
        s_plan = plan[3] if len(plan) > 3 else ()
        dev = params[0].device
        # The composed tensors live in PERSISTENT storage per (plan, parameter storages): the same addresses every step, so the
        # k = 3 convs that use them (the skip-return ResBlocks' dense convs) keep their packed-fragment workspaces and ride in
        # ops.prepack_all() like every leaf weight -- fresh tensors per step meant a pack launch in front of each of those convs.
        # What is returned are new aliases of that storage (a step's autograd graph never sees another step's tensor objects).
        ckey = (repr(plan), tuple(t.data_ptr() for t in params))
        store = ComposeAll._store.get(ckey)
        fresh = store is None
        if fresh:
            if len(ComposeAll._store) > 16:
                ComposeAll._store.clear()
            store = ComposeAll._store[ckey] = []
        it = iter(store)
        ComposeAll._gen[0] += 1
        gen = ComposeAll._gen[0]

        def new(*shape):
            if fresh:
                store.append(torch.empty(shape, dtype=torch.float32, device=dev))
            base = store[-1] if fresh else next(it)
            alias = base.view(shape)
            alias._xh_base = base                          # ops._pack_entry keys a conv's packed fragments on the storage ...
            alias._xh_gen = base._xh_gen = gen             # ... and tells a stale pack from a fresh one by this stamp (ops._wversion)
            return alias
        outs, pi = [], 0
        for ns, ne, e in a_plan:
            k = params[pi].shape[-1]
            outs += [new(2, ne, k, k, k), new(2)]
            pi += 8
        for c in d_plan:
            outs += [new(1, 2 * c, 1, 1, 1), new(1), new(2, 1, 3, 3, 3), new(2)]
            pi += 10
        if has_head:
            wf, _, ws, _ = params[pi:pi + 4]
            outs += [new(wf.shape[0], ws.shape[1], 1, 1, 1), new(wf.shape[0])]
            pi += 4
        for _c in s_plan:
            dw_ = params[pi]
            outs.append(new(dw_.shape[0], dw_.shape[0], *dw_.shape[2:]))
            pi += 2
        # (the gradient buffers below are cleared by the same launch)
        sizes = [(o.numel() + 15) // 16 * 16 for o in outs]
        # two copies of the gradient buffers (both cleared here): a SECOND backward over this forward (retain_graph=True, two
        # losses sharing one generator forward) must not see the first pass's sums again, so backward() moves the composed
        # tensors on to the clean copy once it has consumed the first (a third pass pays one fill)
        total = sum(sizes)
        flat = torch.empty(2 * total, dtype=torch.float32, device=dev) if any(ctx.needs_input_grad) else None
        at_, du_, he_, se_ = ComposeAll._jobs(plan, params, outs, False)
        ops.compose_multi(False, at_, du_, he_, zero=flat, sep=se_)
        ctx.plan, ctx.params = plan, params
        ctx.save_for_backward(*params)
        ctx.out_meta = [tuple(o.shape) for o in outs]
        ctx.set_materialize_grads(False)
        # One zeroed fp32 gradient buffer per composed tensor, alive for the step: the convs that use the tensor accumulate
        # their weight gradients into it (as they do into a leaf's .grad) and hand autograd nothing, so those launches can wait
        # for the end-of-backward batch however many times the tensor is used (autograd would sum per-use gradients the
        # moment the second one is returned -- before a deferred launch has written it).  Private storage, not the per-step
        # arena: a second forward before this one's backward would be handed the same arena slices.
        ctx.gbufs = None
        if flat is not None:
            import weakref
            ctx.gsets, ctx.gflat, ctx.gcur, ctx.passes = [[], []], (flat[:total], flat[total:]), 0, 0
            off = 0
            for o, n_ in zip(outs, sizes):
                for half in (0, 1):
                    ctx.gsets[half].append(flat[half * total + off:half * total + off + o.numel()].view(o.shape))
                o._xh_gbuf = ctx.gsets[0][-1]
                off += n_
            ctx.gbufs = ctx.gsets[0]
            ctx.out_refs = [weakref.ref(o) for o in outs]      # (weak: the outputs own this node)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        params = ctx.saved_tensors
        dev = params[0].device
        ops.join_wgrad_stream()          # the composed tensors' weight gradients may still be queued (functional._direct)
        # the step-long buffers hold what the package's own convs accumulated; a consumer that returned a gradient through
        # autograd instead (a stock op on a composed tensor) is added on top
        gouts = [buf if g is None else buf.add_(g.reshape(buf.shape)) for g, buf in zip(gouts, ctx.gbufs)]
        grads, rets = _targets(ctx.params)
        at_, du_, he_, se_ = ComposeAll._jobs(ctx.plan, params, None, True, grads=grads, gouts=gouts)
        ops.compose_multi(True, at_, du_, he_, sep=se_)
        # this pass's sums are spent: a later backward over the same forward accumulates into the other (clean) copy
        ctx.passes += 1
        ctx.gcur ^= 1
        if ctx.passes >= 2:
            ctx.gflat[ctx.gcur].zero_()                      # third and later passes: the copy being returned to is dirty
        ctx.gbufs = ctx.gsets[ctx.gcur]
        for r, buf in zip(ctx.out_refs, ctx.gbufs):
            o = r()
            if o is not None:
                o._xh_gbuf = buf
        return (None, *rets)


class ComposeDuSE(Function):
    """DuSEAttention's conv_comb o conv_squeeze_ch{1,2} and stacked adjust convs (blocks.DuSEAttention.composed)."""

    @staticmethod
    def forward(ctx, c, *params):
        params = tuple(t.contiguous() for t in params)
        ctx.c = c
        ctx.params = params
        ctx.save_for_backward(*params)
        return ops.compose_duse_fwd(params, c)

    @staticmethod
    def backward(ctx, dsqw, dsqb, dadjw, dadjb):
        grads, rets = _targets(ctx.params)
        ops.compose_duse_bwd(ctx.saved_tensors, ctx.c, dsqw.contiguous(), dsqb.contiguous(), dadjw.contiguous(),
                             dadjb.contiguous(), grads)
        return (None, *rets)


# ------------------------------------------------------------------------------------------------------
# The recon | seg PAIR of the shared decoder (RA_HVED.py:158-201).  At every decoder level the reconstruction stream and the
# segmentation stream run a DoubleConv of identical shape and then meet in DuSEAttention; with one sample per launch (N == 1) a
# (1, 2C, D, H, W) tensor holding [recon | seg] is, byte for byte, a (2, C, D, H, W) batch, so the existing kernels serve the
# pair in ONE launch each: the second convs as a grouped conv with one weight pointer per stream, the DuSE gates / their
# backward / the trilinear resampling as batch-2 or 2C-channel launches, the two BatchNorm modules through the paired-parameter
# entry points (xh_bn_affine_act2 / xh_norm_bwd_fused2).  Same arithmetic per element as the one-stream launches.
class Upsample2(Function):
    """F.interpolate(trilinear) of the pair in one launch; returns the two halves (views of one tensor).  Their gradients are
    collected side by side through a TwinDest, so the adjoint is one launch as well."""

    @staticmethod
    def forward(ctx, p, size):
        c = p.shape[1] // 2
        u = ops.upsample(p, tuple(size))
        ctx.in_size = tuple(p.shape[2:])
        ctx.dest = TwinDest(u.shape, c, u.dtype, u.device)
        return u[:, :c], u[:, c:]

    @staticmethod
    def backward(ctx, ga, gb):
        return ops.upsample_bwd(ctx.dest.joined(ga, gb), ctx.in_size), None


def upsample2(p, size):
    a, b = Upsample2.apply(p, tuple(size))
    dest = a.grad_fn.dest if a.grad_fn is not None and hasattr(a.grad_fn, "dest") else None
    if dest is not None:
        a._xh_dest, b._xh_dest = (dest, 0), (dest, 1)
    return a, b


class Split2(Function):
    """Channels [0, c) and [c, C) of a tensor as separate tensors (the pair's halves for the heads; [X | S] of the five-stream
    encoder's last level); their gradients rejoin through a TwinDest."""

    @staticmethod
    def forward(ctx, p, c):
        ctx.dest = TwinDest(p.shape, c, p.dtype, p.device)
        return p[:, :c], p[:, c:]

    @staticmethod
    def backward(ctx, ga, gb):
        return ctx.dest.joined(ga, gb), None


def split2(p):
    return split_at(p, p.shape[1] // 2)


def split_at(p, c):
    a, b = Split2.apply(p, int(c))
    dest = a.grad_fn.dest if a.grad_fn is not None and hasattr(a.grad_fn, "dest") else None
    if dest is not None:
        a._xh_dest, b._xh_dest = (dest, 0), (dest, 1)
    return a, b


class InLreluConv2(Function):
    """The second SingleConv 'ilc' of BOTH decoder streams (buildingblocks.py:464-507 twice) as one grouped launch: ya | yb are
    the halves of `base` (1, 2C, ...) -- written there by the streams' first convs (in_lrelu_conv(into=...)) --, red_in their
    channel sums (1, 2C, 2); weights / biases per stream.  Returns (y (1, 2Co, ...), its channel sums)."""

    @staticmethod
    def forward(ctx, ya, yb, base, red_in, *wb):
        wa, wb_, ba, bb = wb
        n, c2 = base.shape[:2]
        cout = wa.shape[0] + wb_.shape[0]
        k = wa.shape[-1]
        red_y = ops.zeros_red(base, n, cout)
        y, sc, sh, mean, rstd = ops.conv3d(base, None, [wa, wb_], [ba, bb], k=k, cout=cout, groups=2, in_stats=(red_in, _dhw(base), LEAK),
                                           epi=2, red=red_y)
        ctx.save_for_backward(base, sc, sh, mean, rstd, wa, wb_)
        ctx.params = ([wa, wb_], [ba, bb])
        ctx.k = k
        ctx.mark_non_differentiable(red_y)
        ctx.set_materialize_grads(False)
        return y, red_y

    @staticmethod
    def backward(ctx, dy, _dred=None):
        base, sc, sh, mean, rstd, wa, wb_ = ctx.saved_tensors
        k = ctx.k
        n, c2 = base.shape[:2]
        c = c2 // 2
        dy = _blk(dy)
        dws, rws = _targets(ctx.params[0])
        dbs, rbs = _targets(ctx.params[1])
        ops.conv3d_wgrad(base, None, dy, dws, dbs, k=k, groups=2, pre=(sc, sh, LEAK), side=_direct(*rws, *rbs))
        red = ops.zeros_red(base, n, c2)
        g = ops.conv3d(dy, None, [wa, wb_], None, k=k, cout=c2, groups=2, transposed=True, epi=1, e=(base, None, sc, sh, LEAK), red=red)
        dbase = torch.empty_like(base, memory_format=torch.contiguous_format)
        halves = [dbase[:, :c], dbase[:, c:]]
        if ops._NB_FOLD[0] and n == 1 and c * _dhw(base) <= NB_FOLD_MAX:
            # each stream's first conv takes its half over unwritten (the norm-backward fold, _NB_PENDING)
            for i, h in enumerate(halves):
                sl = slice(i * c, (i + 1) * c)
                _NB_PENDING[h.data_ptr()] = (g[:, sl], base[:, sl], red[:, sl], mean[:, sl], rstd[:, sl], h)
        else:
            ops.in_bwd_apply(g, base, red, mean, rstd, have_g=True, out=dbase)
        return (halves[0], halves[1], None, None, *rws, *rbs)


class DuSE2(Function):
    """DuSEAttention.forward (modules/DuSFE.py:113-155) on the recon | seg pair y = [r | s] (1, 2C, ...): the two gate passes,
    the two BatchNorm passes and their backward counterparts as ONE launch each (see the section comment); the squeeze conv
    reads the pair as its 2C input channels.  Returns the pair [out_r | out_s]."""

    @staticmethod
    def forward(ctx, y, stats, training, rm1, rv1, rm2, rv2, wc, bc, w1, b1, w2, b2, sqw, sqb, adjw, adjb, g1, be1, g2, be2):
        n, c2 = y.shape[:2]
        c = c2 // 2
        sp_shape = tuple(y.shape[2:])
        cnt = _dhw(y)
        mode = MODE_BN_TRAIN if training else MODE_BN_EVAL
        red_r, red_s = stats[:, :c], stats[:, c:]
        fc = dict(wc=wc, bc=bc, w1=w1, b1=b1, w2=w2, b2=b2)
        comb = ops.conv3d(y, None, [sqw], [sqb], k=1, cout=1)
        sp = ops.conv3d(comb, None, [adjw], [adjb], k=3, cout=2, act=ACT_SIGMOID)
        red_u = ops.zeros_red(y, 2, c) if training else None
        if ops.FC_FOLD[0] and n == 1 and c <= 32:
            # the channel excitation's dense layers inside the gate pass (every workgroup derives its own channel's gate)
            u, chb, gvec, means = ops.duse_gate_fc(y, sp, stats, fc, red=red_u)
        else:
            chb = torch.empty((2, c), dtype=torch.float32, device=y.device)
            gvec, ch1, ch2, means = ops.duse_fc_fwd(red_r, red_s, cnt, n, c, fc, ch_out=chb)
            u = ops.duse_gate(y.view((2, c) + sp_shape), chb, sp.view((2, 1) + sp_shape), red=red_u).view(y.shape)
        out, sc, sh, m, rs = ops.bn_affine_act2(mode, u, red_u.view(1, c2, 2) if red_u is not None else None, ACT_NONE, c, gammas=(g1, g2),
                                               betas=(be1, be2), running_means=(rm1, rm2), running_vars=(rv1, rv2), steps=1)
        ctx.save_for_backward(y, means, gvec, chb, comb, sp, u, sc, sh, m, rs, wc, w1, w2, sqw, adjw, g1, g2)
        ctx.mode = mode
        ctx.params = (wc, bc, w1, b1, w2, b2, sqw, sqb, adjw, adjb, g1, be1, g2, be2)
        return out

    @staticmethod
    def backward(ctx, dout):
        (y, means, gvec, chb, comb, sp, u, sc, sh, m, rs, wc, w1, w2, sqw, adjw, g1, g2) = ctx.saved_tensors
        mode = ctx.mode
        n, c2 = y.shape[:2]
        c = c2 // 2
        sp_shape = tuple(y.shape[2:])
        cnt = _dhw(y)
        (dwc, dbc, dw1, db1, dw2, db2, dsqw, dsqb, dadjw, dadjb, dg1, dbe1, dg2, dbe2), rets = _targets(ctx.params)
        dout = _blk(dout)
        red = ops.act_bwd_reduce(dout, u, sc, sh, 1.0)
        du = ops.norm_bwd_fused2(mode, dout, u, red, m, rs, c, gammas=(g1, g2), dgammas=(dg1, dg2), dbetas=(dbe1, dbe2))
        dsp = torch.empty_like(sp)
        fused = ops.duse_gate_bwd_fuses(c)
        dx, dch = ops.duse_gate_bwd(y.view((2, c) + sp_shape), chb, sp.view((2, 1) + sp_shape), du.view((2, c) + sp_shape),
                                    dsp.view((2, 1) + sp_shape), sigmoid_bwd=fused)
        dpre = dsp if fused else ops.act_bwd(dsp, sp, ACT_SIGMOID)
        ops.conv3d_wgrad(comb, None, dpre, [dadjw], [dadjb], k=3, side=_direct(rets[8], rets[9]))
        dcomb = ops.conv3d(dpre, None, [adjw], None, k=3, cout=1, transposed=True)
        ops.conv3d_wgrad(y, None, dcomb, [dsqw], [dsqb], k=1, side=_direct(rets[6], rets[7]))
        fc = dict(wc=wc, w1=w1, w2=w2)
        fcg = dict(wc=dwc, bc=dbc, w1=dw1, b1=db1, w2=dw2, b2=db2)
        dx = dx.view(y.shape)
        if ops.FC_FOLD[0] and n == 1 and c <= 32:
            ops.rank1_add_fc(dx, dcomb, sqw.reshape(-1).contiguous(), means, gvec, chb, dch, fc, fcg)
        else:
            dm = torch.empty((1, c2), dtype=torch.float32, device=y.device)
            ops.duse_fc_bwd(means, cnt, n, c, fc, gvec, chb[0:1], chb[1:2], dch[0:1], dch[1:2], fcg, dm_out=dm)
            ops.rank1_add(dx, dcomb, sqw.reshape(-1).contiguous(), dm)
        return (dx, None, None, None, None, None, None, *rets)

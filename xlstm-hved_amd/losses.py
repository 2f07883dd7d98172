"""Loss and metric epilogues of the training step on the HIP path: same names and call signatures as
the reference's loss.py / metrics.py objects that train.py:171-175,232-262,288-296 uses, so the step reads the same.

Each loss is ONE reduction pass over its tensors (fp64 sums) + a one-workgroup finalisation on the device; its backward
is one linear-combination pass scaled by the upstream gradient, which stays on the device (no .item(), no chain of
full-resolution ATen temporaries).  Predictions may be in a 16-bit storage type while targets are fp32."""
import torch
from torch.autograd import Function

from . import ops
from .model import SUBSETS_MODALITIES


class _Dice(Function):
    @staticmethod
    def forward(ctx, p, t, eps):
        red = ops.pair_sums(p, t)
        loss, ca, cb = ops.loss_finalize(0, red, eps=eps)
        ctx.save_for_backward(p, t, ca, cb)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        p, t, ca, cb = ctx.saved_tensors
        return ops.lincomb(p, t, ca, cb, gscale=g.float().reshape(1).contiguous()), None, None


class DiceLoss(torch.nn.Module):
    """loss.py:188-209 + compute_per_channel_dice (loss.py:257-285): 1 - mean_c 2 sum(p t) / max(sum p^2 + sum t^2, 1e-6),
    sums over batch and space; the input is already a probability (the reference applies no normalisation)."""

    def __init__(self, weight=None, epsilon=1e-6):
        super().__init__()
        if weight is not None:
            raise NotImplementedError("per-class weights are not used by train.py")
        self.epsilon = epsilon

    def forward(self, input, target):
        if input.shape != target.shape:
            raise ValueError("'input' and 'target' must have the same shape")
        return _Dice.apply(input, _as_target(target, input), self.epsilon)


def _as_target(t, like):
    """Targets are consumed as fp32 or in the prediction's dtype (xh_pair_sums reads either)."""
    if t.dtype == like.dtype or t.dtype == torch.float32:
        return t.contiguous()
    return t.float().contiguous()


class _MSE(Function):
    @staticmethod
    def forward(ctx, a, b, bval, scale=1.0):
        red = ops.pair_sums(a, b, bval=bval)
        # scale * mean: the divisor carries the weight, so the loss AND the coefficients of its gradient come out scaled
        loss, ca, cb = ops.loss_finalize(1, red, count=a.numel() / float(scale))
        ctx.save_for_backward(a, b, ca, cb)
        ctx.bval = bval
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b, ca, cb = ctx.saved_tensors
        gs = g.float().reshape(1).contiguous()
        da = ops.lincomb(a, b, ca, cb, bval=ctx.bval, gscale=gs)
        db = None
        if b is not None and ctx.needs_input_grad[1]:
            db = ops.lincomb(b, a, ca, cb, gscale=gs) if b.dtype == a.dtype else None
        return da, db, None, None


def mse_loss(a, b):
    """nn.MSELoss() of train.py:173,234: mean (a - b)^2.  b (the input patch) may be fp32 next to a 16-bit reconstruction."""
    return _MSE.apply(a, _as_target(b, a), 0.0)


class MSELoss(torch.nn.Module):
    def forward(self, input, target):
        return mse_loss(input, target)


class GANLoss(torch.nn.Module):
    """loss.py:167-186 with use_lsgan=True (train.py:172): mean (d - label)^2 against a constant label."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        if not use_lsgan:
            raise NotImplementedError("train.py uses the least-squares GAN loss")
        self.real_label, self.fake_label = float(target_real_label), float(target_fake_label)

    def forward(self, input, target_is_real):
        return _MSE.apply(input, None, self.real_label if target_is_real else self.fake_label)


_LABELS = {}


def gan_pair_loss(out, nb, scale=1.0, fake_label=0.0, real_label=1.0):
    """scale * 0.5 * (GANLoss(out[:nb], False) + GANLoss(out[nb:], True)) (train.py:272-280, least squares) for a batch that
    holds the fake samples' predictions in its first nb rows and the real samples' in the other nb: ONE reduction against a
    resident label tensor and one gradient pass over the whole batch, instead of two losses on slices whose backward is two
    zero-fills, two copies and an add before the discriminator's backward can start."""
    if out.shape[0] != 2 * nb:
        raise ValueError("gan_pair_loss takes the predictions of nb fake and nb real samples")
    key = (tuple(out.shape), out.device, nb, float(fake_label), float(real_label))
    lab = _LABELS.get(key)
    if lab is None:                       # built on the first (eager) call: nothing is created inside a captured step
        lab = torch.empty(out.shape, dtype=torch.float32, device=out.device)
        lab[:nb] = fake_label
        lab[nb:] = real_label
        _LABELS[key] = lab
    return _MSE.apply(out.contiguous(), lab, 0.0, float(scale))


class _Combine(Function):
    """sum_i coefs[i] * terms[i] over device scalars: one launch forward, one backward (ops.scalar_lincomb / scalar_fanout)."""

    @staticmethod
    def forward(ctx, coefs, *terms):
        ctx.coefs = tuple(float(c) for c in coefs)
        ctx.meta = [(t.dtype, tuple(t.shape)) for t in terms]
        return ops.scalar_lincomb([t.detach() for t in terms], ctx.coefs).reshape(())

    @staticmethod
    def backward(ctx, g):
        o32, o64 = ops.scalar_fanout(ctx.coefs, g.float().reshape(1).contiguous())
        return (None,) + tuple((o64 if dt == torch.float64 else o32)[i].reshape(shape) if ctx.needs_input_grad[i + 1] else None
                               for i, (dt, shape) in enumerate(ctx.meta))


def combine(terms, coefs):
    """The weighted sum of loss terms (train.py:262: dice + m_dice + beta * recon + beta * kld + alpha * g_gan) as ONE node."""
    terms = [t if t.dtype in (torch.float32, torch.float64) else t.float() for t in terms]
    return _Combine.apply(tuple(coefs), *terms)


class _KLDLevels(Function):
    """mean over the latent levels of compute_KLD (train.py:235-239) for one keep mask: the levels' reductions land in one zeroed
    block, one launch turns them into the mean; backward is one pass per level scaled by the upstream gradient."""

    @staticmethod
    def forward(ctx, keep, *stacks):
        nl = len(stacks) // 2
        mus, lvs = [m.contiguous() for m in stacks[:nl]], [v.contiguous() for v in stacks[nl:]]
        red = ops.zeros_f64(keep.device, (nl * 16,))                  # one 128-byte line per level: the levels' atomics do not meet
        ctx.scales = []
        for l in range(nl):
            n, _, L_, d, h, w = mus[l].shape
            ctx.scales.append(0.5 / (n * L_ * d * h * w) / nl)
            ops.kld_fwd(mus[l], lvs[l], keep, red=red[16 * l:16 * l + 1])
        ctx.save_for_backward(keep, *mus, *lvs)
        return ops.scalar_lincomb([red[16 * l:16 * l + 1] for l in range(nl)], ctx.scales).reshape(())

    @staticmethod
    def backward(ctx, g):
        keep, *stacks = ctx.saved_tensors
        nl = len(stacks) // 2
        gs = g.float().reshape(1).contiguous()
        dmus, dlvs = [], []
        for l in range(nl):
            dmu, dlv = ops.kld_bwd(stacks[l], stacks[nl + l], keep, ctx.scales[l], gscale=gs)
            dmus.append(dmu); dlvs.append(dlv)
        return (None,) + tuple(dmus) + tuple(dlvs)


def compute_KLD_levels(mu_levels, logvar_levels, subset_index_list=(14,)):
    """sum_l compute_KLD(mu[l], logvar[l], subset) / len(mu) of train.py:235-239 as one node (a device keep mask, or a list of
    subset indices averaged like loss.py:85-115)."""
    if torch.is_tensor(subset_index_list):
        return _KLDLevels.apply(subset_index_list.contiguous(), *mu_levels, *logvar_levels)
    total = None
    for idx in subset_index_list:
        k = _KLDLevels.apply(_keep_of(int(idx), mu_levels[0].shape[0], mu_levels[0].device), *mu_levels, *logvar_levels)
        total = k if total is None else total + k
    return total if len(subset_index_list) == 1 else total / len(subset_index_list)


class _KLD(Function):
    @staticmethod
    def forward(ctx, mu, lv, keep):
        mu, lv = mu.contiguous(), lv.contiguous()
        red = ops.kld_fwd(mu, lv, keep)
        n, _, L_, d, h, w = mu.shape
        ctx.count = n * L_ * d * h * w
        ctx.save_for_backward(mu, lv, keep)
        return (red * (0.5 / ctx.count)).float().reshape(())

    @staticmethod
    def backward(ctx, g):
        mu, lv, keep = ctx.saved_tensors
        dmu, dlv = ops.kld_bwd(mu, lv, keep, 0.5 / ctx.count, gscale=g.float().reshape(1).contiguous())
        return dmu, dlv, None


_KEEP = {}


def _keep_of(idx, n, device):
    key = (idx, n, device)
    keep = _KEEP.get(key)
    if keep is None:                    # built once per (subset, batch, device): no host-to-device copy inside a captured step
        keep = _KEEP[key] = torch.tensor([[1.0 if k in SUBSETS_MODALITIES[idx] else 0.0 for k in range(4)]] * n,
                                         dtype=torch.float32, device=device)
    return keep


def compute_KLD(mu_list, logvar_list, subset_index_list=(14,), choices=(0, 1, 2, 3)):
    """loss.py:85-115 on the (B, 5, L, d, h, w) stacks the model returns per level: for every subset index in the list,
    KL(PoE(prior + subset) || prior), averaged over the list."""
    if torch.is_tensor(subset_index_list):           # an (N, 4) device keep mask (TrainStep under a captured graph)
        return _KLD.apply(mu_list, logvar_list, subset_index_list.contiguous())
    total = None
    for idx in subset_index_list:
        k = _KLD.apply(mu_list, logvar_list, _keep_of(int(idx), mu_list.shape[0], mu_list.device))
        total = k if total is None else total + k
    return total / len(subset_index_list)


def nested_attention(seg, syn):
    """train.py:242-259: syn * (1 + w) with w the nested tumour-region weight map of the DETACHED segmentation output
    (w = p_WT where > .5, overridden by p_TC, p_ET where those exceed .5).  The gradient reaches `syn` only."""
    from . import functional as Fn
    w = ops.nested_weight(seg.detach().contiguous())
    return Fn.Gate.apply(syn, w)


class _Mean(Function):
    @staticmethod
    def forward(ctx, t):
        v = t if t.dim() == 5 else t.reshape(t.shape[0], -1, *t.shape[-3:])       # (N,5,L,d,h,w) stacks -> (N,5L,d,h,w)
        v = v.contiguous()
        red = ops.pair_sums(v)
        ctx.meta = (tuple(t.shape), t.dtype, t.device, t.numel())
        return ops.loss_finalize(3, red, count=t.numel()).reshape(())

    @staticmethod
    def backward(ctx, g):
        shape, dtype, device, numel = ctx.meta
        return ops.fill(shape, 1.0 / numel, dtype, device, gscale=g.float().reshape(1).contiguous())


def mean_of(t):
    """t.float().mean() as one reduction pass (+ one fill in backward): the terms of the SURVEY 8(d) benchmark loss."""
    return _Mean.apply(t)


class _SumOfMeans(Function):
    @staticmethod
    def forward(ctx, *ts):
        dev = ts[0].device
        # a reduction row is summed by at most 64 workgroups (one atomic address): a 128^3 output as ONE row took 47 us.  Large
        # tensors are cut into 16 rows that all carry the tensor's weight 1 / numel
        splits = [16 if t.numel() >= (1 << 20) and t.numel() % 128 == 0 else 1 for t in ts]
        key = (tuple(t.numel() for t in ts), dev)
        w = _WEIGHTS.get(key)
        if w is None:
            w = _WEIGHTS[key] = torch.tensor([1.0 / t.numel() for t, sp in zip(ts, splits) for _ in range(sp)], dtype=torch.float32,
                                             device=dev)
        red = ops.zeros_f64(dev, (sum(splits), 1, 6))
        ctx.meta = [(tuple(t.shape), t.dtype, t.device, t.numel()) for t in ts]
        ctx.multi = len(ts) <= 16 and all(t.dtype == ts[0].dtype for t in ts)
        if ctx.multi:
            # all tensors in ONE launch (twenty launches per step -- ten sums, ten gradient fills -- were mostly launch latency)
            row0, r0 = [], 0
            for sp in splits:
                row0.append(r0)
                r0 += sp
            ops.multi_sum([t.contiguous() for t in ts], red, row0, splits)
        else:
            r0 = 0
            for t, sp in zip(ts, splits):
                ops.pair_sums(t.contiguous().view(1, sp, 1, 1, -1), red=red[r0:r0 + sp].view(1, sp, 6))
                r0 += sp
        return ops.loss_finalize(4, red, count=w).reshape(())

    @staticmethod
    def backward(ctx, g):
        gs = g.float().reshape(1).contiguous()
        if ctx.multi:
            return tuple(ops.multi_fill([(shape, dtype, device) for shape, dtype, device, _ in ctx.meta],
                                        [1.0 / numel for _, _, _, numel in ctx.meta], gscale=gs))
        return tuple(ops.fill(shape, 1.0 / numel, dtype, device, gscale=gs) for shape, dtype, device, numel in ctx.meta)


_WEIGHTS = {}


def sum_of_means(tensors):
    """sum_i t_i.float().mean(): one reduction pass per tensor and ONE finalisation (the SURVEY 8(d) benchmark loss)."""
    return _SumOfMeans.apply(*tensors)


class DiceCoefficient:
    """metrics.py:10-48: mean over channels of the thresholded (> 0.5) per-channel Dice, averaged over the batch."""

    def __init__(self, epsilon=1e-6, **kwargs):
        self.epsilon = epsilon

    def per_channel(self, input, target):
        red = ops.pair_sums(input.detach(), _as_target(target, input), thr=0.5)
        return ops.loss_finalize(2, red, eps=self.epsilon)

    def __call__(self, input, target):
        return self.per_channel(input, target).mean()


class DiceRegion(DiceCoefficient):
    """metrics.py:51-107, mode='sigmoid': the thresholded Dice of one nested region channel (WT / TC / EC)."""

    def __call__(self, input, target, region="WT", mode="sigmoid", epsilon=1e-6):
        if mode != "sigmoid":
            raise NotImplementedError("XLSTM_HVED emits sigmoid region maps (SURVEY F7)")
        return self.per_channel(input, target)[{"WT": 0, "TC": 1, "EC": 2}[region]]

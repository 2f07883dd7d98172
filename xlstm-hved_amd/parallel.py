"""Data-parallel exchange for the hot path: one process per GPU, one flat gradient all-reduce per step.

The reference only has single-process nn.DataParallel (train.py:148-151).  Patches are independent units, so the
path shards by data; the only exchange is the gradient sum.  The generator has 422 588 parameters = 1.69 MB fp32:
a single latency-bound RCCL all-reduce over xGMI per step, issued on the compute stream after backward (it can be
captured in the same hipGraph as the step).  96 366 parameters never receive a gradient in the reference (dead
modules); they contribute zeros instead of requiring find_unused_parameters.
BatchNorm statistics and the reparameterisation noise stay per rank, like the reference's DataParallel replicas."""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, params, world_size=None, group=None):
        self.params = list(params)
        self.group = group
        self.world = world_size if world_size is not None else dist.get_world_size(group)
        p0 = self.params[0]
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=p0.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def pack(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)

    def unpack(self):
        for p, v in zip(self.params, self.views):
            if p.grad is not None:
                p.grad.copy_(v)

    def __call__(self):
        """Averages .grad of every parameter that has one across ranks (in place)."""
        self.pack()
        dist.all_reduce(self.flat, group=self.group)
        self.flat.div_(self.world)
        self.unpack()


class FlatGrads:
    """One flat fp32 bucket holding every parameter's gradient; `p.grad` are views into it.

    The weight-gradient kernels accumulate directly into an existing `.grad` (functional._targets), so with this
    bucket a step needs ONE zero-fill instead of one per parameter tensor, autograd performs no AccumulateGrad adds,
    and the data-parallel exchange is a single in-place all-reduce of the bucket (no pack/unpack copies).
    Parameters the network never reaches simply keep zero gradients."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        p0 = self.params[0]
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=p0.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.attach()

    def attach(self):
        """(Re-)points every p.grad at its view of the bucket.  A gradient that something else has replaced in the
        meantime (optimizer.zero_grad() defaults to set_to_none=True, train.py:264 -> backward then allocates fresh
        per-parameter gradients) is copied into the bucket first, so nothing is lost."""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()                      # no gradient this step = a zero contribution (not last step's values)
                p.grad = v
            elif g.data_ptr() != v.data_ptr() or g.shape != v.shape or g.dtype != torch.float32:
                v.copy_(g)
                p.grad = v

    def check(self):
        """True when every p.grad still aliases the bucket."""
        return all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params, self.views))

    def zero(self):
        """Use this instead of optimizer.zero_grad(): one fill, and the views stay attached (re-attached if lost)."""
        from . import ops
        ops.join_wgrad_stream()
        if not self.check():
            for p in self.params:                      # drop foreign gradients: zero() means zero
                p.grad = None
            self.attach()
        self.flat.zero_()

    def all_reduce(self, world_size=None, group=None):
        """Averages the bucket across ranks in place.  Gradients that no longer alias the bucket (see attach) are
        gathered into it first -- reducing a stale bucket would silently stop synchronising the ranks."""
        if not self.check():
            self.attach()
        from . import ops
        ops.join_wgrad_stream()            # weight-gradient kernels on the side stream (ops.set_wgrad_overlap) finish first
        world = world_size if world_size is not None else dist.get_world_size(group)
        dist.all_reduce(self.flat, group=group)
        self.flat.div_(world)


def shard_windows(n_items, rank, world):
    """Round-robin assignment of independent work items (sliding-window tiles, evaluation.py:311-333) to ranks."""
    return list(range(rank, n_items, world))

"""One optimisation step of the reference's training loop (train.py:208-296) on the HIP path.

    generator:      f_out, f_rec = G(x, [14]);  m_out, (mu, logvar), m_rec = G(x, subset)          train.py:224-225
                    loss = dice(f_out) + dice(m_out) + beta * mse(m_rec, x) + beta * mean_l KLD_l
                           + alpha * lsgan(D(cat[m_out, m_rec * (1 + w(m_out))]), real)             train.py:232-262
    discriminator:  loss_d = alpha/2 * (lsgan(D(fake.detach()), fake) + lsgan(D(real.detach()), real))   train.py:270-280

What is specific to this implementation:
  * the two generator forwards share everything that depends on the input alone (init blocks, encoders, DRBs, skip-return
    path: model.forward_shared) -- their mu / logvar stacks are bit-identical in the reference too;
  * every loss term is a HIP reduction (losses.py), the upstream gradients stay on the device;
  * AMP policy: activations in `storage` (fp16 like the reference's autocast, bf16, or fp32), fp32 arithmetic, the ViL block
    and all parameters / gradients fp32; with fp16 storage the loss is scaled by a DEVICE-resident loss scale (GradScaler's
    initial 65536, train.py:207) and the fp32 gradients are unscaled before the optimizer.  `step()` keeps GradScaler's
    books (train.py:265-268,283-285): each optimizer is skipped on its own when its gradients are not finite, the scale is
    halved after a skipped step and doubled after `growth_interval` (2000) clean ones; because the scale is a device scalar a
    captured hipGraph follows every update;
  * the modality subset (train.py:222-223 draws a new one every step) enters as a DEVICE mask (N, 4) that `step()` /
    `replay()` overwrite before the launch: ONE captured hipGraph serves all 15 subsets (`capture()`);
  * data parallel (`group=`): one process per GPU, patches sharded by data (the reference only has nn.DataParallel,
    train.py:148-151).  The generator's flat gradient bucket (1.69 MB fp32) is all-reduced right after the generator's
    backward ON A COMMUNICATION STREAM, so the collective runs under the discriminator passes; the discriminator's bucket
    (44.3 MB fp32) is all-reduced after its backward; both before the optimizers step.  With a group the captured step is TWO
    hipGraphs split at that point (one memory pool), the collectives are issued eagerly between / after them;
  * GradScaler bookkeeping differs from train.py:265-285 in one documented way: the reference calls scaler.update() between
    the generator's and the discriminator's pass, so after a generator overflow its discriminator loss is already scaled by
    the halved scale.  Here both backward passes of a step use the scale the step started with (they may live in one captured
    graph), so a generator overflow usually overflows the discriminator too; when BOTH optimizers are skipped in one step
    the scale is backed off ONCE, which is what the reference ends up with.  `reset_scaler()` re-creates the scaler state
    like the reference does at every epoch (train.py:207);
  * the Discriminator (RA_HVED.py:204-236; train.py:146 builds it with ks=4) is xlstm_hved_amd.Discriminator
    (csrc/dconv.hip: channels-last implicit-GEMM convolutions on the matrix cores) in every storage mode; its activations
    are 16-bit like under the reference's autocast, so with fp32 storage it takes its input in fp16 (disc.py).
"""
import torch

from . import losses, ops
from .parallel import FlatGrads


SUBSET_ROWS = None


def subset_rows(device):
    """(15, 4) table: row k = which modalities SUBSETS_MODALITIES[k] keeps (RA_HVED.py:733-738)."""
    from .model import SUBSETS_MODALITIES
    return torch.tensor([[1.0 if m in sub else 0.0 for m in range(4)] for sub in SUBSETS_MODALITIES], dtype=torch.float32, device=device)


class TrainStep:
    def __init__(self, model, disc, optimizer=None, optimizer_d=None, alpha=0.1, beta=0.2, storage=torch.bfloat16,
                 loss_scale=None, shared_encoder=True, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 share_disc_pass=True, defer_wgrads=True, group=None, world_size=None):
        """group: a torch.distributed process group (or True for the default group) -> data-parallel step, see the module
        docstring; world_size: its size if it cannot be asked (tests)."""
        self.model, self.disc = model, disc
        self.group = None if group is True else group
        self.dp = group is not None
        self.world = int(world_size) if world_size is not None else None
        self._comm = None                                    # communication stream (device runs only)
        self.defer_wgrads = bool(defer_wgrads)               # the generator's weight gradients in one end-of-backward batch
        self.optimizer, self.optimizer_d = optimizer, optimizer_d
        self.alpha, self.beta = float(alpha), float(beta)
        self.storage = storage
        self.shared_encoder = shared_encoder
        self.share_disc_pass = share_disc_pass              # D(fake.detach()) of train.py:272 reuses the forward of train.py:260
        self.dice, self.gan = losses.DiceLoss(), losses.GANLoss()
        self.grads = FlatGrads(model.parameters())             # p.grad = views of one flat fp32 bucket (one fill, one all-reduce)
        self.grads_d = FlatGrads(disc.parameters())
        dev = self.grads.flat.device
        # GradScaler's state (torch/amp/grad_scaler.py semantics; the reference re-creates it per epoch at 65536, train.py:207)
        init = float(loss_scale if loss_scale is not None else (65536.0 if storage == torch.float16 else 1.0))
        self.scaling = init != 1.0
        self._init_scale = init
        self._scale = torch.full((1,), init, dtype=torch.float32, device=dev)      # device scalar: captured graphs follow it
        self._inv_scale = torch.full((1,), 1.0 / init, dtype=torch.float32, device=dev)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self._clean_steps = 0
        self._rows = subset_rows(dev)
        self._graph = None

    @property
    def loss_scale(self):
        return float(self._scale.item())

    def set_loss_scale(self, v):
        self._scale.fill_(float(v))
        self._inv_scale.fill_(1.0 / float(v))

    def _seed(self, loss):
        """The gradient a backward pass starts from: the loss scale (a device scalar, so a captured graph follows its updates)
        or a resident 1 -- instead of a multiply launch on the loss and the ones-fill of an implicit seed."""
        if self.scaling:
            return self._scale.view(loss.shape)
        one = self.__dict__.get("_one")
        if one is None or one.device != loss.device:
            one = self.__dict__["_one"] = torch.ones((), dtype=torch.float32, device=loss.device)
        return one.view(loss.shape)

    def reset_scaler(self, loss_scale=None):
        """A fresh GradScaler: the reference constructs one per epoch (train.py:207).  The growth counter restarts too."""
        self.set_loss_scale(self._init_scale if loss_scale is None else loss_scale)
        self._clean_steps = 0

    # ------------------------------------------------------------------------------------------------
    def _disc(self, t):
        return self.disc(t)                               # HIP path: 16-bit activations, fp32 arithmetic inside (disc.py)

    def keep_mask(self, subset_index_list, n):
        """(n, 4) device mask of a subset index (the model takes subset_idx_list[0], RA_HVED.py:517-520)."""
        return self._rows[int(subset_index_list[0])].unsqueeze(0).repeat(n, 1).contiguous()

    def generator_forward(self, x, mask, subset, eps_lists=None):
        """subset: a list of subset indices like the reference's `subset_index_list`, or an (N, 4) device mask (the form a
        captured graph uses).  Returns (loss, parts dict, tensors the discriminator step reuses)."""
        xs = x.to(self.storage)
        calls = [dict(subset_idx_list=[14]), dict(subset_idx_list=subset if torch.is_tensor(subset) else list(subset))]
        if eps_lists is not None:
            calls[0]["eps_list"], calls[1]["eps_list"] = eps_lists
        if self.shared_encoder:
            (f_out, _, f_rec), (m_out, (mu, lv), m_rec) = self.model.forward_shared(xs, calls, recon=True)
        else:
            f_out, _, f_rec = self.model(xs, recon=True, **calls[0])
            m_out, (mu, lv), m_rec = self.model(xs, recon=True, **calls[1])
        f_rec = f_rec[0] if len(f_rec) == 1 else torch.cat(f_rec, 1)            # train.py:227-228
        m_rec = m_rec[0] if len(m_rec) == 1 else torch.cat(m_rec, 1)
        dice = self.dice(f_out, mask)                                            # train.py:232-234
        m_dice = self.dice(m_out, mask)
        recon = losses.mse_loss(m_rec, x)
        kld = losses.compute_KLD_levels(mu, lv, subset)                          # train.py:235-239: the mean over the levels, one node
        atten_f_x = losses.nested_attention(f_out, f_rec.detach())              # train.py:242-259
        atten_m_x = losses.nested_attention(m_out, m_rec)
        fake = torch.cat([m_out, atten_m_x], 1)
        # the discriminator is frozen for the generator's pass: its parameter gradients from this backward are discarded
        # by the reference anyway (optimizer_d.zero_grad(), train.py:282), so they are not computed
        dparams = [p for p in self.disc.parameters() if p.requires_grad]
        for p in dparams:
            p.requires_grad_(False)
        from .disc import Discriminator, DiscShare
        share = DiscShare() if (self.share_disc_pass and isinstance(self.disc, Discriminator)) else None
        try:
            pred_fake = self.disc(fake, share=share) if share is not None else self._disc(fake)
            g_gan = self.gan(pred_fake, True)                                   # train.py:260-261 (the loss reads 16-bit predictions as they are)
        finally:
            for p in dparams:
                p.requires_grad_(True)
        # train.py:262 as ONE node (one launch forward, one backward; term by term it is seven scalar launches and as many back)
        loss = losses.combine([dice, m_dice, recon, kld, g_gan], [1.0, 1.0, self.beta, self.beta, self.alpha])
        parts = dict(dice=dice, m_dice=m_dice, recon=recon, kld=kld, g_gan=g_gan)
        real = torch.cat([f_out.detach(), atten_f_x.detach()], 1)
        return loss, parts, (fake.detach(), real, f_out.detach(), share)

    def discriminator_forward(self, fake, real, share=None):
        from .disc import Discriminator
        if share is not None and share.bufs is not None and fake.shape == real.shape:
            # train.py:272 runs D on fake.detach() again: the same values through the same weights as train.py:260 -- the
            # generator pass's activations are reused and only `real` is run forward; the backward is one batch of two
            # [D(fake); D(real)] against a resident label tensor: alpha * 0.5 * (loss_d_fake + loss_d_real) as one reduction, and ONE
            # gradient pass over the batch (two losses on slices cost two zero-fills, two copies and an add in backward)
            out = self.disc.forward_pair(real, share)
            return losses.gan_pair_loss(out, fake.shape[0], self.alpha, self.gan.fake_label, self.gan.real_label)
        elif isinstance(self.disc, Discriminator) and fake.shape == real.shape:
            # both passes as one batch: InstanceNorm is per sample, so this is the same arithmetic as train.py:272-277, with
            # half the launches and twice the rows for the small deep layers (and one weight-gradient pass instead of two)
            out = self._disc(torch.cat([fake, real], 0))
            return losses.gan_pair_loss(out, fake.shape[0], self.alpha, self.gan.fake_label, self.gan.real_label)
        else:
            loss_d_fake = self.gan(self._disc(fake).float(), False)             # train.py:272-277
            loss_d_real = self.gan(self._disc(real).float(), True)
        return self.alpha * (loss_d_fake + loss_d_real) * 0.5                    # train.py:280

    # ------------------------------------------------------------------------------------------------
    def compute_generator(self, x, mask, subset, eps_lists=None):
        """First half of the step: both generator forwards, the loss epilogues, the frozen discriminator's pass on the fake sample
        and the generator's backward.  Leaves the generator's (unscaled fp32) gradients in self.grads and returns
        (parts, carry) -- `carry` is what compute_discriminator() needs.  Call inside a disc.pack_scope()."""
        self.grads.zero()
        self.grads_d.zero()
        loss, parts, carry = self.generator_forward(x, mask, subset, eps_lists)
        # the generator's weight gradients go straight into self.grads and nothing reads them before the join: collected
        # during backward and issued together (xh_conv3d_wgrad_batch: 13 launches instead of one per convolution)
        was = ops._WG["defer"]
        ops.set_wgrad_defer(self.defer_wgrads or was)
        try:
            loss.backward(self._seed(loss))                # d(scale * loss): the device-resident scale (or 1) seeds the pass
            ops.join_wgrad_stream()
        except BaseException:
            # a failed backward (or a failed stream capture) leaves a partial batch queued: launching it now would raise a
            # second error that masks the first -- drop it
            ops.drop_deferred_wgrads()
            raise
        finally:
            ops._WG["defer"] = was
        if self.scaling:
            self.grads.flat.mul_(self._inv_scale)
        parts["loss"] = loss.detach()
        return parts, carry

    def compute_discriminator(self, parts, carry):
        """Second half: the discriminator's loss on (fake.detach(), real) and its backward; gradients in self.grads_d."""
        fake, real, f_out, share = carry
        loss_d = self.discriminator_forward(fake, real, share)
        loss_d.backward(self._seed(loss_d))
        if self.scaling:
            self.grads_d.flat.mul_(self._inv_scale)
        parts["loss_d"], parts["f_out"] = loss_d.detach(), f_out
        return parts

    # ---- the data-parallel exchange: two flat buckets, two collectives ------------------------------
    def _world(self):
        if self.world is None:
            import torch.distributed as dist
            self.world = dist.get_world_size(self.group)
        return self.world

    def reduce_generator(self):
        """All-reduce (mean) of the generator's bucket, issued behind everything queued on the current stream but ON THE
        COMMUNICATION STREAM: the discriminator passes that follow do not wait for it (join_reductions does)."""
        if not self.dp:
            return
        flat = self.grads.flat
        if not self.grads.check():
            self.grads.attach()
        if flat.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream(flat.device)
            self._comm.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._comm):
                self._all_reduce_mean(flat)
        else:
            self._all_reduce_mean(flat)

    def reduce_discriminator(self):
        """Joins the generator's collective, then all-reduces (mean) the discriminator's bucket on the current stream."""
        if not self.dp:
            return
        self.join_reductions()
        if not self.grads_d.check():
            self.grads_d.attach()
        self._all_reduce_mean(self.grads_d.flat)

    def join_reductions(self):
        if self._comm is not None:
            torch.cuda.current_stream(self.grads.flat.device).wait_stream(self._comm)

    def _all_reduce_mean(self, flat):
        import torch.distributed as dist
        dist.all_reduce(flat, group=self.group)
        flat.div_(self._world())

    def compute(self, x, mask, subset, eps_lists=None):
        """Both backward passes of the step (no optimizer): generator gradients in self.grads, discriminator gradients
        in self.grads_d (unscaled fp32; averaged over the ranks of `group` when there is one).  Without a group it is
        capturable into ONE hipGraph when the inputs are device resident and `subset` is a device mask (keep_mask()): nothing of
        the subset or of the loss scale is baked into the capture.  With a group, capture() records the two halves separately
        and issues the collectives between / after them."""
        from .disc import pack_scope
        with pack_scope():                                   # the discriminator's weight images are built once for both passes
            parts, carry = self.compute_generator(x, mask, subset, eps_lists)
            self.reduce_generator()                          # runs under the discriminator passes
            parts = self.compute_discriminator(parts, carry)
            self.reduce_discriminator()
        return parts

    def check_finite(self):
        """GradScaler's overflow check on demand (one host synchronisation): True when every gradient is finite."""
        return bool(torch.isfinite(self.grads.flat).all() and torch.isfinite(self.grads_d.flat).all())

    def _update(self, parts):
        """The two optimizer steps + GradScaler.update() (train.py:264-268,282-285)."""
        if self.scaling:
            ok = torch.stack([torch.isfinite(self.grads.flat).all(), torch.isfinite(self.grads_d.flat).all()]).tolist()   # one sync
        else:
            ok = [True, True]
        if ok[0] and self.optimizer is not None:
            self.optimizer.step()
        if ok[1] and self.optimizer_d is not None:
            self.optimizer_d.step()
        if self.scaling:
            # the reference calls scaler.update() after each scaler.step(): a skipped step backs the scale off once, a clean
            # step counts towards growth.  The second update() of a step sees the found-inf of the discriminator's pass only.
            if not any(ok):
                ok_books = [False]       # both skipped in one step: ONE back-off (see the module docstring)
            else:
                ok_books = ok
            for fin in ok_books:
                if not fin:
                    self.set_loss_scale(self.loss_scale * self.backoff_factor)
                    self._clean_steps = 0
                else:
                    self._clean_steps += 1
                    if self._clean_steps >= self.growth_interval:
                        self.set_loss_scale(self.loss_scale * self.growth_factor)
                        self._clean_steps = 0
        if not all(ok):
            parts["skipped"] = [name for name, fin in zip(("generator", "discriminator"), ok) if not fin]
        return parts

    def step(self, x, mask, subset_index_list, eps_lists=None):
        """compute() + the two optimizer steps (train.py:264-268,282-285), eager."""
        return self._update(self.compute(x, mask, subset_index_list, eps_lists))

    # ------------------------------------------------------------------------------------------------
    def capture(self, x, mask, eps_lists=None, warmup=2):
        """Captures compute() ONCE into a hipGraph on static copies of (x, mask, subset mask[, eps]); `replay()` then serves
        every subset: it overwrites the static buffers and launches the graph (train.py:222-225 draws a new subset per
        step -- no re-capture, no per-subset graphs).  Data parallel (`group`): TWO graphs sharing one memory pool, split
        after the generator's backward; replay() issues the generator bucket's all-reduce on the communication stream between
        them and the discriminator bucket's after the second."""
        from .disc import pack_scope
        self._sx, self._smask = x.detach().clone(), mask.detach().clone()
        self._skeep = self.keep_mask([14], x.shape[0])
        self._seps = None if eps_lists is None else [[e.detach().clone() for e in el] for el in eps_lists]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # warm-up on a side stream (allocator pools, weight packs, arenas)
            for _ in range(warmup):
                self.compute(self._sx, self._smask, self._skeep, self._seps)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        mode = "thread_local" if self.dp else "global"       # RCCL's watchdog thread may query events while this thread captures
        ops.prepare_capture()
        self._graph = torch.cuda.CUDAGraph()
        self._graph2 = None
        if not self.dp:
            with torch.cuda.graph(self._graph, capture_error_mode=mode):
                self._parts = self.compute(self._sx, self._smask, self._skeep, self._seps)
            return self
        with pack_scope():
            with torch.cuda.graph(self._graph, capture_error_mode=mode):
                parts, carry = self.compute_generator(self._sx, self._smask, self._skeep, self._seps)
            self._graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph2, pool=self._graph.pool(), capture_error_mode=mode):
                self._parts = self.compute_discriminator(parts, carry)
        return self

    def replay(self, x, mask, subset_index_list, eps_lists=None, update=True):
        """One training step through the captured graph(s) (capture() first).  Returns the static `parts` tensors."""
        if self._graph is None:
            raise RuntimeError("TrainStep.capture() has not been called")
        self._sx.copy_(x)
        self._smask.copy_(mask)
        self._skeep.copy_(self.keep_mask(subset_index_list, self._sx.shape[0]))
        if self._seps is not None and eps_lists is not None:
            for dst_l, src_l in zip(self._seps, eps_lists):
                for d_, s_ in zip(dst_l, src_l):
                    d_.copy_(s_)
        self._graph.replay()
        if self._graph2 is not None:
            self.reduce_generator()                          # on the communication stream, under the second graph
            self._graph2.replay()
            self.reduce_discriminator()
        return self._update(dict(self._parts)) if update else dict(self._parts)

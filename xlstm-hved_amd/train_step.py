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
    and all parameters / gradients fp32; with fp16 storage the loss is scaled by `loss_scale` (GradScaler's initial 65536,
    train.py:207) and the fp32 gradients are unscaled before the optimizer -- the bookkeeping of GradScaler without its
    per-step host synchronisation (inf check), which `check_finite()` offers on demand;
  * the Discriminator (RA_HVED.py:204-236; train.py:146 builds it with ks=4) is xlstm_hved_amd.Discriminator
    (csrc/dconv.hip: channels-last implicit-GEMM convolutions on the matrix cores) in every storage mode; its activations
    are 16-bit like under the reference's autocast, so with fp32 storage it takes its input in fp16 (disc.py).
"""
import torch

from . import losses, ops
from .parallel import FlatGrads


class TrainStep:
    def __init__(self, model, disc, optimizer=None, optimizer_d=None, alpha=0.1, beta=0.2, storage=torch.bfloat16,
                 loss_scale=None, shared_encoder=True):
        self.model, self.disc = model, disc
        self.optimizer, self.optimizer_d = optimizer, optimizer_d
        self.alpha, self.beta = float(alpha), float(beta)
        self.storage = storage
        self.loss_scale = float(loss_scale if loss_scale is not None else (65536.0 if storage == torch.float16 else 1.0))
        self.shared_encoder = shared_encoder
        self.dice, self.gan = losses.DiceLoss(), losses.GANLoss()
        self.grads = FlatGrads(model.parameters())             # p.grad = views of one flat fp32 bucket (one fill, one all-reduce)
        self.grads_d = FlatGrads(disc.parameters())

    # ------------------------------------------------------------------------------------------------
    def _disc(self, t):
        return self.disc(t)                               # HIP path: 16-bit activations, fp32 arithmetic inside (disc.py)

    def generator_forward(self, x, mask, subset_index_list, eps_lists=None):
        """Returns (loss, parts dict, tensors the discriminator step reuses)."""
        xs = x.to(self.storage)
        calls = [dict(subset_idx_list=[14]), dict(subset_idx_list=list(subset_index_list))]
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
        kld = None
        for level in range(len(mu)):                                             # train.py:235-239
            k = losses.compute_KLD(mu[level], lv[level], subset_index_list)
            kld = k if kld is None else kld + k
        kld = kld / len(mu)
        atten_f_x = losses.nested_attention(f_out, f_rec.detach())              # train.py:242-259
        atten_m_x = losses.nested_attention(m_out, m_rec)
        fake = torch.cat([m_out, atten_m_x], 1)
        # the discriminator is frozen for the generator's pass: its parameter gradients from this backward are discarded
        # by the reference anyway (optimizer_d.zero_grad(), train.py:282), so they are not computed
        dparams = [p for p in self.disc.parameters() if p.requires_grad]
        for p in dparams:
            p.requires_grad_(False)
        try:
            g_gan = self.gan(self._disc(fake).float(), True)                    # train.py:260-261
        finally:
            for p in dparams:
                p.requires_grad_(True)
        loss = dice + m_dice + self.beta * recon + self.beta * kld + self.alpha * g_gan
        parts = dict(dice=dice, m_dice=m_dice, recon=recon, kld=kld, g_gan=g_gan)
        real = torch.cat([f_out.detach(), atten_f_x.detach()], 1)
        return loss, parts, (fake.detach(), real, f_out.detach())

    def discriminator_forward(self, fake, real):
        from .disc import Discriminator
        if isinstance(self.disc, Discriminator) and fake.shape == real.shape:
            # both passes as one batch: InstanceNorm is per sample, so this is the same arithmetic as train.py:272-277, with
            # half the launches and twice the rows for the small deep layers (and one weight-gradient pass instead of two)
            out = self._disc(torch.cat([fake, real], 0)).float()
            nb = fake.shape[0]
            loss_d_fake, loss_d_real = self.gan(out[:nb], False), self.gan(out[nb:], True)
        else:
            loss_d_fake = self.gan(self._disc(fake).float(), False)             # train.py:272-277
            loss_d_real = self.gan(self._disc(real).float(), True)
        return self.alpha * (loss_d_fake + loss_d_real) * 0.5                    # train.py:280

    # ------------------------------------------------------------------------------------------------
    def compute(self, x, mask, subset_index_list, eps_lists=None):
        """Both backward passes of the step (no optimizer): generator gradients in self.grads, discriminator gradients
        in self.grads_d (unscaled fp32).  Capturable into a hipGraph when the inputs are device resident."""
        from .disc import pack_scope
        s = self.loss_scale
        self.grads.zero()
        self.grads_d.zero()
        with pack_scope():                                   # the discriminator's weight images are built once for both passes
            loss, parts, (fake, real, f_out) = self.generator_forward(x, mask, subset_index_list, eps_lists)
            (loss * s if s != 1.0 else loss).backward()
            ops.join_wgrad_stream()
            if s != 1.0:
                self.grads.flat.mul_(1.0 / s)
            loss_d = self.discriminator_forward(fake, real)
            (loss_d * s if s != 1.0 else loss_d).backward()
            if s != 1.0:
                self.grads_d.flat.mul_(1.0 / s)
        parts["loss"], parts["loss_d"], parts["f_out"] = loss.detach(), loss_d.detach(), f_out
        return parts

    def check_finite(self):
        """GradScaler's overflow check on demand (one host synchronisation): True when every gradient is finite."""
        return bool(torch.isfinite(self.grads.flat).all() and torch.isfinite(self.grads_d.flat).all())

    def step(self, x, mask, subset_index_list):
        """compute() + the two optimizer steps (train.py:264-268,282-285).  With fp16 storage a non-finite gradient skips
        both steps and halves the loss scale, like GradScaler."""
        parts = self.compute(x, mask, subset_index_list)
        if self.loss_scale != 1.0 and not self.check_finite():
            self.loss_scale *= 0.5
            parts["skipped"] = True
            return parts
        if self.optimizer is not None:
            self.optimizer.step()
        if self.optimizer_d is not None:
            self.optimizer_d.step()
        return parts

"""Trains the REAL reference for a few hundred CPU steps and stores the weights (evidence for the 16-bit storage
modes that does not rest on a random initialisation).

Run in the build container only (needs /root/reference):  python tests/golden/make_trained_like.py [--steps 300]
The reference XLSTM_HVED (train.py:142-145: seed 1 -> ctor -> init_weights == tests/golden/weights_seed1.npz) is trained with
the generator part of train.py's loss (train.py:224-239,262 without the adversarial term: DiceLoss(full) + DiceLoss(subset) +
0.2 MSE(recon) + 0.2 KLD), Adam(lr 1e-3, weight_decay 1e-5 as train.py:165,177), batch 2 of 32^3 patches from
tests/synth_blobs.py (smooth modalities, nested tumour masks), a new random modality subset every step (train.py:222-223).
Output: tests/golden/weights_trained_like.npz (state_dict) + trained_like_log.json (loss curve, Dice of the last steps)."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch
import torch._dynamo  # noqa: F401  (torch.optim imports it lazily; after the shim's stub modules are registered that import trips over them)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402
import synth_blobs as SB  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--size", type=int, default=32)
a = ap.parse_args()
torch.set_num_threads(8)
ns = ref_shim.load_reference()
import loss as RL  # noqa: E402  (reference module)
import metrics as RM  # noqa: E402
import utils as RU  # noqa: E402

model = ref_shim.build_reference_model(ns, seed=1).train()
z = np.load(os.path.join(HERE, "weights_seed1.npz"))
for k, v in model.state_dict().items():                    # same starting point as every other fixture
    assert torch.equal(v, torch.from_numpy(z[k])), k
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
dice_loss, mse = RL.DiceLoss(), torch.nn.MSELoss()
np.random.seed(4)
torch.manual_seed(4)
log = []
t0 = time.time()
for step in range(a.steps):
    x, mask = SB.blob_case(1000 + step, 2, a.size)
    subset = RU.subset_idx(np.random.choice([1, 2, 3], 1))                        # train.py:222-223
    with contextlib.redirect_stdout(io.StringIO()):        # compute_per_channel_dice prints shapes (loss.py:269-270)
        f_out, _, f_rec = model(x, [14], recon=True)
        m_out, (mu, lv), m_rec = model(x, subset, recon=True)
        m_rec = m_rec[0] if len(m_rec) == 1 else torch.cat(m_rec, 1)
        dice, m_dice = dice_loss(f_out, mask), dice_loss(m_out, mask)
        recon = mse(m_rec, x)
        kld = sum(RL.compute_KLD(mu[l], lv[l], subset) for l in range(len(mu))) / len(mu)
    loss = dice + m_dice + 0.2 * recon + 0.2 * kld
    opt.zero_grad()
    loss.backward()
    opt.step()
    wt = RM.DiceRegion()(f_out.detach(), mask).item()
    log.append(dict(step=step, loss=loss.item(), dice=dice.item(), m_dice=m_dice.item(), recon=recon.item(), kld=kld.item(), wt_dice=wt))
    if step % 20 == 0 or step == a.steps - 1:
        print(f"step {step:4d} loss {loss.item():.4f} dice {dice.item():.4f} m_dice {m_dice.item():.4f} recon {recon.item():.4f} kld {kld.item():.4f} "
              f"WT dice {wt:.3f}  ({time.time() - t0:.0f} s)", flush=True)
np.savez_compressed(os.path.join(HERE, "weights_trained_like.npz"), **{k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
with open(os.path.join(HERE, "trained_like_log.json"), "w") as f:
    json.dump(dict(note="reference XLSTM_HVED trained on tests/synth_blobs.py, see make_trained_like.py", steps=a.steps, size=a.size,
                   torch=torch.__version__, log=log[::10] + log[-5:]), f, indent=1)
print("wrote weights_trained_like.npz")

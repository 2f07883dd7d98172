"""The segmentation mask the REAL reference produces for the benchmark-size parity case bench.py measures its storage modes
against (BASELINE config 2 size: 1 x 4 x 128^3).

Run in the build container only (needs /root/reference):  python tests/golden/make_mask_128.py
Case = tests/test_gpu_network.py's `trained_like_128_blob8_subset14_eval`: weights_trained_like.npz (the reference trained 300
CPU steps, make_trained_like.py), input tests/synth_blobs.blob_case(8, 1, 128), all four modalities, eval mode, posterior mean
(valid=True), fp32.  Written to tests/golden/mask_trained_like_128.npz (data only):
    bits     np.packbits of (seg > 0.5) over (1, 3, 128, 128, 128)      metrics.py:85-107 thresholds at 0.5
    margin   how many voxels of each channel sit within 1e-3 of the threshold (those may legitimately flip)
    pos      positive voxels per channel
The oracle (oracle/xlstm_hved_oracle.py) is run on the same case and must reproduce the reference's mask exactly.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_shim  # noqa: E402
import synth_blobs as SB  # noqa: E402
import xlstm_hved_oracle as O  # noqa: E402

torch.set_num_threads(8)
ns = ref_shim.load_reference()
model = ref_shim.build_reference_model(ns).eval()
z = np.load(os.path.join(HERE, "weights_trained_like.npz"))
w = {k: torch.from_numpy(z[k]) for k in z.files}
model.load_state_dict(w, strict=True)
x, _ = SB.blob_case(8, 1, 128)
with torch.no_grad():
    out = model(x, [14], recon=True, valid=True)
    seg = out[0].float()
    prob_o = O.xlstm_hved_forward({k: v.clone() for k, v in w.items()}, x, 14, eps_list=None, training=False)[0]
mask = seg > 0.5
flips = int((mask != (prob_o > 0.5)).sum())
print(f"reference mask: positives per channel {[int(mask[0, c].sum()) for c in range(3)]}; oracle vs reference: "
      f"max |d| {(seg - prob_o).abs().max().item():.2e}, mask flips {flips}")
assert flips == 0, "the oracle's mask differs from the reference's"
np.savez_compressed(os.path.join(HERE, "mask_trained_like_128.npz"),
                    bits=np.packbits(mask.numpy().reshape(-1)),
                    shape=np.array(mask.shape, dtype=np.int64),
                    margin=np.array([int(((seg[0, c] - 0.5).abs() < 1e-3).sum()) for c in range(3)], dtype=np.int64),
                    pos=np.array([int(mask[0, c].sum()) for c in range(3)], dtype=np.int64))
print("written", os.path.join(HERE, "mask_trained_like_128.npz"))

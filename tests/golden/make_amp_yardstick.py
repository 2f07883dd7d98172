"""Measures the REFERENCE's own mixed-precision deviation (the yardstick for the build's 16-bit storage modes).

Run in the build container only (needs /root/reference):  python tests/golden/make_amp_yardstick.py
The reference trains under `with autocast():` (train.py:20,207,218 -- fp16 on the authors' CUDA box) and keeps the ViL
block in fp32 (`@autocast(enabled=False)` + up-cast, UxLSTMEnc_3d.py:77-80).  There is no GPU here, so the same model
is run under `torch.autocast("cpu", dtype=fp16|bf16)` (the CPU autocast op lists: conv/linear/matmul in 16 bit, the
rest in the input's type) with the ViL block forced to fp32 like the CUDA decorator does ("vil_fp32"), and once as
is ("cpu_autocast_as_is": on CPU that decorator does nothing, so the ViL block runs in 16 bit too).  The outputs are
compared with the same model in fp32 on the SAME seeded weights (tests/golden/weights_seed1.npz) and the SAME inputs the
-m gpu tests use; the numbers go to tests/golden/amp_yardstick.json (data only).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_shim  # noqa: E402
import xlstm_hved_oracle as O  # noqa: E402

torch.set_num_threads(8)
ns = ref_shim.load_reference()
from UxLSTM.nnunetv2.nets import UxLSTMEnc_3d as UX  # noqa: E402

model = ref_shim.build_reference_model(ns).eval()
z = np.load(os.path.join(HERE, "weights_seed1.npz"))
model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)

_orig_vil = UX.ViLLayer.forward


def _vil_fp32(self, x):
    with torch.autocast("cpu", enabled=False):
        return _orig_vil(self, x.float())


def metrics(out, ref):
    seg, rec = out[0].float(), out[2][0].float()
    seg0, rec0 = ref[0], ref[2][0]
    tgt = (seg0 > 0.5).float()
    return {
        "seg_rel_l2": ((seg - seg0).norm() / seg0.norm()).item(),
        "seg_max_abs": (seg - seg0).abs().max().item(),
        "recon_rel_l2": ((rec - rec0).norm() / rec0.norm()).item(),
        "recon_max_rel": ((rec - rec0).abs().max() / rec0.abs().max()).item(),
        "dice_dev": (O.dice_region(seg, tgt) - 1.0).abs().max().item(),
        "mask_flips": int(((seg > 0.5) != (seg0 > 0.5)).sum().item()), "mask_voxels": int(seg.numel()),
    }


cases = {"64_seed5_subset14_eval": (5, 64, 14), "128_seed2_subset14_eval": (2, 128, 14), "64_seed5_subset7_eval": (5, 64, 7)}
res = {"note": "reference XLSTM_HVED (weights_seed1) under torch.autocast('cpu') vs itself in fp32; see make_amp_yardstick.py",
       "torch": torch.__version__, "cases": {}}
with torch.no_grad():
    for name, (seed, s, k) in cases.items():
        torch.manual_seed(seed)
        x = torch.rand(1, 4, s, s, s)
        ref = model(x, [k], recon=True, valid=True)
        entry = {}
        for dn, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            for mode in ("vil_fp32", "cpu_autocast_as_is"):
                UX.ViLLayer.forward = _vil_fp32 if mode == "vil_fp32" else _orig_vil
                with torch.autocast("cpu", dtype=dt):
                    out = model(x, [k], recon=True, valid=True)
                entry[f"{dn}.{mode}"] = metrics(out, ref)
                print(name, dn, mode, entry[f"{dn}.{mode}"], flush=True)
        UX.ViLLayer.forward = _orig_vil
        res["cases"][name] = entry
# ---- the same yardstick on TRAINED-LIKE weights and smooth inputs (tests/golden/make_trained_like.py, tests/synth_blobs.py):
# the randomly initialised network above amplifies any rounding ~1e4x; a network that has seen a few hundred
# optimisation steps on smooth data is the fairer picture of what a 16-bit mode costs in use
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_blobs as SB  # noqa: E402
tl = os.path.join(HERE, "weights_trained_like.npz")
if os.path.exists(tl):
    z = np.load(tl)
    model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    model.eval()
    tcases = {"trained_like_64_blob7_subset14_eval": (7, 64, 14), "trained_like_64_blob7_subset5_eval": (7, 64, 5),
              "trained_like_128_blob8_subset14_eval": (8, 128, 14)}
    with torch.no_grad():
        for name, (seed, s, k) in tcases.items():
            x, _ = SB.blob_case(seed, 1, s)
            if k != 14:
                keep = O.SUBSETS_MODALITIES[k]
                for c in range(4):
                    if c not in keep:
                        x[:, c] = 0                            # evaluation.py:305-307: dropped modalities are zeroed
            ref = model(x, [k], recon=True, valid=True)
            entry = {}
            for dn, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
                UX.ViLLayer.forward = _vil_fp32
                with torch.autocast("cpu", dtype=dt):
                    out = model(x, [k], recon=True, valid=True)
                entry[f"{dn}.vil_fp32"] = metrics(out, ref)
                print(name, dn, entry[f"{dn}.vil_fp32"], flush=True)
            UX.ViLLayer.forward = _orig_vil
            entry["positive_fraction"] = [(ref[0][:, c] > 0.5).float().mean().item() for c in range(3)]
            res["cases"][name] = entry
with open(os.path.join(HERE, "amp_yardstick.json"), "w") as f:
    json.dump(res, f, indent=1)

"""Smooth synthetic BraTS-like patches ("low-pass filtered noise + blobs, zero background"): four modalities in
[0, 1] with a zero background outside an ellipsoidal head, and nested tumour regions WT > TC > ET (irregular ellipsoids) that
change the modality intensities the way the real sequences do (oedema bright on T2 / FLAIR, enhancing core bright on T1ce).
Deterministic in (seed, n, S); used by tests/golden/make_trained_like.py to train the reference for a few hundred CPU steps
and by the -m gpu tests that measure the 16-bit storage modes on those trained-like weights."""
import torch
import torch.nn.functional as F


def _lowpass(g, n, c, S, coarse):
    z = torch.randn(n, c, coarse, coarse, coarse, generator=g)
    z = F.interpolate(z, size=(S, S, S), mode="trilinear", align_corners=False)
    lo, hi = z.amin((2, 3, 4), keepdim=True), z.amax((2, 3, 4), keepdim=True)
    return (z - lo) / (hi - lo).clamp_min(1e-6)


def blob_case(seed, n, S):
    """Returns x (n, 4, S, S, S) float32 in [0, 1] and mask (n, 3, S, S, S) float32 = (WT, TC, ET), nested."""
    g = torch.Generator().manual_seed(int(seed))
    ax = torch.linspace(-1.0, 1.0, S)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    grid = torch.stack([zz, yy, xx])[None]                                   # (1, 3, S, S, S)
    head = ((grid / torch.tensor([0.95, 0.85, 0.9]).view(1, 3, 1, 1, 1)) ** 2).sum(1, keepdim=True) < 1.0
    tex = _lowpass(g, n, 4, S, max(S // 8, 2))
    centre = (torch.rand(n, 3, generator=g) - 0.5) * 0.7
    radii = 0.28 + 0.22 * torch.rand(n, 1, generator=g)
    aniso = 0.8 + 0.4 * torch.rand(n, 3, generator=g)
    wobble = 0.25 * (_lowpass(g, n, 1, S, max(S // 8, 2)) - 0.5)
    dist = (((grid - centre.view(n, 3, 1, 1, 1)) / aniso.view(n, 3, 1, 1, 1)) ** 2).sum(1, keepdim=True).sqrt() + wobble
    r = radii.view(n, 1, 1, 1, 1)
    wt, tc, et = dist < r, dist < 0.62 * r, dist < 0.36 * r
    wt = wt & head
    tc = tc & wt
    et = et & tc
    wtf, tcf, etf = wt.float(), tc.float(), et.float()
    base = 0.25 + 0.35 * tex
    contrast = torch.cat([-0.10 * wtf - 0.05 * tcf,             # T1: hypo-intense lesion
                          -0.08 * wtf + 0.45 * etf,             # T1ce: enhancing core
                          0.30 * wtf + 0.05 * tcf,              # T2: oedema bright
                          0.35 * wtf - 0.12 * tcf], 1)          # FLAIR: oedema bright, core darker
    x = ((base + contrast) * head.float()).clamp(0.0, 1.0)
    return x.contiguous(), torch.cat([wtf, tcf, etf], 1).contiguous()

"""Helpers shared by the -m gpu parity tests."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def sd_of(g, prefix="sd."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)


def check(a, b, tol, what):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
    return e


def rnd(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def l2_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def check_grads(named_got, named_ref, tol, what, l2=False):
    """Parameter gradients of one stage: deviations are scaled by the largest gradient magnitude of the stage so
    that gradients which are mathematically zero (a bias feeding a normalisation) are judged on an absolute scale."""
    scale = max(v.double().abs().max().item() for v in named_ref.values())
    worst = 0.0
    for k, v in named_ref.items():
        assert named_got[k] is not None, f"{what}: no gradient for {k}"
        a, b = named_got[k].detach().double().cpu(), v.double()
        if l2:
            err = (a - b).norm().item() / max(b.norm().item(), 2e-2 * scale * b.numel() ** 0.5)
        else:
            err = (a - b).abs().max().item() / scale
        assert err <= tol, f"{what}.{k}: scaled err {err:.3e} > {tol:.1e}"
        worst = max(worst, err)
    return worst

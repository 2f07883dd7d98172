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

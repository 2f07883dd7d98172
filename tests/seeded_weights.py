"""Deterministic parameter values for the variant fixtures (tests/golden/variant_*.npz).

The variant fixtures do not store weights: both tests/golden/make_golden.py (which loads these values into the REAL
reference before running it) and the replaying tests draw them from this procedure, keyed by state_dict order, so a
fixture only has to carry names, shapes and the reference's outputs."""
import math

import torch


def seeded_state(entries, seed):
    """entries: iterable of (name, shape tuple, is_float).  Returns {name: fp32/long tensor}."""
    out = {}
    for i, (name, shape, is_float) in enumerate(entries):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        shape = tuple(int(s) for s in shape)
        if not is_float:
            out[name] = torch.zeros(shape, dtype=torch.long)
            continue
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "running_var":
            t = 1.0 + 0.2 * torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) <= 1:
            t = 0.1 * torch.randn(shape, generator=g)
            if leaf == "weight":
                t = t + 1.0
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        out[name] = t.float()
    return out


def entries_of(state_dict):
    return [(k, tuple(v.shape), bool(v.is_floating_point())) for k, v in state_dict.items()]

"""Host-side helpers the reference's callers use next to the model (utils.py of the reference)."""
import random

import numpy as np
import torch
from torch import nn
from torch.nn import init


def _draw(fn, **kw):
    return lambda t: fn(t.data, **kw)


# module type -> (weight initialiser, bias initialiser), in the precedence utils.py:191-212 tests them.  One row per stock
# parameter holder the drop-in classes keep their parameters in; a type that is not listed keeps its constructor values.
_INIT_TABLE = (
    (nn.Conv3d, _draw(init.kaiming_normal_), _draw(init.normal_)),
    (nn.ConvTranspose3d, _draw(init.xavier_normal_), _draw(init.normal_)),
    (nn.BatchNorm3d, _draw(init.normal_, mean=1, std=0.02), _draw(init.constant_, val=0)),
    (nn.Linear, _draw(init.xavier_normal_), _draw(init.normal_)),
)


def init_weights(m):
    """Seed-compatible with utils.py:191-215 of the reference: the same initialiser per module type, drawn weight first and
    bias second, so `model.apply(init_weights)` under one seed gives the reference's weights.  The reference also walks the
    DIRECT children of every nn.ModuleList a second time (utils.py:213-215; `apply` has visited them already), which
    advances the RNG: reproduced, or the draws after the first ModuleList would differ."""
    for kind, w_init, b_init in _INIT_TABLE:
        if isinstance(m, kind):
            w_init(m.weight)
            if m.bias is not None:
                b_init(m.bias)
            return
    if isinstance(m, nn.ModuleList):
        for child in m:
            init_weights(child)


def subset_idx(subset_size=(4,)):
    """utils.py:36-51: one random subset index per requested cardinality."""
    ranges = {1: (0, 4), 2: (4, 10), 3: (10, 13), 4: (13, 14)}
    out = []
    for size in subset_size:
        lo, hi = ranges[int(size)]
        k = int(np.random.choice(range(lo, hi)))
        if k not in out:
            out.append(k)
    return out


def seed_everything(seed_value):
    """utils.py:179-189."""
    random.seed(seed_value)
    np.random.seed(seed_value)
    torch.manual_seed(seed_value)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed_value)

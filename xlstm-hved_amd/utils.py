"""Host-side helpers the reference's callers use next to the model (utils.py of the reference)."""
import random

import numpy as np
import torch
from torch import nn
from torch.nn import init


def init_weights(m):
    """utils.py:191-215: kaiming-normal Conv3d weights, N(0,1) biases, xavier Linear, BatchNorm N(1,0.02)/0, and the
    reference's nn.ModuleList branch (utils.py:213-215): under `model.apply(init_weights)` the DIRECT children of every
    ModuleList are initialised a second time (apply visits them as well), which advances the RNG -- reproduced so that
    the same seed draws the same weights as the reference."""
    if isinstance(m, nn.Conv3d):
        init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            init.normal_(m.bias.data)
    elif isinstance(m, nn.ConvTranspose3d):
        init.xavier_normal_(m.weight.data)
        if m.bias is not None:
            init.normal_(m.bias.data)
    elif isinstance(m, nn.BatchNorm3d):
        init.normal_(m.weight.data, mean=1, std=0.02)
        init.constant_(m.bias.data, 0)
    elif isinstance(m, nn.Linear):
        init.xavier_normal_(m.weight.data)
        if m.bias is not None:
            init.normal_(m.bias.data)
    elif isinstance(m, nn.ModuleList):
        for l in m:
            init_weights(l)


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

"""Shared by the Discriminator tests (CPU oracle pin and -m gpu parity): rebuilds what tests/golden/make_golden.py::disc_cases
fed the REAL reference class -- weights from `torch.manual_seed(7)` -> ctor -> `.apply(init_weights)` (train.py:146-147), input
from a seeded generator -- and checks the per-tensor checksums the fixture stores, so a drift in either RNG path fails loudly
instead of comparing against the wrong numbers.  Also holds the stock-module comparator (nn.Conv3d / nn.InstanceNorm3d /
nn.LeakyReLU, i.e. the reference's own layer list, buildingblocks.py:342-358) used for full-size per-layer checks."""
import os

import numpy as np
import torch
from torch import nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
X_SHAPE, X_SEED, W_SEED = (2, 7, 40, 36, 44), 501, 7


def rnd(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


def sample_index(numel, cap=4096):
    return torch.arange(0, numel, max(1, numel // cap))


class StockDiscriminator(nn.Module):
    """RA_HVED.py:204-236 as stock PyTorch modules (CPU / fp32 comparator; never on the product path)."""

    def __init__(self, in_channels=3, f_maps=(64, 128, 256, 512), ks=3, strides=(1, 2, 2, 2)):
        super().__init__()
        blocks = []
        for i, (out_f, st) in enumerate(zip(f_maps, strides)):
            layers = [nn.Conv3d(in_channels, out_f, ks, stride=st, padding=1)]
            if i > 0:
                layers.append(nn.InstanceNorm3d(out_f))
            layers.append(nn.LeakyReLU(0.2, inplace=True))
            blocks.append(nn.Sequential(*layers))
            in_channels = out_f
        self.disc = nn.ModuleList(blocks)
        self.last = nn.Conv3d(512, 1, ks, padding=1, bias=False)

    def forward(self, x):
        for block in self.disc:
            x = block(x)
        return self.last(x)


def load_fixture(ks):
    z = np.load(os.path.join(GOLDEN, f"stage_disc_ks{ks}.npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) if z[k].dtype.kind in "fiu" else z[k] for k in z.files}


def seeded_disc_state(ks, g=None):
    """state_dict of Discriminator(in_channels=7, ks=ks, strides=[1,2,2,2]).apply(init_weights) at seed 7, checked against the
    fixture's checksums when `g` (a loaded fixture) is given."""
    import xlstm_hved_amd as X
    torch.manual_seed(W_SEED)
    m = StockDiscriminator(in_channels=7, ks=ks)
    m.apply(X.init_weights)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    if g is not None:
        assert list(sd.keys()) == [str(n) for n in g["names"]]
        for k, v in sd.items():
            a, b = v.double().sum().item(), float(g["wsum." + k])
            assert abs(a - b) <= 1e-9 * max(1.0, float(g["wabs." + k])), f"seeded weight {k} differs from what the reference drew"
    return sd


def seeded_input(g=None):
    x = rnd(X_SHAPE, X_SEED)
    if g is not None:
        assert abs(x.double().sum().item() - float(g["xsum"])) <= 1e-9 * float(g["xabs"])
    return x

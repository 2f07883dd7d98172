"""Sliding-window tiler (xlstm_hved_amd.inference, the reference's eval_overlap evaluation.py:279-384): host logic, tested on
CPU with a stub model -- window rule, accumulation against a direct restatement of the reference loop, and the 2-rank sharding."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import xlstm_hved_amd as X
from xlstm_hved_amd.inference import eval_overlap_volume, window_list, window_origins


def ref_origins(size, patch, step):
    # evaluation.py:311-321 with the evident fix (patch_size[axis] instead of the list)
    r = list(range(0, size - patch + 1, step))
    if (size - patch) % step != 0:
        r.append(size - patch)
    return r


@pytest.mark.parametrize("size,patch,step,want", [(155, 128, 64, [0, 27]), (240, 128, 64, [0, 64, 112]), (128, 128, 64, [0]),
                                                  (192, 128, 64, [0, 64]), (40, 32, 8, [0, 8]), (41, 32, 8, [0, 8, 9])])
def test_window_origins(size, patch, step, want):
    assert window_origins(size, patch, step) == want == ref_origins(size, patch, step)


def test_survey_c5_window_count():
    # 240 x 240 x 155 with 128^3 windows every 64 voxels -> 3 x 3 x 2 windows
    assert len(window_list((240, 240, 155), (128, 128, 128), (64, 64, 64))) == 18


def stub_model(crop, subset_idx_list=(14,), valid=True):
    """Deterministic stand-in with the generator's call signature: 3 'class' maps from the 4 input channels."""
    w = torch.tensor([[0.5, -0.2, 0.1, 0.3], [0.2, 0.4, -0.3, 0.1], [-0.1, 0.2, 0.6, -0.4]])
    y = torch.einsum("oc,ncdhw->nodhw", w, crop.float()) + 0.01 * subset_idx_list[0]
    return torch.sigmoid(y), []


def reference_loop(x, patch, step, batch_size):
    """eval_overlap's accumulation (evaluation.py:323-381) restated with numpy-style slicing on the host."""
    _, _, D, H, W = x.shape
    sum_tot = torch.zeros((1, 3, D, H, W))
    count_tot = torch.zeros((1, 3, D, H, W), dtype=torch.int32)
    base, crops = [], []

    def flush():
        pred = stub_model(torch.cat(crops), subset_idx_list=[14], valid=True)[0]
        for i, (d2, h2, w2) in enumerate(base):
            sum_tot[:, :, d2:d2 + patch[0], h2:h2 + patch[1], w2:w2 + patch[2]] += pred[i]
            count_tot[:, :, d2:d2 + patch[0], h2:h2 + patch[1], w2:w2 + patch[2]] += 1
    for d in ref_origins(D, patch[0], step[0]):
        for h in ref_origins(H, patch[1], step[1]):
            for w in ref_origins(W, patch[2], step[2]):
                base.append((d, h, w))
                crops.append(x[:, :, d:d + patch[0], h:h + patch[1], w:w + patch[2]])
                if len(base) == batch_size:
                    flush()
                    base, crops = [], []
    if base:
        flush()
    return sum_tot / count_tot


@pytest.mark.parametrize("batch_size", [1, 3])
def test_tiler_matches_reference_loop(batch_size):
    torch.manual_seed(0)
    x = torch.rand(1, 4, 21, 16, 27)
    got = eval_overlap_volume(stub_model, x, 14, (8, 8, 8), (4, 4, 4), batch_size=batch_size)
    want = reference_loop(x, (8, 8, 8), (4, 4, 4), batch_size)
    assert torch.allclose(got, want, atol=1e-6)


def test_tiler_zeroes_modalities_outside_the_subset():
    """evaluation.py:305-307: `x_batch[:, mod_list == False] = 0` (mod_list is a bool ndarray row) before cropping."""
    torch.manual_seed(0)
    x = torch.rand(1, 4, 16, 16, 16)
    seen = []

    def spy(crop, subset_idx_list=(14,), valid=True):
        seen.append(crop.clone())
        return stub_model(crop, subset_idx_list, valid)
    got = eval_overlap_volume(spy, x, 5, (8, 8, 8), (8, 8, 8))           # subset 5 = modalities (0, 2)
    assert X.SUBSETS_MODALITIES[5] == (0, 2)
    for c in seen:
        assert c[:, 1].abs().sum() == 0 and c[:, 3].abs().sum() == 0 and c[:, 0].abs().sum() > 0 and c[:, 2].abs().sum() > 0
    xz = x.clone()
    xz[:, [1, 3]] = 0
    want = stub_model(xz, [5])[0]
    assert torch.allclose(got, want, atol=1e-6)
    assert x[:, 1].abs().sum() > 0                                        # the caller's volume is left alone


def _rank_main(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    x = torch.rand(1, 4, 21, 16, 27)
    y = eval_overlap_volume(stub_model, x, 14, (8, 8, 8), (4, 4, 4), batch_size=2, rank=rank, world=world)
    if rank == 0:
        np.save(out, y.numpy())
    dist.destroy_process_group()


def test_tiler_sharded_over_two_ranks(tmp_path):
    out = str(tmp_path / "y.npy")
    mp.spawn(_rank_main, args=(2, 29517, out), nprocs=2, join=True)
    torch.manual_seed(0)
    x = torch.rand(1, 4, 21, 16, 27)
    want = eval_overlap_volume(stub_model, x, 14, (8, 8, 8), (4, 4, 4), batch_size=2)
    assert np.allclose(np.load(out), want.numpy(), atol=1e-6)

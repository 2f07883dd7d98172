"""Sliding-window inference over a whole volume (the caller on the inference side of the hot path).

Mirrors `eval_overlap` of the reference (evaluation.py:279-384): windows of `patch_size` every `overlap_stepsize` voxels
plus one window flush with the far edge when the stride does not land on it, every window's sigmoid output summed into
the volume and divided by the per-voxel window count.  Differences, all deliberate:

* evaluation.py:316-321 appends `D - patch_size` with `patch_size` a *list* (a TypeError whenever the branch is taken);
  the rule implemented here is the evident intent, `D - patch_size[axis]`;
* like the reference (evaluation.py:305-307) the modalities outside the subset are zeroed in the volume before cropping;
* sums and counts are accumulated on the device, not copied to the host per window;
* the window list can be sharded round-robin over ranks (`rank`, `world`): every rank accumulates its own windows and ONE
  all-reduce of (sum, count) finishes the volume -- windows are independent, so there is no other exchange step.

The model call is `model(crop, subset_idx_list=[subset_idx], valid=valid)[0]` exactly like the reference, so anything with
that signature works (the tests drive the tiler with a stub model on CPU).
"""
import torch

from .model import SUBSETS_MODALITIES
from .parallel import shard_windows


def window_origins(size, patch, step):
    """Origins along one axis: range(0, size - patch + 1, step), plus size - patch if the stride misses the far edge."""
    if patch > size:
        raise ValueError(f"patch {patch} larger than the volume axis {size}")
    origins = list(range(0, size - patch + 1, step))
    if (size - patch) % step != 0:
        origins.append(size - patch)
    return origins


def window_list(shape, patch_size, overlap_stepsize):
    """All (d, h, w) window origins in the reference's loop order (d outermost)."""
    D, H, W = shape
    return [(d, h, w) for d in window_origins(D, patch_size[0], overlap_stepsize[0])
            for h in window_origins(H, patch_size[1], overlap_stepsize[1])
            for w in window_origins(W, patch_size[2], overlap_stepsize[2])]


# Captured window forwards, kept between calls (a whole-volume pass is 18 replays of ~1.3 ms; capturing the graph again for every
# volume cost as much as ten of them).  A captured graph holds its static tensors and a private allocator pool -- at 128^3 the
# whole forward's activations -- so the cache
#   * lives ON the model (`model.__dict__["_xh_window_graphs"]`): it goes when the model goes, there is no process-wide table;
#   * is a small LRU (WINDOW_GRAPHS_MAX entries, default 2: e.g. the fp16 and the fp32 graph of one evaluation): a 15-subset
#     missing-modality evaluation re-captures per subset instead of holding 15 pools (set_window_graphs_max to trade memory for it);
#   * is keyed by everything the captured launches depend on, INCLUDING where the parameters and buffers live: the graph reads
#     them in place (weight updates between volumes are seen), so after model.half() / .to() / load_state_dict(assign=True) the
#     recorded addresses are stale -- the fingerprint differs and the window is captured again.
# A change of the model's structure (anything else that changes which kernels a forward launches) needs clear_window_graphs(model).
WINDOW_GRAPHS_MAX = [2]


def set_window_graphs_max(n):
    WINDOW_GRAPHS_MAX[0] = max(1, int(n))


class _GraphCache(dict):
    """The per-model table; a copy or pickle of the model starts with an empty one (graphs are not copyable)."""

    def __deepcopy__(self, memo):
        return _GraphCache()

    def __reduce__(self):
        return (_GraphCache, ())


def clear_window_graphs(model=None):
    """Drops the captured window forwards of `model` (and their memory pools)."""
    if model is not None:
        model.__dict__.pop("_xh_window_graphs", None)


def _fingerprint(model):
    """Addresses and types of every parameter and buffer: what a captured forward has baked in."""
    return tuple((t.data_ptr(), t.dtype) for t in list(model.parameters()) + list(model.buffers()))


def _window_graph(model, subset_idx, batch_size, channels, patch_size, dtype, device):
    from . import ops
    cache = model.__dict__.setdefault("_xh_window_graphs", _GraphCache())
    key = (bool(model.training), int(subset_idx), int(batch_size), int(channels), tuple(patch_size), dtype, str(device),
           ops.current_arith() if getattr(model, "fp32_arith", None) is None else model.fp32_arith, ops.MIXED[0])
    fp = _fingerprint(model)
    hit = cache.pop(key, None)
    if hit is not None and hit[0] == fp:
        cache[key] = hit                                 # most recently used last
        return hit[1:]
    del hit                                              # (stale addresses: its pool is released before the new capture)
    while len(cache) >= WINDOW_GRAPHS_MAX[0]:
        cache.pop(next(iter(cache)))
    static_in = torch.zeros((batch_size, channels) + tuple(patch_size), dtype=dtype, device=device)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model(static_in, subset_idx_list=[subset_idx], valid=True)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    ops.prepare_capture()                                # (weight-pack job table, statistics fan-in block of the capture)
    with torch.cuda.graph(graph):
        static_out = model(static_in, subset_idx_list=[subset_idx], valid=True)[0]
    cache[key] = (fp, graph, static_in, static_out)
    return graph, static_in, static_out


@torch.no_grad()
def eval_overlap_volume(model, x, subset_idx=14, patch_size=(128, 128, 128), overlap_stepsize=(64, 64, 64), batch_size=1,
                        draw=None, num_classes=3, rank=0, world=1, group=None, use_graph=False):
    """x: (1, 4, D, H, W) on the model's device.  Returns the (1, num_classes, D, H, W) fp32 overlap-averaged
    probabilities (on every rank when world > 1).  `draw=None` uses the posterior mean (valid=True); an integer averages
    that many random draws per window (evaluation.py:286-291,339-349).  use_graph=True (device tensors, posterior mean
    only) captures the window forward once into a hipGraph (kept on the model for later volumes: clear_window_graphs(model)) and replays it per window batch: the eager forward is bound
    by ~300 host-side launches, the replay by the GPU."""
    if x.dim() != 5 or x.shape[0] != 1:
        raise ValueError("expected one volume shaped (1, C, D, H, W)")
    valid = draw is None
    ndraw = 1 if draw is None else int(draw)
    # evaluation.py:305-307: `x_batch[:, mod_list == False] = 0` -- SUBSETS_MODALITIES there is a (15, 4) bool ndarray
    # (evaluation.py:15-21), so `mod_list == False` is a boolean channel mask: the modalities outside the subset are zeroed
    # in the input BEFORE cropping, which is what the skip-return path (x0_init(x), RA_HVED.py:621) then sees.
    keep = torch.zeros(x.shape[1], dtype=torch.bool, device=x.device)
    keep[list(SUBSETS_MODALITIES[subset_idx])] = True
    if not bool(keep.all()):
        x = x * keep.view(1, -1, 1, 1, 1).to(x.dtype)
    D, H, W = x.shape[2:]
    pd, ph, pw = patch_size
    wins = window_list((D, H, W), patch_size, overlap_stepsize)
    mine = [wins[i] for i in shard_windows(len(wins), rank, world)]
    sum_tot = torch.zeros((1, num_classes, D, H, W), dtype=torch.float32, device=x.device)
    count_tot = torch.zeros((1, 1, D, H, W), dtype=torch.float32, device=x.device)
    graph = static_in = static_out = None
    if use_graph and valid and x.is_cuda and mine:
        graph, static_in, static_out = _window_graph(model, subset_idx, batch_size, x.shape[1], (pd, ph, pw), x.dtype, x.device)
    for i0 in range(0, len(mine), batch_size):
        chunk = mine[i0:i0 + batch_size]
        crops = torch.cat([x[:, :, d:d + pd, h:h + ph, w:w + pw] for d, h, w in chunk], 0).contiguous()
        if graph is not None:
            static_in[:len(chunk)].copy_(crops)               # a short last batch reuses stale rows; they are ignored below
            graph.replay()
            pred = static_out.float()
        else:
            pred = None
            for _ in range(ndraw):
                p = model(crops, subset_idx_list=[subset_idx], valid=valid)[0].float()
                pred = p if pred is None else pred + p
            pred = pred / ndraw
        for j, (d, h, w) in enumerate(chunk):
            sum_tot[:, :, d:d + pd, h:h + ph, w:w + pw] += pred[j]
            count_tot[:, :, d:d + pd, h:h + ph, w:w + pw] += 1
    if world > 1 or group is not None:            # one exchange step: the (sum, count) all-reduce
        import torch.distributed as dist
        dist.all_reduce(sum_tot, group=group)
        dist.all_reduce(count_tot, group=group)
    return sum_tot / count_tot

"""-m gpu: weight gradients deferred and batched (ops.set_wgrad_defer -> xh_conv3d_wgrad_batch, the multi-problem MFMA
kernel) or on the side stream (ops.set_wgrad_overlap) give the same gradients as the plain per-call step, eagerly and
through a hipGraph replay, for bf16 and fp16 storage."""
import pytest
import torch

from gpu_common import load

pytestmark = pytest.mark.gpu

import xlstm_hved_amd as X  # noqa: E402

DEV = "cuda"


def _grads(mode, graph, dtype=torch.bfloat16, size=64):
    m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    m.load_state_dict(load("weights_seed1"), strict=True)
    m = m.to(DEV).train()
    torch.manual_seed(3)
    x = torch.rand(1, 4, size, size, size).to(DEV, dtype)
    h = size // 2
    eps = [torch.randn(1, 2 ** l, h >> l, h >> l, h >> l).to(DEV, dtype) for l in range(4)]
    fg = X.parallel.FlatGrads(m.parameters())
    X.ops.set_wgrad_overlap(mode == "side")
    X.ops.set_wgrad_defer(mode == "defer")
    try:
        def step():
            fg.zero()
            seg, (mu, lv), rec = m(x, [14], recon=True, eps_list=eps)
            (seg.float().mean() + rec[0].float().mean() + sum(a.float().mean() + b.float().mean() for a, b in zip(mu, lv))).backward()
            X.ops.join_wgrad_stream()
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step()
            for _ in range(3):
                g.replay()
        else:
            step()
        torch.cuda.synchronize()
        return fg.flat.clone()
    finally:
        X.ops.set_wgrad_overlap(False)
        X.ops.set_wgrad_defer(False)


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graph"])
@pytest.mark.parametrize("mode", ["side", "defer"])
def test_wgrad_side_stream_and_deferred_batch_match_plain_step(mode, graph):
    a = _grads("plain", graph)
    b = _grads(mode, graph)
    assert torch.isfinite(b).all()
    scale = a.abs().max().item()
    # fp32 atomics accumulate in a different order on every run: compare to round-off of the largest gradient
    assert (a - b).abs().max().item() <= 2e-4 * scale, (a - b).abs().max().item() / scale


def test_deferred_batch_fp16_and_batch2():
    a = _grads("plain", False, torch.float16)
    b = _grads("defer", False, torch.float16)
    assert (a - b).abs().max().item() <= 2e-4 * a.abs().max().item()


def test_bench_configuration_128_graph_replay_matches_eager():
    """The benchmarked configuration itself (128^3, bf16, deferred weight gradients, prepacked fragments, statistics fan-in --
    the launches with >= 256 workgroups per channel block only exist at this size): three hipGraph replays give the gradients
    of the eager step, and the fan-in path gives those of the direct-atomics path (xh_set_option(2, 64))."""
    lib = X._lib.load()
    a = _grads("defer", False, size=128)
    b = _grads("defer", True, size=128)
    lib.xh_set_option(2, 64)
    try:
        c = _grads("defer", False, size=128)
    finally:
        lib.xh_set_option(2, 0)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 5e-4 * scale, (a - b).abs().max().item() / scale
    assert (a - c).abs().max().item() <= 5e-4 * scale, (a - c).abs().max().item() / scale

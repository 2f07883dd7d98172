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


def test_level_streams_switch_is_inert_under_capture_and_matches_eagerly():
    """ops.set_level_streams (the coarse latent-path chains on side streams, an A/B switch).  Eagerly it gives the gradients of the
    one-stream step.  Under stream capture it is INERT (round 6): as parallel graph branches the chains run truly concurrently, and
    besides the statistics fan-in block (ops.fan_block now hands the capture's block to its origin stream only) tensors
    allocated on one stream and last read on another can be recycled while the other branch still reads them -- one capture in three
    came back with gradients off by 1e-3..1e-2.  So a captured step with the switch on is the default step."""
    a = _grads("defer", True, size=128)
    X.ops.set_level_streams(True)
    try:
        b = _grads("defer", True, size=128)
        c = _grads("defer", False, size=128)
    finally:
        X.ops.set_level_streams(False)
    assert torch.isfinite(b).all() and torch.isfinite(c).all()
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 5e-4 * scale, (a - b).abs().max().item() / scale
    assert (a - c).abs().max().item() <= 5e-4 * scale, (a - c).abs().max().item() / scale


def _batch_vs_autograd(cases, k, stride, dtype, kernel):
    """Weight gradients of `cases` (N, Ca, Cb, Cout, groups, S, pre) through the deferred batch, against autograd's weight
    gradient of the same fp32 values and against the one-by-one path (xh_set_option(2, 512))."""
    import torch.nn.functional as F
    torch.manual_seed(11)
    lib = X._lib.load()
    probs = []
    for n, ca, cb, cout, g, s, pre in cases:
        so = s // stride
        xa = torch.randn(n, ca, s, s, s, device=DEV).to(dtype)
        xb = torch.randn(n, cb, s, s, s, device=DEV).to(dtype) if cb else None
        dy = torch.randn(n, cout, so, so, so, device=DEV).to(dtype)
        cin = ca + cb
        p = (torch.rand(n, cin, device=DEV) + 0.5, torch.randn(n, cin, device=DEV), 0.01) if pre else None
        probs.append((xa, xb, dy, g, p, cin, cout))

    def run():
        outs = []
        X.ops.set_wgrad_defer(True)
        try:
            for xa, xb, dy, g, p, cin, cout in probs:
                dws = [torch.zeros(cout // g, cin // g, k, k, k, device=DEV) for _ in range(g)]
                dbs = [torch.zeros(cout // g, device=DEV) for _ in range(g)]
                X.ops.conv3d_wgrad(xa, xb, dy, dws, dbs, k=k, stride=stride, groups=g, pre=p, side=True)
                outs.append((dws, dbs))
            X.ops.join_wgrad_stream()
        finally:
            X.ops.set_wgrad_defer(False)
        torch.cuda.synchronize()
        return outs

    got = run()
    assert kernel in X.ops.last_conv_kernel(), X.ops.last_conv_kernel()
    lib.xh_set_option(2, 512)
    try:
        single = run()
        assert "multi" not in X.ops.last_conv_kernel()
    finally:
        lib.xh_set_option(2, 0)
    for (xa, xb, dy, g, p, cin, cout), (dws, dbs), (sws, sbs) in zip(probs, got, single):
        x = torch.cat([xa, xb], 1).float() if xb is not None else xa.float()
        if p is not None:
            x = F.leaky_relu(x * p[0][:, :, None, None, None] + p[1][:, :, None, None, None], p[2])
        w = torch.zeros(cout, cin // g, k, k, k, device=DEV, requires_grad=True)
        b = torch.zeros(cout, device=DEV, requires_grad=True)
        F.conv3d(x, w, b, stride=stride, padding=k // 2, groups=g).backward(dy.float())
        dw, db = torch.cat(dws, 0), torch.cat(dbs, 0)
        sw, sb = w.grad.abs().max().item() + 1e-6, b.grad.abs().max().item() + 1e-6
        # the products are exact fp32 in every path; the sums differ in order
        assert (dw - w.grad).abs().max().item() <= 1e-3 * sw, ((dw - w.grad).abs().max().item() / sw, xa.shape)
        assert (db - b.grad).abs().max().item() <= 1e-3 * sb
        assert (dw - torch.cat(sws, 0)).abs().max().item() <= 1e-3 * sw
        assert (db - torch.cat(sbs, 0)).abs().max().item() <= 1e-3 * sb


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_k1_weight_gradients_of_a_batch_share_one_launch(dtype):
    """xh_conv3d_wgrad_batch sends the k = 1 problems of a step through conv1x1_wgrad_multi_kernel (one launch for up to 20
    problems): mixed shapes, groups with one weight tensor each, a channel-concatenated input, the producer's norm + LeakyReLU
    applied on load, batch 2."""
    #        N  Ca Cb Cout groups S   pre
    cases = [(1, 4, 0, 4, 1, 32, True), (2, 3, 5, 12, 1, 16, False), (1, 16, 0, 16, 4, 16, True), (1, 2, 0, 1, 1, 32, False),
             (1, 8, 8, 1, 1, 24, True), (2, 32, 0, 8, 1, 8, False)]
    _batch_vs_autograd(cases, 1, 1, dtype, "conv1x1_wgrad_multi_kernel")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_stride2_weight_gradients_of_a_batch_share_launches(dtype):
    """... and the k = 3 stride-2 problems (the DRB convs, RA_HVED.py:569: 4 streams as groups) through
    conv3_s2_wgrad_vec_multi_kernel, 4 problems per launch: 6 problems = two launches.  Groups of TWO output channels (the
    level-0 DRB conv, 16 -> 8 in 4 groups) share the launches of the groups of four since round 6."""
    cases = [(1, 16, 0, 16, 4, 32, True), (1, 32, 0, 32, 4, 16, True), (2, 8, 0, 4, 1, 32, False), (1, 8, 0, 8, 2, 64, True),
             (1, 4, 4, 8, 1, 32, False), (1, 16, 0, 8, 4, 32, True)]
    _batch_vs_autograd(cases, 3, 2, dtype, "conv3_s2_wgrad_vec_multi_kernel")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_k7_gate_weight_gradients_of_a_batch_share_one_launch(dtype):
    """... and the 7^3 gate convs of AttenModule2 (buildingblocks.py:283-296, collapsed to 4 -> 2 channels) through
    conv7_wgrad_mfma_multi_kernel + one second-stage reduction for all of them."""
    cases = [(1, 4, 0, 2, 1, 64, False), (1, 4, 0, 2, 1, 32, False), (2, 4, 0, 2, 1, 32, False)]
    _batch_vs_autograd(cases, 7, 1, dtype, "conv7_wgrad_mfma_multi_kernel")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cin", [1, 2])
def test_tiny_channel_k3_weight_gradients_of_a_batch_share_one_launch(dtype, cin):
    """... and the 1 -> 2 / 2 -> 1 channel 3^3 stencils (DuSE's spatial adjust convs, modules/DuSFE.py:113-155) through
    conv3_tiny_wgrad_multi_kernel."""
    cases = [(1, cin, 0, 3 - cin, 1, 64, False), (1, cin, 0, 3 - cin, 1, 32, False), (2, cin, 0, 3 - cin, 1, 16, False)]
    _batch_vs_autograd(cases, 3, 1, dtype, "conv3_tiny_wgrad_multi_kernel")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_deferred_batch_with_a_composition_used_by_two_decoder_passes(dtype):
    """forward_shared (train.py:224-225: two decoder passes over one encoder pass) uses every composed tensor (7^3 gates, DuSE,
    head) TWICE.  Their weight gradients accumulate into one buffer per composed tensor, so the deferred batch gives the plain
    step's gradients (autograd summing per-use gradients would read buffers the deferred launches have not written yet)."""
    m = X.XLSTM_HVED(1, 3, **X.TRAIN_KWARGS)
    m.load_state_dict(load("weights_seed1"), strict=True)
    m = m.to(DEV).train()
    torch.manual_seed(3)
    size = 32
    x = torch.rand(1, 4, size, size, size).to(DEV, dtype)
    h = size // 2
    eps = [torch.randn(1, 2 ** l, h >> l, h >> l, h >> l).to(DEV, dtype) for l in range(4)]
    fg = X.parallel.FlatGrads(m.parameters())

    def run(defer):
        X.ops.set_wgrad_defer(defer)
        try:
            fg.zero()
            outs = m.forward_shared(x, [dict(subset_idx_list=[14], eps_list=eps), dict(subset_idx_list=[6], eps_list=eps)], recon=True)
            loss = 0.0
            for k, (seg, (mu, lv), rec) in enumerate(outs):
                loss = loss + (1 + k) * (seg.float().mean() + rec[0].float().mean() + sum(a.float().mean() + b.float().mean() for a, b in zip(mu, lv)))
            loss.backward()
            X.ops.join_wgrad_stream()
            torch.cuda.synchronize()
            return fg.flat.clone()
        finally:
            X.ops.set_wgrad_defer(False)
    a, b = run(False), run(True)
    scale = a.abs().max().item()
    assert torch.isfinite(b).all() and scale > 0
    assert (a - b).abs().max().item() <= (2e-5 if dtype == torch.float32 else 2e-4) * scale, (a - b).abs().max().item() / scale

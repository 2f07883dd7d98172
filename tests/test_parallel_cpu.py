"""world_size-2 gloo test of the data-parallel gradient exchange (runs on CPU)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import xlstm_hved_amd as X
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    dead = torch.nn.Parameter(torch.ones(4))              # never reached: .grad stays None on every rank
    half = torch.nn.Parameter(torch.ones(2))              # has a gradient on rank 0 only
    x = torch.full((2, 5), float(rank + 1))
    loss = lin(x).sum() + (half.sum() if rank == 0 else 0.0)
    loss.backward()
    ar = X.parallel.FlatGradAllReduce([lin.weight, lin.bias, dead, half], world)
    ar()
    # flat bucket variant: gradients accumulate into views of one buffer, all-reduced in place
    lin2 = torch.nn.Linear(5, 3)
    dead2 = torch.nn.Parameter(torch.ones(4))
    fg = X.parallel.FlatGrads([lin2.weight, lin2.bias, dead2])
    for _ in range(2):                                    # second step checks zero() really resets the views
        fg.zero()
        lin2(x).sum().backward()
        fg.all_reduce(world)
    assert lin2.weight.grad.data_ptr() == fg.flat.data_ptr()
    # optimizer.zero_grad() (set_to_none=True by default; the reference calls it at train.py:264) detaches the views:
    # backward allocates fresh gradients, and all_reduce must gather them instead of reducing a stale bucket
    opt = torch.optim.SGD([lin2.weight, lin2.bias, dead2], lr=0.1)
    opt.zero_grad()
    lin2(x).sum().backward()
    assert lin2.weight.grad.data_ptr() != fg.flat.data_ptr() and not fg.check()
    fg.all_reduce(world)
    assert fg.check() and lin2.weight.grad.data_ptr() == fg.flat.data_ptr()
    # plain numpy payloads: a tensor in a Queue is shared by file descriptor and needs the sender alive when it is read
    q.put((rank, lin.weight.grad.numpy().copy(), lin.bias.grad.numpy().copy(), dead.grad,
           None if half.grad is None else half.grad.numpy().copy(),
           X.parallel.shard_windows(7, rank, world), lin2.weight.grad.numpy().copy(), dead2.grad.numpy().copy()))
    dist.destroy_process_group()


def test_flat_grad_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # d/dW of sum(W x + b) = sum over batch of x: rank r contributes 2*(r+1) per entry -> average (2+4)/2 = 3
    res = [tuple(torch.from_numpy(v) if hasattr(v, "dtype") else v for v in r) for r in res]
    for rank, wg, bg, dead, half, shard, wg2, dead2 in res:
        assert torch.allclose(wg, torch.full((3, 5), 3.0)) and torch.allclose(bg, torch.full((3,), 2.0))
        assert dead is None
        assert shard == list(range(rank, 7, 2))
        assert torch.allclose(wg2, torch.full((3, 5), 3.0)) and torch.equal(dead2, torch.zeros(4))
    assert torch.allclose(res[0][4], torch.full((2,), 0.5))     # rank 0: (1 + 0) / 2
    assert res[1][4] is None                                     # rank 1 had no gradient; its zeros were counted


def _ts_worker(rank, world, port, q):
    """TrainStep's data-parallel plumbing with the two halves of the step stubbed (no HIP needed): what is under test is which
    bucket is reduced when, the mean over ranks, the re-attachment of gradients and the consistent skip decision."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xlstm_hved_amd.train_step import TrainStep
    torch.manual_seed(0)
    gen, disc = torch.nn.Linear(4, 3), torch.nn.Linear(3, 2)
    opt_g, opt_d = torch.optim.SGD(gen.parameters(), lr=1.0), torch.optim.SGD(disc.parameters(), lr=1.0)
    ts = TrainStep(gen, disc, opt_g, opt_d, storage=torch.float16, group=dist.group.WORLD)
    assert ts.dp and ts.scaling and ts.loss_scale == 65536.0
    order = []

    def gen_half(x, mask, subset, eps_lists=None):
        ts.grads.zero(); ts.grads_d.zero()
        ts.grads.flat.fill_(float(rank + 1))                       # "generator backward" of this rank
        if subset == "overflow" and rank == 1:
            ts.grads.flat[0] = float("inf")                          # one rank overflows: every rank must skip
        order.append("g")
        return {"loss": torch.zeros(())}, None

    def disc_half(parts, carry):
        assert torch.allclose(ts.grads.flat[1:], torch.full_like(ts.grads.flat[1:], 1.5))     # already reduced (CPU: synchronous)
        ts.grads_d.flat.fill_(10.0 * (rank + 1))
        order.append("d")
        parts["loss_d"] = torch.zeros(())
        return parts
    ts.compute_generator, ts.compute_discriminator = gen_half, disc_half
    w0, wd0 = gen.weight.detach().clone(), disc.weight.detach().clone()
    parts = ts.step(None, None, "plain")
    g_after, d_after = ts.grads.flat.clone(), ts.grads_d.flat.clone()
    w1, wd1 = gen.weight.detach().clone(), disc.weight.detach().clone()
    parts2 = ts.step(None, None, "overflow")
    q.put((rank, order, g_after.numpy().copy(), d_after.numpy().copy(), (w0 - w1).numpy().copy(), (wd0 - wd1).numpy().copy(),
           parts2.get("skipped"), ts.loss_scale, (gen.weight.detach() - w1).abs().max().item(),
           (disc.weight.detach() - wd1).abs().max().item()))
    dist.destroy_process_group()


def test_train_step_buckets_all_reduced_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ts_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, order, g, d, dw, dwd, skipped, scale, gen_moved, disc_moved in res:
        assert order == ["g", "d", "g", "d"]
        assert (g == 1.5).all() and (d == 15.0).all()               # means of (1, 2) and (10, 20) on BOTH ranks
        assert (abs(dw - 1.5) < 1e-6).all() and (abs(dwd - 15.0) < 1e-6).all()     # SGD lr 1: the step applied the reduced gradients
        # the overflow of rank 1 reaches rank 0 through the sum: both skip the generator, neither skips the discriminator,
        # one back-off of the loss scale
        assert skipped == ["generator"] and scale == 32768.0
        assert gen_moved == 0.0 and disc_moved > 0.0

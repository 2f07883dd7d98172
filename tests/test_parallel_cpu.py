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

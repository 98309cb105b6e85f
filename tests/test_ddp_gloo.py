"""world_size-2 data-parallel plumbing on CPU (gloo): parameter broadcast, batch sharding and the
flat-arena gradient all-reduce must reproduce the single-process large-batch gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fovealseg import train


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, lr, w = train.ddp_setup(backend="gloo")
    assert (r, w) == (rank, world)
    net = _toy(seed=100 + rank)                       # ranks start from DIFFERENT weights
    opt = train.FlatAdam(list(net.parameters()), lr=1e-3, weight_decay=1e-4, lr_mult=0.001, zoom=False)
    train.broadcast_parameters([opt])                  # ...and must end up with rank 0's
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    idx = train.shard_indices(8, rank, world, shuffle=False)
    opt.zero_grad()
    loss = ((net(X[idx]) - Y[idx]) ** 2).mean()
    loss.backward()
    train.allreduce_gradients([opt])
    out[rank] = (opt.flat.data.clone(), opt.flat.grad.clone() * opt.grad_scale)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_matches_full_batch():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    ref = _toy(seed=100)
    opt = train.FlatAdam(list(ref.parameters()), lr=1e-3, weight_decay=1e-4, lr_mult=0.001, zoom=False)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    opt.zero_grad()
    ((ref(X) - Y) ** 2).mean().backward()
    for rank in range(world):
        data, grad = out[rank]
        assert torch.equal(data, opt.flat.data)                       # broadcast from rank 0
        assert torch.allclose(grad, opt.flat.grad, rtol=1e-5, atol=1e-7)   # mean of shard means == full mean

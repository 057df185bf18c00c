"""world_size-2 gloo test (CPU) of the data-parallel helper: sharded batch + ONE flat-bucket gradient all-reduce gives
exactly the single-process full-batch gradient, and two ranks stepping Adam stay bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def make_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3)).double()


def data():
    g = torch.Generator().manual_seed(7)
    return torch.randn(8, 6, generator=g, dtype=torch.float64), torch.randn(8, 3, generator=g, dtype=torch.float64)


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nns.data_parallel import FlatGradAllReduce, broadcast_parameters, shard
        model = make_model()
        if rank == 1:                                  # deliberately different start: broadcast must fix it
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        broadcast_parameters(list(model.parameters()))
        x, y = data()
        xs, ys = shard(x), shard(y)
        assert xs.shape[0] == 8 // world
        bucket = FlatGradAllReduce(model.parameters(), average=True)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        grads0 = None
        for step in range(3):
            bucket.zero_()
            loss = ((model(xs) - ys) ** 2).mean()      # mean over the LOCAL shard; averaging ranks = mean over the batch
            loss.backward()
            assert all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in model.parameters())     # still views of the bucket
            bucket.reduce_()
            if step == 0:
                grads0 = bucket.flat.clone()
            opt.step()
        torch.save({'g0': grads0, 'params': [p.detach().clone() for p in model.parameters()]}, os.path.join(out, 'r%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_flat_bucket_allreduce_world2(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in (0, 1))
    # single-process reference on the full batch
    model = make_model()
    x, y = data()
    ((model(x) - y) ** 2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert torch.allclose(r0['g0'], ref, rtol=1e-12, atol=1e-14)
    assert torch.equal(r0['g0'], r1['g0'])
    for a, b in zip(r0['params'], r1['params']):
        assert torch.equal(a, b)                       # replicas stay identical after 3 Adam steps


def test_flat_bucket_single_process_noop():
    from nns.data_parallel import FlatGradAllReduce, shard
    model = make_model()
    b = FlatGradAllReduce(model.parameters())
    x, y = data()
    ((model(x) - y) ** 2).mean().backward()
    before = b.flat.clone()
    assert torch.equal(b.reduce_(), before) and shard(x) is x
    with pytest.raises(ValueError):
        FlatGradAllReduce([torch.zeros(3)])

"""world_size-2 `gloo` tests of the view-sharded step's only exchange (gags_amd/dist.py)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, n, d, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gags_amd.dist import reduce_feature_grad, shard_views
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(n, d, generator=g)
    expect = sum(torch.randn(n, d, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    reduce_feature_grad(grad, mode=mode, bucket_bytes=4096)
    ok = torch.allclose(grad, expect, rtol=1e-6, atol=1e-6)
    views = shard_views(8)
    q.put((rank, bool(ok), views))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rs_ag", "allreduce"])
def test_feature_grad_reduction_two_ranks(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, 1037, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5, 7]


def test_single_process_is_noop():
    from gags_amd.dist import reduce_feature_grad, shard_views
    g = torch.arange(12.0).reshape(3, 4)
    assert reduce_feature_grad(g.clone()).equal(g)
    assert shard_views(5, rank=0, world_size=1) == [0, 1, 2, 3, 4]

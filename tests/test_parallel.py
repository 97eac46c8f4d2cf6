"""world_size-2 gloo tests (CPU): the data-parallel contract bench.py --mode train relies on.
SURVEY.md section 4 (iv): same seed, 1 vs k ranks -- gradients after the all-reduce equal the single-rank
gradients on the concatenated batch; every rank ends a step with identical weights; list sharding is a partition."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flownet2_amd import parallel  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(6, 4, generator=g, requires_grad=True), torch.randn(6, generator=g, requires_grad=True),
            torch.randn(3, 6, generator=g, requires_grad=True)]


def _loss(params, x, y):
    h = torch.nn.functional.leaky_relu(x @ params[0].t() + params[1], 0.1)
    return ((h @ params[2].t() - y) ** 2).mean()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _model(seed=100 + rank)                       # deliberately different -> broadcast must fix it
        parallel.broadcast_params(params, src=0)
        g = torch.Generator().manual_seed(7)
        X, Y = torch.randn(8, 4, generator=g), torch.randn(8, 3, generator=g)
        xs, ys = X[rank::world], Y[rank::world]               # each rank its own shard of the batch
        opt = torch.optim.Adam(params, lr=1e-2)
        for _ in range(3):
            opt.zero_grad()
            _loss(params, xs, ys).backward()
            parallel.allreduce_gradients(params, bucket_bytes=64)   # tiny buckets: exercise the bucketing
            opt.step()
        torch.save((rank, [p.detach().clone() for p in params], [p.grad.clone() for p in params],
                    parallel.shard(list(range(11))), parallel.max_over_ranks(float(rank), "cpu")), os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_dp2_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    # single-process reference: mean of the per-shard losses == the all-reduced, 1/world-scaled gradient
    params = _model(seed=100)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 4, generator=g), torch.randn(8, 3, generator=g)
    opt = torch.optim.Adam(params, lr=1e-2)
    for _ in range(3):
        opt.zero_grad()
        (sum(_loss(params, X[r::world], Y[r::world]) for r in range(world)) / world).backward()
        opt.step()
    for r in range(world):
        for a, b in zip(res[r][1], params):
            assert torch.allclose(a, b, atol=1e-6), "weights diverged from the single-process run"
        for a, b in zip(res[r][2], [p.grad for p in params]):
            assert torch.allclose(a, b, atol=1e-6)
    assert all(torch.equal(a, b) for a, b in zip(res[0][1], res[1][1])), "ranks must hold identical weights"
    assert sorted(res[0][3] + res[1][3]) == list(range(11)) and not set(res[0][3]) & set(res[1][3])
    assert res[0][4] == res[1][4] == 1.0          # max over ranks


def test_shard_without_process_group():
    assert parallel.world() == 1 and parallel.rank() == 0
    assert parallel.shard(list(range(5))) == [0, 1, 2, 3, 4]
    assert parallel.shard(list(range(5)), 1, 2) == [1, 3]

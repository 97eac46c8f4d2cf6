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


# ---------------------------------------------------------------------------------------------------------------------
# The real graph under 2 ranks: a small FlowNetC (128x128 pairs) + the multi-scale loss, gradients exchanged by
# GradientExchange (reverse-order buckets launched from gradient hooks), against a single-process evaluation.
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_train_backend():
    """Differentiable CPU backend for this test: the C oracle's forward AND backward restatements behind autograd
    (test infrastructure; the product path is flownet2_amd.functional on the GPU)."""
    import types

    import numpy as np

    import oracle
    from oracle import backend as fwd

    f32 = lambda t: t.detach().numpy().astype(np.float32, copy=False)

    class Corr(torch.autograd.Function):
        @staticmethod
        def forward(ctx, b0, b1, p):
            ctx.p = p
            ctx.save_for_backward(b0, b1)
            return torch.from_numpy(oracle.correlation_forward(p, f32(b0), f32(b1)))

        @staticmethod
        def backward(ctx, g):
            b0, b1 = ctx.saved_tensors
            d0, d1 = oracle.correlation_backward(ctx.p, f32(b0), f32(b1), f32(g.contiguous()))
            return torch.from_numpy(d0), torch.from_numpy(d1), None

    class L1(torch.autograd.Function):
        @staticmethod
        def forward(ctx, b0, b1, p):
            loss, ncoef = oracle.l1loss_forward(p, f32(b0), f32(b1))
            ctx.p, ctx.ncoef = p, ncoef
            ctx.save_for_backward(b0, b1)
            return torch.tensor(loss)

        @staticmethod
        def backward(ctx, g):
            b0, b1 = ctx.saved_tensors
            d0, _ = oracle.l1loss_backward(ctx.p, f32(b0), f32(b1), float(g), ctx.ncoef)
            return torch.from_numpy(d0), None, None

    be = types.SimpleNamespace()
    be.correlation = lambda b0, b1, pad=0, kernel_size=1, max_displacement=0, stride_1=1, stride_2=1: Corr.apply(
        b0.contiguous(), b1.contiguous(), oracle.corr_params(pad, kernel_size, max_displacement, stride_1, stride_2))
    be.l1_loss = lambda b0, b1, l2_per_location=False, normalize_by_num_entries=False: L1.apply(
        b0.contiguous(), b1, oracle.l1_params(l2_per_location, False, normalize_by_num_entries))
    be.downsample = fwd.downsample
    return be


def _flownet_batch(seed, nan_frac):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(1, 3, 128, 128, generator=g) - 0.4
    b = torch.roll(a, (1, -2), (2, 3)) + 0.01 * torch.randn(1, 3, 128, 128, generator=g)
    gt = torch.randn(1, 2, 128, 128, generator=g) * 4
    gt[:, :, :, :int(128 * nan_frac)] = float("nan")          # an occluded band (the Downsample vote turns i.i.d. NaNs above 1/3 into all-NaN tops)
    return a, b, gt


_NAN_FRAC = (0.02, 0.45)          # very different numbers of valid pixels per rank: normalize_by_num_entries is per rank


def _flownet_loss(P, batch, be):
    from flownet2_amd import nets
    a, b, gt = batch
    return nets.multiscale_loss(nets.flownet_c_core(P, a, b, be), gt, be)


def _flownet_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flownet2_amd import nets
        P = nets.init_params("C", seed=3 + rank)                 # different per rank: the broadcast must fix it
        for v in P.values():
            v.requires_grad_(True)
        plist = list(P.values())
        parallel.broadcast_params(plist, src=0)
        ex = parallel.GradientExchange(plist, bucket_bytes=24 << 20)
        assert len(ex.buckets) >= 4 and sum(b["flat"].numel() for b in ex.buckets) == nets.num_params(P)
        # reverse order: the first bucket holds the decoder's last layers, the last bucket conv1
        assert any(p is P["Convolution5.w"] for p in ex.buckets[0]["params"]) and any(p is P["conv1.w"] for p in ex.buckets[-1]["params"])
        be = _cpu_train_backend()
        ex.zero_grad()
        loss = _flownet_loss(P, _flownet_batch(50 + rank, _NAN_FRAC[rank]), be)
        loss.backward()
        launched_in_backward = ex.finish()
        torch.save({"loss": float(loss.detach()), "launched": launched_in_backward, "nb": len(ex.buckets),
                    "grads": {k: v.grad.clone() for k, v in P.items()}}, os.path.join(out, f"fn_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_dp2_flownetc_gradient_exchange_matches_single_process(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_flownet_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = [torch.load(os.path.join(str(tmp_path), f"fn_rank{r}.pt")) for r in range(world)]
    from flownet2_amd import nets
    torch.set_num_threads(4)
    be = _cpu_train_backend()
    P = nets.init_params("C", seed=3)
    for v in P.values():
        v.requires_grad_(True)
    # the reference's semantics (parallel.cpp:334-378): every solver normalises ITS loss by ITS number of valid entries, the
    # gradients are summed and scaled by 1/solver_count
    losses = [_flownet_loss(P, _flownet_batch(50 + r, _NAN_FRAC[r]), be) for r in range(world)]
    (sum(losses) / world).backward()
    for r in range(world):
        assert abs(res[r]["loss"] - float(losses[r].detach())) <= 1e-6 * max(1.0, abs(float(losses[r].detach())))
        assert res[r]["launched"] == res[r]["nb"]            # every bucket left from inside backward (hooks), none in finish()
        for k, v in P.items():
            s = max(1e-3, float(v.grad.abs().max()))
            assert float((res[r]["grads"][k] - v.grad).abs().max()) <= 2e-5 * s, k
    for k in P:
        assert torch.equal(res[0]["grads"][k], res[1]["grads"][k]), "ranks must end the exchange with identical gradients"
    # ... and that is NOT the gradient of one loss over the concatenated batch when the valid-pixel counts differ
    P2 = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    a = torch.cat([_flownet_batch(50 + r, _NAN_FRAC[r])[0] for r in range(world)])
    b = torch.cat([_flownet_batch(50 + r, _NAN_FRAC[r])[1] for r in range(world)])
    gt = torch.cat([_flownet_batch(50 + r, _NAN_FRAC[r])[2] for r in range(world)])
    _flownet_loss(P2, (a, b, gt), be).backward()
    k = "Convolution5.w"
    assert float((P2[k].grad - P[k].grad).abs().max()) > 1e-3 * float(P[k].grad.abs().max())


def test_gradient_exchange_accumulation_needs_no_sync():
    """A second backward() before finish() must not add into a bucket whose all-reduce has been launched (round-2 advisor finding):
    it raises; inside no_sync() the passes accumulate and the exchange leaves with finish()."""
    import torch
    from flownet2_amd import parallel
    torch.manual_seed(0)
    w = [torch.randn(4, 3, requires_grad=True), torch.randn(3, requires_grad=True)]
    ex = parallel.GradientExchange(w, bucket_bytes=1 << 20)
    x = torch.randn(5, 3)
    loss = lambda: ((x @ w[0].t()).sum() + (w[1] * 2).sum())
    ex.zero_grad()
    loss().backward()
    with pytest.raises(RuntimeError, match="no_sync"):
        loss().backward()
    ex.finish()
    # accumulation: two passes inside no_sync, the third outside launches from its hooks; gradient = 3 x one pass
    ex.zero_grad()
    loss().backward()
    ex.finish()
    one = [p.grad.clone() for p in w]
    ex.zero_grad()
    with ex.no_sync():
        loss().backward()
        loss().backward()
    loss().backward()
    assert ex.finish() == 1
    for g1, p in zip(one, w):
        assert torch.allclose(p.grad, 3 * g1)
    # everything inside no_sync: finish() launches
    ex.zero_grad()
    with ex.no_sync():
        loss().backward()
        loss().backward()
    ex.finish()
    for g1, p in zip(one, w):
        assert torch.allclose(p.grad, 2 * g1)
    ex.remove()

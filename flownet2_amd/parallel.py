"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  The hot path shards by image pair: inference needs NO collective; training
needs exactly one exchange per step, the sum-all-reduce of the gradients.

Replaces the reference's P2PSync (src/caffe/parallel.cpp:117-437): a hand-rolled CUDA peer-to-peer TREE with one host
thread per GPU that broadcasts the flat weight buffer down the tree every iteration (:287-322), reduces the flat
gradient buffer up the tree (:325-380), scales by 1/solver_count on the root (:377) and updates on the root only.
Here every rank applies the same update after the all-reduce, so no weight broadcast is needed after the initial one.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard(items: Sequence, r: int | None = None, w: int | None = None) -> List:
    """Round-robin shard of a work list (image pairs of a run-flownet-many list) for rank r of w."""
    r = rank() if r is None else r
    w = world() if w is None else w
    return [it for i, it in enumerate(items) if i % w == r]


def broadcast_params(params: Iterable[torch.Tensor], src: int = 0) -> None:
    """Initial weight broadcast (the reference repeats it every iteration, parallel.cpp:304-320)."""
    if world() == 1:
        return
    for p in params:
        dist.broadcast(p.data, src)


def allreduce_gradients(params: Sequence[torch.Tensor], bucket_bytes: int = 256 << 20) -> None:
    """Sum the gradients over ranks and scale by 1/world (parallel.cpp:377), in flat fp32 buckets.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, so few large
    buckets (default 256 MB: FlowNetC's 156.7 MB of gradients travel as ONE bucket) beat many small ones.
    Gradients stay fp32 (parity with the reference)."""
    w = world()
    if w == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0
    buckets = []
    for g in grads:
        nb = g.numel() * g.element_size()
        if bucket and size + nb > bucket_bytes:
            buckets.append(bucket)
            bucket, size = [], 0
        bucket.append(g)
        size += nb
    if bucket:
        buckets.append(bucket)
    works = []
    flats = []
    for b in buckets:
        flat = torch._utils._flatten_dense_tensors(b)
        flats.append((flat, b))
        works.append(dist.all_reduce(flat, async_op=True))
    for wk, (flat, b) in zip(works, flats):
        wk.wait()
        flat.mul_(1.0 / w)
        for g, s in zip(b, torch._utils._unflatten_dense_tensors(flat, b)):
            g.copy_(s)


def max_over_ranks(value: float, device) -> float:
    if world() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)

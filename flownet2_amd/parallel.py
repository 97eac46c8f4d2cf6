"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in
the CPU tests).  The hot path shards by image pair: inference needs NO collective; training
needs exactly one exchange per step, the sum-all-reduce of the gradients.

Replaces the reference's P2PSync (src/caffe/parallel.cpp:117-437): a hand-rolled CUDA peer-to-peer TREE with one host
thread per GPU that broadcasts the flat weight buffer down the tree every iteration (on_start, :287-322), reduces the flat
gradient buffer up the tree once the whole backward pass is over (on_gradients_ready, :325-380), scales by 1/solver_count
on the root (:377) and updates on the root only.  Here every rank applies the same update after the all-reduce, so no weight
broadcast is needed after the initial one, and the exchange is cut into buckets that leave while backward is still running
(GradientExchange).
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
    with torch.no_grad():
        for p in params:
            dist.broadcast(p, src)      # in place on the parameter itself: bumps p._version, which the packed-weight caches check
    from . import functional
    functional.invalidate_weight_caches()


class GradientExchange:
    """Bucketed sum-all-reduce of the fp32 gradients, overlapped with backward; result scaled by 1/world (parallel.cpp:377).

    * The parameters (given in forward order) are cut into flat fp32 buckets in REVERSE order -- backward produces the
      decoder's gradients first.  With world > 1 a gradient is copied into its bucket slot the moment it is produced and `.grad`
      becomes a view of that slot (one copy instead of zero-fill + accumulate; see zero_grad()); with one rank nothing is copied.
    * A post-accumulate hook per parameter counts its bucket down; the last gradient of a bucket launches that bucket's
      `all_reduce(async_op=True)`: with the "nccl" (= RCCL) backend it runs on the communicator's own HIP stream behind an
      event on the compute stream, so the exchange of bucket k overlaps the backward kernels of buckets k+1...
    * `finish()` (after `loss.backward()`) launches what has not been launched (parameters that got no gradient this step),
      waits, and scales by 1/world.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound and every collective pays a
    fixed latency, so buckets are few and large (default 48 MB: FlowNetC's 156.7 MB travel as 4 buckets; the first leaves
    after the decoder, ~25 % into backward).  Gradients stay fp32 (parity with the reference)."""

    def __init__(self, params: Sequence[torch.Tensor], bucket_bytes: int = 48 << 20, local_only: bool = False, wgrad_side_pixels: int = 36000):
        """local_only: keep the gradients of this rank (no collective, no bucket copies) although the process group has several ranks --
        the comparison step bench.py times to report how much of the all-reduce is NOT hidden behind backward.
        wgrad_side_pixels: in a single-rank JOB the weight gradients of maps up to this size run on a second HIP stream beside the
        data-gradient chain (functional.set_wgrad_side_stream; 0 = off).  With several ranks the gradient hooks below read every gradient the
        moment it is produced, so it stays off (also for the local_only comparison leg: the two legs must differ by the collective alone)."""
        self.params = [p for p in params if p.requires_grad]
        self.world = 1 if local_only else world()
        from . import functional
        functional.set_wgrad_side_stream(wgrad_side_pixels if world() == 1 else 0)
        self.launched_in_backward = 0       # buckets whose all-reduce left from a gradient hook during the last backward pass
        self.buckets: List[dict] = []
        cur, size = [], 0
        for p in reversed(self.params):
            nb = p.numel() * p.element_size()
            if cur and size + nb > bucket_bytes:
                self._close(cur)
                cur, size = [], 0
            cur.append(p)
            size += nb
        if cur:
            self._close(cur)
        self._index = {}
        self._handles = []
        self._defer = False
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._index[id(p)] = bi
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=ps[0].dtype, device=ps[0].device)
        views, off = {}, 0
        for p in ps:
            views[id(p)] = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.buckets.append({"params": ps, "flat": flat, "views": views, "pending": len(ps), "work": None, "launched": False})

    def reset(self):
        for b in self.buckets:
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False

    def zero_grad(self):
        """Start an iteration: the gradients are dropped (None), not zero-filled.  The first backward pass then hands its gradient
        tensors to the parameters as they are (autograd's AccumulateGrad moves the tensor in: no fill of 157 MB, no `grad += new` pass
        over them -- 50 + 17 launches and 0.4 ms of a FlowNetC step at batch 8), and the gradient hook copies each into its slot of the
        flat bucket and re-points `p.grad` at that slot (world > 1: the all-reduce needs them contiguous).  Further backward passes of
        the same iteration (no_sync) find a gradient in place and accumulate into it -- into the bucket slot, where there is one."""
        for b in self.buckets:
            for p in b["params"]:
                p.grad = None
        self.reset()

    def _launch(self, b):
        b["launched"] = True
        if self.world > 1:
            b["work"] = dist.all_reduce(b["flat"], async_op=True)

    def no_sync(self):
        """Context manager for gradient ACCUMULATION (iter_size > 1 in the reference's solver: several backward passes before
        on_gradients_ready, solver.cpp:221-226): backward passes inside it only accumulate into the buckets; the exchange is launched
        by finish() -- or by the hooks of the first backward pass after the context, which must be the last before finish()."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._defer = self._defer, True
            try:
                yield self
            finally:
                self._defer = prev
                self.reset()                # the passes inside the context launched nothing: count the next pass from scratch
        return ctx()

    def _on_grad(self, p):
        b = self.buckets[self._index[id(p)]]
        if b["launched"]:
            raise RuntimeError("GradientExchange: a gradient arrived for a bucket whose all-reduce is already in flight (a second "
                               "backward() before finish()); wrap accumulation passes in no_sync()")
        if self.world > 1:
            view = b["views"][id(p)]
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
        if self._defer:
            return
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self):
        from . import functional
        functional.join_side_streams()          # (already done by the engine's end-of-pass callback; idempotent)
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["flat"].mul_(1.0 / self.world)
        n = sum(1 for b in self.buckets if b["pending"] == 0)
        self.launched_in_backward = n
        self.reset()
        return n           # buckets whose exchange was launched from inside backward

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def allreduce_gradients(params: Sequence[torch.Tensor], bucket_bytes: int = 256 << 20) -> None:
    """Post-backward form of the exchange (no overlap): sum over ranks, scale by 1/world, in flat fp32 buckets."""
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


def ranks_seen(device) -> int:
    """One all-reduce of ones: the number of ranks the collective library actually connected."""
    if world() == 1:
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(t)
    return int(round(float(t)))

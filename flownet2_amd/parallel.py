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

from . import functional


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
    functional.invalidate_weight_caches()


class GradientExchange:
    """Bucketed sum-all-reduce of the fp32 gradients, overlapped with backward; result scaled by 1/world (parallel.cpp:377).

    * The parameters (given in forward order) are cut into flat fp32 buckets in REVERSE order -- backward produces the
      decoder's gradients first.  With a collective the weight-gradient kernels write their result straight into the bucket slot
      (functional.set_grad_slots: `.grad` becomes a view of the slot, nothing is copied); the few gradients that are not produced
      by those kernels (biases, the 2-channel flow heads: 0.1 % of the bytes) are copied into their slots by the hook.  Without a
      collective (one rank) nothing is flattened or copied.
    * A post-accumulate hook per parameter counts its bucket down; the last gradient of a bucket launches that bucket's
      `all_reduce(async_op=True)`: with the "nccl" (= RCCL) backend it runs on the communicator's own HIP stream behind an
      event on the stream it is launched under, so the exchange of bucket k overlaps the backward kernels of buckets k+1...
    * The weight gradients run on a second HIP stream beside the data-gradient chain (functional.set_wgrad_side_stream) with ANY
      number of ranks: a bucket's all-reduce is launched under that second stream after it has been made to wait for the main
      stream (functional.side_stream_for_collective), i.e. behind every kernel that writes into the bucket, whichever stream ran it.
    * `finish()` (after `loss.backward()`) launches what has not been launched (parameters that got no gradient this step),
      waits, and scales by 1/world.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound and every collective pays a
    fixed latency, so buckets are few and large (default 48 MB: FlowNetC's 156.7 MB travel as 4 buckets; the first leaves
    after the decoder, ~25 % into backward).  Gradients stay fp32 (parity with the reference)."""

    def __init__(self, params: Sequence[torch.Tensor], bucket_bytes: int = 48 << 20, local_only: bool = False, wgrad_side_pixels: int = 36000,
                 force_collective: bool = False):
        """local_only: keep the gradients of this rank (no collective, no buckets) although the process group has several ranks --
        the comparison step bench.py times to report how much of the all-reduce is NOT hidden behind backward.
        wgrad_side_pixels: the weight gradients of maps up to this size run on a second HIP stream beside the data-gradient chain
        (functional.set_wgrad_side_stream; 0 = off) -- with one rank and with several.
        force_collective: run the full bucket path (slots, hooks, all_reduce on the collective library's stream, wait, scale) although the
        process group has ONE rank -- the GPU test that executes RCCL on a single-GPU box (tests/test_parallel.py)."""
        self.params = [p for p in params if p.requires_grad]
        self.world = 1 if local_only else world()
        self.collective = (not local_only) and (world() > 1 or (force_collective and dist.is_available() and dist.is_initialized()))
        functional.set_wgrad_side_stream(wgrad_side_pixels)
        self.launched_in_backward = 0       # buckets whose all-reduce left from a gradient hook during the last backward pass
        self.copied_bytes = 0               # gradient bytes the hooks copied into their slots during the last iteration (not produced there)
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
        functional.set_grad_slots(self._slot if self.collective else None)
        self.reset()

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        # (one rank, no collective: the bucket is bookkeeping only -- no flat buffer is allocated)
        flat = torch.zeros(n if self.collective else 0, dtype=ps[0].dtype, device=ps[0].device)
        span, off = {}, 0
        for p in ps:
            span[id(p)] = (off, p.numel())
            off += p.numel()
        self.buckets.append({"params": ps, "flat": flat, "span": span, "pending": len(ps), "work": None, "launched": False})

    def _slot(self, p):
        """A fresh view of p's slot in its bucket (fresh: autograd moves a gradient into `.grad` only when nobody else holds the tensor)."""
        bi = self._index.get(id(p))
        if bi is None:
            return None
        b = self.buckets[bi]
        off, n = b["span"][id(p)]
        return b["flat"][off:off + n].view(p.shape)

    def reset(self):
        for b in self.buckets:
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False

    def zero_grad(self):
        """Start an iteration: the gradients are dropped (None), not zero-filled.  The first backward pass then hands its gradient
        tensors to the parameters as they are (autograd's AccumulateGrad moves the tensor in: no fill of 157 MB, no `grad += new` pass
        over them -- 50 + 17 launches and 0.4 ms of a FlowNetC step at batch 8); with a collective that tensor already IS the slot of the
        flat bucket (see _slot).  Further backward passes of the same iteration (no_sync) find a gradient in place and accumulate into
        it -- into the bucket slot."""
        for b in self.buckets:
            for p in b["params"]:
                p.grad = None
        self.copied_bytes = 0
        self.reset()

    def _launch(self, b):
        b["launched"] = True
        if not self.collective:
            return
        flat = b["flat"]
        side = functional.side_stream_for_collective(flat.device) if flat.is_cuda else None
        if side is None:
            b["work"] = dist.all_reduce(flat, async_op=True)
        else:
            # weight gradients of this bucket may still be running on the second stream: the collective is ordered behind THAT stream
            # (which has just been made to wait for the main one), and the main stream -- the data-gradient chain -- is not held up
            with torch.cuda.stream(side):
                b["work"] = dist.all_reduce(flat, async_op=True)
            flat.record_stream(side)

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
        if self.collective and p.grad is not None:
            off, n = b["span"][id(p)]
            if p.grad.data_ptr() != b["flat"].data_ptr() + off * b["flat"].element_size():
                view = self._slot(p)            # not produced in its slot (a bias, a flow head, a library fallback): one copy
                side = functional.side_stream_for_collective(view.device) if view.is_cuda else None
                if side is None:
                    view.copy_(p.grad)
                else:
                    # the gradient may have been computed on the second stream (the flow heads' weight gradients): the copy is ordered behind
                    # BOTH streams by running on the second one, which has just been made to wait for the main one
                    with torch.cuda.stream(side):
                        view.copy_(p.grad)
                    p.grad.record_stream(side)
                p.grad = view
                self.copied_bytes += n * b["flat"].element_size()
        if self._defer:
            return
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self):
        n = sum(1 for b in self.buckets if b["pending"] == 0)          # buckets whose exchange was launched from inside backward
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b)
        functional.join_side_streams()          # (already done by the engine's end-of-pass callback; idempotent)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()                # the current stream waits for the collective's stream
                if self.world > 1:
                    b["flat"].mul_(1.0 / self.world)
        self.launched_in_backward = n
        self.reset()
        return n

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        functional.set_grad_slots(None)


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
    """One all-reduce of ones: the number of ranks the collective library actually connected.  Without a process group there is no
    collective library in the job: 1 by definition (bench.py then says `"collective": "none (world 1)"`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    t = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(t)
    return int(round(float(t)))

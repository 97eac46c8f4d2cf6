#!/usr/bin/env python
"""The training input pipeline of the reference, end to end on one MI355X, with synthetic records:

  LMDB values (Datum: 2 x uint8 BGR image + int16 flow + 1-bit occlusions)  --CustomData-->  img0, img1, flow, occ          [GPU decode]
  img * 1/255  --DataAugmentation(img0; coefficients drawn on the host)-->  img0_aug, params0                               [one kernel]
  params0  --GenerateAugmentationParameters(add)-->  params1 ;  img1  --DataAugmentation(params1)-->  img1_aug               [host + one kernel]
  flow  --FlowAugmentation(params0, params1)-->  flow_aug                                                                    [one kernel]
  FlowNetC forward + multi-scale L1 loss + backward + Adam (the step bench.py --mode train times)

Prints the time per stage (HIP events, median of the timed iterations).  The coefficient DRAWS come from a counter-based generator
(Philox4x32-10, flownet2_amd/augment.py: the reference's boost stream cannot be reproduced, its distributions are pinned): the draws
of iteration i are a function of (seed, i), so a worker PROCESS (augment.CoefficientPrefetcher(process=True)) produces them ahead of the
step and the "draw" stages below are what the step WAITS for them; --prefetch-thread uses a background thread instead (no gain: it competes
with the step's own Python for the interpreter lock), --no-prefetch draws inline (the r02 behaviour).  Everything downstream of
the coefficient blobs is pinned against the reference's layers.
Usage: python scripts/train_pipeline.py [--batch 8] [--iters 10] [--no-train] [--no-prefetch]"""
import argparse
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import augment, nets, ops, sample_format as SF   # noqa: E402
from flownet2_amd import functional as Fn                           # noqa: E402
from flownet2_amd.layers import Blob, LayerParameter, LayerRegistry  # noqa: E402

AUG0 = dict(mirror=dict(rand_type="bernoulli", prob=0.5), translate=dict(rand_type="uniform_bernoulli", mean=0, spread=0.4, prob=1.0),
            rotate=dict(rand_type="uniform_bernoulli", mean=0, spread=0.4, prob=1.0), zoom=dict(rand_type="uniform_bernoulli", exp=True, mean=0.2, spread=0.4, prob=1.0),
            squeeze=dict(rand_type="uniform_bernoulli", exp=True, mean=0, spread=0.3, prob=1.0), gamma=dict(rand_type="uniform_bernoulli", exp=True, mean=0, spread=0.02, prob=1.0),
            brightness=dict(rand_type="gaussian_bernoulli", mean=0, spread=0.02, prob=1.0), contrast=dict(rand_type="uniform_bernoulli", exp=True, mean=0, spread=0.4, prob=1.0),
            color=dict(rand_type="gaussian_bernoulli", exp=True, mean=0, spread=0.02, prob=1.0))
AUG1 = dict(translate=dict(rand_type="gaussian_bernoulli", mean=0, spread=0.03, prob=1.0), rotate=dict(rand_type="gaussian_bernoulli", mean=0, spread=0.03, prob=1.0),
            zoom=dict(rand_type="gaussian_bernoulli", exp=True, mean=0, spread=0.03, prob=1.0), gamma=dict(rand_type="gaussian_bernoulli", exp=True, mean=0, spread=0.02, prob=1.0))


def synthetic_records(n, H, W, seed=0):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        b = np.roll(a, (2, -3), (0, 1))
        f = np.stack([np.full((H, W), -3.0, np.float32), np.full((H, W), 2.0, np.float32)]) + rng.normal(0, 0.2, (2, H, W)).astype(np.float32)
        f[:, rng.random((H, W)) < 0.02] = np.nan
        recs.append(("%08d_pair%d" % (i, i), SF.make_record(a, b, f, rng.random((H, W)) < 0.1, label=i)))
    return recs


def draw_pair(it, B, W, H, cw, ch):
    """Both coefficient blobs of iteration `it`: a pure function of (seed, it) (module level: the prefetch process pickles it)."""
    p0 = augment.draw_batch(augment.make_rng(1, 2 * it), AUG0, B, W, H, cw, ch, discount=augment.discount_coeff(it + 1))
    p1 = augment.draw_batch(augment.make_rng(1, 2 * it + 1), AUG1, B, W, H, cw, ch, discount=1.0, in_params=p0, mode="add")
    return p0, p1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--crop-height", type=int, default=320)
    ap.add_argument("--crop-width", type=int, default=448)
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-prefetch", action="store_true")
    ap.add_argument("--prefetch-thread", action="store_true", help="draw on a background thread instead of a worker process")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B, H, W, ch, cw = a.batch, a.height, a.width, a.crop_height, a.crop_width
    recs = synthetic_records(2 * B, H, W)
    data = LayerRegistry.CreateLayer(LayerParameter(type="CustomData", data_param=dict(source=recs, backend="LMDB", batch_size=B, slice_point=[3, 6, 8],
                                                                                          encoding=["UINT8", "UINT8", "UINT16FLOW", "BOOL1"])))
    dtop = [Blob() for _ in range(4)]
    data.SetUp([], dtop)
    aug_p = dict(crop_width=cw, crop_height=ch, mean=[0.411, 0.433, 0.45], mean_per_pixel=False)
    aug0 = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", augmentation_param=aug_p))
    aug1 = LayerRegistry.CreateLayer(LayerParameter(type="DataAugmentation", augmentation_param=aug_p))
    faug = LayerRegistry.CreateLayer(LayerParameter(type="FlowAugmentation", augmentation_param=dict(crop_width=cw, crop_height=ch)))
    import functools
    draw = functools.partial(draw_pair, B=B, W=W, H=H, cw=cw, ch=ch)
    pre = None if a.no_prefetch else augment.CoefficientPrefetcher(draw, depth=4, process=not a.prefetch_thread)
    P = {k: v.to(dev).requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-5, fused=True)
    stages = ["decode", "scale", "draw0 (host)", "augment0", "draw1 (host)", "augment1", "flow_aug", "train step"]
    times = {s: [] for s in stages}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    setup_done = False
    for it in range(a.iters + 3):
        marks = [ev() for _ in range(len(stages) + 1)]
        host_ms = {}
        marks[0].record()
        data.Forward([], dtop)
        marks[1].record()
        img0, img1 = dtop[0].data * (1.0 / 255.0), dtop[1].data * (1.0 / 255.0)
        marks[2].record()
        t0 = time.perf_counter()
        p0, p1 = pre.get() if pre is not None else draw(it)
        host_ms["draw0 (host)"] = (time.perf_counter() - t0) * 1e3
        marks[3].record()
        b0 = [Blob.from_tensor(img0), Blob.from_tensor(torch.from_numpy(p0).view(B, 42, 1, 1))]
        b0[1].data = torch.from_numpy(p0).view(B, 42, 1, 1)          # coefficient blobs live on the host side, as the reference reads them
        t_img0, t_img1, t_flow = [Blob()], [Blob()], [Blob()]
        if not setup_done:
            aug0.SetUp(b0, t_img0)
        aug0.Forward(b0, t_img0)
        marks[4].record()
        host_ms["draw1 (host)"] = 0.0          # drawn together with p0
        marks[5].record()
        b1 = [Blob.from_tensor(img1), Blob.from_tensor(torch.zeros(1))]
        b1[1].data = torch.from_numpy(p1).view(B, 42, 1, 1)
        b1[1]._shape = [B, 42, 1, 1]
        if not setup_done:
            aug1.SetUp(b1, t_img1)
        aug1.Forward(b1, t_img1)
        marks[6].record()
        bf = [Blob.from_tensor(dtop[2].data), b0[1], b1[1]]
        if not setup_done:
            faug.SetUp(bf, t_flow)
            setup_done = True
        faug.Forward(bf, t_flow)
        marks[7].record()
        if not a.no_train:
            opt.zero_grad(set_to_none=True)          # gradients handed over, not zero-filled + accumulated (as bench.py --mode train)
            loss = nets.multiscale_loss(nets.flownet_c_core(P, t_img0[0].data, t_img1[0].data, Fn), t_flow[0].data, Fn)
            loss.backward()
            opt.step()
        marks[8].record()
        torch.cuda.synchronize()
        if it >= 3:
            for i, s in enumerate(stages):
                times[s].append(host_ms.get(s, marks[i].elapsed_time(marks[i + 1])))
    print("| stage | ms per batch of %d (%dx%d -> %dx%d) |\n|---|---|" % (B, W, H, cw, ch))
    for s in stages:
        print("| %s | %.3f |" % (s, statistics.median(times[s])))
    if pre is not None:
        pre.close()
    how = "inline (draw0 = both blobs)" if pre is None else ("prefetch %s, 4 iterations deep (draw0 = time the step waited)" % ("thread" if a.prefetch_thread else "process"))
    print("coefficient draws: %s" % how)
    print("iteration (sum of the stages): %.3f ms" % sum(statistics.median(times[s_]) for s_ in stages))
    if not a.no_train:
        print("loss %.4f, NaN ground truth kept: %s" % (float(loss), bool(torch.isnan(t_flow[0].data).any())))


if __name__ == "__main__":
    main()

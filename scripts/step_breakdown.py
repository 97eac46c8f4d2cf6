#!/usr/bin/env python
"""Per-layer GPU time of one deploy forward as the graph routes it (own kernels, GEMM route, library): wraps nets._conv_routed / nets._deconv
and the backend's flow heads / correlation with HIP events.  Each layer is timed in isolation (synchronised), so the sum exceeds the
pipelined step a little; the point is the ranking.
    python scripts/step_breakdown.py [--net C|2] [--batch 8 --height 320 --width 448] [--iters 10]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flownet2_amd import functional as Fn, nets  # noqa: E402

T = collections.OrderedDict()


def timed(label_fn, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = fn(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        lab = label_fn(a, k, y)
        T.setdefault(lab, []).append(e0.elapsed_time(e1) * 1e3)
        return y
    return wrapper


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="C")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=448)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    P = nets.init_params_flownet2(0, dev) if a.net == "2" else nets.init_params(a.net, 0, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    i0 = torch.rand(a.batch, 3, a.height, a.width, device=dev, generator=g) * 255
    i1 = torch.rand(a.batch, 3, a.height, a.width, device=dev, generator=g) * 255

    def conv_label(args, k, y):
        x, PP, name = args[0], args[1], args[2]
        w = PP[name + ".w"]
        gf = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3] * w.shape[1] * w.shape[2] * w.shape[3] / 1e9
        pre = getattr(PP, "prefix", "") if hasattr(PP, "prefix") else ""
        return "%s%-12s %-20s k%ds%d %7.2f GF  %s" % (pre, name, tuple(x.shape), w.shape[2], args[3], gf, nets._LAST_ROUTE[0])

    def deconv_label(args, k, y):
        x, PP, name = args[0], args[1], args[2]
        w = PP[name + ".w"]
        gf = 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * w.shape[0] * w.shape[1] * 16 / 1e9
        return "%-12s %-20s deconv %7.2f GF" % (name, tuple(x.shape), gf)

    nets._conv_routed = timed(conv_label, nets._conv_routed)
    nets._deconv = timed(deconv_label, nets._deconv)

    def into_label(args, k, y):
        x, PP, name = args[0], args[1], args[2]
        w = PP[name + ".w"]
        o = y[1]
        gf = 2.0 * o.shape[0] * o.shape[1] * o.shape[2] * o.shape[3] * w.shape[1] * w.shape[2] * w.shape[3] / 1e9
        return "%-12s %-20s k%ds%d %7.2f GF  %s" % (name, tuple(x.shape), w.shape[2], args[3], gf, "into concat blob" if y[0] is not None else "-> " + nets._LAST_ROUTE[0])

    nets._conv_into_concat = timed(into_label, nets._conv_into_concat)
    Fn.deconv_mfma_relu = timed(lambda args, k, y: "%-12s %-20s -> %d  %s" % ("deconv_mfma", tuple(args[0].shape), args[1].shape[1], "own kernel" if y is not None else "(declined)"),
                                Fn.deconv_mfma_relu)
    for nm in ("predict_flow_conv", "upsample_flow_deconv", "correlation_relu_into", "resample", "flow_warp", "channel_norm"):
        if hasattr(Fn, nm):
            setattr(Fn, nm, timed(lambda args, k, y, nm=nm: "%-12s %s" % (nm, tuple(args[0].shape)), getattr(Fn, nm)))
    cat = torch.cat
    torch.cat = timed(lambda args, k, y: "torch.cat -> %s" % (tuple(y.shape),), cat)
    fwd = (lambda: nets.flownet2_deploy_forward(P, i0, i1, Fn)) if a.net == "2" else (lambda: nets.deploy_forward(a.net, P, i0, i1, Fn))
    with torch.no_grad():
        for _ in range(3):
            fwd()
        T.clear()
        for _ in range(a.iters):
            fwd()
    tot = 0.0
    for lab, v in T.items():
        per_step = sum(v) / a.iters
        tot += per_step
        print("%8.1f us  x%-2d %s" % (per_step, len(v) // a.iters, lab))
    print("%8.1f us  sum of the wrapped calls" % tot)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Does MIOpen's own tuning (MIOPEN_FIND_ENFORCE=4 with a user perf-db) speed up the weak FlowNetC layers on gfx950?"""
import os, sys, time, torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
layers = {"deconv3": ("deconv", 770, 128, 20, 28, 4, 2, 1), "conv5": ("conv", 512, 512, 20, 28, 3, 2, 1),
          "conv6_1": ("conv", 1024, 1024, 5, 7, 3, 1, 1), "deconv2": ("deconv", 386, 64, 40, 56, 4, 2, 1)}
which = sys.argv[1:] or list(layers)
for name in which:
    kind, ci, co, h, w, k, s, p = layers[name]
    x = torch.randn(8, ci, h, w, device="cuda")
    wt = torch.randn((co, ci, k, k) if kind == "conv" else (ci, co, k, k), device="cuda") * 0.01
    f = (lambda: F.conv2d(x, wt, None, stride=s, padding=p)) if kind == "conv" else (lambda: F.conv_transpose2d(x, wt, None, stride=2, padding=1))
    t0 = time.time(); f(); torch.cuda.synchronize(); first = time.time() - t0
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("%-8s first call %.1f s, steady %.1f us" % (name, first, e0.elapsed_time(e1) / 20 * 1e3), flush=True)

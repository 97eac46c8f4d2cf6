#!/usr/bin/env python
"""Interleaved A/B timing of correlation-forward variants (fn2_debug_set_correlation_impl codes) in one process:
boxes and DVFS state differ between runs, so only within-run, interleaved comparisons are meaningful."""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops, _lib
impls = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 3]
shape = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (8, 256, 40, 56)
N, C, H, W = shape
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, C, H, W, device="cuda", generator=g); y = torch.randn(N, C, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
out = torch.empty(N, 441, H, W, device="cuda")
def timeit(iters=100):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.correlation_forward(p, x, y, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
res = {i: [] for i in impls}
for i in impls:
    _lib.lib().fn2_debug_set_correlation_impl(i); timeit(50)
for rnd in range(8):
    for i in (impls if rnd % 2 == 0 else impls[::-1]):
        _lib.lib().fn2_debug_set_correlation_impl(i)
        res[i].append(timeit())
_lib.lib().fn2_debug_set_correlation_impl(0)
for i in impls:
    print("impl %3d: min %.2f  median %.2f  max %.2f us" % (i, min(res[i]), statistics.median(res[i]), max(res[i])))

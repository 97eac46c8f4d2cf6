#!/usr/bin/env python
"""Is the correlation kernel running against the power budget?  Same launch, inputs of different toggle activity."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops, _lib
N, C, H, W = 8, 256, 40, 56
p = ops.corr_params(20, 1, 20, 1, 2)
out = torch.empty(N, 441, H, W, device="cuda")
def timeit(x, y, iters=300):
    for _ in range(20): ops.correlation_forward(p, x, y, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.correlation_forward(p, x, y, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for impl in (0, 2):
    _lib.lib().fn2_debug_set_correlation_impl(impl)
    g = torch.Generator(device="cuda").manual_seed(0)
    xr = torch.randn(N, C, H, W, device="cuda", generator=g); yr = torch.randn(N, C, H, W, device="cuda", generator=g)
    z = torch.zeros_like(xr); o = torch.ones_like(xr)
    print("impl", impl, " randn %.2f us   zeros %.2f us   ones %.2f us   relu(randn) %.2f us" %
          (timeit(xr, yr), timeit(z, z), timeit(o, o), timeit(xr.relu(), yr.relu())))

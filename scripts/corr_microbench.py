#!/usr/bin/env python
"""Correlation-only micro-benchmark (the kernel bench.py's `roofline` object is quoted on).
Used under rocprofv3:  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python scripts/corr_microbench.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="8,256,40,56")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--generic", action="store_true")
ap.add_argument("--impl", type=int, default=0, help="fn2_debug_set_correlation_impl value (3 = general dword LDS-DMA MFMA kernel)")
ap.add_argument("--backward", action="store_true")
ap.add_argument("--ablation", type=int, default=0, help="FN2_ABLATION builds: 1 no MFMA, 2 no loads, 4 no stores (bit-or)")
a = ap.parse_args()
N, C, H, W = map(int, a.shape.split(","))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, C, H, W, device="cuda", generator=g)
y = torch.randn(N, C, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
out = torch.empty(N, 441, H, W, device="cuda")
ops.set_correlation_impl(a.generic)
if a.ablation:
    from flownet2_amd import _lib
    _lib.lib().fn2_debug_set_correlation_impl(64 + a.ablation)
elif a.impl:
    from flownet2_amd import _lib
    _lib.lib().fn2_debug_set_correlation_impl(a.impl)
for _ in range(5):
    ops.correlation_forward(p, x, y, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    ops.correlation_forward(p, x, y, out=out)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / a.iters * 1e-3
fl = 2.0 * C * 441 * N * H * W
by = 4.0 * N * H * W * (2 * C + 441)
print(f"corr fwd [{N},{C},{H},{W}] {'generic' if a.generic else 'mfma'}: {t*1e6:.2f} us/launch  {fl/t/1e12:.2f} TFLOP/s(alg)  {by/t/1e9:.1f} GB/s(alg)")
if a.backward:
    gg = torch.randn(out.shape, device="cuda", generator=g)
    for _ in range(2):
        ops.correlation_backward(p, x, y, gg)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(max(1, a.iters // 5)):
        ops.correlation_backward(p, x, y, gg)
    e1.record()
    torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / max(1, a.iters // 5) * 1e-3
    print(f"corr bwd: {tb*1e6:.2f} us/call (both bottoms)  {2*fl/tb/1e12:.2f} TFLOP/s(alg)")

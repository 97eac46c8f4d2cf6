#!/usr/bin/env python
"""Quick GPU self-check of the correlation fast paths against the generic kernel (max abs difference per shape)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops, _lib
SHAPES = [(1, 16, 13, 17, 20, 2), (2, 64, 24, 40, 20, 2), (1, 256, 16, 24, 20, 2), (1, 32, 9, 30, 8, 1), (1, 16, 11, 13, 8, 2),
          (2, 32, 10, 35, 4, 1), (1, 32, 41, 57, 20, 2), (1, 16, 9, 70, 21, 2), (1, 32, 12, 72, 20, 2), (8, 256, 40, 56, 20, 2), (4, 256, 48, 96, 20, 2), (1, 256, 56, 128, 20, 2), (3, 32, 20, 28, 20, 2), (2, 16, 7, 8, 20, 2)]
bad = 0
for (N, C, H, W, md, s2) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, C, H, W, device="cuda", generator=g); y = torch.randn(N, C, H, W, device="cuda", generator=g)
    p = ops.corr_params(md, 1, md, 1, s2)
    ops.set_correlation_impl(True); ref = ops.correlation_forward(p, x, y); ops.set_correlation_impl(False)
    res = {}
    for impl in (0, 3):
        _lib.lib().fn2_debug_set_correlation_impl(impl)
        worst = 0.0
        for rep in range(3):
            out = ops.correlation_forward(p, x, y)
            worst = max(worst, (out - ref).abs().max().item())
        res[impl] = worst
    _lib.lib().fn2_debug_set_correlation_impl(0)
    ok = all(v < 2e-6 for v in res.values()); bad += not ok
    print((N, C, H, W, md, s2), "auto %.2e  general (dword LDS-DMA) %.2e" % (res[0], res[3]), "OK" if ok else "MISMATCH")
sys.exit(1 if bad else 0)

"""Resample LINEAR at unit tap scale (identity / non-integer sizes): the lean kernel against the per-output-pixel kernel, kernel times from
HIP events around 200 back-to-back launches into a preallocated top (no allocator, no Python in the loop beyond the ctypes call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops  # noqa: E402


def t(f, it=200):
    for _ in range(50):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


for (N, C, H, W, Ho, Wo) in [(4, 3, 384, 768, 384, 768), (4, 3, 436, 1024, 448, 1024), (1, 3, 375, 1242, 384, 1280), (4, 2, 96, 192, 384, 768), (32, 3, 384, 768, 384, 768)]:
    x = torch.randn(N, C, H, W, device="cuda")
    mb = 4 * N * C * (H * W + Ho * Wo) / 1e6
    res = []
    for generic in (False, True):
        ops.set_resample_generic(generic)
        us = t(lambda: ops.resample_forward(x, Ho, Wo))
        res.append("%s %.1f us %.0f GB/s" % ("per-pixel" if generic else "fast path", us, mb / us * 1e3))
    ops.set_resample_generic(False)
    y = x.clone()
    us = t(lambda: y.copy_(x))
    print("[%d,%d,%d,%d]->[%d,%d] %.1f MB | %s | %s | torch copy_ of the bottom %.1f us" % (N, C, H, W, Ho, Wo, mb, res[0], res[1], us), flush=True)

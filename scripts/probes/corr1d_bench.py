"""Correlation1D forward at the DispNet shape: the MFMA kernel (default), the LDS-tiled VALU kernel (impl 17), the generic kernel (impl 1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
def t(f, it=100):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for (N, C, H, W, md, sd) in ((4, 256, 48, 96, 40, -1), (8, 256, 40, 56, 20, 0), (4, 128, 96, 192, 40, -1)):
    a, b = torch.randn(N, C, H, W, device="cuda"), torch.randn(N, C, H, W, device="cuda")
    p = ops.corr_params(md, 1, md, 1, 1, 0, False, sd)
    out = []
    for impl in (0, 17, 1):
        ops.set_correlation_impl(impl)
        out.append(t(lambda: ops.correlation1d_forward(p, a, b)))
    ops.set_correlation_impl(0)
    print("[%d,%d,%d,%d] md %d direction %d: MFMA %.1f us, LDS-tiled %.1f us, generic %.1f us" % (N, C, H, W, md, sd, *out))

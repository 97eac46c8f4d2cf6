"""What would a split-bf16 (bf16 x 3) correlation forward gain on this kernel's data movement?  Times corr_fwd_pair at [8,256,40,56] as
built, with 3 of every 8 MFMAs (the matrix-pipe time of 6 bf16 products per fp32 product, operand split taken as free) and with no
MFMA at all (the staging + LDS + epilogue floor).  The projection builds compute wrong results; they exist for this measurement only.
    python scripts/probes/corr_bf16x3_projection.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
N, C, H, W = 8, 256, 40, 56
g = torch.Generator(device="cuda").manual_seed(0)
b0 = torch.randn(N, C, H, W, device="cuda", generator=g)
b1 = torch.randn(N, C, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
alg_bytes = 4.0 * N * H * W * (2 * C + 441)
for impl, what in [(0, "as built (exact fp32)"), (7, "3/8 of the MFMAs (bf16 x 3 matrix time)"), (8, "no MFMA (data-movement floor)")]:
    ops.set_correlation_impl(impl)
    for _ in range(5):
        ops.correlation_forward(p, b0, b1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.correlation_forward(p, b0, b1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 5
    print("%-44s %6.2f us/launch   %5.2f TB/s algorithmic = %4.1f %% of 8 TB/s" % (what, us, alg_bytes / us / 1e6, 100 * alg_bytes / us / 1e6 / 8.0))
ops.set_correlation_impl(0)

"""1x1 MFMA convolution (csrc/conv_mfma.hip, kernel_size 1) per tile variant on the shapes it serves -- conv_redir and the
weight^T x bottom GEMMs of deconv3 / deconv2 -- next to the library GEMM."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flownet2_amd  # noqa: E402
from flownet2_amd import ops  # noqa: E402


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


VERBOSE = "--variants" in sys.argv
for name, (N, Cin, H, W, Cout) in {"conv_redir": (8, 256, 40, 56, 32), "deconv4 gemm": (8, 1026, 10, 14, 4096), "deconv3 gemm": (8, 770, 20, 28, 2048),
                                   "deconv2 gemm": (8, 386, 40, 56, 1024), "fn2 deconv5": (4, 1024, 6, 12, 8192), "fn2 deconv4": (4, 1026, 12, 24, 4096),
                                   "fn2 deconv2": (4, 386, 96, 192, 1024), "fn2 deconv3": (4, 770, 48, 96, 2048)}.items():
    x = torch.randn(N, Cin, H, W, device="cuda")
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05
    pw = ops.conv_mfma_pack_weights(w)
    flops = 2.0 * N * H * W * Cin * Cout
    out = torch.empty(N, Cout, H, W, device="cuda")
    t = timeit(lambda: ops.conv_mfma_forward(x, pw, None, Cout, 1, 1, 0, False, 0.0, out=out))
    w2 = w.view(Cout, Cin)
    tl = timeit(lambda: torch.matmul(w2, x.view(N, Cin, H * W)))
    print("%-14s [%d,%d,%d,%d] -> %4d  own (autotuned) %7.1f us %6.1f TF | library GEMM %7.1f us %6.1f TF" % (name, N, Cin, H, W, Cout, t, flops / t / 1e6, tl, flops / tl / 1e6), flush=True)
    if not VERBOSE:
        continue
    nv = ops.conv_num_variants()
    for v in list(range(nv)) + [1000 + i for i in range(nv)]:
        ops.set_conv_variant(v)
        try:
            tv = timeit(lambda: ops.conv_mfma_forward(x, pw, None, Cout, 1, 1, 0, False, 0.0, out=out), 10)
            print("      variant %4d: %7.1f us %6.1f TF" % (v, tv, flops / tv / 1e6))
        except flownet2_amd.Fn2Error:
            pass
    ops.set_conv_variant(-1)

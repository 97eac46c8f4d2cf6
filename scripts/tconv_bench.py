"""Transposed-convolution kernel (csrc/tconv_mfma.hip) at the FlowNetC shapes: every applicable variant, the first-use pick, and the
library (conv_transpose2d = the GEMM + col2im route / MIOpen)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flownet2_amd import ops  # noqa: E402
from test_tconv import LAYERS  # noqa: E402


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


def main():
    per_variant = "--variants" in sys.argv
    for name, (sx, Cout, k, p, hw) in LAYERS.items():
        x = torch.randn(sx, device="cuda")
        w = torch.randn((sx[1], Cout, k, k), device="cuda") * 0.05
        pw = ops.tconv_pack_weights(w)
        b = torch.zeros(Cout, device="cuda")
        flops = 2.0 * sx[0] * sx[1] * sx[2] * sx[3] * Cout * k * k
        out = torch.empty((sx[0], Cout) + hw, device="cuda")
        t_own = timeit(lambda: ops.tconv_forward(x, pw, b, Cout, k, p, out_hw=hw, relu=True, out=out))
        opad = (hw[0] - (2 * (sx[2] - 1) + k - 2 * p), hw[1] - (2 * (sx[3] - 1) + k - 2 * p))
        t_lib = timeit(lambda: torch.nn.functional.leaky_relu_(torch.nn.functional.conv_transpose2d(x, w, b, stride=2, padding=p, output_padding=opad), 0.1))
        print("%-12s x %-20s -> %3d ch %-10s k%d  own %8.1f us %6.1f TF | library %8.1f us %6.1f TF" %
              (name, sx, Cout, hw, k, t_own, flops / t_own / 1e6, t_lib, flops / t_lib / 1e6), flush=True)
        if per_variant:
            for v in range(ops.tconv_num_variants()):
                ops.set_tconv_variant(v)
                try:
                    t = timeit(lambda: ops.tconv_forward(x, pw, b, Cout, k, p, out_hw=hw, relu=True, out=out), 10)
                    print("      variant %2d: %8.1f us %6.1f TF" % (v, t, flops / t / 1e6))
                except Exception:
                    pass
            ops.set_tconv_variant(-1)


if __name__ == "__main__":
    main()

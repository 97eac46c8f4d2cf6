"""Weight-gradient kernel (csrc/conv_wgrad.hip) at the FlowNetC training shapes: time per call and TFLOP/s next to the library's
weight gradient (aten::convolution_backward = MIOpen incl. its NCHW<->NHWC transposes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from flownet2_amd import ops  # noqa: E402
from test_conv_wgrad import LAYERS  # noqa: E402

EXTRA = {"conv5_1": ((8, 512, 10, 14), (8, 512, 10, 14), 3, 1, 1), "conv6": ((8, 1024, 5, 7), (8, 512, 10, 14), 3, 2, 1),
         "deconv4": ((8, 1026, 10, 14), (8, 256, 20, 28), 4, 2, 1), "deconv3": ((8, 770, 20, 28), (8, 128, 40, 56), 4, 2, 1)}


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
    from flownet2_amd import _lib
    only = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else None
    if len(sys.argv) > 2:
        _lib.lib().fn2_debug_set_wgrad_buffers(int(sys.argv[2]))
    if len(sys.argv) > 3:
        _lib.lib().fn2_debug_set_wgrad_chunk(int(sys.argv[3]))
    print("buffers", sys.argv[2] if len(sys.argv) > 2 else "default", "chunk", sys.argv[3] if len(sys.argv) > 3 else "default")
    tot_own = tot_lib = 0.0
    for name, (sa, sb, k, s, p) in {**LAYERS, **EXTRA}.items():
        if only and name not in only:
            continue
        a, b = torch.randn(sa, device="cuda"), torch.randn(sb, device="cuda")
        flops = 2.0 * sa[0] * sa[1] * sa[2] * sa[3] * sb[1] * k * k
        w = torch.zeros((sa[1], sb[1], k, k), device="cuda")
        t_own = timeit(lambda: ops.conv_wgrad(a, b, k, s, p))
        if k == 4:
            t_lib = timeit(lambda: torch.ops.aten.convolution_backward(b, a, w, None, [s, s], [p, p], [1, 1], True, [0, 0], 1, [False, True, False]))
        else:
            t_lib = timeit(lambda: torch.ops.aten.convolution_backward(a, b, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False]))
        ks = ops.conv_wgrad_ksplit(sa[0], sa[1], sa[2], sa[3], sb[1], sb[2], sb[3], k, s, p)
        print("%-10s a %-20s b %-20s k%d s%d  ksplit %3d  own %8.1f us %6.1f TF | library %8.1f us %6.1f TF" %
              (name, sa, sb, k, s, ks, t_own, flops / t_own / 1e6, t_lib, flops / t_lib / 1e6), flush=True)
        tot_own += t_own
        tot_lib += t_lib
    print("total own %.1f us, library %.1f us" % (tot_own, tot_lib))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-variant timing of the small-map deconvolution kernel (csrc/conv_plane.hip, MODE 1) on the refinement layers, next to the GEMM +
col2im route (functional.deconv_gemm_relu) and the library's transposed convolution.
    python scripts/deconv_bench.py [--net C|2] [--layers deconv5,...] [--ksplit 1,2,4]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flownet2_amd  # noqa: E402
from flownet2_amd import functional as Fn, nets, ops  # noqa: E402

LAYERS = {"C": [("deconv5", 8, 1024, 5, 7, 512), ("deconv4", 8, 1026, 10, 14, 256), ("deconv3", 8, 770, 20, 28, 128), ("deconv2", 8, 386, 40, 56, 64)],
          "2": [("deconv5", 4, 1024, 6, 12, 512), ("deconv4", 4, 1026, 12, 24, 256), ("deconv3", 4, 770, 24, 48, 128), ("deconv2", 4, 386, 48, 96, 64)]}


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net", default="C")
    ap.add_argument("--layers", default="")
    ap.add_argument("--ksplit", default="")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    for (name, N, Cin, H, W, Cout) in LAYERS[a.net]:
        if a.layers and name not in a.layers.split(","):
            continue
        x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
        w = torch.randn(Cin, Cout, 4, 4, device="cuda", generator=g) * (2.0 / (Cin * 4)) ** 0.5
        b = torch.randn(Cout, device="cuda", generator=g) * 0.1
        gf = 2.0 * N * H * W * Cin * Cout * 16 / 1e9
        want = F.leaky_relu(F.conv_transpose2d(x, w, b, stride=2, padding=1), 0.1)
        t_lib = timeit(lambda: Fn.conv_bias_leaky_relu(F.conv_transpose2d(x, w, None, stride=2, padding=1), b, 0.1), a.iters)
        wt = nets._transposed_deconv_weight(w)
        t_gemm = timeit(lambda: Fn.deconv_gemm_relu(x, wt, b, Cout), a.iters)
        print(f"{name:8s} [{N},{Cin},{H},{W}]->{Cout}  {gf:6.2f} GF | MIOpen+bias/act {t_lib:7.1f} us {gf / t_lib * 1e3:6.1f} TF | GEMM+col2im {t_gemm:7.1f} us {gf / t_gemm * 1e3:6.1f} TF",
              flush=True)
        if not ops.deconv_plane_supported(N, Cin, H, W, Cout):
            print("   (deconv_plane: unsupported geometry)")
            continue
        pw = ops.deconv_plane_pack_weights(w)
        out = torch.empty_like(want)
        for ksp in ([int(v) for v in a.ksplit.split(",")] if a.ksplit else [0]):
            ops.set_plane_ksplit(ksp)
            ks_used = ops.deconv_plane_ksplit(N, Cin, H, W, Cout)
            first = None
            for v in range(ops.plane_num_variants()):
                ops.set_plane_variant(v)
                try:
                    ops.deconv_plane_forward(x, pw, b, Cout, True, 0.1, out=out)
                except flownet2_amd.Fn2Error:
                    continue
                torch.cuda.synchronize()
                err = float((out - want).abs().max())
                same = "" if first is None else ("  bits==first" if torch.equal(out, first) else "  BITS DIFFER")
                if first is None:
                    first = out.clone()
                t = timeit(lambda: ops.deconv_plane_forward(x, pw, b, Cout, True, 0.1, out=out), a.iters)
                print(f"   plane ksplit {ks_used:2d} variant {v:3d}: {t:7.1f} us {gf / t * 1e3:6.1f} TF   max|diff vs torch| {err:.2e}{same}", flush=True)
            ops.set_plane_variant(-1)
        ops.set_plane_ksplit(0)


if __name__ == "__main__":
    main()

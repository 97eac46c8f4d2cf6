#!/usr/bin/env python
"""Per-variant timing of the direct MFMA convolution (csrc/conv_mfma.hip) on the FlowNet layer shapes, next to the library
path it replaces (MIOpen conv2d + bias/activation pass, or im2col + GEMM).  Run on the GPU box:
    python scripts/conv_bench.py [--net C|2] [--check]
Prints one line per (layer, variant): us, TFLOP/s, and the max abs difference against torch's fp32 convolution."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flownet2_amd  # noqa: E402
from flownet2_amd import functional as Fn, nets, ops  # noqa: E402

LAYERS_C = [  # name, N, Cin, H, W, Cout, k, s, p     (FlowNetC batch 8 @448x320; the siamese towers run as batch 16)
    ("conv2", 16, 64, 160, 224, 128, 5, 2, 2), ("conv3", 16, 128, 80, 112, 256, 5, 2, 2), ("conv3_1", 8, 473, 40, 56, 256, 3, 1, 1),
    ("conv4", 8, 256, 40, 56, 512, 3, 2, 1), ("conv4_1", 8, 512, 20, 28, 512, 3, 1, 1), ("conv5", 8, 512, 20, 28, 512, 3, 2, 1),
    ("conv5_1", 8, 512, 10, 14, 512, 3, 1, 1), ("conv6", 8, 512, 10, 14, 1024, 3, 2, 1), ("conv6_1", 8, 1024, 5, 7, 1024, 3, 1, 1)]
LAYERS_SD = [  # FlowNet2-SD / fusion / interconv layers and the 12-channel stem, batch 4 @768x384
    ("net2_conv1", 4, 12, 384, 768, 64, 7, 2, 3), ("sd_conv0", 4, 6, 384, 768, 64, 3, 1, 0 + 1), ("sd_conv1_1", 4, 64, 192, 384, 128, 3, 1, 1), ("sd_conv2_1", 4, 128, 96, 192, 128, 3, 1, 1),
    ("sd_ic2", 4, 194, 96, 192, 64, 3, 1, 1), ("sd_ic3", 4, 386, 48, 96, 128, 3, 1, 1), ("sd_ic4", 4, 770, 24, 48, 256, 3, 1, 1),
    ("fuse_conv0", 4, 11, 384, 768, 64, 3, 1, 1)]
LAYERS_2 = [  # FlowNet2 batch 4 @768x384 (FlowNetS stage)
    ("conv2", 4, 64, 192, 384, 128, 5, 2, 2), ("conv3", 4, 128, 96, 192, 256, 5, 2, 2), ("conv3_1", 4, 256, 48, 96, 256, 3, 1, 1),
    ("conv4", 4, 256, 48, 96, 512, 3, 2, 1), ("conv4_1", 4, 512, 24, 48, 512, 3, 1, 1), ("conv5", 4, 512, 24, 48, 512, 3, 2, 1),
    ("conv5_1", 4, 512, 12, 24, 512, 3, 1, 1), ("conv6", 4, 512, 12, 24, 1024, 3, 2, 1), ("conv6_1", 4, 1024, 6, 12, 1024, 3, 1, 1)]



def lib_gemm_conv(x, w, b, s, p):
    """The route these layers took until round 4, kept HERE as the A/B reference only: own im2col, one batched library GEMM, own bias +
    ReLU pass (the product no longer calls a library GEMM)."""
    N, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    col = ops.im2col_forward(x, k, p, s)
    y = torch.matmul(w.reshape(Cout, Cin * k * k), col).view(N, Cout, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
    return ops.bias_leaky_relu_(y, b, 0.1)


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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--plane-ksplit", default="", help="comma list of K splits to time for the small-map kernel (default: the geometry's own)")
    ap.add_argument("--only-plane", action="store_true")
    a = ap.parse_args()
    layers = {"C": LAYERS_C, "2": LAYERS_2, "SD": LAYERS_SD}[a.net]
    if a.layers:
        layers = [l for l in layers if l[0] in a.layers.split(",")]
    g = torch.Generator(device="cuda").manual_seed(0)
    for (name, N, Cin, H, W, Cout, k, s, p) in layers:
        x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
        w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * (2.0 / (Cin * k * k)) ** 0.5
        b = torch.randn(Cout, device="cuda", generator=g) * 0.1
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        gf = 2.0 * N * Cout * Ho * Wo * Cin * k * k / 1e9
        want = F.leaky_relu(F.conv2d(x, w, b, stride=s, padding=p), 0.1)
        t_lib = timeit(lambda: Fn.conv_bias_leaky_relu(F.conv2d(x, w, None, stride=s, padding=p), b, 0.1), a.iters)
        line = f"{name:8s} [{N},{Cin},{H},{W}]->{Cout} k{k}s{s}  {gf:6.2f} GF | MIOpen+bias/act {t_lib:7.1f} us {gf / t_lib * 1e3:6.1f} TF"
        if k == 3:
            t_gemm = timeit(lambda: lib_gemm_conv(x, w, b, s, p), a.iters)
            line += f" | im2col+GEMM {t_gemm:7.1f} us {gf / t_gemm * 1e3:6.1f} TF"
        print(line, flush=True)
        if k == 3 and s == 1 and ops.conv_wino_supported(Cin, H, W, Cout, p):
            pu = ops.conv_wino_pack_weights(w)
            outw = torch.empty_like(want)
            firstw = None
            nv = ops.wino_num_variants()
            for v in list(range(nv)) + [1000 + i for i in range(nv)]:
                ops.set_wino_variant(v)
                try:
                    ops.conv_wino_forward(x, pu, b, Cout, p, True, 0.1, out=outw)
                except flownet2_amd.Fn2Error:
                    continue
                torch.cuda.synchronize()
                err = float((outw - want).abs().max())
                same = "" if firstw is None else ("  bits==first" if torch.equal(outw, firstw) else "  BITS DIFFER")
                if firstw is None:
                    firstw = outw.clone()
                t = timeit(lambda: ops.conv_wino_forward(x, pu, b, Cout, p, True, 0.1, out=outw), a.iters)
                print(f"   winograd {v:4d}: {t:7.1f} us {gf / t * 1e3:6.1f} TF(direct-equivalent)   max|diff vs torch| {err:.2e}{same}", flush=True)
            ops.set_wino_variant(-1)
            t = timeit(lambda: ops.conv_wino_forward(x, pu, b, Cout, p, True, 0.1, out=outw), a.iters)
            print(f"   winograd cost-model choice: {t:7.1f} us {gf / t * 1e3:6.1f} TF(direct-equivalent)", flush=True)
        if k == 3 and ops.conv_plane_supported(N, Cin, H, W, Cout, s, p):
            pwp = ops.conv_mfma_pack_weights(w)
            outp = torch.empty_like(want)
            for ksp in ([int(v) for v in a.plane_ksplit.split(",")] if a.plane_ksplit else [0]):
                ops.set_plane_ksplit(ksp)
                ks_used = ops.conv_plane_ksplit(N, Cin, H, W, Cout, s, p)
                firstp = None
                for v in range(ops.plane_num_variants()):
                    ops.set_plane_variant(v)
                    try:
                        ops.conv_plane_forward(x, pwp, b, Cout, s, p, True, 0.1, out=outp)
                    except flownet2_amd.Fn2Error:
                        continue
                    torch.cuda.synchronize()
                    err = float((outp - want).abs().max())
                    same = "" if firstp is None else ("  bits==first" if torch.equal(outp, firstp) else "  BITS DIFFER")
                    if firstp is None:
                        firstp = outp.clone()
                    t = timeit(lambda: ops.conv_plane_forward(x, pwp, b, Cout, s, p, True, 0.1, out=outp), a.iters)
                    print(f"   plane ksplit {ks_used:2d} variant {v:3d}: {t:7.1f} us {gf / t * 1e3:6.1f} TF   max|diff vs torch| {err:.2e}{same}", flush=True)
                ops.set_plane_variant(-1)
            ops.set_plane_ksplit(0)
        if a.only_plane:
            continue
        if not ops.conv_mfma_supported(Cin, H, W, Cout, k, s, p):
            print("   (conv_mfma: unsupported geometry)")
            continue
        pw = ops.conv_mfma_pack_weights(w)
        out = torch.empty_like(want)
        first = None
        nv = ops.conv_num_variants()
        for v in list(range(nv)) + [1000 + i for i in range(nv)]:          # 1000 + i: split-tail launch of variant i
            ops.set_conv_variant(v)
            try:
                ops.conv_mfma_forward(x, pw, b, Cout, k, s, p, True, 0.1, out=out)
            except flownet2_amd.Fn2Error:
                continue
            torch.cuda.synchronize()
            err = float((out - want).abs().max())
            same = "" if first is None else ("  bits==v%d" % first[0] if torch.equal(out, first[1]) else "  BITS DIFFER")
            if first is None:
                first = (v, out.clone())
            t = timeit(lambda: ops.conv_mfma_forward(x, pw, b, Cout, k, s, p, True, 0.1, out=out), a.iters)
            print(f"   variant {v:4d}: {t:7.1f} us {gf / t * 1e3:6.1f} TF   max|diff vs torch| {err:.2e}{same}", flush=True)
        ops.set_conv_variant(-1)
        t = timeit(lambda: ops.conv_mfma_forward(x, pw, b, Cout, k, s, p, True, 0.1, out=out), a.iters)
        print(f"   cost-model choice: {t:7.1f} us {gf / t * 1e3:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()

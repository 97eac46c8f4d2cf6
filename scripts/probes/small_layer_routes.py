"""FlowNet2 at batch 1 @1024x448 (BASELINE config 5): every 3x3 layer of the FlowNetC / FlowNetS / SD encoders on the production route
(what functional.conv_mfma_relu picks, tile variant autotuned) next to the im2col + library GEMM route it replaced as the default in round 4,
and the small-map kernel under forced K splits (is the geometry's own split the right one at batch 1?)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import flownet2_amd  # noqa: E402
from flownet2_amd import functional as Fn, ops  # noqa: E402

B1 = [  # name, N, Cin, H, W, Cout, k, s, p     batch 1 @1024x448
    ("C.conv3_1", 1, 473, 56, 128, 256, 3, 1, 1), ("S.conv3_1", 1, 256, 56, 128, 256, 3, 1, 1), ("conv4", 1, 256, 56, 128, 512, 3, 2, 1),
    ("conv4_1", 1, 512, 28, 64, 512, 3, 1, 1), ("conv5", 1, 512, 28, 64, 512, 3, 2, 1), ("conv5_1", 1, 512, 14, 32, 512, 3, 1, 1),
    ("conv6", 1, 512, 14, 32, 1024, 3, 2, 1), ("conv6_1", 1, 1024, 7, 16, 1024, 3, 1, 1),
    ("conv2", 1, 64, 224, 512, 128, 5, 2, 2), ("conv3", 1, 128, 112, 256, 256, 5, 2, 2), ("S.conv1", 1, 12, 448, 1024, 64, 7, 2, 3),
    ("sd_conv1_1", 1, 64, 224, 512, 128, 3, 1, 1), ("sd_conv2", 1, 128, 224, 512, 128, 3, 2, 1), ("sd_conv2_1", 1, 128, 112, 256, 128, 3, 1, 1),
    ("sd_conv3", 1, 128, 112, 256, 256, 3, 2, 1), ("sd_ic4", 1, 770, 28, 64, 256, 3, 1, 1), ("sd_ic3", 1, 386, 56, 128, 128, 3, 1, 1), ("sd_ic2", 1, 194, 112, 256, 64, 3, 1, 1)]
C8 = [("conv4", 8, 256, 40, 56, 512, 3, 2, 1), ("conv5", 8, 512, 20, 28, 512, 3, 2, 1), ("conv5_1", 8, 512, 10, 14, 512, 3, 1, 1),
      ("conv6", 8, 512, 10, 14, 1024, 3, 2, 1), ("conv6_1", 8, 1024, 5, 7, 1024, 3, 1, 1), ("conv4_1", 8, 512, 20, 28, 512, 3, 1, 1)]      # FlowNetC, batch 8 @448x320
B4 = [("conv3_1", 4, 256, 48, 96, 256, 3, 1, 1), ("conv4_1", 4, 512, 24, 48, 512, 3, 1, 1), ("conv5_1", 4, 512, 12, 24, 512, 3, 1, 1), ("conv6_1", 4, 1024, 6, 12, 1024, 3, 1, 1)]



def lib_gemm_conv(x, w, b, s, p):
    """The route these layers took until round 4, kept HERE as the A/B reference only: own im2col, one batched library GEMM, own bias +
    ReLU pass (the product no longer calls a library GEMM)."""
    N, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    col = ops.im2col_forward(x, k, p, s)
    y = torch.matmul(w.reshape(Cout, Cin * k * k), col).view(N, Cout, (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1)
    return ops.bias_leaky_relu_(y, b, 0.1)


def timeit(fn, iters=30, warm=5):
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


g = torch.Generator(device="cuda").manual_seed(0)
tot = {"own": 0.0, "lib": 0.0, "best": 0.0}
for name, N, Cin, H, W, Cout, k, s, p in (B4 if "--b4" in sys.argv else C8 if "--c8" in sys.argv else B1):
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g) * 0.1
    gf = 2.0 * N * Cout * ((H + 2 * p - k) // s + 1) * ((W + 2 * p - k) // s + 1) * Cin * k * k / 1e9
    with torch.no_grad():
        kind = Fn._conv_mfma_pick(x, w, s, p)
        t_own = timeit(lambda: Fn.conv_mfma_relu(x, w, b, s, p, 0.1, True))
        if k == 3:
            t_lib = timeit(lambda: lib_gemm_conv(x, w, b, s, p))
        else:           # what FN2_CONV_SMALL=lib runs for 5x5 / 7x7 layers: MIOpen's convolution + the fused bias / activation pass
            t_lib = timeit(lambda: Fn.conv_bias_leaky_relu(torch.nn.functional.conv2d(x, w, None, stride=s, padding=p), b, 0.1))
        alts = {}
        for alt in ("wino", "plane", "direct"):
            try:
                if alt == "wino" and not (s == 1 and ops.conv_wino_supported(Cin, H, W, Cout, p)):
                    continue
                if alt == "plane" and not ops.conv_plane_k_supported(N, Cin, H, W, Cout, k, s, p):
                    continue
                if alt == "direct" and not ops.conv_mfma_supported(Cin, H, W, Cout, k, s, p):
                    continue
                alts[alt] = timeit(lambda: Fn._conv_mfma_run(alt, x, w, b, s, p, 0.1, True))
            except flownet2_amd.Fn2Error:
                pass
        ks_txt = ""
        if "plane" in alts:
            pw = ops.conv_mfma_pack_weights(w)
            for ksp in (1, 2, 4, 8, 16):
                ops.set_plane_ksplit(ksp)
                try:
                    used = ops.conv_plane_k_ksplit(N, Cin, H, W, Cout, k, s, p)
                    if used == ksp:
                        ks_txt += " k%d:%.0f" % (ksp, timeit(lambda: ops.conv_plane_forward(x, pw, b, Cout, s, p, True, 0.1, kernel=k), 15, 3))
                except flownet2_amd.Fn2Error:
                    pass
            ops.set_plane_ksplit(0)
            ks_txt = " | plane own-split k%d;%s" % (ops.conv_plane_k_ksplit(N, Cin, H, W, Cout, k, s, p), ks_txt)
    best = min([t_own, t_lib] + list(alts.values()))
    tot["own"] += t_own; tot["lib"] += t_lib; tot["best"] += best
    print("%-11s [%d,%d,%d,%d]->%d s%d %6.2f GF | route %-6s %7.1f us %5.1f TF | im2col+lib GEMM %7.1f us | %s%s" % (
        name, N, Cin, H, W, Cout, s, gf, kind, t_own, gf / t_own * 1e3, t_lib, "  ".join("%s %.1f" % kv for kv in alts.items()), ks_txt), flush=True)
print("sum: production route %.1f us, library route %.1f us, best of all per layer %.1f us" % (tot["own"], tot["lib"], tot["best"]))

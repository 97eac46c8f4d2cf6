#!/usr/bin/env python
"""Small-map 3x3 layers of FlowNetC at batch 8: MIOpen conv2d vs im2col (F.unfold) + one hipBLASLt/rocBLAS GEMM."""
import torch, torch.nn.functional as F
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
layers = {"conv4": (256, 512, 40, 56, 2), "conv4_1": (512, 512, 20, 28, 1), "conv5": (512, 512, 20, 28, 2), "conv5_1": (512, 512, 10, 14, 1),
          "conv6": (512, 1024, 10, 14, 2), "conv6_1": (1024, 1024, 5, 7, 1)}
for name, (ci, co, h, w, s) in layers.items():
    x = torch.randn(8, ci, h, w, device="cuda"); wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.01
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    w2 = wt.view(co, ci * 9)
    def gemm():
        cols = F.unfold(x, 3, padding=1, stride=s)            # [8, ci*9, ho*wo]
        return torch.matmul(w2, cols).view(8, co, ho, wo)
    def gemm_only(cols=F.unfold(x, 3, padding=1, stride=s)):
        return torch.matmul(w2, cols)
    colsT = F.unfold(x, 3, padding=1, stride=s).permute(1, 0, 2).reshape(ci * 9, 8 * ho * wo).contiguous()
    def gemm_flat():
        return torch.mm(w2, colsT)                           # [co, 8*ho*wo] one GEMM
    ref = F.conv2d(x, wt, None, stride=s, padding=1)
    err = (gemm() - ref).abs().max().item()
    fl = 2.0 * 8 * ho * wo * co * ci * 9
    tm, tg, tgo, tf = t(lambda: F.conv2d(x, wt, None, stride=s, padding=1)), t(gemm), t(gemm_only), t(gemm_flat)
    print("%-8s MIOpen %.1f us (%.0f TF) | unfold+bmm %.1f | bmm only %.1f (%.0f TF) | single mm only %.1f (%.0f TF) | err %.1e" %
          (name, tm, fl / tm / 1e6, tg, tgo, fl / tgo / 1e6, tf, fl / tf / 1e6, err))

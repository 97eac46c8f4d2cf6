#!/usr/bin/env python
"""Per-layer MIOpen fp32 timing of the FlowNetC conv stack at batch 8 @448x320 (towers batched: 16 images)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import nets
dev = torch.device("cuda")
B, H, W = 8, 320, 448
res = {"conv1": 1, "conv2": 2, "conv3": 4, "conv_redir": 8, "conv3_1": 8, "conv4": 8, "conv4_1": 16, "conv5": 16, "conv5_1": 32, "conv6": 32, "conv6_1": 64,
       "Convolution1": 64, "deconv5": 64, "upsample_flow6to5": 64, "Convolution2": 32, "deconv4": 32, "upsample_flow5to4": 32, "Convolution3": 16,
       "deconv3": 16, "upsample_flow4to3": 16, "Convolution4": 8, "deconv2": 8, "upsample_flow3to2": 8, "Convolution5": 4}   # INPUT resolution divisor
tot = 0.0; totf = 0.0
for (name, kind, ci, co, k, s, p) in nets.layer_table("C"):
    n = B * 2 if name in ("conv1", "conv2", "conv3") else B
    hi, wi = H // res[name], W // res[name]
    x = torch.randn(n, ci, hi, wi, device=dev)
    w = torch.randn((co, ci, k, k) if kind == "conv" else (ci, co, k, k), device=dev) * 0.01
    b = torch.zeros(co, device=dev)
    f = (lambda: F.conv2d(x, w, b, stride=s, padding=p)) if kind == "conv" else (lambda: F.conv_transpose2d(x, w, b, stride=2, padding=1))
    for _ in range(3): y = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    ho, wo = y.shape[2], y.shape[3]
    fl = 2.0 * n * co * ci * k * k * (ho * wo if kind == "conv" else hi * wi)
    tot += t; totf += fl
    print(f"{name:20s} {kind:6s} in[{n},{ci},{hi},{wi}] -> [{co},{ho},{wo}] k{k}s{s}  {t*1e6:8.1f} us  {fl/1e9:7.2f} GF  {fl/t/1e12:6.1f} TF")
print(f"TOTAL {tot*1e3:.3f} ms  {totf/1e9:.1f} GF  {totf/tot/1e12:.1f} TF")

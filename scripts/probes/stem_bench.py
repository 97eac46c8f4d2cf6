#!/usr/bin/env python
"""conv1 (7x7/2, 3 -> 64) + bias + leaky ReLU at batch 16 @448x320: fused HIP kernel vs MIOpen conv + the fused bias/activation pass."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
x = torch.randn(16, 3, 320, 448, device="cuda"); w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05; b = torch.randn(64, device="cuda")
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=3), 0.1)
out = ops.conv_k7s2_relu_forward(x, w, b, 0.1)
print("max abs diff vs MIOpen: %.3e (max |ref| %.2f)" % ((out - ref).abs().max().item(), ref.abs().max().item()))
fl = 2.0 * 16 * 64 * 160 * 224 * 147
tf = t(lambda: ops.conv_k7s2_relu_forward(x, w, b, 0.1))
tm = t(lambda: ops.bias_leaky_relu_(F.conv2d(x, w, None, stride=2, padding=3), b, 0.1))
print("fused HIP %.1f us (%.1f TFLOP/s)   MIOpen conv + bias/act pass %.1f us" % (tf, fl / tf / 1e6, tm))

"""Relative-L2 accuracy of the backward kernels on the coarse levels of FlowNetC (batch 8 @448x320), each against torch's fp64 gradient on
the same random tensors -- the per-op follow-up to tests/test_train_parity.py (round 4: conv5 .. conv6_1 / deconv5 gradients of the own
path were 30 x further from fp64 than the library's)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flownet2_amd import functional as Fn, ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(1)
R = lambda *s: torch.randn(s, device="cuda", generator=g)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def wgrad_ref(d, x, wshape, s, p, transposed):
    w0 = torch.zeros(wshape, dtype=torch.float64, device="cuda")
    return torch.ops.aten.convolution_backward(d.double(), x.double(), w0, None, [s, s], [p, p], [1, 1], transposed, [0, 0], 1, [False, True, False])[1]


def dgrad_ref(d, xshape, w, s, p, transposed):
    x0 = torch.zeros(xshape, dtype=torch.float64, device="cuda")
    return torch.ops.aten.convolution_backward(d.double(), x0, w.double(), None, [s, s], [p, p], [1, 1], transposed, [0, 0], 1, [True, False, False])[0]


for name, xs, ws, s, p, tr in [("conv5", (8, 512, 20, 28), (512, 512, 3, 3), 2, 1, False), ("conv5_1", (8, 512, 10, 14), (512, 512, 3, 3), 1, 1, False),
                               ("conv6", (8, 512, 10, 14), (1024, 512, 3, 3), 2, 1, False), ("conv6_1", (8, 1024, 5, 7), (1024, 1024, 3, 3), 1, 1, False),
                               ("conv4_1", (8, 512, 20, 28), (512, 512, 3, 3), 1, 1, False),
                               ("deconv5", (8, 1024, 5, 7), (1024, 512, 4, 4), 2, 1, True), ("deconv4", (8, 1026, 10, 14), (1026, 256, 4, 4), 2, 1, True),
                               ("deconv3", (8, 770, 20, 28), (770, 128, 4, 4), 2, 1, True), ("deconv2", (8, 386, 40, 56), (386, 64, 4, 4), 2, 1, True)]:
    x, w = R(*xs), R(*ws) * 0.02
    if tr:
        dshape = (xs[0], ws[1], 2 * xs[2], 2 * xs[3])
    else:
        dshape = (xs[0], ws[0], (xs[2] + 2 * p - ws[2]) // s + 1, (xs[3] + 2 * p - ws[2]) // s + 1)
    d = R(*dshape)
    gw = Fn._own_bwd_weight(d, x, w, s, p, tr)
    gx = Fn._own_bwd_data(d, w, s, p, tr, xs)
    lw = torch.ops.aten.convolution_backward(d, x, w, None, [s, s], [p, p], [1, 1], tr, [0, 0], 1, [True, True, False])
    print("%-8s weight grad: own %s  library %.2e | data grad: own %s  library %.2e" % (
        name, "%.2e" % rel(gw, wgrad_ref(d, x, ws, s, p, tr)) if gw is not None else "  (lib) ", rel(lw[1], wgrad_ref(d, x, ws, s, p, tr)),
        "%.2e" % rel(gx, dgrad_ref(d, xs, w, s, p, tr)) if gx is not None else "  (lib) ", rel(lw[0], dgrad_ref(d, xs, w, s, p, tr))), flush=True)

# forward kernels of the same levels
import torch.nn.functional as F
for name, xs, ws, s, p in [("conv5", (8, 512, 20, 28), (512, 512, 3, 3), 2, 1), ("conv5_1", (8, 512, 10, 14), (512, 512, 3, 3), 1, 1),
                           ("conv6", (8, 512, 10, 14), (1024, 512, 3, 3), 2, 1), ("conv6_1", (8, 1024, 5, 7), (1024, 1024, 3, 3), 1, 1)]:
    x, w, b = R(*xs), R(*ws) * 0.02, R(ws[0]) * 0.1
    with torch.no_grad():
        y = Fn.conv_mfma_relu(x, w, b, s, p, 0.1, True)
        ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p), 0.1)
        lib = F.leaky_relu(F.conv2d(x, w, b, stride=s, padding=p), 0.1)
    print("%-8s forward: own %.2e  library %.2e" % (name, rel(y, ref), rel(lib, ref)), flush=True)
# flow heads
for name, C, H, W in [("Convolution1", 1024, 5, 7), ("Convolution2", 1026, 10, 14), ("Convolution3", 770, 20, 28), ("Convolution4", 386, 40, 56), ("Convolution5", 194, 80, 112)]:
    x, w, d = R(8, C, H, W), R(2, C, 3, 3) * 0.02, R(8, 2, H, W)
    dx, dw, db = ops.predict_flow_conv_backward(x, w, d)
    r = torch.ops.aten.convolution_backward(d.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False])
    print("%-12s head: data grad %.2e  weight grad %.2e" % (name, rel(dx, r[0]), rel(dw, r[1])), flush=True)

# upsample_flow heads
for name, H, W in [("upsample_flow6to5", 5, 7), ("upsample_flow5to4", 10, 14), ("upsample_flow4to3", 20, 28), ("upsample_flow3to2", 40, 56)]:
    x, w, d = R(8, 2, H, W), R(2, 2, 4, 4) * 0.25, R(8, 2, 2 * H, 2 * W)
    dx, dw, db = ops.upsample_flow_deconv_backward(x, w, d)
    r = torch.ops.aten.convolution_backward(d.double(), x.double(), w.double(), None, [2, 2], [1, 1], [1, 1], True, [0, 0], 1, [True, True, False])
    print("%-18s head: data grad %.2e  weight grad %.2e" % (name, rel(dx, r[0]), rel(dw, r[1])), flush=True)
# fused bias + leaky-ReLU backward on a channel slice of a Concat gradient
for C, H, W, c0, Ct in [(128, 40, 56, 256, 386), (256, 20, 28, 512, 770)]:
    y, g = R(8, C, H, W), R(8, Ct, H, W)
    d, db = ops.bias_leaky_relu_backward(y, (g, c0, C), 0.1, True)
    ref = g[:, c0:c0 + C].double() * torch.where(y > 0, 1.0, 0.1).double()
    print("bias_leaky_relu_backward slice C=%d: d %.2e  db %.2e" % (C, rel(d, ref), rel(db, ref.sum((0, 2, 3)))), flush=True)

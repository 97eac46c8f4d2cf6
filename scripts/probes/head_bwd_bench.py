"""Backward of the 2-channel flow heads (csrc/flow_head_bwd.hip) at the FlowNetC training shapes, next to aten::convolution_backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
def t(f, it=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
for (C, H, W) in ((1024, 5, 7), (1026, 10, 14), (770, 20, 28), (386, 40, 56), (194, 80, 112)):
    x, w, g = torch.randn(8, C, H, W, device="cuda"), torch.randn(2, C, 3, 3, device="cuda"), torch.randn(8, 2, H, W, device="cuda")
    own_w = t(lambda: ops.predict_flow_conv_backward(x, w, g, need_x=False))
    own_x = t(lambda: ops.predict_flow_conv_backward(x, w, g, need_w=False, need_b=False))
    lib = t(lambda: torch.ops.aten.convolution_backward(g, x, w, [2], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True]))
    print("predict_flow C=%4d %3dx%3d: own weight+bias %6.1f us, own data %6.1f us | library (all three) %6.1f us" % (C, H, W, own_w, own_x, lib))
for (H, W) in ((5, 7), (10, 14), (20, 28), (40, 56)):
    x, w, g = torch.randn(8, 2, H, W, device="cuda"), torch.randn(2, 2, 4, 4, device="cuda"), torch.randn(8, 2, 2 * H, 2 * W, device="cuda")
    own = t(lambda: ops.upsample_flow_deconv_backward(x, w, g))
    lib = t(lambda: torch.ops.aten.convolution_backward(g, x, w, [2], [2, 2], [1, 1], [1, 1], True, [0, 0], 1, [True, True, True]))
    print("upsample_flow %3dx%3d: own %6.1f us | library %6.1f us" % (H, W, own, lib))

import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (n, ci, co, h, w, k, s, p) in {"conv2": (16, 64, 128, 160, 224, 5, 2, 2), "conv3": (16, 128, 256, 80, 112, 5, 2, 2), "conv3_1": (8, 473, 256, 40, 56, 3, 1, 1),
                                          "conv4_1": (8, 512, 512, 20, 28, 3, 1, 1), "conv_redir": (8, 256, 32, 40, 56, 1, 1, 0)}.items():
    x = torch.randn(n, ci, h, w, device="cuda"); wt = torch.randn(co, ci, k, k, device="cuda") * 0.01; b = torch.randn(co, device="cuda")
    tb = t(lambda: F.leaky_relu(F.conv2d(x, wt, b, stride=s, padding=p), 0.1))
    tn = t(lambda: ops.bias_leaky_relu_(F.conv2d(x, wt, None, stride=s, padding=p), b, 0.1))
    tc = t(lambda: F.conv2d(x, wt, None, stride=s, padding=p))
    print("%-10s conv+bias+leaky (torch) %.1f us | bias-free conv + fused pass %.1f us | bias-free conv alone %.1f us" % (name, tb, tn, tc))

import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import functional as Fn, ops
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = {"conv3_1 s1 [4,256,48,96]": (4, 256, 256, 48, 96, 1), "conv4 s2 [4,256,48,96]": (4, 256, 512, 48, 96, 2), "conv4_1 s1 [4,512,24,48]": (4, 512, 512, 24, 48, 1),
          "conv5 s2 [4,512,24,48]": (4, 512, 512, 24, 48, 2), "conv5_1 s1 [4,512,12,24]": (4, 512, 512, 12, 24, 1), "conv6 s2 [4,512,12,24]": (4, 512, 1024, 12, 24, 2),
          "conv6_1 s1 [4,1024,6,12]": (4, 1024, 1024, 6, 12, 1), "SD conv2 s2 [4,128,192,384]": (4, 128, 128, 192, 384, 2), "SD conv3 s2 [4,128,96,192]": (4, 128, 256, 96, 192, 2),
          "conv4_1 s1 b8 [8,512,20,28]": (8, 512, 512, 20, 28, 1), "conv3_1 s1 b8 [8,473,40,56]": (8, 473, 256, 40, 56, 1)}
for name, (n, ci, co, h, w, s) in shapes.items():
    x = torch.randn(n, ci, h, w, device="cuda"); wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.01; b = torch.randn(co, device="cuda")
    tm = t(lambda: ops.bias_leaky_relu_(F.conv2d(x, wt, None, stride=s, padding=1), b, 0.1))
    tg = t(lambda: Fn.conv_gemm_relu(x, wt, b, s, 1, 0.1))
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    print("%-30s MIOpen + bias/act %.1f us | im2col + GEMM + bias/act %.1f us | cols %.0f MB" % (name, tm, tg, 4e-6 * n * ci * 9 * ho * wo))

import os, sys, torch
sys.path.insert(0, os.getcwd())
from flownet2_amd import ops
N, C, H, W = 8, 256, 40, 56
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, C, H, W, device="cuda", generator=g); y = torch.randn(N, C, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
gg = torch.randn(N, 441, H, W, device="cuda", generator=g)
for which in ((True, False), (False, True)):
    for _ in range(300): ops.correlation_backward(p, x, y, gg, need0=which[0], need1=which[1])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1500): ops.correlation_backward(p, x, y, gg, need0=which[0], need1=which[1])
    e1.record(); torch.cuda.synchronize()
    print("bottom %d: %.2f us" % (0 if which[0] else 1, e0.elapsed_time(e1) / 1500 * 1e3))

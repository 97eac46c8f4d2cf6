import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
def t(f, it=100):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
x = torch.randn(8, 2, 320, 448, device="cuda")
for s in (4, 8, 16, 32, 64):
    print("downsample [8,2,320,448] -> 1/%d: %.1f us" % (s, t(lambda: ops.downsample_forward(x, 320 // s, 448 // s))))
sizes = [(320 // s, 448 // s) for s in (4, 8, 16, 32, 64)]
print("the five as five launches: %.1f us" % t(lambda: [ops.downsample_forward(x, h, w) for h, w in sizes]))
print("the five in ONE launch (fn2_downsample_forward_multi): %.1f us" % t(lambda: ops.downsample_forward_multi(x, sizes)))
y = torch.randn(2, 2, 12, 1200, device="cuda")
print("wide window [2,2,12,1200] -> (9, 20) (107 x 5 taps): %.1f us;  -> (11, 30) (83 x 5): %.1f us" % (t(lambda: ops.downsample_forward(y, 9, 20)), t(lambda: ops.downsample_forward(y, 11, 30))))

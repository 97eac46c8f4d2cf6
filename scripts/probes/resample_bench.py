import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
def t(f, it=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
x = torch.randn(8, 2, 80, 112, device="cuda")
print("x4 up [8,2,80,112]: plain %.1f us  scaled %.1f us" % (t(lambda: ops.resample_forward(x, 320, 448)), t(lambda: ops.resample_forward_slices(x, 320, 448, in_scale=20.0))))
x = torch.randn(4, 2, 96, 192, device="cuda")
print("x4 up [4,2,96,192]: plain %.1f us" % t(lambda: ops.resample_forward(x, 384, 768)))
print("nearest: %.1f us" % t(lambda: ops.resample_forward(x, 384, 768, 1)))

#!/usr/bin/env python
"""FN2_ABLATION builds: per-workgroup timeline of the persistent correlation forward (wave 0: start, per task {K loop + scatter done, image-ready barrier passed}, end)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops, _lib
L = _lib.lib()
L.fn2_debug_set_correlation_trace.argtypes = [C.c_void_p]
N, Cc, H, W = 8, 256, 40, 56
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, Cc, H, W, device="cuda", generator=g); y = torch.randn(N, Cc, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
out = torch.empty(N, 441, H, W, device="cuda")
IMPL = int(sys.argv[1]) if len(sys.argv) > 1 else 17
L.fn2_debug_set_correlation_impl(IMPL)
print("impl", IMPL)
for _ in range(300):
    ops.correlation_forward(p, x, y, out=out)
dbg = torch.zeros(10 * 256, dtype=torch.int64, device="cuda")
L.fn2_debug_set_correlation_trace(C.c_void_p(dbg.data_ptr()))
ops.correlation_forward(p, x, y, out=out); torch.cuda.synchronize()
L.fn2_debug_set_correlation_trace(None)
r = dbg.cpu().numpy().reshape(256, 10)
t0 = r[:, 0:1]
rel = r[:, :9] - t0
names = ["start", "task0 loop+scatter", "task0 E", "task1 loop+scatter", "task1 E", "task2 loop+scatter", "task2 E", "(task3)", "end"]
for i, nm in enumerate(names):
    col = rel[:, i]
    if (r[:, i] != 0).any():
        print("%-20s min %7d med %7d max %7d" % (nm, col[r[:, i] != 0].min(), np.median(col[r[:, i] != 0]), col[r[:, i] != 0].max()))
print("per-task K loop (median): ", [int(np.median(rel[:, 1 + 2 * t] - (rel[:, 2 * t] if t else 0))) for t in range(3)], " -> cycles per chunk step", [int(np.median(rel[:, 1 + 2 * t] - (rel[:, 2 * t] if t else 0)) / 32) for t in range(3)])

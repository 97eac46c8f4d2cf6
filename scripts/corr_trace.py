#!/usr/bin/env python
"""FN2_ABLATION builds only: per-workgroup timeline of the MFMA correlation kernels (start / K-loop end / end, CU id).
usage: corr_trace.py [impl]   (impl 0 = automatic, 3 = general kernel, 64 + bits = its ablations)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops, _lib
N, Cc, H, W = 8, 256, 40, 56
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, Cc, H, W, device="cuda", generator=g); y = torch.randn(N, Cc, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
out = torch.empty(N, 441, H, W, device="cuda")
for _ in range(3): ops.correlation_forward(p, x, y, out=out)
dbg = torch.zeros(4 * 1024 + 4 * 8 * 1024 + 2 * 1024, dtype=torch.int64, device="cuda")
if len(sys.argv) > 1: _lib.lib().fn2_debug_set_correlation_impl(int(sys.argv[1]))
L = _lib.lib(); L.fn2_debug_set_correlation_trace.argtypes = [C.c_void_p]
for _ in range(50): ops.correlation_forward(p, x, y, out=out)      # steady state: clocks ramped, inputs in the Infinity Cache
L.fn2_debug_set_correlation_trace(C.c_void_p(dbg.data_ptr()))
ops.correlation_forward(p, x, y, out=out); torch.cuda.synchronize()
L.fn2_debug_set_correlation_trace(None)
raw = dbg.cpu().numpy()
d = raw[:4 * 1024].reshape(-1, 4)
rt = raw[4 * 1024 + 4 * 8 * 1024:].reshape(1024, 2)
if rt.any():
    rr = rt[rt[:, 0] != 0]
    print("wall-clock span of all workgroups: %.2f us (100 MHz ticks)" % ((rr[:, 1].max() - rr[:, 0].min()) / 100.0))
bidx = np.nonzero(d[:, 0] != 0)[0]
d = d[d[:, 0] != 0]
hw = d[:, 3] & 0xffffffff; xcc = (d[:, 3] >> 32) & 0xf; heavy = (d[:, 3] >> 63) & 1
t0 = np.zeros(len(d), dtype=np.int64)
cu_ = (hw >> 8) & 0xf; sh_ = (hw >> 12) & 1; se_ = (hw >> 13) & 0x7
grp = (bidx % 8) * 1000 + se_ * 100 + sh_ * 10 + cu_       # one clock domain per CU is the safe assumption
for k in set(grp.tolist()):
    t0[grp == k] = d[grp == k, 0].min()
start, loop, end = (d[:, 0] - t0), (d[:, 1] - t0), (d[:, 2] - t0)
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = grp
print("blocks", len(d), "heavy", int(heavy.sum()), "distinct CUs", len(set(key.tolist())))
print("span of kernel (max over XCDs)", int(end.max()), "shader cycles")
for name, arr in [("start", start), ("loop_end", loop), ("end", end)]:
    print(name, ": min/med/max", int(arr.min()), int(np.median(arr)), int(arr.max()))
dur = end - start
print("duration med", int(np.median(dur)))
# concurrency per CU at the median time
from collections import Counter
c = Counter(key[heavy == 1].tolist())
print("heavy blocks per CU histogram:", sorted(Counter(c.values()).items()))
tm = np.median(loop[heavy == 1]) * 0.5
alive = (start <= tm) & (end >= tm)
print("blocks alive at t=%d: %d (heavy %d)" % (tm, alive.sum(), (alive & (heavy == 1)).sum()))
ca = Counter(key[alive].tolist()); print("alive per CU histogram:", sorted(Counter(ca.values()).items()))

for q in (0.25, 0.5, 0.75):
    tm = end.max() * q
    alive = (start <= tm) & (end >= tm)
    ca = Counter(key[alive].tolist())
    print("t=%d: alive %d (heavy %d); per-CU histogram %s" % (tm, alive.sum(), (alive & (heavy == 1)).sum(), sorted(Counter(ca.values()).items())))
print("heavy: loop phase med", int(np.median((loop - start)[heavy == 1])), " epilogue med", int(np.median((end - loop)[heavy == 1])))


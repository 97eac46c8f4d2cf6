#!/usr/bin/env python
"""FN2_ABLATION builds: (1) interleaved timing of the ablation builds of corr_fwd_units (impl 100 + bits: 1 no MFMA, 2 no LDS-DMA, 4 no stores,
8 no K-loop barriers, 16 no operand reads); (2) per-wave timeline of ONE launch (start / K loop end / end, CU, SIMD)."""
import ctypes as C, json, os, statistics, sys, time
from collections import Counter, defaultdict
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops, _lib
L = _lib.lib()
L.fn2_debug_set_correlation_trace.argtypes = [C.c_void_p]
shape = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (8, 256, 40, 56)
N, Cc, H, W = shape
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, Cc, H, W, device="cuda", generator=g); y = torch.randn(N, Cc, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
out = torch.empty(N, 441, H, W, device="cuda")


def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.correlation_forward(p, x, y, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


impls = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [20, 19, 201, 202, 203]
t0 = time.time()
while time.time() - t0 < 1.0:
    run(500)
res = {i: [] for i in impls}
for rnd in range(4):
    for i in (impls if rnd % 2 == 0 else impls[::-1]):
        L.fn2_debug_set_correlation_impl(i)
        run(200)
        res[i].append(run(1000))
names = {1: "no MFMA", 2: "no DMA", 4: "no stores", 8: "no loop barriers", 16: "no operand reads"}
for i in impls:
    what = "corr_fwd_pair" if i == 19 else "units" if i == 20 else "units, policy %d flags %d" % ((i - 200) >> 3, (i - 200) & 7) if i >= 200 else "units without " + ", ".join(v for k, v in names.items() if (i - 100) & k)
    print("impl %3d %-60s median %.2f us (min %.2f)" % (i, what, statistics.median(res[i]), min(res[i])), flush=True)

# ---- per-wave trace of one launch of the real kernel ----
TR = int(os.environ.get("TRACE_IMPL", "20"))
L.fn2_debug_set_correlation_impl(TR)
run(300)
dbg = torch.zeros(6 * 5 * 4096, dtype=torch.int64, device="cuda")
L.fn2_debug_set_correlation_trace(C.c_void_p(dbg.data_ptr()))
ops.correlation_forward(p, x, y, out=out); torch.cuda.synchronize()
L.fn2_debug_set_correlation_trace(None)
L.fn2_debug_set_correlation_impl(0)
raw = dbg.cpu().numpy().reshape(-1, 5, 6)
blk = np.nonzero(raw[:, 0, 0] != 0)[0]
d = raw[blk]                                                   # [block][wave][start, loop end, end, info]
hw = d[:, :, 3] & 0xffffffff
cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7)
xcc = (d[:, :, 3] >> 32) & 0xf
simd = (hw >> 4) & 3
col = (d[:, 0, 3] >> 40) & 7
key = xcc[:, 0] * 1000 + cu[:, 0]
t0c = {}
for k in set(key.tolist()):
    t0c[k] = d[key == k][:, :, 0].min()
base = np.array([t0c[k] for k in key.tolist()])[:, None]
start, loop, end = d[:, :, 0] - base, d[:, :, 1] - base, d[:, :, 2] - base
tbar, tsync = d[:, :, 4] - base, d[:, :, 5] - base
print("trace of impl", TR)
print("phases, median over workgroups (cycles): start->last barrier %d | last barrier->scatter done %d | ->sync %d | ->end (stores) %d" % (
    np.median(tbar[:, :4].max(1) - start.min(1)), np.median(loop[:, :4].max(1) - tbar[:, :4].max(1)), np.median(tsync.max(1) - loop[:, :4].max(1)), np.median(end.max(1) - tsync.max(1))))
old = start.min(1) < 2000
for nm, m in (("first-round", old), ("later", ~old)):
    if m.any():
        print("  %s workgroups (%d): loop %d | scatter %d | sync %d | stores %d" % (nm, m.sum(), np.median((tbar[:, :4].max(1) - start.min(1))[m]), np.median((loop[:, :4].max(1) - tbar[:, :4].max(1))[m]),
                                                                              np.median((tsync.max(1) - loop[:, :4].max(1))[m]), np.median((end.max(1) - tsync.max(1))[m])))
print("live workgroups traced %d on %d CUs; kernel span %d cycles (max end over CUs, per-CU clock origin)" % (len(blk), len(t0c), int(end.max())))
print("workgroup start (wave 0): min/med/max %d %d %d" % (start[:, 0].min(), np.median(start[:, 0]), start[:, 0].max()))
print("K loop end (consumers):   min/med/max %d %d %d" % (loop[:, :4].min(), np.median(loop[:, :4]), loop[:, :4].max()))
print("workgroup end:            min/med/max %d %d %d" % (end.min(), np.median(end), end.max()))
dur = end.max(1) - start.min(1)
for c in sorted(set(col.tolist())):
    m = col == c
    print("column task %d: %4d workgroups, duration med %6d (loop %6d, epilogue %5d), prologue-to-first-barrier n/a" % (
        c, m.sum(), np.median(dur[m]), np.median((loop[:, :4].max(1) - start.min(1))[m]), np.median((end.max(1) - loop[:, :4].max(1))[m])))
# per CU: when does each of its workgroups end, how many at a time
per_cu = defaultdict(list)
for i, k in enumerate(key.tolist()):
    per_cu[k].append((int(start[i].min()), int(loop[i, :4].max()), int(end[i].max())))
ends = np.array([max(e for _, _, e in v) for v in per_cu.values()])
print("per-CU last end: min/med/max %d %d %d; workgroups per CU histogram %s" % (ends.min(), np.median(ends), ends.max(), sorted(Counter(len(v) for v in per_cu.values()).items())))
for q in (0.1, 0.25, 0.5, 0.75, 0.9, 0.97):
    tm = end.max() * q
    inloop = ((start.min(1) <= tm) & (loop[:, :4].max(1) >= tm)).sum()
    inepi = ((loop[:, :4].max(1) < tm) & (end.max(1) >= tm)).sum()
    print("t = %6d (%.0f %%): workgroups in K loop %4d, in epilogue %4d" % (tm, 100 * q, inloop, inepi))
# SIMD placement of the five waves
pat = Counter(tuple(simd[i].tolist()) for i in range(len(blk)))
print("SIMD of waves 0-4, most common patterns:", pat.most_common(6))
np.save("gpurun_out/corr_units_trace_%d.npy" % TR, raw[blk])

#!/usr/bin/env python
"""FN2_ABLATION builds only: per-WAVE timeline of corr_fwd_pair at [8,256,40,56]: which SIMD every wave ran on, how many tile units it had,
when its K loop ended -- per-SIMD matrix work against the time the SIMD was occupied."""
import ctypes as C, os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops, _lib
N, Cc, H, W = 8, 256, 40, 56
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, Cc, H, W, device="cuda", generator=g); y = torch.randn(N, Cc, H, W, device="cuda", generator=g)
p = ops.corr_params(20, 1, 20, 1, 2)
out = torch.empty(N, 441, H, W, device="cuda")
dbg = torch.zeros(4 * 1024 + 4 * 8 * 1024 + 2 * 1024, dtype=torch.int64, device="cuda")
L = _lib.lib(); L.fn2_debug_set_correlation_trace.argtypes = [C.c_void_p]
for _ in range(50): ops.correlation_forward(p, x, y, out=out)
L.fn2_debug_set_correlation_trace(C.c_void_p(dbg.data_ptr()))
ops.correlation_forward(p, x, y, out=out); torch.cuda.synchronize()
L.fn2_debug_set_correlation_trace(None)
raw = dbg.cpu().numpy()
wg = raw[:4 * 1024].reshape(-1, 4)
wv = raw[4 * 1024:4 * 1024 + 4 * 8 * 1024].reshape(1024, 8, 4)
NT = {0: 4, 1: 5, 2: 6, 3: 3, 4: 4, 5: 5, 6: 2, 7: 3, 8: 4, 9: 0}     # tiles of selector lo*3 + (hi - 3): lo in {0,1,2}, hi in {3,4,5}
rows = []
for b in range(1024):
    if wg[b, 0] == 0: continue
    xcc = (int(wg[b, 3]) >> 32) & 0xf
    for w in range(4):
        r = wv[b, w]
        if r[0] == 0: continue
        hw = int(r[3]) & 0xffffffff
        sel = (int(r[3]) >> 32) & 0xff
        cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
        rows.append((cu, (hw >> 4) & 3, hw & 0xf, w, NT[sel], int(r[0]), int(r[1]), int(r[2]), b))
print("waves traced", len(rows))
by_cu = collections.defaultdict(list)
for r in rows: by_cu[r[0]].append(r)
simd_of_jw = collections.Counter((r[3], r[1]) for r in rows)
print("wave Jw -> SIMD histogram:", sorted(simd_of_jw.items()))
slot_hist = collections.Counter(r[2] for r in rows)
print("wave slot histogram:", sorted(slot_hist.items()))
eff, ends, units = [], [], []
for cu, rs in by_cu.items():
    t0 = min(r[5] for r in rs)
    for simd in range(4):
        ws = [r for r in rs if r[1] == simd]
        if not ws: continue
        u = sum(r[4] for r in ws)
        end = max(r[6] for r in ws) - t0
        units.append(u); ends.append(end)
        eff.append(u * 4 * 32 * 32 / max(1, end))
units, ends, eff = np.array(units), np.array(ends), np.array(eff)
print("SIMDs", len(units), " tile units per SIMD: min/mean/max", units.min(), units.mean(), units.max(), " histogram", sorted(collections.Counter(units.tolist()).items()))
print("loop end per SIMD (cycles): min/median/max", ends.min(), int(np.median(ends)), ends.max())
print("matrix-pipe occupancy until the SIMD's last loop end: min/median/max %.2f %.2f %.2f" % (eff.min(), np.median(eff), eff.max()))
for u in sorted(set(units.tolist())):
    m = units == u
    print("  SIMDs with %2d units: n=%4d  loop end median %6d  (pure MFMA time %6d)  occupancy %.2f" % (u, m.sum(), np.median(ends[m]), u * 4096, np.median(eff[m])))
# a few CUs in detail
for cu in list(by_cu)[:3]:
    print("CU", cu)
    t0 = min(r[5] for r in by_cu[cu])
    for r in sorted(by_cu[cu], key=lambda r: (r[1], r[2])):
        print("   simd %d slot %2d  block %4d Jw %d tiles %d  start %6d loop_end %6d end %6d" % (r[1], r[2], r[8], r[3], r[4], r[5] - t0, r[6] - t0, r[7] - t0))

#!/usr/bin/env python
"""conv2 / conv3 / conv3_1 of FlowNetC under MIOpen's solver-family switches (run once per environment setting)."""
import os, torch, torch.nn.functional as F
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tag = " ".join(f"{k[13:]}={v}" for k, v in os.environ.items() if k.startswith("MIOPEN_DEBUG_")) or "default"
res = []
for name, (n, ci, co, h, w, k, s, p) in {"conv2": (16, 64, 128, 160, 224, 5, 2, 2), "conv3": (16, 128, 256, 80, 112, 5, 2, 2), "conv3_1": (8, 473, 256, 40, 56, 3, 1, 1)}.items():
    x = torch.randn(n, ci, h, w, device="cuda"); wt = torch.randn(co, ci, k, k, device="cuda") * 0.01
    try:
        res.append("%s %.0f us" % (name, t(lambda: F.conv2d(x, wt, None, stride=s, padding=p))))
    except Exception as e:
        res.append("%s failed (%s)" % (name, type(e).__name__))
print("%-40s %s" % (tag, " | ".join(res)))

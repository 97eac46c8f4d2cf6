"""Do the packed-weight caches of functional.py notice the optimizer's in-place update (torch.optim.Adam(fused=True))?  Trains a few steps
twice from the same seed -- once as bench.py does, once dropping every cache before each forward -- and compares the losses."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import functional as Fn, nets  # noqa: E402


def run(invalidate, steps=4, lr=1e-3):
    torch.manual_seed(0)
    P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
    opt = torch.optim.Adam(list(P.values()), lr=lr, fused=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.rand(2, 3, 128, 192, device="cuda", generator=g); b = torch.rand(2, 3, 128, 192, device="cuda", generator=g)
    gt = torch.randn(2, 2, 128, 192, device="cuda", generator=g)
    out = []
    w = P["conv3_1.w"]
    for _ in range(steps):
        if invalidate:
            Fn.invalidate_weight_caches()
        opt.zero_grad(set_to_none=True)
        loss = nets.multiscale_loss(nets.flownet_c_core(P, a - 0.43, b - 0.43, Fn), gt, Fn)
        loss.backward()
        v0 = w._version
        opt.step()
        out.append((float(loss), v0, w._version))
    return out


a = run(False)
b = run(True)
for x, y in zip(a, b):
    print("cached: loss %.9f version %d -> %d | caches dropped every step: loss %.9f | %s" % (x[0], x[1], x[2], y[0], "same" if x[0] == y[0] else "DIFFERENT"))

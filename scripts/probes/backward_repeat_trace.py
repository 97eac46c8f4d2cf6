"""Which tensor of a FlowNetC training step is the first whose value (forward) or gradient (backward) differs between two runs of the same
process on the same inputs?  Every operator output of flownet2_amd.functional is recorded (value + gradient hook)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import functional as Fn, nets  # noqa: E402

B, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2x128x192").split("x"))
P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.rand(B, 3, H, W, device="cuda", generator=g); b = torch.rand(B, 3, H, W, device="cuda", generator=g)
gt = torch.randn(B, 2, H, W, device="cuda", generator=g)

NAMES = ["conv_mfma_relu", "conv_k7s2_relu", "deconv_gemm_relu", "deconv_mfma_relu", "predict_flow_conv", "upsample_flow_deconv", "correlation",
         "conv_bias_leaky_relu", "l1_loss_multi", "downsample"]
rec = None


def wrap(name, fn):
    def f(*args, **kw):
        out = fn(*args, **kw)
        t = out[0] if isinstance(out, tuple) else out
        if isinstance(t, torch.Tensor):
            idx = len(rec["fwd"])
            tag = "%02d %s %s" % (idx, name, tuple(t.shape))
            rec["fwd"].append((tag, t.detach().clone()))
            if t.requires_grad:
                t.register_hook(lambda gr, tag=tag: rec["bwd"].append((tag, gr.detach().clone())))
        return out
    return f


orig = {n: getattr(Fn, n) for n in NAMES if hasattr(Fn, n)}
for n, fn in orig.items():
    setattr(Fn, n, wrap(n, fn))


def run():
    global rec
    rec = {"fwd": [], "bwd": []}
    for v in P.values():
        v.grad = None
    loss = nets.multiscale_loss(nets.flownet_c_core(P, a - 0.43, b - 0.43, Fn), gt, Fn)
    loss.backward()
    torch.cuda.synchronize()
    return rec, {k: v.grad.clone() for k, v in P.items()}


run()
r1, g1 = run()
r2, g2 = run()
print("forward tensors:", len(r1["fwd"]), " differing:", [t for (t, x), (_, y) in zip(r1["fwd"], r2["fwd"]) if not torch.equal(x, y)][:10])
print("gradients w.r.t. operator outputs, in backward order:")
for (t, x), (t2, y) in zip(r1["bwd"], r2["bwd"]):
    assert t == t2
    same = torch.equal(x, y)
    print("  %-60s %s" % (t, "same" if same else "DIFFERENT %.2e" % float((x - y).abs().max() / x.abs().max())))
print("parameter gradients differing:", [k for k in g1 if not torch.equal(g1[k], g2[k])])

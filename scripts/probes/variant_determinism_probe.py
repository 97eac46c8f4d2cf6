"""Are the gradients of a training step independent of the tile variants the autotuner picks?  Runs one FlowNetC step at 2x3x128x192 and
writes every parameter gradient to a file; run twice (FN2_AUTOTUNE=1 / 0) and compare with --compare."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def grads(B=2, H=128, W=192):
    from flownet2_amd import functional as Fn, nets
    P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.rand(B, 3, H, W, device="cuda", generator=g); b = torch.rand(B, 3, H, W, device="cuda", generator=g)
    gt = torch.randn(B, 2, H, W, device="cuda", generator=g)
    for _ in range(2):          # the second pass runs the picked variants
        for v in P.values():
            v.grad = None
        loss = nets.multiscale_loss(nets.flownet_c_core(P, a - 0.43, b - 0.43, Fn), gt, Fn)
        loss.backward()
    return {k: v.grad.cpu() for k, v in P.items()}, float(loss)


if sys.argv[1] == "--compare":
    A, B = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    print("loss", A["loss"], B["loss"])
    for k in A["g"]:
        if not torch.equal(A["g"][k], B["g"][k]):
            d = (A["g"][k] - B["g"][k]).abs().max() / A["g"][k].abs().max()
            print("DIFFERENT %-24s rel max diff %.2e" % (k, float(d)))
    print("compared", len(A["g"]), "parameters")
else:
    shape = tuple(int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (2, 128, 192)
    g, loss = grads(*shape)
    torch.save({"g": g, "loss": loss}, sys.argv[2])

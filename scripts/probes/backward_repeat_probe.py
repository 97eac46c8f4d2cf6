"""Same inputs, same weights, same process: is every parameter gradient of a FlowNetC training step bit-identical from run to run?
Then the same question per operator for the ones that are not."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import functional as Fn, nets, ops  # noqa: E402

B, H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2x128x192").split("x"))
P = {k: v.cuda().requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.rand(B, 3, H, W, device="cuda", generator=g); b = torch.rand(B, 3, H, W, device="cuda", generator=g)
gt = torch.randn(B, 2, H, W, device="cuda", generator=g)


def grads():
    for v in P.values():
        v.grad = None
    loss = nets.multiscale_loss(nets.flownet_c_core(P, a - 0.43, b - 0.43, Fn), gt, Fn)
    loss.backward()
    torch.cuda.synchronize()
    return {k: v.grad.clone() for k, v in P.items()}


grads(); ref = grads()
for rep in range(4):
    cur = grads()
    bad = [k for k in ref if not torch.equal(ref[k], cur[k])]
    print("repeat %d: %d of %d parameter gradients differ from the first run%s" % (rep, len(bad), len(ref), (": " + ", ".join(bad[:8])) if bad else ""))

# per operator
x = torch.randn(B, 194, H // 4, W // 4, device="cuda", generator=g); w = torch.randn(2, 194, 3, 3, device="cuda", generator=g) * 0.05
gg = torch.randn(B, 2, H // 4, W // 4, device="cuda", generator=g)
r0 = ops.predict_flow_conv_backward((x, 0, 194), w, gg, True, True, True)
same = all(all(torch.equal(u, v) for u, v in zip(r0, ops.predict_flow_conv_backward((x, 0, 194), w, gg, True, True, True))) for _ in range(5))
print("predict_flow backward repeatable:", same)
y = torch.randn(B, 64, H // 4, W // 4, device="cuda", generator=g); gy = torch.randn(B, 64, H // 4, W // 4, device="cuda", generator=g)
r0 = ops.bias_leaky_relu_backward(y, (gy, 0, 64), 0.1, True)
same = all(all(torch.equal(u, v) for u, v in zip(r0, ops.bias_leaky_relu_backward(y, (gy, 0, 64), 0.1, True))) for _ in range(5))
print("bias_leaky_relu backward repeatable:", same)

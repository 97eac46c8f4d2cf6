"""Can the whole FlowNetC training step (forward + backward on two streams + fused Adam, capturable) be captured into ONE hipGraph, and what
does the replay cost against host launches?  Round 6, one MI355X: the capture works (warm-up on the capture stream: AccumulateGrad nodes
remember their stream); 8.67 ms per replayed step against 8.46 ms launched from the host -- the 0.66 ms of gaps on the main queue are not
host time a graph removes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flownet2_amd import functional as Fn, nets, parallel

B, H, W = 8, 320, 448
dev = torch.device("cuda:0")


def build(seed=0):
    P = {k: v.to(dev).requires_grad_(True) for k, v in nets.init_params("C", seed=seed).items()}
    opt = torch.optim.Adam(list(P.values()), lr=1e-5, fused=True, capturable=True)
    ex = parallel.GradientExchange(list(P.values()))
    return P, opt, ex


g = torch.Generator(device="cuda").manual_seed(3)
img0 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
img1 = torch.rand(B, 3, H, W, device=dev, generator=g) * 255
gt = torch.randn(B, 2, H, W, device=dev, generator=g) * 5
MEAN = torch.full((3,), -0.43, device=dev)


def make_step(P, opt, ex):
    def step():
        ex.zero_grad()
        tg = nets.loss_targets_ahead(gt, Fn)
        towers = torch.empty((2 * B, 3, H, W), device=dev)
        Fn.scale_shift(img0, 1.0 / 255.0, MEAN, out=towers[:B])
        Fn.scale_shift(img1, 1.0 / 255.0, MEAN, out=towers[B:])
        loss = nets.multiscale_loss(nets.flownet_c_core(P, None, None, Fn, towers=towers), gt, Fn, targets=tg)
        loss.backward()
        ex.finish()
        opt.step()
        return loss
    return step


def timed(f, n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


import functools
print = functools.partial(print, flush=True)
P, opt, ex = build()
step = make_step(P, opt, ex)
for _ in range(8): step()
print("host launches: %.3f ms/step" % timed(step, 40))
torch.cuda.synchronize()
# the whole-network capture recipe: warm up on the side stream the capture will run on (AccumulateGrad nodes remember their stream)
P, opt, ex = build()
step = make_step(P, opt, ex)
cs = torch.cuda.Stream()
cs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cs):
    for _ in range(5): step()
torch.cuda.current_stream().wait_stream(cs)
torch.cuda.synchronize()
print("warm-up on the capture stream done")
gr = torch.cuda.CUDAGraph()
ex.zero_grad()
try:
    with torch.cuda.graph(gr, stream=cs):
        static_loss = step()
except Exception as e:
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:600])
    sys.exit(0)
print("captured")
gr.replay(); torch.cuda.synchronize()
print("hipGraph replay: %.3f ms/step, loss %.6f" % (timed(gr.replay, 40), float(static_loss.detach())))

"""Host-side issue time of one FlowNetC training step next to its device time (is the step launch-bound?) and of opt.step()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flownet2_amd import functional as Fn, nets, parallel

B, H, W = 8, 320, 448
dev = torch.device("cuda:0")
P = {k: v.to(dev).requires_grad_(True) for k, v in nets.init_params("C", seed=0).items()}
opt = torch.optim.Adam(list(P.values()), lr=1e-5, fused=True)
ex = parallel.GradientExchange(list(P.values()))
a = torch.rand(B, 3, H, W, device=dev) - 0.43
b = torch.rand(B, 3, H, W, device=dev) - 0.43
gt = torch.randn(B, 2, H, W, device=dev)
hook_t = [0.0]



def step():
    ex.zero_grad()
    tg = nets.loss_targets_ahead(gt, Fn)
    loss = nets.multiscale_loss(nets.flownet_c_core(P, a, b, Fn), gt, Fn, targets=tg)
    loss.backward()
    ex.finish()
    t = time.perf_counter()
    opt.step()
    hook_t[0] += time.perf_counter() - t


for _ in range(20):
    step()
torch.cuda.synchronize()
hook_t[0] = 0.0
t0 = time.perf_counter()
for _ in range(50):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host issue %.3f ms/step, wall %.3f ms/step, opt.step() host %.3f ms/step" % (
    t_host / 50 * 1e3, t_all / 50 * 1e3, hook_t[0] / 50 * 1e3))

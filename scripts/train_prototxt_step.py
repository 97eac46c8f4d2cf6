#!/usr/bin/env python
"""Time of one FlowNetC TRAIN iteration through the prototxt executor (Net.ClearParamDiffs + ForwardBackward: Caffe-style Backward_gpu of every
layer mirror) next to the same iteration through nets.py + autograd -- batch 8 @448x320, forward + backward only (no optimizer)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import functional as Fn, net as fnet, nets, templates

N, H, W = 8, 320, 448
P = {k: v.cuda() for k, v in nets.init_params("C", seed=0).items()}
g = torch.Generator().manual_seed(1)
img0, img1 = (torch.rand((N, 3, H, W), generator=g) - 0.43).cuda(), (torch.rand((N, 3, H, W), generator=g) - 0.43).cuda()
gt = (torch.randn((N, 2, H, W), generator=g) * 5).cuda()
net = fnet.Net(templates.flownet_c_train_prototxt(N, H, W), phase="TRAIN", device="cuda")
assert net.load_param_dict(P) == []
Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}


def proto_step():
    net.ClearParamDiffs()
    return net.ForwardBackward(img0_nomean=img0, img1_nomean=img1, flow_gt=gt)


def autograd_step():
    for p in Pg.values():
        p.grad = None
    loss = nets.multiscale_loss(nets.flownet_c_core(Pg, img0, img1, Fn), gt, Fn)
    loss.backward()
    return loss


def timeit(fn, n=20, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b = timeit(autograd_step), timeit(proto_step)
print("FlowNetC forward + backward, batch 8 @448x320: nets.py + autograd %.2f ms, prototxt Net.ForwardBackward %.2f ms (losses %.6f / %.6f)"
      % (a, b, float(autograd_step().detach()), float(proto_step())))

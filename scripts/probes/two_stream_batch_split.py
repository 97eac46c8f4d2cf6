#!/usr/bin/env python
"""Round 5 probe: one FlowNetC deploy step of 8 pairs as ONE batch against two half batches (or four quarter batches) on their own HIP
streams -- do independent chains fill each other's underfilled coarse layers?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import functional as Fn, nets

N, H, W = 8, 320, 448
P = {k: v.cuda() for k, v in nets.init_params("C", seed=0).items()}
g = torch.Generator().manual_seed(1)
img0 = (torch.rand((N, 3, H, W), generator=g) * 255).cuda()
img1 = (torch.rand((N, 3, H, W), generator=g) * 255).cuda()
streams = [torch.cuda.Stream() for _ in range(4)]


def one():
    return nets.deploy_forward("C", P, img0, img1, Fn)


def split(k):
    main = torch.cuda.current_stream()
    outs = []
    n = N // k
    for i in range(k):
        st = streams[i]
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(nets.deploy_forward("C", P, img0[i * n:(i + 1) * n], img1[i * n:(i + 1) * n], Fn))
    for i in range(k):
        main.wait_stream(streams[i])
    return outs


def timeit(fn, n=200, warm=60):
    with torch.no_grad():
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = one()
    two = torch.cat(split(2), 0)
print("max |one batch - two half batches| = %.3e px" % float((ref - two).abs().max()))
for rep in range(2):
    print("one batch of 8: %.3f ms   two streams x 4: %.3f ms   four streams x 2: %.3f ms" % (timeit(one), timeit(lambda: split(2)), timeit(lambda: split(4))))

"""Which kernel serves every convolution / deconvolution of a net (FN2_TRACE_CONV=1 prints the convolution routes; library kernels
launched during one forward are listed from torch's profiler)."""
import os
import sys
os.environ["FN2_TRACE_CONV"] = "1"
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flownet2_amd import functional as Fn, nets  # noqa: E402

net = sys.argv[1] if len(sys.argv) > 1 else "2"
B, H, W = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (4, 384, 768)))
P = nets.init_params_flownet2(0) if net == "2" else nets.init_params(net, 0)
Pd = {k: v.cuda() for k, v in P.items()}
i0 = torch.rand(B, 3, H, W, device="cuda") * 255
i1 = torch.rand(B, 3, H, W, device="cuda") * 255
run = (lambda: nets.flownet2_deploy_forward(Pd, i0, i1, Fn)) if net == "2" else (lambda: nets.deploy_forward(net, Pd, i0, i1, Fn))
with torch.no_grad():
    run()
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
names = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        names[e.name] = names.get(e.name, 0) + 1
print("---- kernels of one forward that are not fn2:: ----")
for n, c in sorted(names.items()):
    if "fn2::" not in n:
        print("%3d x %s" % (c, n[:150]))

#!/usr/bin/env python
"""Conv-stack experiments for the FlowNetC forward (memory formats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import functional as Fn, nets
import numpy as np

def run(tag, channels_last=False, steps=20):
    dev = torch.device("cuda")
    P = {k: v.to(dev) for k, v in nets.init_params("C", 0).items()}
    if channels_last:
        P = {k: (v.contiguous(memory_format=torch.channels_last) if (v.dim() == 4 and not k.startswith(("Convolution", "upsample"))) else v) for k, v in P.items()}
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.integers(0, 256, (8, 3, 320, 448)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.integers(0, 256, (8, 3, 320, 448)).astype(np.float32)).to(dev)
    def fwd():
        if not channels_last:
            return nets.deploy_forward("C", P, a, b, Fn)
        # same graph, activations kept channels_last between convs
        orig_conv = nets._conv
        return nets.deploy_forward("C", P, a, b, Fn)
    with torch.no_grad():
        for _ in range(4):
            out = fwd()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            out = fwd()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
    print(f"{tag}: {dt*1e3:.3f} ms/step  {8/dt:.1f} pairs/s  checksum {float(out.abs().mean()):.6f}", flush=True)

run("nchw")
run("channels_last weights", channels_last=True)

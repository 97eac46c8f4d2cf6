#!/usr/bin/env python
"""Known-byte-count streaming kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports wide coalesced reads by 2x; other access
widths must be calibrated).  channel_norm_fwd reads 4 B per lane, coalesced (the same access width as the
correlation staging loads) and writes 1/C of that; the tensor is larger than the 256 MiB Infinity Cache."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import ops
N, C, H, W = 8, 64, 512, 512          # 512 MiB read, 8 MiB written
x = torch.randn(N, C, H, W, device="cuda")
for _ in range(3):
    y = ops.channel_norm_forward(x)
torch.cuda.synchronize()
print("channel_norm_fwd read_bytes", x.numel() * 4, "write_bytes", y.numel() * 4)
# write calibration: flow_warp_fwd with zero flow writes as many bytes as it reads
img = torch.randn(4, 32, 512, 512, device="cuda")     # 128 MiB
flow = torch.zeros(4, 2, 512, 512, device="cuda")
for _ in range(3):
    w = ops.flow_warp_forward(img, flow)
torch.cuda.synchronize()
print("flow_warp_fwd read_bytes", img.numel() * 4 + flow.numel() * 4, "write_bytes", w.numel() * 4)

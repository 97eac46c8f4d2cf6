"""One Winograd variant on one layer shape, a few launches (for rocprofv3 --pmc): python scripts/probes/wino_pmc_probe.py <variant> [layer]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
v = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, Cin, H, W, Cout = (8, 473, 40, 56, 256) if (len(sys.argv) < 3 or sys.argv[2] == "conv3_1") else (8, 512, 20, 28, 512)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.02
b = torch.zeros(Cout, device="cuda")
pu = ops.conv_wino_pack_weights(w)
out = torch.empty(N, Cout, H, W, device="cuda")
ops.set_wino_variant(v)
for _ in range(5):
    ops.conv_wino_forward(x, pu, b, Cout, 1, True, 0.1, out=out)
torch.cuda.synchronize()

"""One small-map convolution variant on one layer shape, a few launches (for rocprofv3 --pmc):
python scripts/probes/plane_pmc_probe.py <variant> [conv5_1|conv6_1|conv5|conv6|conv4] [ksplit]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
layer = sys.argv[2] if len(sys.argv) > 2 else "conv5_1"
N, Cin, H, W, Cout, s = {"conv5_1": (8, 512, 10, 14, 512, 1), "conv6_1": (8, 1024, 5, 7, 1024, 1), "conv5": (8, 512, 20, 28, 512, 2),
                         "conv6": (8, 512, 10, 14, 1024, 2), "conv4": (8, 256, 40, 56, 512, 2)}[layer]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.02
b = torch.zeros(Cout, device="cuda")
pw = ops.conv_mfma_pack_weights(w)
if len(sys.argv) > 3:
    ops.set_plane_ksplit(int(sys.argv[3]))
ops.set_plane_variant(v)
for _ in range(5):
    ops.conv_plane_forward(x, pw, b, Cout, s, 1, True, 0.1)
torch.cuda.synchronize()

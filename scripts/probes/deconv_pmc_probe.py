"""One small-map deconvolution variant on one layer shape, a few launches (for rocprofv3 --pmc):
python scripts/probes/deconv_pmc_probe.py <variant> [deconv2|deconv3|deconv4|deconv5]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
v = int(sys.argv[1]) if len(sys.argv) > 1 else 41
layer = sys.argv[2] if len(sys.argv) > 2 else "deconv2"
N, Cin, H, W, Cout = {"deconv5": (8, 1024, 5, 7, 512), "deconv4": (8, 1026, 10, 14, 256), "deconv3": (8, 770, 20, 28, 128), "deconv2": (8, 386, 40, 56, 64)}[layer]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
w = torch.randn(Cin, Cout, 4, 4, device="cuda", generator=g) * 0.02
b = torch.zeros(Cout, device="cuda")
pw = ops.deconv_plane_pack_weights(w)
ops.set_plane_variant(v)
for _ in range(5):
    ops.deconv_plane_forward(x, pw, b, Cout, True, 0.1)
torch.cuda.synchronize()

#!/usr/bin/env python
"""Records flownet2_amd/tuning/gemm_gfx950.csv: PyTorch TunableOp times the rocBLAS / hipBLASLt candidates for every GEMM
shape the GEMM route issues in a FlowNetC (batch 8 @448x320) and a FlowNet2 (batch 4 @768x384) forward.  Run on the GPU box:
    python scripts/tune_gemms.py gpurun_out/gemm_gfx950.csv      and copy the file into flownet2_amd/tuning/."""
import os, sys
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "gemm_gfx950.csv")
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "100")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "10")
import torch
import torch.cuda.tunable as tn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flownet2_amd import functional as Fn, nets
tn.set_filename(out, insert_device_ordinal=False)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    P = {k: v.to(dev) for k, v in nets.init_params("C", seed=0).items()}
    a, b = (torch.rand(8, 3, 320, 448, device=dev, generator=g) * 255 for _ in range(2))
    for _ in range(2): nets.deploy_forward("C", P, a, b, Fn)
    del P
    P2 = {k: v.to(dev) for k, v in nets.init_params_flownet2(seed=0).items()}
    a, b = (torch.rand(4, 3, 384, 768, device=dev, generator=g) * 255 for _ in range(2))
    for _ in range(2): nets.flownet2_deploy_forward(P2, a, b, Fn)
torch.cuda.synchronize()
tn.write_file(out) if hasattr(tn, "write_file") else None
print("wrote", out, os.path.getsize(out) if os.path.exists(out) else "(missing)")

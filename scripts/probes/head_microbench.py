import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flownet2_amd import ops
for (N, C, H, W) in [(8,1024,5,7),(8,1026,10,14),(8,770,20,28),(8,386,40,56),(8,194,80,112)]:
    x = torch.randn(N,C,H,W,device="cuda"); w = torch.randn(2,C,3,3,device="cuda")*0.01; b = torch.zeros(2,device="cuda")
    for _ in range(3): ops.predict_flow_conv_forward(x,w,b)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.predict_flow_conv_forward(x,w,b)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1)/50*1e-3
    print(f"predict_flow [{N},{C},{H},{W}]: {t*1e6:.1f} us  {x.numel()*4/t/1e9:.0f} GB/s")

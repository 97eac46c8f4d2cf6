import torch, torch.nn.functional as F
dev="cuda"
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for name,(ci,co,h,w) in {"deconv5":(1024,512,5,7),"deconv4":(1026,256,10,14),"deconv3":(770,128,20,28),"deconv2":(386,64,40,56)}.items():
    x=torch.randn(8,ci,h,w,device=dev); wt=torch.randn(ci,co,4,4,device=dev)*0.01
    ref=F.conv_transpose2d(x,wt,None,stride=2,padding=1)
    # sub-pixel weights: out[2y+py, 2x+px] = sum_{dy,dx in {0,1}} in[y+py-1+dy, x+px-1+dx] * W[ky,kx], ky = 3-py-2dy? derive: Y = 2*iy - 1 + ky -> ky = Y+1-2iy
    w2=torch.empty(4*co,ci,2,2,device=dev)
    for py in range(2):
        for px in range(2):
            for dy in range(2):
                for dx in range(2):
                    # conv2d(k=2,pad=1): out'[yy,xx] = sum in[yy-1+dy, xx-1+dx]*w[dy,dx]; class output y uses yy = y+py  -> iy = y+py-1+dy ; ky = (2y+py)+1-2iy = 3 - py - 2dy... 
                    ky = (py+1) - 2*(py-1+dy); kx = (px+1) - 2*(px-1+dx)
                    w2[(py*2+px)*co:(py*2+px+1)*co,:,dy,dx] = wt[:,:,ky,kx].t()
    def sub():
        r=F.conv2d(x,w2,None,padding=1)     # [8,4co,h+1,w+1]
        out=torch.empty(8,co,2*h,2*w,device=dev)
        for py in range(2):
            for px in range(2):
                out[:,:,py::2,px::2]=r[:,(py*2+px)*co:(py*2+px+1)*co,py:py+h,px:px+w]
        return out
    out=sub()
    err=(out-ref).abs().max().item()
    tc=t(lambda: F.conv2d(x,w2,None,padding=1))
    print(name,"conv_transpose %.1f us | k2 conv only %.1f us | k2 conv + 4 slice copies %.1f us | max err %.2e (ref max %.2f)"%(t(lambda: F.conv_transpose2d(x,wt,None,stride=2,padding=1)), tc, t(sub), err, ref.abs().max().item()))

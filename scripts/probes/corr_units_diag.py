import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
from flownet2_amd import ops, _lib
L=_lib.lib(); P=ops.corr_params(20,1,20,1,2)
H,W=40,56
shape=(1,32,H,W)
g=torch.Generator(device="cuda").manual_seed(0)
xr=torch.randn(*shape,device="cuda",generator=g); yr=torch.randn(*shape,device="cuda",generator=g)
def run(impl,x,y):
    L.fn2_debug_set_correlation_impl(impl); got=torch.full((1,441,H,W),float('nan'),device="cuda"); ops.correlation_forward(P,x,y,out=got); torch.cuda.synchronize(); return got
A=torch.zeros(*shape,device="cuda"); B=torch.zeros(*shape,device="cuda")
pix=(torch.arange(H*W,device="cuda",dtype=torch.float32)+1).reshape(H,W)
A[0,0]=pix*32; B[0,0]=1.0
want=run(19,A,B)
for impl in (20,27):
    run(impl,xr,yr)
    got=run(impl,A,B)
    d=(got!=want).cpu().numpy()[0].reshape(21,21,H,W); gw=got.cpu().numpy()[0].reshape(21,21,H,W); ww=want.cpu().numpy()[0].reshape(21,21,H,W)
    print("impl",impl,"differing",d.sum())
    idx=np.argwhere(d)
    sel=idx[::max(1,len(idx)//40)][:40]
    for i in sel:
        q,o,y,x=[int(v) for v in i]; gv=gw[q,o,y,x]; wv=ww[q,o,y,x]
        src=int(round(gv))-1
        print("  (qq %2d oo %2d y %2d x %2d): want %7.1f (pixel y %d x %d)  got %12.4f -> pixel y %d x %d" % (q,o,y,x,wv,(int(wv)-1)//W if wv>0 else -1,(int(wv)-1)%W if wv>0 else -1,gv,src//W if 0<=src<H*W else -9,src%W if 0<=src<H*W else -9))

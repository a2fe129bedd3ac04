import sys, os, torch
sys.path.insert(0, os.getcwd())
from sudo_rm_rf_amd import ops, _lib
DEV="cuda:0"
def rnd(*s, seed=0, scale=1.0):
    g=torch.Generator().manual_seed(seed); return (torch.randn(*s, generator=g)*scale).to(DEV)
for (Bt,Cin,Cout,L) in [(32,512,256,3200),(32,256,512,3200),(32,512,256,3200),(16,512,512,3200),(32,128,256,3200),(32,512,256,1600)]:
    for pro in (0,2,1,3):
        x=rnd(Bt,Cin,L,seed=1); w=rnd(Cout,Cin,1,seed=2,scale=Cin**-0.5); b=rnd(Cout,seed=3)
        kw={}
        if pro in (1,2):
            sums=torch.zeros(Bt,64,2,dtype=torch.float64,device=DEV)
            xf=x.double().reshape(Bt,-1); sums[:,0,0]=xf.sum(1); sums[:,0,1]=(xf*xf).sum(1)
            kw.update(in_sums=sums, in_gamma=rnd(Cin,seed=4)+1, in_beta=rnd(Cin,seed=5))
        if pro in (2,3): kw.update(in_prelu=torch.tensor([0.2],device=DEV))
        res=rnd(Bt,Cout,L,seed=6)
        ops.set_debug_flags(0); a=ops.pw_conv(x,w,b,residual=res,**kw)
        ops.set_debug_flags(1<<27); c=ops.pw_conv(x,w,b,residual=res,**kw)
        ops.set_debug_flags(2048); d=ops.pw_conv(x,w,b,residual=res,**kw)
        ops.set_debug_flags(0)
        print((Bt,Cin,Cout,L),'pro',pro,'buf-vs-ptr max diff %.3e'%(a-c).abs().max().item(),'ptr-vs-w8 %.3e'%(c-d).abs().max().item(), 'nan' if not torch.isfinite(a).all() else '')

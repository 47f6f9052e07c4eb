import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surface_texture_inpainting_net_amd import functional as SF
dev='cuda:0'
def t(fn, n=10):
    for _ in range(3): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
for (M,Nc,K) in [(131072,1024,256),(131072,256,256),(131072,1024,64),(16384,1024,256),(18063,1024,256)]:
    A=torch.randn(M,K,device=dev); W=torch.randn(Nc,K,device=dev)*0.05
    o=torch.empty(M,Nc,device=dev)
    Ws=SF.split_weights(W, SF.GEMM_F16X3|0x400)
    ts=t(lambda: SF.gemm_nt(A,Ws,out=o,precision=SF.GEMM_F16X3|SF.GEMM_W_PRESPLIT|0x400))
    Wk=SF.split_weights(W, SF.GEMM_F16X3)
    tt=t(lambda: SF.gemm_nt(A,Wk,out=o,precision=SF.GEMM_F16X3|SF.GEMM_W_PRESPLIT))
    fl=3*2.0*M*Nc*K
    print(M,Nc,K,'tiled %.0f us (%.0f TF)  strip %.0f us (%.0f TF = %.2f of 2.5 PF)' % (tt, fl/tt/1e6, ts, fl/ts/1e6, fl/ts/1e6/2500))

import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd'); K = pkg.kernels
dev='cuda'; B,T=8,6656
def timeit(fn, flop, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/n
    return ms, flop/ms/1e9
M,C=1024,4096
for kind in ('random','zeros','ones','small-int'):
    if kind=='random': x=torch.randn(B,C,T,device=dev); w=torch.randn(C,M,device=dev)*0.02
    elif kind=='zeros': x=torch.zeros(B,C,T,device=dev); w=torch.zeros(C,M,device=dev)
    elif kind=='ones': x=torch.ones(B,C,T,device=dev); w=torch.ones(C,M,device=dev)
    else: x=torch.randint(-2,3,(B,C,T),device=dev).float(); w=torch.randint(-2,3,(C,M),device=dev).float()
    out=torch.empty(B,M,T,device=dev)
    for tile in (22,):
        ms,tf=timeit(lambda: K.conv_gemm(x0=x,w=w,out0=out,B=B,T_in=T,T_out=T,M=M,C0=C,taps=[0],tile=tile), 2.0*B*T*C*M, n=10)
        print('%-10s STORE M=%4d C=%4d tile=%d  %.3f ms %.1f TF/s' % (kind,M,C,tile,ms,tf), flush=True)
    del x,w,out

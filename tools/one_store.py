import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd'); K = pkg.kernels
dev='cuda'; B,T=8,6656; M,C=1024,4096; tile=int(sys.argv[1]) if len(sys.argv)>1 else 22
x=torch.randn(B,C,T,device=dev); w=torch.randn(C,M,device=dev)*0.02; out=torch.empty(B,M,T,device=dev)
for _ in range(3):
    K.conv_gemm(x0=x,w=w,out0=out,B=B,T_in=T,T_out=T,M=M,C0=C,taps=[0],tile=tile)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Run one engine configuration a few times (for rocprofv3 --pmc passes).
usage: one_kernel.py <gate|out|dgrad|wgrad> <tile-or-splits> [dilation]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
what, tile = sys.argv[1], int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = 'cuda'
B, T, R, S = 8, 6656, 256, 512
Tz = T // 64
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
net, gated, th, sg = r(B, R, T), r(B, R, T), r(B, R, T), r(B, R, T)
skip, net2, dpre = r(B, S, T), r(B, R, T), r(B, 2 * R, T)
wg, bg, cond = r(3, R, 2 * R) * 0.03, r(2 * R), r(B, 2 * R, Tz)
wo, bo = r(R, S + R) * 0.05, r(S + R)
wgT = r(3, 2 * R, R) * 0.03
dwg = torch.zeros(3, R, 2 * R, device=dev)
for _ in range(5):
    if what == 'gate':
        K.conv_gemm(x0=net, w=wg, bias=bg, out0=gated, save0=th, save1=sg, cond=cond, cond_T=Tz, B=B, T_in=T, T_out=T,
                    M=2 * R, C0=R, taps=[-2 * d, -d, 0], epilogue=K.EPI_GATE, tile=tile)
    elif what == 'out':
        K.conv_gemm(x0=gated, w=wo, bias=bo, out0=skip, out1=net2, aux1=net, B=B, T_in=T, T_out=T, M=S + R, M0=S, C0=R,
                    taps=[0], epilogue=K.EPI_ACCUM_SPLIT, tile=tile)
    elif what == 'dgrad':
        K.conv_gemm(x0=dpre, w=wgT, out1=net2, aux1=net2, out0=net2, B=B, T_in=T, T_out=T, M=R, M0=0, C0=2 * R,
                    taps=[2 * d, d, 0], epilogue=K.EPI_ACCUM_SPLIT, tile=tile)
    else:
        K.wgrad_gemm(p=net, q0=dpre, dw=dwg, B=B, T_q=T, T_p=T, Cp=R, Q0=2 * R, taps=[-2 * d, -d, 0], splits=tile)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Micro-benchmark of the MFMA engines on the decoder-layer shapes (B=8, T=6656, fp32).
Prints TFLOP/s per (kernel, tile) measured with HIP events on the launch stream."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
dev = 'cuda'
B, T, R, S = 8, 6656, 256, 512
Tz = T // 64
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
net, gated, th, sg = r(B, R, T), r(B, R, T), r(B, R, T).tanh(), r(B, R, T).sigmoid()
skip, net2, dpre = r(B, S, T), r(B, R, T), r(B, 2 * R, T)
wg, bg, cond = r(3, R, 2 * R) * 0.03, r(2 * R), r(B, 2 * R, Tz)
wo, bo = r(R, S + R) * 0.05, r(S + R)
woT, wgT = r(S + R, R) * 0.05, r(3, 2 * R, R) * 0.03
dwg, dwo = torch.zeros(3, R, 2 * R, device=dev), torch.zeros(R, S + R, device=dev)


def timeit(fn, flop, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms, flop / ms / 1e9


N = B * T
for d in (1, 64, 512):
    for tile in (24, 22, 21):
        ms, tf = timeit(lambda: K.conv_gemm(x0=net, w=wg, bias=bg, out0=gated, save0=th, save1=sg, cond=cond, cond_T=Tz,
                                            B=B, T_in=T, T_out=T, M=2 * R, C0=R, taps=[-2 * d, -d, 0],
                                            epilogue=K.EPI_GATE, tile=tile), 2.0 * N * 3 * R * 2 * R)
        print('gate fwd   d=%-3d tile=%d  %.3f ms  %.1f TF/s' % (d, tile, ms, tf), flush=True)
for tile in (24, 22, 21, 14, 12):
    ms, tf = timeit(lambda: K.conv_gemm(x0=gated, w=wo, bias=bo, out0=skip, out1=net2, aux1=net, B=B, T_in=T, T_out=T,
                                        M=S + R, M0=S, C0=R, taps=[0], epilogue=K.EPI_ACCUM_SPLIT, tile=tile),
                    2.0 * N * R * (S + R))
    print('out 1x1    tile=%d  %.3f ms  %.1f TF/s' % (tile, ms, tf), flush=True)
for tile in (24, 22, 21, 12):
    ms, tf = timeit(lambda: K.conv_gemm(x0=skip, x1=net2, w=woT, out0=dpre, aux0=th, aux1=sg, B=B, T_in=T, T_out=T, M=R,
                                        C0=S, C1=R, taps=[0], epilogue=K.EPI_GATE_BWD, tile=tile), 2.0 * N * R * (S + R))
    print('gate bwd   tile=%d  %.3f ms  %.1f TF/s' % (tile, ms, tf), flush=True)
for d in (1, 512):
    for tile in (24, 22, 21, 12):
        ms, tf = timeit(lambda: K.conv_gemm(x0=dpre, w=wgT, out1=net2, aux1=net2, out0=net2, B=B, T_in=T, T_out=T, M=R,
                                            M0=0, C0=2 * R, taps=[2 * d, d, 0], epilogue=K.EPI_ACCUM_SPLIT, tile=tile),
                        2.0 * N * 3 * R * 2 * R)
        print('dgrad      d=%-3d tile=%d  %.3f ms  %.1f TF/s' % (d, tile, ms, tf), flush=True)
for d in (1, 512):
    for splits in (0, 5, 6, 7, 8, 10, 13):
        ms, tf = timeit(lambda: K.wgrad_gemm(p=net, q0=dpre, dw=dwg, B=B, T_q=T, T_p=T, Cp=R, Q0=2 * R,
                                             taps=[-2 * d, -d, 0], splits=splits), 2.0 * N * 3 * R * 2 * R)
        print('wgrad gate d=%-3d splits=%d  %.3f ms  %.1f TF/s' % (d, splits, ms, tf), flush=True)
for splits in (0, 6, 7, 8, 9, 10, 11, 12, 16):
    ms, tf = timeit(lambda: K.wgrad_gemm(p=gated, q0=skip, q1=net2, dw=dwo, B=B, T_q=T, T_p=T, Cp=R, Q0=S, Q1=R, lddw=S + R,
                                         taps=[0], splits=splits), 2.0 * N * R * (S + R))
    print('wgrad out  splits=%d  %.3f ms  %.1f TF/s' % (splits, ms, tf), flush=True)

#!/usr/bin/env python3
"""The condition projection (all 31 add_condition 1x1s as one GEMM, wavenet_ops.py:93-101) and its gradients at the benchmark shape,
per tile of the fp32 conv engine:  python tools/cond_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
B, Cc, Tz, M = 8, 128, 104, 30 * 512 + 512
dev = 'cuda'
cond, w = torch.randn(B, Cc, Tz, device=dev), torch.randn(Cc, M, device=dev) * 0.05
wt = w.t().contiguous()
out, dce, dcond = torch.empty(B, M, Tz, device=dev), torch.randn(B, M, Tz, device=dev), torch.empty(B, Cc, Tz, device=dev)
dw = torch.zeros(Cc, M, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for tile in (0, 11, 12, 21, 22):
    f = timeit(lambda: K.conv_gemm(x0=cond, w=w, out0=out, B=B, T_in=Tz, T_out=Tz, M=M, C0=Cc, taps=[0], tile=tile))
    g = timeit(lambda: K.conv_gemm(x0=dce, w=wt, out0=dcond, B=B, T_in=Tz, T_out=Tz, M=Cc, C0=M, taps=[0], tile=tile))
    print('tile %2d: forward %6.1f us   input gradient %6.1f us' % (tile, f, g), flush=True)
for sk in (0, 2, 4, 8):
    try:
        g = timeit(lambda: (dcond.zero_() if sk else None, K.conv_gemm(x0=dce, w=wt, out0=dcond, B=B, T_in=Tz, T_out=Tz, M=Cc, C0=M, taps=[0], tile=12, split_k=-sk if sk else 0)))
        print('input gradient, tile 12, split_k %d: %6.1f us' % (sk, g), flush=True)
    except Exception as e:
        print('split_k', sk, type(e).__name__, e)
print('weight gradient %6.1f us' % timeit(lambda: K.wgrad_gemm(p=cond, q0=dce, dw=dw, B=B, T_q=Tz, T_p=Tz, Cp=Cc, Q0=M, taps=[0])))
# the dedicated fp32-MFMA kernels (csrc/cond_proj.hip: one tile per wave, operands straight from global memory)
for cc in (80, 128):
    c2, w2 = cond[:, :cc].contiguous(), w[:cc].contiguous()
    dc2, dw2 = torch.empty(B, cc, Tz, device=dev), torch.zeros(cc, M, device=dev)
    scratch = torch.empty(K.cond_proj_dgrad_scratch(B, cc, M, Tz), device=dev)
    f = timeit(lambda: K.cond_proj_fwd(c2, w2, out, B=B, Cc=cc, Mall=M, Tz=Tz))
    g = timeit(lambda: K.cond_proj_dgrad(w2, dce, dc2, scratch, B=B, Cc=cc, Mall=M, Tz=Tz))
    h = timeit(lambda: K.cond_proj_wgrad(c2, dce, dw2, B=B, Cc=cc, Mall=M, Tz=Tz))
    print('cond_proj kernels, Cc = %3d: forward %6.1f us   input gradient (+ reduce) %6.1f us   weight gradient %6.1f us' % (cc, f, g, h), flush=True)

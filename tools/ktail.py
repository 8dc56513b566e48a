#!/usr/bin/env python3
"""Gate conv / input gradient at B=8, T=6656 with different numbers of main-tile columns handed to the
half-width tail tiles (tile = main + 10000*n)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
dev = 'cuda'
B, T, R = 8, 6656, 256
Tz = T // 64
torch.manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
net, gated, th, sg = r(B, R, T), r(B, R, T), r(B, R, T), r(B, R, T)
net2, dpre = r(B, R, T), r(B, 2 * R, T)
wg, bg, cond = r(3, R, 2 * R) * 0.03, r(2 * R), r(B, 2 * R, Tz)
wgT = r(3, 2 * R, R) * 0.03
N = B * T


def timeit(fn, flop, n=20):
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


for d in (2, 64):
    row = 'd=%-3d gate t22:' % d
    for nt in (0, 2, 4, 6, 8, 12, 16, 26):
        tile = 22 + 10000 * nt if nt else 22 + 10000 * 99   # 99 >= n_nt: no tail
        ms, tf = timeit(lambda: K.conv_gemm(x0=net, w=wg, bias=bg, out0=gated, save0=th, save1=sg, cond=cond, cond_T=Tz,
                                            B=B, T_in=T, T_out=T, M=2 * R, C0=R, taps=[-2 * d, -d, 0],
                                            epilogue=K.EPI_GATE, tile=tile), 2.0 * N * 3 * R * 2 * R)
        row += ' n%d %.0f' % (nt, tf)
    print(row, flush=True)
    row = 'd=%-3d dgrad t12:' % d
    for nt in (0, 2, 4, 6, 8, 12, 16, 26):
        tile = 12 + 10000 * nt if nt else 12 + 10000 * 99
        ms, tf = timeit(lambda: K.conv_gemm(x0=dpre, w=wgT, out1=net2, aux1=net2, out0=net2, B=B, T_in=T, T_out=T, M=R,
                                            M0=0, C0=2 * R, taps=[2 * d, d, 0], epilogue=K.EPI_ACCUM_SPLIT, tile=tile),
                        2.0 * N * 3 * R * 2 * R)
        row += ' n%d %.0f' % (nt, tf)
    print(row, flush=True)

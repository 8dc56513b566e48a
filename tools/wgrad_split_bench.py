"""vqw_f16x3_wgrad at the decoder's shapes as a function of the number of K splits (blocks = tiles x splits; a multiple of 8
puts the tiles of one K range on the same XCD, where they share operand panels in L2).  GPU box: python tools/wgrad_split_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
DEV = 'cuda'


def timeit(fn, n=30):
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


def main():
    B, T, R, S = 8, 6656, 256, 512
    net = torch.randn(B, R, T, device=DEV)
    dpre = torch.randn(B, 2 * R, T, device=DEV) * 1e-5
    dskip = torch.randn(B, S, T, device=DEV) * 1e-5
    dnet = torch.randn(B, R, T, device=DEV) * 1e-5
    slab = torch.empty(320 * 65536, device=DEV)
    sc = torch.tensor([2.0 ** 10, 2.0 ** 27], device=DEV)
    dw = torch.zeros(3, R, 2 * R, device=DEV)
    for d in (64, 1):
        for ns, grp in ((0, 2), (42, 2), (40, 2), (32, 2), (40, 1), (42, 1), (32, 1)):
            t = timeit(lambda: K.f16x3_wgrad(p=net, q0=dpre, dw=dw, slab=slab, B=B, T=T, Cp=R, Q0=2 * R, taps=[-2 * d, -d, 0],
                                             p_scale=sc[0:1], q0_scale=sc[1:2], nsplit=ns, mode=0, xcd_group=grp))
            print('gate wgrad d=%d nsplit %2d %s: %6.1f us' % (d, ns, 'tap groups per XCD' if grp == 1 else '', t), flush=True)
    dw2 = torch.zeros(R, S + R, device=DEV)
    for ns in (0, 85, 80, 72, 64, 88, 96):
        t = timeit(lambda: K.f16x3_wgrad(p=net, q0=dskip, q1=dnet, Q1=R, dw=dw2, slab=slab, B=B, T=T, Cp=R, Q0=S, taps=[0],
                                         p_scale=sc[0:1], q0_scale=sc[1:2], q1_scale=sc[1:2], nsplit=ns, mode=0))
        print('1x1 wgrad nsplit %2d: %6.1f us' % (ns, t), flush=True)
    F = 768
    x = torch.randn(B, F, 3328, device=DEV)
    dy = torch.randn(B, F, 1664, device=DEV) * 1e-5
    dw3 = torch.zeros(5, F, F, device=DEV)
    for ns, grp in ((5, 2), (4, 2), (5, 1), (4, 1), (3, 1)):
        t = timeit(lambda: K.f16x3_wgrad(p=x, q0=dy, dw=dw3, slab=slab, B=B, T=1664, Cp=F, Q0=F, taps=[j - 1 for j in range(5)],
                                         p_stride=2, T_p=3328, nsplit=ns, mode=0, xcd_group=grp))
        print('encoder layer-1 wgrad nsplit %2d %s: %6.1f us' % (ns, 'tap groups per XCD' if grp == 1 else '', t), flush=True)


if __name__ == '__main__':
    main()

"""Stand-alone timings of the encoder's strided convs (layers 1..3 of the benchmark: 768 -> 768, k=5, stride 2, batch 8) on the
fp16x3 engine, per block shape, next to the fp32 engine's kernels.  GPU box: python tools/sconv_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
DEV = 'cuda'


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


def main():
    B, F, ks, pl = 8, 768, 5, 1
    for Tout in (1664, 832, 416, 208, 104):
        Tin = 2 * Tout
        x = torch.randn(B, F, Tin, device=DEV)
        w = torch.randn(ks, F, F, device=DEV) * 0.02
        wt = w.permute(0, 2, 1).contiguous()
        dy = torch.randn(B, F, Tout, device=DEV)
        bias = torch.randn(F, device=DEV)
        xp = torch.empty(2 * B * F * Tin, dtype=torch.float16, device=DEV)
        dyp = torch.empty(2 * B * F * Tout, dtype=torch.float16, device=DEV)
        wp = torch.empty(2 * ks * F * F, dtype=torch.float16, device=DEV)
        wtp = torch.empty_like(wp)
        K.f16x3_split_activations(x, xp, B, F, Tin, mode=K.X3_S2D)
        K.f16x3_split_activations(dy, dyp, B, F, Tout, mode=0)
        K.f16x3_pack_weights(w, wp, ks * F, F, F, 16.0, mode=0)
        K.f16x3_pack_weights(wt, wtp, ks * F, F, F, 16.0, mode=0)
        out, r, dx = torch.empty(B, F, Tout, device=DEV), torch.empty(B, F, Tout, device=DEV), torch.empty(B, F, Tin, device=DEV)
        gf = 2.0 * B * Tout * F * F * ks / 1e9
        row = ['T_out %4d (%.1f GFLOP):' % (Tout, gf)]
        for shape in (1, 2, 3, 4, 5):
            tf = timeit(lambda: K.f16x3_strided_conv(xp=xp, wp=wp, out=out, save_r=r, B=B, T=Tout, Cin=F, M=F, ks=ks, pad_left=pl, bias=bias,
                                                     bn_scale=bias, bn_shift=bias, relu=True, w_scale_inv=1 / 16.0, shape=shape))
            tb = timeit(lambda: K.f16x3_strided_conv(xp=dyp, wp=wtp, out=dx, B=B, T=Tout, Cin=F, M=F, ks=ks, pad_left=pl, dgrad=True,
                                                     w_scale_inv=1 / 16.0, shape=shape))
            row.append('shape %d fwd %6.1f us dgrad %6.1f us |' % (shape, tf, tb))
        sslab = torch.empty(256 * 65536, device=DEV)
        cnt = torch.zeros(1024, dtype=torch.int32, device=DEV)
        if os.environ.get('SCONV_SPLIT_SWEEP', '1') != '0':
            for shape in (2,):
                for S in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 15):
                    res = []
                    for dg in (False, True):
                        kw = dict(xp=dyp, wp=wtp, out=dx, dgrad=True) if dg else dict(xp=xp, wp=wp, out=out, save_r=r, bias=bias, bn_scale=bias, bn_shift=bias, relu=True)
                        try:
                            res.append('%6.1f' % timeit(lambda: K.f16x3_strided_conv(B=B, T=Tout, Cin=F, M=F, ks=ks, pad_left=pl, w_scale_inv=1 / 16.0,
                                                                                     shape=0 if S == 0 else shape, split_slab=sslab, split_counters=cnt, ksplit=S, **kw)))
                        except RuntimeError:
                            res.append('   n/a')
                    print('   T_out %4d split: shape %d ksplit %2d fwd %s us dgrad %s us' % (Tout, shape, S, res[0], res[1]), flush=True)
        t32 = timeit(lambda: K.conv_gemm(x0=x, w=w, bias=bias, out0=out, save0=r, scale=bias, shift=bias, B=B, T_in=Tin, T_out=Tout, M=F, C0=F,
                                         in_stride=2, taps=[j - pl for j in range(ks)], out_relu=True))
        row.append('fp32 engine fwd %6.1f us' % t32)

        def fp32_dgrad():
            for p_ in (0, 1):
                j0 = (p_ + pl) % 2
                js = list(range(j0, ks, 2))
                K.conv_gemm(x0=dy, w=wt[j0:], w_tap_stride=2 * F * F, out0=dx, B=B, T_in=Tout, T_out=(Tin - p_ + 1) // 2, M=F, C0=F,
                            taps=[(p_ + pl - j) // 2 for j in js], out_tstride=2, out_toffset=p_, T_store=Tin)
        row.append('dgrad %6.1f us' % timeit(fp32_dgrad))
        ts = timeit(lambda: K.f16x3_split_activations(x, xp, B, F, Tin, mode=K.X3_S2D))
        row.append('| s2d split %5.1f us' % ts)
        slab = torch.empty(256 * 65536, device=DEV)
        dw = torch.zeros(ks, F, F, device=DEV)
        if Tout % 32:
            print(' '.join(row), flush=True)
            continue
        tw = timeit(lambda: K.f16x3_wgrad(p=x, q0=dy, dw=dw, slab=slab, B=B, T=Tout, Cp=F, Q0=F, taps=[j - pl for j in range(ks)], p_stride=2, T_p=Tin, mode=0))
        row.append('| wgrad %6.1f us' % tw)
        print(' '.join(row), flush=True)


if __name__ == '__main__':
    main()

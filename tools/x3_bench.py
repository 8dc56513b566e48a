#!/usr/bin/env python3
"""Stand-alone timings of the fp16x3 engine's kernels at the benchmark shapes (B=8, T=6656, R=256, S=512), one kernel
at a time on an otherwise idle GPU (HIP events, median of `reps`):  python tools/x3_bench.py [reps]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print('VQW_X3_HALF =', os.environ.get('VQW_X3_HALF', K.DEFAULT_X3_HALF))
B, T, R, S, ks = int(os.environ.get('XB', '8')), int(os.environ.get('XT', '6656')), 256, 512, 3
print('B, T =', B, T)
dev = 'cuda'
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731


def timeit(name, fn, flop):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    print('%-34s %8.1f us   %6.1f TFLOP/s fp32-equivalent   (min %.1f)' % (name, med, flop / med / 1e6, ts[0]), flush=True)


net, gated = rnd(B, R, T), rnd(B, R, T, sc=0.3)
dpre, dskip, dnet = rnd(B, 2 * R, T, sc=1e-5), rnd(B, S, T, sc=1e-5), rnd(B, R, T, sc=1e-5)
th, sg = torch.tanh(rnd(B, R, T)), torch.sigmoid(rnd(B, R, T))
gw, ow = rnd(ks, R, 2 * R, sc=0.06), rnd(R, S + R, sc=0.06)
xp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
gp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
dp = torch.empty(2 * B * 2 * R * T, dtype=torch.float16, device=dev)
gr = torch.empty(2 * B * (S + R) * T, dtype=torch.float16, device=dev)
wp = torch.empty(2 * ks * R * 2 * R, dtype=torch.float16, device=dev)
wres = torch.empty(2 * R * R, dtype=torch.float16, device=dev)
wdg = torch.empty(2 * ks * 2 * R * R, dtype=torch.float16, device=dev)
wgb = torch.empty(2 * (S + R) * R, dtype=torch.float16, device=dev)
K.f16x3_split_activations(net, xp, B, R, T)
K.f16x3_split_activations(gated, gp, B, R, T)
K.f16x3_split_activations(dpre, dp, B, 2 * R, T, scale=2.0 ** 20)
K.f16x3_split_activations(dskip, gr, B, S, T, scale=2.0 ** 20, kc0=0, KC=(S + R) // 8)
K.f16x3_split_activations(dnet, gr, B, R, T, scale=2.0 ** 20, kc0=S // 8, KC=(S + R) // 8)
K.f16x3_pack_gate_weights(gw, wp, ks, R, 2 * R, 256.0)
K.f16x3_pack_weights(ow.view(-1)[S:], wres, R, R, S + R, 256.0)
gwT = gw.permute(0, 2, 1).contiguous()
K.f16x3_pack_weights(gwT, wdg, ks * 2 * R, R, R, 256.0)
owT = ow.t().contiguous()
K.f16x3_pack_weights(owT, wgb, S + R, R, R, 256.0)
out, s0, s1 = torch.empty(B, R, T, device=dev), torch.empty(B, R, T, device=dev), torch.empty(B, R, T, device=dev)
net2, dpre2, dnet2 = torch.empty(B, R, T, device=dev), torch.empty(B, 2 * R, T, device=dev), torch.empty(B, R, T, device=dev)
slab = torch.empty(256 * 65536, device=dev)
dwg, dwo = torch.zeros(ks, R, 2 * R, device=dev), torch.zeros(R, S + R, device=dev)
N = B * T
for d in (1, 64):
    timeit('gate conv d=%d' % d, lambda: K.f16x3_gate_conv(xp=xp, wp=wp, out0=out, save0=s0, save1=s1, B=B, T=T, R=R, ks=ks, dilation=d,
                                                           w_scale_inv=1 / 256.0, out_planes=gp), 2.0 * N * ks * R * 2 * R)
timeit('gate conv d=64, tanh not saved', lambda: K.f16x3_gate_conv(xp=xp, wp=wp, out0=out, save1=s1, B=B, T=T, R=R, ks=ks, dilation=64,
                                                                    w_scale_inv=1 / 256.0, out_planes=gp), 2.0 * N * ks * R * 2 * R)
timeit('residual 1x1 (+planes)', lambda: K.f16x3_out_conv(xp=gp, wp=wres, net_in=net, net_out=net2, net_out_planes=xp, B=B, T=T, R=R, S=0,
                                                          w_scale_inv=1 / 256.0), 2.0 * N * R * R)
timeit('gate backward', lambda: K.f16x3_out_conv(epi=1, xp=gr, Cin=S + R, xp_KC=(S + R) // 8, wp=wgb, aux0=th, aux1=sg, net_out=dpre2,
                                                 net_out_planes=dp, plane_scale=2.0 ** 20, B=B, T=T, R=R, S=0, w_scale_inv=2.0 ** -28), 2.0 * N * (S + R) * R)
timeit('gate backward from gated / sigmoid', lambda: K.f16x3_out_conv(epi=1, xp=gr, Cin=S + R, xp_KC=(S + R) // 8, wp=wgb, aux0=out, aux1=sg, net_out=dpre2,
                                                                      net_out_planes=dp, plane_scale=2.0 ** 20, B=B, T=T, R=R, S=0, w_scale_inv=2.0 ** -28,
                                                                      aux0_is_gated=True), 2.0 * N * (S + R) * R)
timeit('input gradient d=64', lambda: K.f16x3_out_conv(xp=dp, Cin=2 * R, ks=ks, dilation=64, direction=-1, wp=wdg, net_in=dnet, net_out=dnet2, B=B,
                                                       T=T, R=R, S=0, w_scale_inv=2.0 ** -28, net_out_planes=gr, planes_kc0=S // 8,
                                                       planes_KC=(S + R) // 8, plane_scale=2.0 ** 20), 2.0 * N * ks * 2 * R * R)
for d in (1, 64):
    timeit('wgrad gate d=%d (+reduce)' % d, lambda: K.f16x3_wgrad(p=net, q0=dpre, dw=dwg, slab=slab, B=B, T=T, Cp=R, Q0=2 * R,
                                                                  taps=[-2 * d, -d, 0]), 2.0 * N * ks * R * 2 * R)
timeit('wgrad 1x1 (+reduce)', lambda: K.f16x3_wgrad(p=gated, q0=dskip, q1=dnet, Q1=R, dw=dwo, slab=slab, B=B, T=T, Cp=R, Q0=S, lddw=S + R,
                                                    taps=[0]), 2.0 * N * R * (S + R))
timeit('fp32 engine wgrad gate d=64', lambda: K.wgrad_gemm(p=net, q0=dpre, dw=dwg, B=B, T_q=T, T_p=T, Cp=R, Q0=2 * R, taps=[-128, -64, 0]),
       2.0 * N * ks * R * 2 * R)

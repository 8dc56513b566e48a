#!/usr/bin/env python3
"""Gate conv of the benchmark shape (B=8, T=6656, 256 -> 512, k=3): the fp32-MFMA engine against the experimental
fp16x3 kernel (+ its activation split pass).  usage: gate_f16x3_bench.py [dilation]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
dev = torch.device('cuda', 0)
B, T, R, ks = 8, 6656, 256, 3
d = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Tz = T // 64
g = torch.Generator().manual_seed(0)
x = torch.randn(B, R, T, generator=g).to(dev)
w = (torch.randn(ks, R, 2 * R, generator=g) * 0.05).to(dev)
b = (torch.randn(2 * R, generator=g) * 0.3).to(dev)
cond = (torch.randn(B, 2 * R, Tz, generator=g) * 0.3).to(dev)
out = torch.empty(B, R, T, device=dev); th = torch.empty_like(out); sg = torch.empty_like(out)
xp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
wp = torch.empty(2 * ks * R * 2 * R, dtype=torch.float16, device=dev)
taps = [-(ks - 1 - j) * d for j in range(ks)]


def fp32():
    K.conv_gemm(x0=x, w=w, bias=b, out0=out, save0=th, save1=sg, B=B, T_in=T, T_out=T, M=2 * R, C0=R, taps=taps,
                epilogue=K.EPI_GATE, cond=cond, cond_T=Tz)


def split():
    K.f16x3_split_activations(x, xp, B, R, T)


def pack():
    K.f16x3_pack_gate_weights(w, wp, ks, R, 2 * R, 256.0)


def gate():
    K.f16x3_gate_conv(xp=xp, wp=wp, out0=out, save0=th, save1=sg, bias=b, cond=cond, cond_T=Tz, B=B, T=T, R=R, ks=ks,
                      dilation=d, w_scale_inv=1.0 / 256.0)


def timeit(f, n=30):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


S = 512
wo = (torch.randn(R, S + R, generator=g) * 0.08).to(dev)
bo = (torch.randn(S + R, generator=g) * 0.3).to(dev)
skip = torch.zeros(B, S, T, device=dev); net2 = torch.empty(B, R, T, device=dev)
gp = torch.empty_like(xp); np_ = torch.empty_like(xp)
wop = torch.empty(2 * R * (S + R), dtype=torch.float16, device=dev)


def gate_planes():
    K.f16x3_gate_conv(xp=xp, wp=wp, out0=out, save0=th, save1=sg, bias=b, cond=cond, cond_T=Tz, B=B, T=T, R=R, ks=ks,
                      dilation=d, w_scale_inv=1.0 / 256.0, out_planes=gp)


def out_fp32():
    K.conv_gemm(x0=out, w=wo, bias=bo, out0=skip, out1=net2, aux1=x, B=B, T_in=T, T_out=T, M=S + R, M0=S, C0=R, taps=[0],
                epilogue=K.EPI_ACCUM_SPLIT, tile=12)


def out_x3():
    K.f16x3_out_conv(xp=gp, wp=wop, bias=bo, skip=skip, net_in=x, net_out=net2, net_out_planes=np_, B=B, T=T, R=R, S=S,
                     w_scale_inv=1.0 / 256.0)


split(); pack(); gate_planes()
K.f16x3_pack_weights(wo, wop, R, S + R, S + R, 256.0)
flop = 2.0 * B * T * 2 * R * ks * R
flop_out = 2.0 * B * T * R * (S + R)
for name, f in (('fp32-MFMA gate conv', fp32), ('f16x3 gate conv', gate), ('f16x3 gate conv + planes', gate_planes),
                ('activation split pass', split), ('weight pack', pack), ('fp32-MFMA out conv', out_fp32), ('f16x3 out conv', out_x3)):
    us = timeit(f)
    fl = flop if 'gate' in name else (flop_out if 'out' in name else 0)
    print('%-26s %8.1f us%s' % (name, us, '  %.1f TFLOP/s fp32-equivalent' % (fl / us * 1e-6) if fl else ''))

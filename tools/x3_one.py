#!/usr/bin/env python3
"""A few launches of ONE kernel of the fp16x3 engine at the benchmark shape, for rocprofv3 --pmc passes.
usage: x3_one.py <gate|gate_nosave|gate_r3|gate_bwd_r3|wgrad|wgrad_batch|sconv|sconv_dgrad|wgrad_s2> [dilation]
(gate_nosave: tanh not stored, as round 2 ran it; gate_r3 / gate_bwd_r3 / wgrad_batch: as the default engine runs them since round 3 --
no fp32 gated output; gate backward from the gated planes, dpre as planes only; the gate kernels' weight gradients of six layers in
one launch, both operands from planes)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
what = sys.argv[1]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, T, R, ks = 8, 6656, 256, 3
dev = 'cuda'
torch.manual_seed(0)
net, dpre = torch.randn(B, R, T, device=dev), torch.randn(B, 2 * R, T, device=dev) * 1e-5
gw = torch.randn(ks, R, 2 * R, device=dev) * 0.06
bias, cond = torch.randn(2 * R, device=dev), torch.randn(B, 2 * R, T // 64, device=dev)
xp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
gp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
wp = torch.empty(2 * ks * R * 2 * R, dtype=torch.float16, device=dev)
K.f16x3_split_activations(net, xp, B, R, T)
K.f16x3_pack_gate_weights(gw, wp, ks, R, 2 * R, 256.0)
out, s0, s1 = (torch.empty(B, R, T, device=dev) for _ in range(3))
slab, dw = torch.empty(256 * 65536, device=dev), torch.zeros(ks, R, 2 * R, device=dev)
if what in ('sconv', 'sconv_dgrad', 'wgrad_s2'):      # encoder layer 1: 768 -> 768, k=5, stride 2, T 3328 -> 1664
    F, To = 768, 1664
    x, dy = torch.randn(B, F, 2 * To, device=dev), torch.randn(B, F, To, device=dev) * 1e-5
    w = torch.randn(5, F, F, device=dev) * 0.02
    ep = torch.empty(2 * B * F * 2 * To, dtype=torch.float16, device=dev)
    ewp = torch.empty(2 * 5 * F * F, dtype=torch.float16, device=dev)
    o1, r1, dx = torch.empty(B, F, To, device=dev), torch.empty(B, F, To, device=dev), torch.empty(B, F, 2 * To, device=dev)
    bias3 = torch.randn(F, device=dev)
    dw5 = torch.zeros(5, F, F, device=dev)
    if what == 'sconv':
        K.f16x3_split_activations(x, ep, B, F, 2 * To, mode=K.X3_S2D)
    else:
        K.f16x3_split_activations(dy, ep, B, F, To, scale=2.0 ** 20, mode=0)
    K.f16x3_pack_weights(w, ewp, 5 * F, F, F, 16.0, mode=0)
    for _ in range(5):
        if what == 'sconv':
            K.f16x3_strided_conv(xp=ep, wp=ewp, out=o1, save_r=r1, B=B, T=To, Cin=F, M=F, ks=5, pad_left=1, bias=bias3, bn_scale=bias3,
                                 bn_shift=bias3, relu=True, w_scale_inv=1 / 16.0)
        elif what == 'sconv_dgrad':
            K.f16x3_strided_conv(xp=ep, wp=ewp, out=dx, B=B, T=To, Cin=F, M=F, ks=5, pad_left=1, dgrad=True, w_scale_inv=2.0 ** -24)
        else:
            K.f16x3_wgrad(p=x, q0=dy, dw=dw5, slab=slab, B=B, T=To, Cp=F, Q0=F, taps=[j - 1 for j in range(5)], p_stride=2, T_p=2 * To, mode=0)
    torch.cuda.synchronize()
    sys.exit(0)
if what in ('gate_r3', 'gate_bwd_r3', 'wgrad_batch'):
    S, L6 = 512, 6
    sc = torch.tensor([2.0 ** 10, 2.0 ** 27], device=dev)
    if what == 'gate_r3':
        for _ in range(5):
            K.f16x3_gate_conv(xp=xp, wp=wp, out0=None, save1=s1, bias=bias, cond=cond, cond_T=T // 64, B=B, T=T, R=R, ks=ks, dilation=d,
                              w_scale_inv=1 / 256.0, out_planes=gp, mode=K.X3_HALF_BLOCKS)
    elif what == 'gate_bwd_r3':
        gr = torch.empty(2 * B * (S + R) * T, dtype=torch.float16, device=dev)
        K.f16x3_split_activations(torch.randn(B, S + R, T, device=dev) * 1e-5, gr, B, S + R, T, scale=2.0 ** 20)
        ow = torch.randn(S + R, R, device=dev) * 0.06
        wgb = torch.empty(2 * (S + R) * R, dtype=torch.float16, device=dev)
        K.f16x3_pack_weights(ow, wgb, S + R, R, R, 256.0)
        K.f16x3_split_activations(torch.tanh(net) * 0.5, gp, B, R, T)
        sg = torch.sigmoid(torch.randn(B, R, T, device=dev))
        dpl = torch.empty(2 * B * 2 * R * T, dtype=torch.float16, device=dev)
        for _ in range(5):
            K.f16x3_out_conv(epi=1, xp=gr, Cin=S + R, wp=wgb, aux0_planes=gp, aux0_is_gated=True, aux1=sg, net_out=None, net_out_planes=dpl,
                             plane_scale=2.0 ** 20, B=B, T=T, R=R, S=0, w_scale_inv=2.0 ** -28, mode=K.X3_HALF_BLOCKS)
    else:
        nets = [torch.randn(B, R, T, device=dev) for _ in range(L6)]
        xps = [torch.empty(2 * B * R * T, dtype=torch.float16, device=dev) for _ in range(L6)]
        dps = [torch.empty(2 * B * 2 * R * T, dtype=torch.float16, device=dev) for _ in range(L6)]
        for i in range(L6):
            K.f16x3_split_activations(nets[i], xps[i], B, R, T, scale_dev=sc[0:1])
            K.f16x3_split_activations(dpre, dps[i], B, 2 * R, T, scale_dev=sc[1:2])
        dws = [torch.zeros(ks, R, 2 * R, device=dev) for _ in range(L6)]
        dils = [4, 8, 16, 32, 64, 128]
        for _ in range(5):
            K.f16x3_wgrad_batch([dict(p_planes=xps[i], q_planes=dps[i], dw=dws[i], taps=[-2 * dd, -dd, 0]) for i, dd in enumerate(dils)],
                                slab=slab, B=B, T=T, Cp=R, Q0=2 * R, p_scale=sc[0:1], q0_scale=sc[1:2])
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(5):
    if what in ('gate', 'gate_nosave'):
        K.f16x3_gate_conv(xp=xp, wp=wp, out0=out, save0=None if what == 'gate_nosave' else s0, save1=s1, bias=bias, cond=cond, cond_T=T // 64, B=B, T=T, R=R, ks=ks,
                          dilation=d, w_scale_inv=1 / 256.0, out_planes=gp)
    else:
        K.f16x3_wgrad(p=net, q0=dpre, dw=dw, slab=slab, B=B, T=T, Cp=R, Q0=2 * R, taps=[-2 * d, -d, 0])
torch.cuda.synchronize()

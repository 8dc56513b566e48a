#!/usr/bin/env python3
"""How much would the engine's kernels gain if the loop phase (MFMA) of one block overlapped the epilogue phase (HBM) of its CU
neighbour?  Co-resident blocks of ONE launch start together and stay in lockstep; blocks of two launches on two streams drift apart.
The probe runs 2n independent launches of one kernel on one stream and as n + n on two streams (separate outputs):
    python tools/overlap_probe.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
B, T, R, S, ks = 8, 6656, 256, 512, 3
dev = 'cuda'
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
HB = K.X3_HALF_BLOCKS
net = rnd(B, R, T)
xp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
K.f16x3_split_activations(net, xp, B, R, T)
wp = torch.empty(2 * ks * R * 2 * R, dtype=torch.float16, device=dev)
K.f16x3_pack_gate_weights(rnd(ks, R, 2 * R, sc=0.06), wp, ks, R, 2 * R, 256.0, mode=HB)
gr = torch.empty(2 * B * (S + R) * T, dtype=torch.float16, device=dev)
K.f16x3_split_activations(rnd(B, S + R, T, sc=1e-5), gr, B, S + R, T, scale=2.0 ** 20)
wgb = torch.empty(2 * (S + R) * R, dtype=torch.float16, device=dev)
K.f16x3_pack_weights(rnd(S + R, R, sc=0.06), wgb, S + R, R, R, 256.0)
dp = torch.empty(2 * B * 2 * R * T, dtype=torch.float16, device=dev)
K.f16x3_split_activations(rnd(B, 2 * R, T, sc=1e-5), dp, B, 2 * R, T, scale=2.0 ** 20)
wdg = torch.empty(2 * ks * 2 * R * R, dtype=torch.float16, device=dev)
K.f16x3_pack_weights(rnd(ks * 2 * R, R, sc=0.06), wdg, ks * 2 * R, R, R, 256.0)
sg = torch.sigmoid(rnd(B, R, T))
dnet = rnd(B, R, T, sc=1e-5)


def bufs():
    return dict(s1=torch.empty(B, R, T, device=dev), gp=torch.empty(2 * B * R * T, dtype=torch.float16, device=dev),
                dpl=torch.empty(2 * B * 2 * R * T, dtype=torch.float16, device=dev), dn=torch.empty(B, R, T, device=dev),
                grp=torch.empty(2 * B * (S + R) * T, dtype=torch.float16, device=dev))


def gate(b):
    K.f16x3_gate_conv(xp=xp, wp=wp, out0=None, save1=b['s1'], B=B, T=T, R=R, ks=ks, dilation=8, w_scale_inv=1 / 256.0, out_planes=b['gp'], mode=HB)


def gbwd(b):
    K.f16x3_out_conv(epi=1, xp=gr, Cin=S + R, wp=wgb, aux0_planes=b['gp'], aux0_is_gated=True, aux1=sg, net_out=None, net_out_planes=b['dpl'],
                     plane_scale=2.0 ** 20, B=B, T=T, R=R, S=0, w_scale_inv=2.0 ** -28, mode=HB)


def dgrad(b):
    K.f16x3_out_conv(xp=dp, Cin=2 * R, ks=ks, dilation=8, direction=-1, wp=wdg, net_in=dnet, net_out=b['dn'], B=B, T=T, R=R, S=0,
                     w_scale_inv=2.0 ** -28, net_out_planes=b['grp'], planes_kc0=S // 8, planes_KC=(S + R) // 8, plane_scale=2.0 ** 20)


A, Bb = bufs(), bufs()
gate(A); gate(Bb)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n = 10
for name, fn in (('gate conv', gate), ('gate backward', gbwd), ('input gradient', dgrad)):
    for _ in range(3):
        fn(A)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2 * n):
        fn(A)
    e1.record()
    torch.cuda.synchronize()
    one = e0.elapsed_time(e1) / (2 * n) * 1e3
    start, end1, end2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    s1.wait_event(start); s2.wait_event(start)
    with torch.cuda.stream(s1):
        for _ in range(n):
            fn(A)
        end1.record()
    with torch.cuda.stream(s2):
        for _ in range(n):
            fn(Bb)
        end2.record()
    torch.cuda.synchronize()
    two = max(start.elapsed_time(end1), start.elapsed_time(end2)) / (2 * n) * 1e3
    print('%-16s one stream %6.1f us per launch   two streams %6.1f us per launch   (%.0f %%)' % (name, one, two, 100 * two / one), flush=True)


# Mixed pairs: the backward main chain (gate backward + input gradient, n layers) on one stream with an independent MFMA-heavy
# 256-register kernel (the 128-row gate conv: a stand-in for a register-light weight-gradient kernel) on the other, against the
# same launches back to back on one stream.
def dgrad_half(b):
    K.f16x3_out_conv(xp=dp, Cin=2 * R, ks=ks, dilation=8, direction=-1, wp=wdg, net_in=dnet, net_out=b['dn'], B=B, T=T, R=R, S=0,
                     w_scale_inv=2.0 ** -28, net_out_planes=b['grp'], planes_kc0=S // 8, planes_KC=(S + R) // 8, plane_scale=2.0 ** 20, mode=HB)


def timed(fn_main, fn_side, two):
    torch.cuda.synchronize()
    start, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    start.record()
    if two:
        s1.wait_event(start); s2.wait_event(start)
        with torch.cuda.stream(s1):
            fn_main()
            e1.record()
        with torch.cuda.stream(s2):
            fn_side()
            e2.record()
        torch.cuda.synchronize()
        return max(start.elapsed_time(e1), start.elapsed_time(e2)) * 1e3
    fn_main(); fn_side()
    e1.record()
    torch.cuda.synchronize()
    return start.elapsed_time(e1) * 1e3


for name, dg in (('input gradient 256-row blocks', dgrad), ('input gradient 128-row blocks', dgrad_half)):
    def chain():
        for _ in range(n):
            gbwd(A); dg(A)
    for m in (n, 2 * n):
        def side():
            for _ in range(m):
                gate(Bb)
        timed(chain, side, False)
        seq = timed(chain, side, False)
        timed(chain, side, True)
        par = timed(chain, side, True)
        print('%d x (gate backward + %s) with %d gate convs beside them: one stream %7.1f us   two streams %7.1f us   (%.0f %%)' % (
            n, name, m, seq, par, 100 * par / seq), flush=True)

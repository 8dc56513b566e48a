#!/usr/bin/env python3
"""Would two half-batch chains on two streams beat one full-batch chain?  Forward decoder loop (gate conv -> residual 1x1,
30 layers) and backward main chain (gate backward -> input gradient) at B=8 on one stream vs 2 x B=4 on two streams."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
T, R, S, ks, L = 6656, 256, 512, 3, 30
dev = 'cuda'
g = torch.Generator().manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3


class Chain:
    def __init__(self, B):
        self.B = B
        self.net = [rnd(B, R, T) for _ in range(2)]
        self.gated, self.th, self.sg = torch.empty(B, R, T, device=dev), torch.tanh(rnd(B, R, T)), torch.sigmoid(rnd(B, R, T))
        self.xp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
        self.gp = torch.empty(2 * B * R * T, dtype=torch.float16, device=dev)
        self.dp = torch.empty(2 * B * 2 * R * T, dtype=torch.float16, device=dev)
        self.gr = torch.empty(2 * B * (S + R) * T, dtype=torch.float16, device=dev)
        self.dpre, self.dnet = torch.empty(B, 2 * R, T, device=dev), [rnd(B, R, T, sc=1e-5) for _ in range(2)]
        K.f16x3_split_activations(self.net[0], self.xp, B, R, T)
        K.f16x3_split_activations(self.th * self.sg, self.gp, B, R, T)
        K.f16x3_split_activations(rnd(B, S, T, sc=1e-5), self.gr, B, S, T, scale=2.0 ** 20, kc0=0, KC=(S + R) // 8)
        K.f16x3_split_activations(self.dnet[0], self.gr, B, R, T, scale=2.0 ** 20, kc0=S // 8, KC=(S + R) // 8)

    def forward(self, W):      # as model.py runs it since round 3: no fp32 gated output, tanh not stored, 128-row blocks
        B = self.B
        for l in range(L):
            K.f16x3_gate_conv(xp=self.xp, wp=W['wp'], out0=None, save1=self.sg, B=B, T=T, R=R, ks=ks,
                              dilation=dil[l], w_scale_inv=1 / 256.0, out_planes=self.gp, mode=HB)
            K.f16x3_out_conv(xp=self.gp, wp=W['wres'], net_in=self.net[l % 2], net_out=self.net[(l + 1) % 2], net_out_planes=self.xp,
                             B=B, T=T, R=R, S=0, w_scale_inv=1 / 256.0, mode=HB)

    def backward(self, W):     # gate backward from the gated planes, dpre as planes only (128-row blocks); input gradient 256-row blocks
        B = self.B
        for l in range(L):
            K.f16x3_out_conv(epi=1, xp=self.gr, Cin=S + R, xp_KC=(S + R) // 8, wp=W['wgb'], aux0_planes=self.gp, aux0_is_gated=True, aux1=self.sg,
                             net_out=None, net_out_planes=self.dp, plane_scale=2.0 ** 20, B=B, T=T, R=R, S=0, w_scale_inv=2.0 ** -28, mode=HB)
            K.f16x3_out_conv(xp=self.dp, Cin=2 * R, ks=ks, dilation=dil[l], direction=-1, wp=W['wdg'], net_in=self.dnet[l % 2],
                             net_out=self.dnet[(l + 1) % 2], B=B, T=T, R=R, S=0, w_scale_inv=2.0 ** -28, net_out_planes=self.gr,
                             planes_kc0=S // 8, planes_KC=(S + R) // 8, plane_scale=2.0 ** 20, mode=0)


HB = K.X3_HALF_BLOCKS
gw, ow = rnd(ks, R, 2 * R, sc=0.06), rnd(R, S + R, sc=0.06)
W = {'wp': torch.empty(2 * ks * R * 2 * R, dtype=torch.float16, device=dev), 'wres': torch.empty(2 * R * R, dtype=torch.float16, device=dev),
     'wdg': torch.empty(2 * ks * 2 * R * R, dtype=torch.float16, device=dev), 'wgb': torch.empty(2 * (S + R) * R, dtype=torch.float16, device=dev)}
K.f16x3_pack_gate_weights(gw, W['wp'], ks, R, 2 * R, 256.0, mode=HB)
K.f16x3_pack_weights(ow.view(-1)[S:], W['wres'], R, R, S + R, 256.0)
K.f16x3_pack_weights(gw.permute(0, 2, 1).contiguous(), W['wdg'], ks * 2 * R, R, R, 256.0)
K.f16x3_pack_weights(ow.t().contiguous(), W['wgb'], S + R, R, R, 256.0)
full, ha, hb = Chain(8), Chain(4), Chain(4)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def two(which):
    with torch.cuda.stream(s1):
        getattr(ha, which)(W)
    with torch.cuda.stream(s2):
        getattr(hb, which)(W)


for which in ('forward', 'backward'):
    a = timed(lambda: getattr(full, which)(W))
    b = timed(lambda: two(which))
    c = timed(lambda: (getattr(ha, which)(W), getattr(hb, which)(W)))
    print('%-8s one chain B=8: %.2f ms   two chains 2 x B=4 on two streams: %.2f ms   the two half chains one after the other: %.2f ms' % (which, a, b, c), flush=True)

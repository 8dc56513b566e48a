#!/usr/bin/env python3
"""The two Cin = 1 convs' weight gradients (preprocess conv: k = 32, stride 1, 256 channels; encoder layer 0: k = 5, stride 2, 768
channels) at the benchmark shape:  python tools/cin1_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('vq-vae-wavenet_amd')
K = pkg.kernels
dev = 'cuda'


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


B, T = 8, 6656
x = torch.randn(B, T, device=dev)
for name, F, k, stride, off, To in (('preprocess', 256, 32, 1, -31, T), ('encoder layer 0', 768, 5, 2, -1, T // 2)):
    dy = torch.randn(B, F, To, device=dev)
    dw = torch.zeros(k, F, device=dev)
    t = timeit(lambda: K.conv_cin1_wgrad(x, dy, dw, k=k, stride=stride, offset=off))
    print('%-16s weight gradient %6.1f us  (%.0f MB of dy: %.2f TB/s)' % (name, t, dy.numel() * 4 / 1e6, dy.numel() * 4 / t / 1e6))

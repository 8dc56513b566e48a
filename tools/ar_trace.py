#!/usr/bin/env python3
"""Where a generated sample's time goes inside the persistent AR kernel.
Build:  VQW_BUILD_SUFFIX=_trace VQW_LIB_NAME=libvqwave_trace.so VQW_EXTRA_FLAGS=-DVQW_AR_TRACE python vq-vae-wavenet_amd/build.py
Run:    VQW_LIB_NAME=libvqwave_trace.so VQW_AR_TRACE_PRINT=1 python tools/ar_trace.py [batch] [steps]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module('vq-vae-wavenet_amd')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m, w = bench.default_configs()
dev = torch.device('cuda', 0)
model = pkg.model.VQVAE(m, w, 109, device=dev, seed=0)
x, spk = bench.synthetic_batch(max(B, 1), 6656, 109, 1234, dev)
enc = model.encode(x[:B].contiguous(), spk[:B].contiguous())
g = pkg.generator.FastGenerator(model, batch=B)
g.generate(enc, 64)
torch.cuda.synchronize()
g.reset()
t = time.perf_counter()
g.generate(enc, n)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print('batch %d: %.1f us/step, %.0f samples/s' % (B, dt / n * 1e6, B * n / dt))
g.close()

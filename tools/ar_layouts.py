#!/usr/bin/env python3
"""Fast generation at the reference widths: us per step for a batch of utterances against how its rows are grouped into
persistent handles (rows per handle) and the handles' decomposition (channels per workgroup):
    python tools/ar_layouts.py [steps]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('vq-vae-wavenet_amd')
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m, w = bench.default_configs()
model = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
x, spk = bench.synthetic_batch(8, 6656, 109, 1234, 'cuda')
enc8 = model.encode(x, spk)
print('VQW_AR_PLACE =', os.environ.get('VQW_AR_PLACE', 'channel'))
for batch, rows, cpb in ((1, 1, 4), (2, 1, 4), (2, 1, 8), (4, 2, 4), (4, 1, 4), (4, 1, 8),
                         (8, 4, 4), (8, 2, 4), (8, 2, 8), (8, 1, 8), (8, None, None)):
    for k in ('VQW_AR_ROWS', 'VQW_AR_CPB'):
        os.environ.pop(k, None)
    if rows:
        os.environ['VQW_AR_ROWS'], os.environ['VQW_AR_CPB'] = str(rows), str(cpb)
    try:
        g = pkg.generator.FastGenerator(model, batch=batch)
        enc = enc8[:batch].contiguous()
        g.generate(enc, 64)
        torch.cuda.synchronize()
        g.reset()
        t0 = time.perf_counter()
        g.generate(enc, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nwg = pkg._lib.lib().vqw_ar_decode_workgroups(g._hs[0])
        print('batch %d  rows/handle %s  cpb %s: parts %s waves %s  %d workgroups/handle  %.1f us/step  %.0f samples/s'
              % (batch, rows, cpb, g._parts, g._waves, nwg, dt / steps * 1e6, batch * steps / dt), flush=True)
        g.close()
    except Exception as e:
        print('batch %d rows %s cpb %s: %s: %s' % (batch, rows, cpb, type(e).__name__, e), flush=True)

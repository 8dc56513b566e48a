"""Loss over the first N training steps of the default model on synthetic batches, default engine next to VQW_ENGINE=fp32
(same weights, same batches).  GPU box: python tools/loss_curve.py [steps]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(engine, steps):
    os.environ['VQW_ENGINE'] = engine
    pkg = importlib.import_module('vq-vae-wavenet_amd')
    m, w = bench.default_configs()
    model = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
    out = []
    for step in range(steps):
        x, spk = bench.synthetic_batch(8, 6656, 109, 1000 + step, 'cuda')
        ws = model.train_step(x, spk)
        if step % 10 == 0 or step == steps - 1:
            out.append((step, model.losses(ws)[0]))
    return out, getattr(model, 'x3_fallbacks', 0)


if __name__ == '__main__':
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    a, fa = run('f16x3', steps)
    b, _ = run('fp32', steps)
    print('step   f16x3 engine   fp32 engine   (steps repeated on fp32 by the range guard: %d)' % fa)
    for (s, la), (_, lb) in zip(a, b):
        print('%4d   %.6f      %.6f' % (s, la, lb))

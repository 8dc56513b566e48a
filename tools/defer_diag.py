"""Deferred range-flag mode against the immediate mode, next to the run-to-run distance of the immediate mode itself
(six steps, two of them flagged).  GPU box: python tools/defer_diag.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
os.environ['VQW_ENGINE'] = 'f16x3'
pkg = importlib.import_module('vq-vae-wavenet_amd')
from oracle import ref_model as M
m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
P = M.init_params(m, w, 109, seed=3, randomize_all=True)
batches = []
for i in range(6):
    x, spk, _ = M.synthetic_batch(1, 1024, 109, 4321 + i)
    batches.append((x[:, :, 0].contiguous().cuda(), spk.cuda()))
def run(defer, tamper=(1, 3)):
    model = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
    model.load_named(P)
    model.defer_guard = defer
    hist = []
    for i, (xd, sd) in enumerate(batches):
        if i in tamper:
            model.x3_scale[model.SL['X'] + 2] *= 2.0 ** 24
        model.train_step(xd, sd)
        if not defer:
            hist.append(model.x3_scale.clone())
    model.finish_steps()
    return model, hist
def l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())
a, ha = run(False)
b, hb = run(False)
c, _ = run(True)
for nm, (p, q) in (('imm vs imm', (a, b)), ('imm vs deferred', (a, c))):
    d = (p.x3_scale != q.x3_scale).nonzero().flatten().tolist()
    print(nm, 'scale diffs', d, [(p.x3_scale[i].item(), q.x3_scale[i].item()) for i in d], 'flat', l2(p.flat, q.flat), 'm', l2(p.adam_m, q.adam_m), 'fallbacks', p.x3_fallbacks, q.x3_fallbacks, p.x3_steps, q.x3_steps)
for i, (u, v) in enumerate(zip(ha, hb)):
    print('imm step', i, 'scale diffs', (u != v).nonzero().flatten().tolist())

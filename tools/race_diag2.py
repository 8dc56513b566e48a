import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_model as M  # noqa: E402


def l2err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


pkg = importlib.import_module('vq-vae-wavenet_amd')
m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
P = M.init_params(m, w, 109, seed=3, randomize_all=True)
x, spk, _ = M.synthetic_batch(1, 1024, 109, 1234)
xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
os.environ['VQW_ENGINE'] = 'fp32'
mdl = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
mdl.load_named(P)
st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
out, grads = M.train_step(x, spk, P, m, w, st, 0)


def report(tag):
    got = mdl.named_gradients()
    errs = sorted(((l2err(got[n], g), n) for n, g in grads.items()), reverse=True)
    print('%-28s worst %.2e (%s) median %.2e' % (tag, errs[0][0], errs[0][1], errs[len(errs) // 2][0]), flush=True)


ws = mdl.forward(xd, sd); mdl.backward(xd, sd, ws); report('fwd+bwd')
ws = mdl.forward(xd, sd); mdl.backward(xd, sd, ws); report('fwd+bwd again')
ws = mdl.forward(xd, sd, compute_grad_seed=False); report('after fwd(no seed)')
ws = mdl.forward(xd, sd); mdl.backward(xd, sd, ws); report('fwd+bwd after no-seed fwd')
mdl.global_step = 0
ws = mdl.train_step(xd, sd); report('train_step (Adam applied)')

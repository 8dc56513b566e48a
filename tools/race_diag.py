"""Is the fp32 engine's distance from the oracle a property of the arithmetic or of the two-stream backward?"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_model as M  # noqa: E402


def l2err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


def main():
    pkg = importlib.import_module('vq-vae-wavenet_amd')
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    for what in ('plain', 'gradients'):
        P = M.init_params(m, w, 109, seed=3, randomize_all=True)
        if what == 'gradients':
            P['decoder/postprocess2/kernel'] *= 8.0
            P['decoder/postprocess1/kernel'] *= 8.0
        x, spk, _ = M.synthetic_batch(1, 1024, 109, 1234)
        xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
        P0 = {k: v.clone() for k, v in P.items()}
        st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
        out, grads = M.train_step(x, spk, P, m, w, st, 0)
        for overlap in ('1', '0'):
            os.environ['VQW_OVERLAP'] = overlap
            os.environ['VQW_ENGINE'] = 'fp32'
            mdl = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
            mdl.load_named(P0)
            for rep in range(2):
                ws = mdl.forward(xd, sd)
                mdl.backward(xd, sd, ws)
                torch.cuda.synchronize()
                got = mdl.named_gradients()
                errs = sorted(((l2err(got[n], g), n) for n, g in grads.items()), reverse=True)
                print('%s overlap=%s rep %d: worst %.2e (%s)  median %.2e  best %.2e' % (
                    what, overlap, rep, errs[0][0], errs[0][1], errs[len(errs) // 2][0], errs[-1][0]))


if __name__ == '__main__':
    main()

"""Per-tensor distance from the oracle of the fp16x3 engine and of the fp32 engine on the same problem (GPU box)."""
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
    what = sys.argv[1] if len(sys.argv) > 1 else 'plain'
    pkg = importlib.import_module('vq-vae-wavenet_amd')
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=3, randomize_all=True)
    if what == 'gradients':
        P['decoder/postprocess2/kernel'] *= 8.0
        P['decoder/postprocess1/kernel'] *= 8.0
    if what == 'weights':
        P['decoder/cycle_2/layer_3/gated/kernel'] *= 1e4
    if what == 'activations':
        P['decoder/preprocess/kernel'] *= 3e5
        for name in P:
            if name.endswith('/gated/kernel'):
                P[name] *= 1e-5
    x, spk, _ = M.synthetic_batch(1, 1024, 109, 1234)
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    models = {}
    for eng in ('f16x3', 'fp32'):
        os.environ['VQW_ENGINE'] = eng
        mdl = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
        mdl.load_named(P)
        models[eng] = mdl
    st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
    for step in range(3):
        for mdl in models.values():          # every step starts from the oracle's parameters: the trajectories of two fp32
            mdl.load_named(P, also_ema=False)  # evaluations part after one Adam step (sign of gradients at rounding level)
        out, grads = M.train_step(x, spk, P, m, w, st, step)
        res = {}
        for eng, mdl in models.items():
            ws = mdl.train_step(xd, sd)
            res[eng] = (mdl.losses(ws)[0], mdl.named_gradients(), ws.get('x3_used'))
        rows = sorted(((l2err(res['f16x3'][1][n], g), l2err(res['fp32'][1][n], g), n) for n, g in grads.items()), reverse=True)
        print('step %d: x3_used=%s loss oracle %.7f  f16x3 %.7f  fp32 %.7f' % (step, res['f16x3'][2], out['loss'].item(), res['f16x3'][0], res['fp32'][0]))
        for e3, e32, n in rows[:6]:
            print('   %-55s f16x3 %.2e  fp32 %.2e' % (n, e3, e32))
        import statistics
        print('   median ratio f16x3/fp32: %.2f' % statistics.median(e3 / max(e32, 1e-12) for e3, e32, _ in rows))
        mx = models['f16x3']
        sc = mx.x3_scale.cpu()
        print('   scales: WG %g WO %g G %g  X %s  DP %s' % (sc[0], sc[1], sc[2], ' '.join('%g' % v for v in sc[3:3 + mx.L + 1:5]),
                                                        ' '.join('%g' % v for v in sc[3 + mx.L + 1::5])))
    print('fallbacks', models['f16x3'].x3_fallbacks, 'scales G %g' % float(models['f16x3'].x3_scale[2]))


if __name__ == '__main__':
    main()

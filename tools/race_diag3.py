import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_model as M  # noqa: E402

pkg = importlib.import_module('vq-vae-wavenet_amd')
m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
P = M.init_params(m, w, 109, seed=3, randomize_all=True)
x, spk, _ = M.synthetic_batch(1, 1024, 109, 1234)
xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
os.environ['VQW_ENGINE'] = 'fp32'
os.environ['VQW_OVERLAP'] = os.environ.get('VQW_OVERLAP', '1')
mdl = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
mdl.load_named(P)


def snap_fwd(ws):
    torch.cuda.synchronize()
    d = {k: ws[k].clone() for k in ('z_e', 'cond', 'condenc', 'skip', 'h1', 'logits', 'inputs')}
    for l in (0, 1, 15, 30):
        d['net%d' % l] = ws['net'][l].clone()
    for l in (0, 29):
        d['gated%d' % l] = ws['gated'][l].clone()
    d['loss_buf'] = mdl.loss_buf.clone()
    return d


def snap_bwd(ws):
    torch.cuda.synchronize()
    d = {'dskip': ws['skip'].clone(), 'dh1': ws['h1'].clone(), 'dce': ws['dcondenc'].clone(), 'dcond': ws['dcond'].clone(), 'dz': ws['dz'].clone()}
    for k, v in mdl.G.items():
        d['G:' + k] = v.clone()
    return d


snaps = []
for rep in range(3):
    ws = mdl.forward(xd, sd, compute_grad_seed=False)
    f = snap_fwd(ws)
    ws = mdl.forward(xd, sd)
    f['dlogits'] = ws['logits'].clone()
    mdl.backward(xd, sd, ws)
    f.update(snap_bwd(ws))
    snaps.append(f)
for a, b in ((0, 1), (1, 2)):
    print('--- call %d vs call %d' % (a, b))
    for k in snaps[0]:
        d = float((snaps[a][k] - snaps[b][k]).abs().max())
        s = float(snaps[b][k].abs().max())
        if d > 0:
            print('  %-22s max|diff| %.3e  (max %.3e, rel %.2e)' % (k, d, s, d / max(s, 1e-30)))

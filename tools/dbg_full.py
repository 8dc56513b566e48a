import importlib, sys, torch, numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module('vq-vae-wavenet_amd')
from oracle import ref_model as M
m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
S, B, T = 109, 1, 1024
P = M.init_params(m, w, S, seed=3, randomize_all=True)
x, spk, _ = M.synthetic_batch(B, T, S, 1234)
model = pkg.model.VQVAE(m, w, S, device='cuda', seed=0); model.load_named(P)
xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
for n, p in P.items(): p.requires_grad_(M.is_trainable(n))
col = {}
out = M.forward(x, spk, P, m, w, collect=col)
for k in ('skip_sum','net0','net_1','net_15','net_29'): col[k].retain_grad()
out['loss'].backward()
ws = model.forward(xd, sd, compute_grad_seed=False)
def rel(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
print('net0', rel(ws['net'][0].permute(0,2,1), col['net0']))
for l in (1, 10, 20, 30):
    print('net', l, rel(ws['net'][l].permute(0,2,1), col['net_%d' % l]))
print('skip_sum', rel(ws['skip'].permute(0,2,1), col['skip_sum']))
print('logits', rel(ws['logits'].permute(0,2,1).reshape(-1, 256), out['logits']))
ws = model.forward(xd, sd); model.backward(xd, sd, ws)
print('dskip', rel(ws['skip'].permute(0,2,1), col['skip_sum'].grad))
ds = ws['skip'].permute(0,2,1).cpu() ; dr = col['skip_sum'].grad
bad = ((ds-dr).abs() > 1e-3*dr.abs().max()).nonzero()
print('n bad', len(bad), 'of', ds.numel(), 'examples', bad[:10].tolist())
if len(bad):
    import collections
    print('bad t hist', collections.Counter((bad[:,1]//64).tolist()).most_common(8), 'bad ch hist', collections.Counter((bad[:,2]//64).tolist()).most_common(8))
    i=bad[0]; print('got', ds[i[0],i[1],i[2]].item(), 'want', dr[i[0],i[1],i[2]].item(), 'skipval', col['skip_sum'][i[0],i[1],i[2]].item())
print('dnet0', rel(ws['dnet'].permute(0,2,1), col['net0'].grad))
got = model.named_gradients()
errs = sorted(((rel(got[n], p.grad), n) for n, p in P.items() if p.grad is not None), reverse=True)
for e, n in errs[:25]: print('%.3e %s' % (e, n))
print('median', errs[len(errs)//2])

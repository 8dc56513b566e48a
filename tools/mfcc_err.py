import importlib, sys, torch
sys.path.insert(0, '.')
from oracle import ref_ops as R, ref_model as M
pkg = importlib.import_module('vq-vae-wavenet_amd')
x, _, _ = M.synthetic_batch(3, 6400 + 160 * 3 + 57, 10, 5)
xb = x[:, :, 0].contiguous().cuda()
frames = -(-xb.shape[1] // 160)
out = torch.empty(3, 16, frames, device='cuda')
pkg.kernels.mfcc(xb, pkg.encoders.mel_weight_matrix().cuda(), out)
want = R.mfcc(x[:, :, 0])
got = out[:, :13].permute(0, 2, 1).cpu()
print('mfcc max abs err %.3e, max |want| %.3e, rel %.3e' % (float((got - want).abs().max()), float(want.abs().max()), float((got - want).abs().max() / want.abs().max())))

#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/ (run from the repo root, in the build
container -- /root/reference is only read as DATA here: the result WAVs).

  mu_law_pcm_labels.npy   uint8[65536]: oracle labels of every int16 PCM code (pcm+0.5)/32767.5
  ref_wav_levels.npz      unique float32 sample values of the reference's five result WAVs
                          (results/VCTK/p225_001/110640_*.wav) -- data files shipped by the reference
  tiny_model.npz          tiny-config train step + VQ indices + 48 greedy AR steps of the oracle
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_model as M, ref_ops as R  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def tiny_cfg():
    w = dict(M.DEFAULT_WAVENET)
    w.update(dilation_rates=[1, 2, 4, 8, 1, 2, 4, 8], num_cycles=2, num_cycle_layers=4, dilation_filters=32,
             skip_filters=64, residual_filters=32, preprocess={"kernel_size": 32, "filters": 32})
    m = dict(M.DEFAULT_MODEL)
    m.update(k=32, latent_dim=16, speaker_embedding=16, encoder_filters=48)
    return m, w


def main():
    pcm = np.arange(-32768, 32768, dtype=np.int64)
    xs = ((pcm.astype(np.float32) + np.float32(0.5)) / np.float32(32767.5)).astype(np.float32)
    np.save(os.path.join(OUT, 'mu_law_pcm_labels.npy'), R.mu_law_encode_np(xs, to_int=True).astype(np.uint8))

    ref = '/root/reference/results/VCTK/p225_001'
    if os.path.isdir(ref):
        from scipy.io import wavfile
        lv = {}
        for f in sorted(os.listdir(ref)):
            sr, a = wavfile.read(os.path.join(ref, f))
            assert sr == 16000 and a.dtype == np.float32
            lv[f.replace('.wav', '')] = np.unique(a)
        np.savez(os.path.join(OUT, 'ref_wav_levels.npz'), **lv)

    m, w = tiny_cfg()
    S, B, T = 10, 2, 512
    P = M.init_params(m, w, S, seed=11, randomize_all=True)
    x, spk, _ = M.synthetic_batch(B, T, S, 1234)
    P0 = {k: v.clone() for k, v in P.items()}
    st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
    out, grads = M.train_step(x, spk, P, m, w, st, 0)
    fx = {'x': x.numpy(), 'spk': spk.numpy(), 'q': out['q'].numpy(), 'labels': out['labels'].numpy(),
          'loss': np.float32(out['loss'].item()), 'recon': np.float32(out['reconstruction_loss'].item()),
          'vq_loss': np.float32(out['vq_loss'].item()), 'commit': np.float32(out['commitment_loss'].item()),
          'z_e': out['z_e'].detach().numpy(), 'logits_head': out['logits'].detach().numpy()[:64]}
    for n in ('decoder/cycle_1/layer_1/gated/kernel', 'decoder/preprocess/kernel', 'encoder/conv1d_2/kernel',
              'embedding/embedding', 'speaker_embedding', 'encoder/batch_normalization_3/gamma'):
        fx['grad:' + n] = grads[n].numpy()
        fx['new:' + n] = P[n].detach().numpy()
    with torch.no_grad():
        enc = M.forward(x, spk, P0, m, w)['local_condition']
    idx, audio = M.generate(P0, w, enc, 48, 'greedy')
    fx['gen_idx'] = idx
    fx['gen_audio'] = audio
    np.savez_compressed(os.path.join(OUT, 'tiny_model.npz'), **fx)
    print('fixtures written to', OUT, {k: os.path.getsize(os.path.join(OUT, k)) for k in os.listdir(OUT)})


if __name__ == '__main__':
    main()

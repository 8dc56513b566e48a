#!/usr/bin/env python3
"""generate.py -- same command line as the reference's generate.py:14-32, running on MI355X.

    python3 generate.py -restore saved_model/weights-110640.pt -audio data/p225_001.wav \\
        -speakers p225 p226 p227 p228 -mode sample

Encodes the utterance once (encoder + VQ), then generates it back sample by sample with the
fast WaveNet generator conditioned on each requested speaker, using the EMA weights, and writes
`<dir>/<step>_<speaker>.wav` (float32, 16 kHz) plus `embedding_<step>.npy` /
`speaker_embedding_<step>.npy` like the reference (generate.py:94-117).  Under torchrun the
speakers are sharded over the GPUs (rows of the batch never interact).
"""
import importlib
import json
import os
import sys
from argparse import ArgumentParser

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    parser = ArgumentParser()
    parser.add_argument('-restore', dest='restore_path', help='path to weights')
    parser.add_argument('-audio', dest='audio_path', help='path to audio')
    parser.add_argument('-speakers', nargs='+', dest='speakers', help='speaker id')
    parser.add_argument('-mode', default='sample', dest='mode', help='decode mode, sample or greedy')
    parser.add_argument('-params', default='model_parameters.json', dest='parameter_path', metavar='str', help='path to parameters file')
    parser.add_argument('-seed', default=None, type=int, help='seed of the sampling uniforms (reference: unseeded)')
    args = parser.parse_args()
    if args.mode not in ('sample', 'greedy'):
        raise NotImplementedError('decode mode %s not implemented' % args.mode)

    pkg = importlib.import_module('vq-vae-wavenet_amd')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)   # no collective: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    gs = int(args.restore_path.split('-')[-1].split('.')[0])
    from scipy.io import wavfile
    sr, wav = wavfile.read(args.audio_path)
    if wav.ndim > 1:
        wav = wav[:, 0]
    wav = wav.astype(np.float32) / 32768.0 if wav.dtype == np.int16 else wav.astype(np.float32)
    if sr != 16000:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(sr, 16000)
        wav = resample_poly(wav, 16000 // g, sr // g).astype(np.float32)
    wav = wav[:len(wav) // 512 * 512]          # generate.py:39 (512 = largest dilation)
    length = len(wav)

    first = args.speakers[0]
    if first[0] == 'p':
        spk_file, num_speakers = 'data/vctk_speakers.txt', 109
    elif first[0].lower() == 's':
        spk_file, num_speakers = 'data/aishell_speakers.txt', 340
    else:
        spk_file, num_speakers = 'data/librispeech_speakers.txt', 251
    speaker_to_int = pkg.data.get_speaker_to_int(spk_file) if os.path.exists(spk_file) else {}
    # 'None' -> all-zero one-hot -> argmax 0 (generate.py:59-60, model.py:22)
    ids = []
    for sp in args.speakers:
        if sp.lower() != 'none' and sp not in speaker_to_int:
            raise ValueError('unknown speaker %r (not in %s)' % (sp, spk_file))
        i = 0 if sp.lower() == 'none' else speaker_to_int[sp]
        if not 0 <= i < num_speakers:
            raise ValueError('speaker %r maps to %d, outside the %d-row speaker table' % (sp, i, num_speakers))
        ids.append(i)

    parameters, wavenet_parameters = pkg.model.load_configs(args.parameter_path)
    model = pkg.model.VQVAE(parameters, wavenet_parameters, num_speakers, device=dev, seed=0)
    if args.restore_path.endswith(('.safetensors', '.npz')):      # TF variable names (checkpoint.py); EMA shadows -> live
        pkg.checkpoint.load(model, args.restore_path, ema_to_live=True)
    else:
        model.load_state_dict(torch.load(args.restore_path, map_location='cpu', weights_only=True))
        model.use_ema_weights()                # generate.py:88-90
    save_path = args.restore_path.split('/weights')[0]
    if rank == 0:
        if model.use_vq:                      # generate.py:96-101
            np.save(save_path + '/embedding_%d.npy' % gs, model.P['embedding'].cpu().numpy())
        if model.spk_table:
            np.save(save_path + '/speaker_embedding_%d.npy' % gs, model.P['speaker_embedding'].cpu().numpy())

    mine = list(range(rank, len(ids), world))  # shard speakers over GPUs: no collective needed
    if mine:
        B = len(mine)
        x = torch.from_numpy(wav).to(dev).unsqueeze(0).contiguous()       # ONE utterance: encoder + VQ run once,
        spk = torch.tensor([ids[i] for i in mine], dtype=torch.int64, device=dev)
        enc = model.encode(x, spk)             # only the speaker rows differ (model.encoding, generate.py:40,92)
        out = np.zeros([B, length], dtype=np.float32)
        uniforms = None
        if args.mode == 'sample':      # one row of uniforms per requested speaker, whatever the sharding
            g = torch.Generator().manual_seed(args.seed) if args.seed is not None else None
            uniforms = torch.rand(len(ids), length, generator=g)[mine].contiguous().to(dev)
        for b0 in range(0, B, 12):             # 12 rows = three 4-row handles in one persistent launch at R=256
            rows = slice(b0, min(b0 + 12, B))
            gen = pkg.generator.FastGenerator(model, batch=rows.stop - rows.start)
            audio, _ = gen.generate(enc[rows].contiguous(), length, mode=args.mode, ratio=length // enc.shape[2],
                                    uniforms=None if uniforms is None else uniforms[rows].contiguous())
            out[rows] = audio.cpu().numpy()
            gen.close()
        for j, i in enumerate(mine):
            s = 'no_speaker' if args.speakers[i] == 'None' else args.speakers[i]
            wavfile.write(save_path + '/%d_%s.wav' % (gs, s), 16000, out[j])
            print('wrote', save_path + '/%d_%s.wav' % (gs, s))


if __name__ == '__main__':
    main()

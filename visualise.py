#!/usr/bin/env python3
"""visualise.py -- same command line as the reference's visualise.py:6-19: turns the `embedding_<step>.npy` /
`speaker_embedding_<step>.npy` files that generate.py writes into `<name>_vecs.tsv` + `<name>_meta.tsv` for the
TensorFlow embedding projector (one tab-separated vector per line; metadata = the 1-based code index, or the speaker's
info line with its fields joined by '#', 'missing_info' where the info file has none).

    python3 visualise.py -embedding saved_model/embedding_110640.npy -speaker saved_model/speaker_embedding_110640.npy \
        -dataset VCTK -save projector
"""
import importlib
import os
import sys
from argparse import ArgumentParser

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INFO_FILES = {'VCTK': ('data/vctk_speakers.txt', 'data/vctk_speaker_info.txt'),
              'LibriSpeech': ('data/librispeech_speakers.txt', 'data/librispeech_speaker_info.txt')}


def speaker_info(speaker_to_int, info_path):
    """index -> info string.  The VCTK table lists bare numbers ('225  23  F ...') for speakers named 'p225'; tables
    with '|' separators (LibriSpeech) carry the ids as they are.  The first line is a header."""
    with open(info_path) as fh:
        rows = fh.read().splitlines()
    prefix = '' if rows and '|' in rows[0] else 'p'
    info = {}
    for row in rows[1:]:
        fields = row.split()
        if fields and prefix + fields[0] in speaker_to_int:
            info[speaker_to_int[prefix + fields[0]]] = '#'.join(fields[1:])
    return {i: info.get(i, 'missing_info') for i in speaker_to_int.values()}


def export(npy_path, out_dir, meta_of):
    vectors = np.load(npy_path)
    stem = os.path.basename(npy_path)
    stem = stem[:-4] if stem.endswith('.npy') else stem
    with open(os.path.join(out_dir, stem + '_vecs.tsv'), 'w', encoding='utf-8') as fv, \
            open(os.path.join(out_dir, stem + '_meta.tsv'), 'w', encoding='utf-8') as fm:
        for i, vec in enumerate(vectors):
            fv.write('\t'.join(str(x) for x in vec) + '\n')
            fm.write(meta_of(i) + '\n')
    return stem


def main():
    parser = ArgumentParser()
    parser.add_argument('-embedding', dest='embedding', help='embedding space')
    parser.add_argument('-speaker', dest='speaker', help='speaker embedding space')
    parser.add_argument('-dataset', default='VCTK', dest='dataset', help='VCTK or LibriSpeech')
    parser.add_argument('-save', dest='save', help='save to folder')
    args = parser.parse_args()
    os.makedirs(args.save, exist_ok=True)
    if args.embedding:
        export(args.embedding, args.save, lambda i: str(i + 1))
    if args.speaker:
        if args.dataset not in INFO_FILES:
            raise NotImplementedError('dataset %s not implemented' % args.dataset)
        pkg = importlib.import_module('vq-vae-wavenet_amd')
        speakers_path, info_path = INFO_FILES[args.dataset]
        info = speaker_info(pkg.data.get_speaker_to_int(speakers_path), info_path)
        export(args.speaker, args.save, lambda i: info.get(i, 'missing_info'))
    print('upload to http://projector.tensorflow.org')


if __name__ == '__main__':
    main()

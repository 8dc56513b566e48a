"""GPU test of the drop-in command lines: train.py (synthetic data, tiny config) -> checkpoint
-> generate.py (speaker conversion of a WAV with the EMA weights)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_generate_roundtrip(tmp_path):
    from scipy.io import wavfile
    w = {"verbose": False, "quantization_channels": 256, "num_cycles": 1, "num_cycle_layers": 4,
         "dilation_rates": [1, 2, 4, 8], "kernel_size": 3, "dilation_filters": 32, "skip_filters": 64,
         "residual_filters": 32, "preprocess": {"kernel_size": 32, "filters": 32}}
    m = {"encoder": "64", "use_vq": True, "speaker_embedding": 16, "k": 32, "latent_dim": 16, "beta": 0.25,
         "encoder_filters": 48, "wavenet_parameters": str(tmp_path / 'w.json'), "verbose": False,
         "learning_rate_schedule": {"0": 1e-3}}
    (tmp_path / 'w.json').write_text(json.dumps(w))
    (tmp_path / 'm.json').write_text(json.dumps(m))
    env = dict(os.environ, PYTHONPATH=ROOT)
    cwd = str(tmp_path)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '-dataset', 'synthetic', '-length', '512',
                          '-batch', '2', '-step', '4', '-interval', '2', '-save', 'saved_model/weights', '-params',
                          str(tmp_path / 'm.json')], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert '[step 4]' in out.stdout and 'recons' in out.stdout
    ckpt = tmp_path / 'saved_model' / 'weights-4.pt'
    assert ckpt.exists()
    # the reference's TensorBoard tags (model.py:28-69,95-104), one JSON line per logged step
    lines = [json.loads(ln) for ln in (tmp_path / 'saved_model' / 'summaries.jsonl').read_text().splitlines()]
    assert [ln['global_step'] for ln in lines] == [2, 4]
    for tag in ('z_e', 'z_e_u', 'z_e_v', 'embedding', 'embedding_u', 'speaker_embedding', 'e_k', 'q(z|x)'):
        assert len(lines[-1][tag]['counts']) == 30 and lines[-1][tag]['min'] <= lines[-1][tag]['mean'] <= lines[-1][tag]['max'], tag
    assert lines[-1]['reconstruction_loss'] > 0 and lines[-1]['vq_loss'] >= 0 and sum(lines[-1]['q(z|x)']['counts']) == 2 * 512 // 64
    sd = torch.load(str(ckpt), map_location='cpu', weights_only=True)
    assert int(sd['global_step']) == 4 and not torch.equal(sd['flat'], sd['ema'])
    # resume continues the step counter
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '-dataset', 'synthetic', '-length', '512',
                          '-batch', '2', '-step', '2', '-interval', '1', '-restore', str(ckpt), '-save',
                          'saved_model/weights', '-params', str(tmp_path / 'm.json')], cwd=cwd, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'last global step: 4' in out.stdout and (tmp_path / 'saved_model' / 'weights-6.pt').exists()
    # generate: 1024-sample utterance, two speakers ('None' -> index 0); the dataset (and so the
    # size of the speaker table) is chosen by the first letter of the first speaker id (generate.py:46-57)
    (tmp_path / 'data').mkdir()
    (tmp_path / 'data' / 'vctk_speakers.txt').write_text('p225, 3\np226, 5\n')
    t = np.arange(1100) / 16000.0
    wavfile.write(str(tmp_path / 'a.wav'), 16000, (np.sin(2 * np.pi * 220 * t) * 8000).astype(np.int16))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'generate.py'), '-restore', str(ckpt), '-audio',
                          str(tmp_path / 'a.wav'), '-speakers', 'p226', 'None', '-mode', 'greedy', '-params',
                          str(tmp_path / 'm.json')], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    sr, gen = wavfile.read(str(tmp_path / 'saved_model' / '4_no_speaker.wav'))
    assert sr == 16000 and gen.dtype == np.float32 and gen.shape == (1024,) and np.isfinite(gen).all()
    assert (tmp_path / 'saved_model' / '4_p226.wav').exists()
    # the same checkpoint under the reference's TF variable names (checkpoint.py): restoring it (EMA shadows -> live
    # variables, generate.py:88-90) generates the same samples; train.py resumes from it as well
    st_path = tmp_path / 'saved_model' / 'weights-4.safetensors'
    assert st_path.exists()
    os.rename(str(tmp_path / 'saved_model' / '4_p226.wav'), str(tmp_path / 'saved_model' / 'pt_p226.wav'))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'generate.py'), '-restore', str(st_path), '-audio',
                          str(tmp_path / 'a.wav'), '-speakers', 'p226', '-mode', 'greedy', '-params',
                          str(tmp_path / 'm.json')], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert np.array_equal(wavfile.read(str(tmp_path / 'saved_model' / '4_p226.wav'))[1],
                          wavfile.read(str(tmp_path / 'saved_model' / 'pt_p226.wav'))[1])
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '-dataset', 'synthetic', '-length', '512',
                          '-batch', '2', '-step', '1', '-interval', '1', '-restore', str(st_path), '-save',
                          'saved_model/weights', '-params', str(tmp_path / 'm.json')], cwd=cwd, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'last global step: 4' in out.stdout, out.stderr[-2000:]
    assert (tmp_path / 'saved_model' / 'embedding_4.npy').exists()
    assert np.load(str(tmp_path / 'saved_model' / 'speaker_embedding_4.npy')).shape == (109, 16)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'generate.py'), '-restore', str(ckpt), '-audio',
                          str(tmp_path / 'a.wav'), '-speakers', 'p225', '-mode', 'beam'], cwd=cwd, env=env,
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and 'NotImplementedError' in bad.stderr

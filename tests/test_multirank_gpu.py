"""N > 1 on the GPU box (ONE GPU there): fresh child ranks over gloo.

* two data-parallel ranks, half the batch each, through model.train_step with parallel.GradAllReduce: the summed
  gradient / world equals the single-rank full-batch gradient, both ranks end with bit-identical parameters, and those
  equal the full-batch step's (what RCCL all-reduce + 1/world does on a node).
* generate.py with WORLD_SIZE=2: speakers are sharded over the ranks with no collective (generate.py:40,103-113 rows
  never interact) and every rank's WAVs equal the single-process run's bit for bit.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def run_ranks(argv, world, extra_env=None, cwd=None):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=cwd, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    return outs


def test_two_ranks_average_to_full_batch_step(pkg, tmp_path):
    run_ranks([os.path.join(ROOT, 'tests', 'dp_worker.py'), str(tmp_path), 'gloo'], 2)
    r0 = torch.load(str(tmp_path / 'rank0.pt'), weights_only=True)
    r1 = torch.load(str(tmp_path / 'rank1.pt'), weights_only=True)
    assert torch.equal(r0['grad_sum'], r1['grad_sum']), 'ranks hold different reduced gradients'
    assert torch.equal(r0['flat'], r1['flat']) and torch.equal(r0['ema'], r1['ema']), 'ranks diverged after the step'
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_worker
    m, w, P, x, spk = dp_worker.shared_problem()
    full = pkg.model.VQVAE(m, w, 10, device='cuda', seed=0)
    full.load_named(P)
    full.train_step(x.cuda(), spk.cuda())
    g_full, g_avg = full.grad.cpu(), r0['grad_sum'] / 2
    assert float((g_avg - g_full).abs().max()) < 1e-4 * float(g_full.abs().max()), 'sum / world != full-batch gradient'
    # Adam normalises by sqrt(v): compare the stepped parameters where the gradient is not at rounding level
    big = g_full.abs() > 1e-3 * g_full.abs().max()
    assert float((r0['flat'] - full.flat.cpu())[big].abs().max()) < 1e-5
    assert not torch.equal(r0['grad_sum'], g_full)        # it really was a sum of two different shards


def test_range_flag_of_one_rank_makes_every_rank_repeat_the_step(pkg, tmp_path):
    """The guarded fp16x3 engine under data parallelism (model.train_step / _x3_overflowed): rank 1's residual stream leaves
    fp16's range, rank 0's (silence) does not; the flag is MAX-all-reduced, BOTH ranks repeat the step on the fp32 engine --
    exchanging their gradient buckets a second time -- and end with bit-identical parameters, equal to the full-batch step."""
    run_ranks([os.path.join(ROOT, 'tests', 'dp_worker.py'), str(tmp_path), 'gloo', 'guard'], 2)
    r0 = torch.load(str(tmp_path / 'rank0.pt'), weights_only=True)
    r1 = torch.load(str(tmp_path / 'rank1.pt'), weights_only=True)
    assert bool(r0['x3_guard']) and bool(r1['x3_guard'])
    assert int(r0['own_flag']) == 0 and int(r1['own_flag']) != 0, 'the set-up must overflow on rank 1 only'
    assert int(r0['fallbacks']) == 1 and int(r1['fallbacks']) == 1, 'both ranks must repeat the step'
    assert not bool(r0['x3_used']) and not bool(r1['x3_used'])          # the step that was kept ran on the fp32 engine
    assert torch.isfinite(r0['grad_sum']).all()
    assert torch.equal(r0['grad_sum'], r1['grad_sum']), 'ranks hold different reduced gradients'
    assert torch.equal(r0['flat'], r1['flat']) and torch.equal(r0['ema'], r1['ema']), 'ranks diverged after the step'
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_worker
    m, w, P, x, spk = dp_worker.guard_problem()
    full = pkg.model.VQVAE(m, w, 10, device='cuda', seed=0)
    full.load_named(P)
    full.train_step(x.cuda(), spk.cuda())
    assert full.x3_fallbacks == 1
    g_full, g_avg = full.grad.cpu().double(), r0['grad_sum'].double() / 2
    assert float((g_avg - g_full).norm()) < 1e-4 * float(g_full.norm()), 'sum / world != full-batch gradient'


def test_deferred_range_flag_under_data_parallelism(pkg, tmp_path):
    """The same problem with model.defer_guard (train.py / bench.py): the flag is MAX-all-reduced on the device, latched into
    the sticky guard in front of the optimiser, and read by the host afterwards -- both ranks find the step voided, repeat it on
    the fp32 engine and end with bit-identical parameters, equal to the immediate mode's."""
    run_ranks([os.path.join(ROOT, 'tests', 'dp_worker.py'), str(tmp_path), 'gloo', 'guard_deferred'], 2)
    r0 = torch.load(str(tmp_path / 'rank0.pt'), weights_only=True)
    r1 = torch.load(str(tmp_path / 'rank1.pt'), weights_only=True)
    assert int(r0['fallbacks']) == 1 and int(r1['fallbacks']) == 1, 'both ranks must repeat the step'
    assert not bool(r0['x3_used']) and not bool(r1['x3_used'])
    assert torch.isfinite(r0['grad_sum']).all() and torch.isfinite(r0['flat']).all()
    assert torch.equal(r0['grad_sum'], r1['grad_sum']), 'ranks hold different reduced gradients'
    assert torch.equal(r0['flat'], r1['flat']) and torch.equal(r0['ema'], r1['ema']), 'ranks diverged after the step'
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import dp_worker
    m, w, P, x, spk = dp_worker.guard_problem()
    full = pkg.model.VQVAE(m, w, 10, device='cuda', seed=0)
    full.load_named(P)
    full.train_step(x.cuda(), spk.cuda())
    g_full, g_avg = full.grad.cpu().double(), r0['grad_sum'].double() / 2
    assert float((g_avg - g_full).norm()) < 1e-4 * float(g_full.norm()), 'sum / world != full-batch gradient'
    big = g_full.abs() > 1e-3 * g_full.abs().max()
    assert float((r0['flat'] - full.flat.cpu())[big].abs().max()) < 1e-5


@pytest.mark.parametrize('mode', ['rccl1', 'rccl1_deferred'])
def test_one_rank_on_rccl_runs_the_bucketed_exchange(pkg, tmp_path, mode):
    """RCCL itself, on the one GPU this box has: a fresh child with WORLD_SIZE=1, backend nccl, GradAllReduce(force=True).
    The decoder bucket, the per-layer encoder buckets and the rest are all-reduced on the side stream by the real
    communicator and the flag MAX-all-reduce runs on it too (the step is flagged and repeated: every bucket goes twice).
    A 1-rank sum is the identity: every bucket comes back bit-identical, the buckets tile [0, n_flat) exactly once per
    pass, and the step equals the same step without a grad_sync (not bitwise: the fp32 engine's split-K atomics make two
    evaluations differ in the last bits).  rccl1_deferred: the same with model.defer_guard -- the flag's MAX all-reduce is enqueued
    on the device in front of the guarded optimiser and the host learns of the flagged step in finish_steps()."""
    run_ranks([os.path.join(ROOT, 'tests', 'dp_worker.py'), str(tmp_path), 'nccl', mode], 1)
    r = torch.load(str(tmp_path / 'rccl1.pt'), weights_only=True)
    assert bytes(r['backend'].tolist()).decode() == 'nccl'
    assert bool(r['identical']), 'a bucket changed under a 1-rank all-reduce'
    n = int(r['n_flat'])
    cover = torch.zeros(n, dtype=torch.int32)
    for lo, hi in r['buckets'].tolist():
        cover[lo:hi] += 1
    assert len(r['buckets']) >= 3 and bool((cover == 1).all()), 'buckets must tile the flat gradient exactly once'
    assert r['fallbacks'].tolist() == [1, 1] and int(r['max_calls']) == 1
    g, gp = r['grad'].double(), r['grad_plain'].double()
    assert float((g - gp).norm()) < 1e-5 * float(gp.norm())
    assert abs(float(r['loss'][0]) - float(r['loss'][1])) <= 1e-6 * abs(float(r['loss'][1]))
    assert float((r['flat'] - r['flat_plain']).abs().max()) < 1e-5


def _tiny_checkpoint(tmp_path):
    w = {"verbose": False, "quantization_channels": 256, "num_cycles": 1, "num_cycle_layers": 4,
         "dilation_rates": [1, 2, 4, 8], "kernel_size": 3, "dilation_filters": 32, "skip_filters": 64,
         "residual_filters": 32, "preprocess": {"kernel_size": 32, "filters": 32}}
    m = {"encoder": "64", "use_vq": True, "speaker_embedding": 16, "k": 32, "latent_dim": 16, "beta": 0.25,
         "encoder_filters": 48, "wavenet_parameters": str(tmp_path / 'w.json'), "verbose": False,
         "learning_rate_schedule": {"0": 1e-3}}
    (tmp_path / 'w.json').write_text(json.dumps(w))
    (tmp_path / 'm.json').write_text(json.dumps(m))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '-dataset', 'synthetic', '-length', '512',
                          '-batch', '2', '-step', '2', '-interval', '2', '-save', 'saved_model/weights', '-params',
                          str(tmp_path / 'm.json')], cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=ROOT),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return tmp_path / 'saved_model' / 'weights-2.pt'


def test_generate_shards_speakers_over_ranks(tmp_path):
    from scipy.io import wavfile
    ckpt = _tiny_checkpoint(tmp_path)
    (tmp_path / 'data').mkdir()
    (tmp_path / 'data' / 'vctk_speakers.txt').write_text('p225, 3\np226, 5\np227, 7\n')
    t = np.arange(1100) / 16000.0
    wavfile.write(str(tmp_path / 'a.wav'), 16000, (np.sin(2 * np.pi * 220 * t) * 8000).astype(np.int16))
    argv = [os.path.join(ROOT, 'generate.py'), '-restore', str(ckpt), '-audio', str(tmp_path / 'a.wav'), '-speakers',
            'p225', 'p226', 'p227', '-mode', 'sample', '-seed', '3', '-params', str(tmp_path / 'm.json')]
    run_ranks(argv, 1, cwd=str(tmp_path))
    names = ['2_p225.wav', '2_p226.wav', '2_p227.wav']
    single = [wavfile.read(str(tmp_path / 'saved_model' / n))[1] for n in names]
    for n in names:
        os.remove(str(tmp_path / 'saved_model' / n))
    outs = run_ranks(argv, 2, cwd=str(tmp_path))
    assert '2_p225.wav' in outs[0][0] and '2_p227.wav' in outs[0][0] and '2_p226.wav' in outs[1][0]   # rank 0: rows 0, 2
    for n, want in zip(names, single):
        got = wavfile.read(str(tmp_path / 'saved_model' / n))[1]
        assert np.array_equal(got, want), n

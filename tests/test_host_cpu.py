"""CPU tests of the host-side mirror: parameter inventory under the reference's variable names,
LR schedule, data pipeline, CLI surface and the data-parallel gradient exchange (gloo, world 2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ref_model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parameter_inventory_matches_reference_names_and_count(pkg):
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    model = pkg.model.VQVAE(m, w, 109, device='cpu', seed=0)
    ref = M.init_params(m, w, 109, seed=0)
    mine = model.named_parameters()
    assert list(sorted(mine)) == list(sorted(ref))
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
    trainable = sum(v.numel() for k, v in mine.items() if M.is_trainable(k))
    assert trainable == 35151104 == model.n_flat
    assert model.receptive_field == 6170                  # wavenet.py:16-17
    # initialiser ranges (model.py:26,49; wavenet_ops.py:69)
    assert abs(float(mine['embedding/embedding'].abs().max()) - 1.7 * (3 / 512) ** 0.5) < 2e-3
    assert float(mine['decoder/preprocess/kernel'].abs().max()) <= (3 / 32) ** 0.5 + 1e-6
    assert float(mine['decoder/cycle_1/layer_1/gated/bias'].abs().max()) == 0.0
    assert float(mine['encoder/batch_normalization/gamma'].min()) == 1.0


def test_load_named_roundtrip_and_ema(pkg):
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    m.update(k=32, latent_dim=16, speaker_embedding=16, encoder_filters=48)
    w.update(dilation_rates=[1, 2], num_cycles=1, num_cycle_layers=2, dilation_filters=32, skip_filters=64,
             residual_filters=32, preprocess={"kernel_size": 32, "filters": 32})
    P = M.init_params(m, w, 7, seed=3, randomize_all=True)
    model = pkg.model.VQVAE(m, w, 7, device='cpu', seed=0)
    model.load_named(P)
    back = model.named_parameters()
    for k, v in P.items():
        assert torch.equal(back[k], v), k
    ema = model.named_parameters(ema=True)
    assert torch.equal(ema['decoder/postprocess1/local_condition/kernel'], P['decoder/postprocess1/local_condition/kernel'])
    with pytest.raises(KeyError):
        model.load_named({'speaker_embedding': P['speaker_embedding']})
    assert model.lr_at(0) == 8e-5 and model.lr_at(80000) == 6e-5 and model.lr_at(10 ** 7) == 8e-6
    with pytest.raises(NotImplementedError):
        pkg.model.VQVAE(dict(m, encoder='1984'), w, 7, device='cpu')
    with pytest.raises(ValueError):
        model._workspace(2, 100)       # length must be a multiple of 64 (Encoder_64)


def test_checkpoint_under_reference_variable_names(pkg, tmp_path):
    """checkpoint.py: the whole state under the TF variable names of SURVEY Appendix B (+ /ExponentialMovingAverage shadows,
    /Adam slots, global_step) in safetensors and npz; round trip, generate.py:88-90's shadow restore, slot-scope prefixes."""
    from safetensors import safe_open
    C = pkg.checkpoint
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    m.update(k=32, latent_dim=16, speaker_embedding=16, encoder_filters=48)
    w.update(dilation_rates=[1, 2], num_cycles=1, num_cycle_layers=2, dilation_filters=32, skip_filters=64,
             residual_filters=32, preprocess={"kernel_size": 32, "filters": 32})
    a = pkg.model.VQVAE(m, w, 7, device='cpu', seed=1)
    g = torch.Generator().manual_seed(0)
    for t in (a.ema, a.adam_m, a.adam_v):
        t.copy_(torch.randn(a.n_flat, generator=g))
    a.bn_mean.copy_(torch.randn(a.bn_mean.numel(), generator=g))
    a.global_step = 1234
    ref_names = set(M.init_params(m, w, 7, seed=0))
    for ext in ('safetensors', 'npz'):
        path = str(tmp_path / ('weights-1234.' + ext))
        C.save(a, path)
        b = pkg.model.VQVAE(m, w, 7, device='cpu', seed=2)
        used = C.load(b, path)
        for attr in ('flat', 'ema', 'adam_m', 'adam_v', 'bn_mean', 'bn_var'):
            assert torch.equal(getattr(a, attr), getattr(b, attr)), (ext, attr)
        assert b.global_step == 1234 and 'global_step' in used
    with safe_open(str(tmp_path / 'weights-1234.safetensors'), 'pt') as f:
        keys = set(f.keys())
        assert f.get_tensor('decoder/cycle_1/layer_2/skip/kernel').shape == (1, 32, 64)
    trainable = {k for k in ref_names if M.is_trainable(k)}
    assert keys == ref_names | {k + s for k in trainable for s in (C.EMA, C.M1, C.M2)} | {'global_step'}
    c = pkg.model.VQVAE(m, w, 7, device='cpu', seed=3)
    C.load(c, str(tmp_path / 'weights-1234.npz'), ema_to_live=True)           # generate.py:88-90
    assert torch.equal(c.flat, a.ema)
    # a file whose slots carry a scope prefix (as a real TF checkpoint may) and has no Adam slots
    st = {('optimiser/' + k if k.endswith(C.EMA) else k): v.numpy() for k, v in C.named_state(a).items()
          if not k.endswith((C.M1, C.M2))}
    np.savez(str(tmp_path / 'tf.npz'), **st)
    d = pkg.model.VQVAE(m, w, 7, device='cpu', seed=4)
    C.load(d, str(tmp_path / 'tf.npz'))
    assert torch.equal(d.ema, a.ema) and torch.equal(d.flat, a.flat) and not torch.equal(d.adam_m, a.adam_m)
    del st['speaker_embedding']
    np.savez(str(tmp_path / 'bad.npz'), **st)
    with pytest.raises(KeyError):
        C.load(d, str(tmp_path / 'bad.npz'))


def _worker(rank, world, port, q):
    import importlib
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pkg = importlib.import_module('vq-vae-wavenet_amd')
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    sync = pkg.parallel.GradAllReduce(flat)
    sync.bucket_ready(600, 1000)       # "decoder" bucket first, as in backward
    sync.bucket_ready(0, 600)
    w = sync.finish()
    q.put((rank, w, flat.clone()))
    dist.destroy_process_group()


def test_grad_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.arange(1000, dtype=torch.float32) * 3          # (1 + 2) * base
    for rank, w, flat in res:
        assert w == 2
        assert torch.equal(flat, want)
        assert torch.equal(flat / w, torch.arange(1000, dtype=torch.float32) * 1.5)   # the averaged gradient


def test_wav_dataset_and_speaker_map(pkg, tmp_path):
    from scipy.io import wavfile
    root = tmp_path / 'data'
    (root / 'VCTK-Corpus' / 'wav48' / 'p225').mkdir(parents=True)
    (root / 'VCTK-Corpus' / 'wav48' / 'p226').mkdir(parents=True)
    rng = np.random.RandomState(0)
    pcm16 = (rng.randn(16000 * 2) * 3000).astype(np.int16)
    wavfile.write(str(root / 'VCTK-Corpus' / 'wav48' / 'p225' / 'p225_001.wav'), 16000, pcm16)
    wavfile.write(str(root / 'VCTK-Corpus' / 'wav48' / 'p226' / 'p226_001.wav'), 48000,
                  (rng.randn(48000 * 2) * 3000).astype(np.int16))
    wavfile.write(str(root / 'VCTK-Corpus' / 'wav48' / 'p226' / 'short.wav'), 16000, pcm16[:100])
    (root / 'vctk_train.txt').write_text('p225/p225_001.wav\np226/p226_001.wav\np226/short.wav\n')
    (root / 'vctk_speakers.txt').write_text('p225, 0\np226, 1\n')
    ds = pkg.data.VCTK(4, 6656, relative_path=str(root) + '/', device='cpu', seed=1)
    assert ds.num_speakers == 2 and ds.speaker_to_int == {'p225': 0, 'p226': 1}
    x, spk = ds.next()
    assert x.shape == (4, 6656) and x.dtype == torch.float32 and spk.dtype == torch.int64
    assert float(x.abs().max()) <= 1.0 and set(spk.tolist()) <= {0, 1}
    direct = ds._read('p225/p225_001.wav')
    np.testing.assert_array_equal(direct, ((pcm16.astype(np.float32) + 0.5) / 32767.5).astype(np.float32))
    assert len(ds._read('p226/p226_001.wav')) == 32000                  # 48 kHz -> 16 kHz
    syn = pkg.data.Synthetic(2, 512, num_speakers=10, seed=1234, device='cpu')
    xs, ss = syn.next()
    xo, so, _ = M.synthetic_batch(2, 512, 10, 1234)
    assert torch.equal(xs, xo[:, :, 0]) and torch.equal(ss, so)           # same segments as the oracle / bench
    # a speaker id outside the table is refused where the ids are still Python ints
    (root / 'vctk_speakers.txt').write_text('p225, 0\np226, 7\n')
    with pytest.raises(ValueError, match='speaker ids outside'):
        pkg.data.VCTK(4, 6656, relative_path=str(root) + '/', device='cpu', seed=1)
    (root / 'vctk_speakers.txt').write_text('p225, 0\np226, 1\n')
    # Prefetcher: a background thread prepares the next batches (48 kHz files are resampled there) while the consumer is
    # busy for a step's time; same batches in the same order as the dataset alone, and the step loop never waits
    import time
    want = pkg.data.VCTK(8, 6656, relative_path=str(root) + '/', device='cpu', seed=5)
    pf = pkg.data.Prefetcher(pkg.data.VCTK(8, 6656, relative_path=str(root) + '/', device='cpu', seed=5), depth=3, device='cpu')
    t0 = time.time()
    want.next()
    per_batch = time.time() - t0                      # host time to build one batch
    want = pkg.data.VCTK(8, 6656, relative_path=str(root) + '/', device='cpu', seed=5)
    time.sleep(3 * per_batch + 0.05)                  # let the ring fill, as the model build does in train.py
    step = max(0.059, 1.5 * per_batch)                # 59 ms per step (BENCH_r01), or slower if this host is
    for _ in range(6):
        xa, sa = pf.next()
        xb, sb = want.next()
        assert torch.equal(xa, xb) and torch.equal(sa, sb)
        time.sleep(step)
    assert pf.waits == 0 and pf.served == 6, 'the step loop waited %d times on the input pipeline' % pf.waits
    pf.close()


def test_visualise_exports_projector_tsv(tmp_path):
    """visualise.py (reference visualise.py:6-49): <name>_vecs.tsv + <name>_meta.tsv for the embedding projector."""
    (tmp_path / 'data').mkdir()
    (tmp_path / 'data' / 'vctk_speakers.txt').write_text('p225, 0\np226, 1\np300, 2\n')
    (tmp_path / 'data' / 'vctk_speaker_info.txt').write_text(
        'ID  AGE  GENDER  ACCENTS  REGION\n225  23  F    English    Southern  England\n226  22  M    English    Surrey\n')
    np.save(str(tmp_path / 'embedding_7.npy'), np.arange(6, dtype=np.float32).reshape(3, 2))
    np.save(str(tmp_path / 'speaker_embedding_7.npy'), np.ones((3, 2), np.float32))
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'visualise.py'), '-embedding', str(tmp_path / 'embedding_7.npy'),
                          '-speaker', str(tmp_path / 'speaker_embedding_7.npy'), '-save', str(tmp_path / 'proj')],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert (tmp_path / 'proj' / 'embedding_7_vecs.tsv').read_text() == '0.0\t1.0\n2.0\t3.0\n4.0\t5.0\n'
    assert (tmp_path / 'proj' / 'embedding_7_meta.tsv').read_text() == '1\n2\n3\n'
    assert (tmp_path / 'proj' / 'speaker_embedding_7_meta.tsv').read_text() == \
        '23#F#English#Southern#England\n22#M#English#Surrey\nmissing_info\n'


def test_cli_surface_matches_reference_flags():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'train.py'), '-h'], text=True)
    for flag in ('-dataset', '-length', '-step', '-batch', '-interval', '-restore', '-save', '-params'):
        assert flag in out
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'generate.py'), '-h'], text=True)
    for flag in ('-restore', '-audio', '-speakers', '-mode', '-params'):
        assert flag in out
    m = __import__('json').load(open(os.path.join(ROOT, 'model_parameters.json')))
    w = __import__('json').load(open(os.path.join(ROOT, 'wavenet_parameters.json')))
    assert m['encoder'] == '64' and m['k'] == 512 and m['beta'] == 0.25 and m['wavenet_parameters'] == 'wavenet_parameters.json'
    assert len(w['dilation_rates']) == w['num_cycles'] * w['num_cycle_layers'] == 30


def test_host_sampling_utils_match_oracle():
    """utils.py:13-46 mirror (vq-vae-wavenet_amd/utils.py) against the oracle's restatement: same cumsum /
    searchsorted semantics (index 256 possible when u > cdf[-1]), greedy = first maximum, decode levels equal."""
    import importlib
    from oracle import ref_ops as R
    U = importlib.import_module('vq-vae-wavenet_amd').utils
    rng = np.random.RandomState(0)
    pdf = rng.rand(5, 256).astype(np.float32)
    pdf /= pdf.sum(1, keepdims=True)
    u = rng.rand(5)
    u[0] = 1.0                                      # beyond cdf[-1] in fp32 -> index 256 (utils.py:24)
    pred, dec = R.sample_with_uniforms(pdf, u)
    np.testing.assert_array_equal(U.sample(pdf, uniforms=u), dec)
    np.testing.assert_array_equal(U.decode(pdf, mode='greedy'), R.mu_law_decode_np(np.argmax(pdf, -1).astype(np.float32)))
    np.testing.assert_array_equal(U.mu_law_decode_np(np.arange(257, dtype=np.float32)), R.mu_law_decode_np(np.arange(257, dtype=np.float32)))
    with pytest.raises(NotImplementedError):
        U.decode(pdf, mode='beam')


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher starts two fresh ranks under torch.distributed.run (probe mode:
    gloo group only, no GPU work anywhere); a WORLD_SIZE that contradicts --gpus is refused, never downgraded."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--probe-ranks'],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert rec == {"probe": "ranks", "n_gpus": 2, "ranks_seen": 2}
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--probe-ranks'],
                         env=dict(os.environ, WORLD_SIZE='2', RANK='0'), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and 'does not match' in bad.stderr


def test_relu_flip_detector():
    """tests/flips.py: what the model parity tests use to tell a relu mask flipped by summation-order noise from an error."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from flips import describe, relu_flips
    ref = torch.tensor([[5.0, -3.0, 7.45e-7, -2.0e-7], [1.0, 2.0, -1.0, 0.5]])
    same = ref.clone()
    assert relu_flips({'t': (same, ref)}) == [] and 'no relu mask' in describe([])
    noisy = ref.clone()
    noisy[0, 2] = -3.0e-7                      # within 2e-6 * 5.0 of zero on both sides: benign
    f = relu_flips({'skip_sum': (noisy, ref)})
    assert len(f) == 1 and f[0]['index'] == 2 and f[0]['benign'] and 'skip_sum[2]' in describe(f)
    wrong = ref.clone()
    wrong[1, 1] = -2.0                         # a real sign error
    f = relu_flips({'skip_sum': (wrong, ref)})
    assert len(f) == 1 and not f[0]['benign'] and 'NOT within noise' in describe(f)
    with pytest.raises(ValueError):
        relu_flips({'t': (ref[:1], ref)})


def test_generator_layout_policy(pkg, monkeypatch):
    """generator.pick_layout: how a batch of utterances is cut into persistent handles (rows never interact, generate.py:40,103-113).
    Up to 8 rows that fit the chip run as one-row handles with 8 channels per workgroup (R/8 workgroups each, ONE launch: measured
    118 us per step for 8 utterances against 129 as two 4-row handles, profiles/round3_ar_layouts.txt); larger batches keep handles of
    up to 4 rows; the environment overrides."""
    g = pkg.generator
    for k in ('VQW_AR_ROWS', 'VQW_AR_CPB'):
        monkeypatch.delenv(k, raising=False)
    assert g.pick_layout(1, 256, 256) == (1, 0)                      # one utterance: the library's choice (R/4 workgroups)
    assert g.pick_layout(2, 256, 256) == (1, 8) and g.pick_layout(8, 256, 256) == (1, 8)      # 8 x 32 workgroups = 256 CUs
    assert g.pick_layout(8, 256, 128) == (4, 0)                      # a chip with 128 CUs cannot hold 8 x 32 workgroups
    assert g.pick_layout(9, 256, 256) == (3, 0) and g.pick_layout(20, 256, 256) == (4, 0)
    assert g.pick_layout(6, 100, 256) == (3, 0)                      # R % 8 != 0: no 8-channel decomposition
    monkeypatch.setenv('VQW_AR_ROWS', '2')
    monkeypatch.setenv('VQW_AR_CPB', '4')
    assert g.pick_layout(8, 256, 256) == (2, 4)


def test_poisoned_allocations(pkg, monkeypatch):
    """_alloc: with VQW_POISON every float buffer starts as NaN and the recorded ones are refilled by repoison(); integer buffers
    (indices, labels: they are used as addresses) are left alone."""
    A = pkg._alloc
    monkeypatch.setattr(A, 'POISON', True)
    rec = []
    with A.record(rec):
        a = A.empty(3, 4)
        h = A.empty(8, dtype=torch.float16)
        i = A.empty(5, dtype=torch.int64)
    outside = A.empty(2)
    assert len(rec) == 3 and torch.isnan(a).all() and torch.isnan(h).all() and torch.isnan(outside).all()
    a.zero_(); h.zero_(); i.zero_()
    A.repoison(rec)
    assert torch.isnan(a).all() and torch.isnan(h).all() and int(i.abs().sum()) == 0
    monkeypatch.setattr(A, 'POISON', False)
    assert not torch.isnan(A.empty(4).fill_(1.0)).any()


def test_grad_allreduce_force_runs_the_collectives_in_a_world_of_one(pkg, tmp_path):
    """parallel.GradAllReduce(force=True): bucket slices, finish() and the flag's MAX all-reduce go through the process group even
    with one rank (how a 1-GPU box exercises the data-parallel path on RCCL, tests/test_multirank_gpu.py); without `force` a world of
    one short-circuits."""
    import torch.distributed as dist
    dist.init_process_group('gloo', init_method='file://%s' % (tmp_path / 'rdv'), rank=0, world_size=1)
    try:
        flat = torch.arange(10, dtype=torch.float32)
        calls = []
        real = dist.all_reduce

        def spy(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        dist.all_reduce = spy
        try:
            plain = pkg.parallel.GradAllReduce(flat)
            plain.bucket_ready(0, 4)
            assert plain.finish() == 1 and not plain.active and calls == []
            forced = pkg.parallel.GradAllReduce(flat, force=True)
            forced.bucket_ready(4, 10)
            forced.bucket_ready(0, 4)
            forced.bucket_ready(3, 3)                               # empty: ignored
            assert forced.finish() == 1 and forced.last_buckets == [(4, 10), (0, 4)] and calls == [6, 4]
            flag = torch.tensor([1], dtype=torch.int32)
            assert int(forced.all_reduce_max(flag)) == 1 and calls == [6, 4, 1]
        finally:
            dist.all_reduce = real
        assert torch.equal(flat, torch.arange(10, dtype=torch.float32))
    finally:
        dist.destroy_process_group()

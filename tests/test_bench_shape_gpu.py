"""GPU parity AT THE BENCHMARKED SHAPES (BASELINE.json configs[1] and configs[3]) against the CPU oracle.

* One full training step at the reference widths, B=8, T=6656, with exactly the weights (model seed 0) and the
  synthetic batch (seed 1234) `bench.py` times: the loss that bench prints, the VQ indices, the mu-law labels and
  every gradient are compared with `oracle.ref_model.train_step` (wavenet.py:24-100, model.py:90-130).  This is the
  shape with tail tiles, LDS-DMA interior blocks, the two-stream backward and the split-K encoder layers.
* The reference-width generator teacher-forced for 2 200 steps, so that the d=512 rings (1 025 slots) wrap twice
  (wavenet_ops.py:163-195), on the persistent kernel AND on the launch-per-phase path (`VQW_AR_PERSISTENT=0`).

Bars (unchanged from tests/test_model_gpu.py): VQ indices + labels bit-exact, losses rtol 2e-5, logits 5e-4 of the
tensor max, gradients 5e-3 in relative L2; every GPU decision = the oracle's argmax within 2e-6 in probability.
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import ref_model as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('vqw_bench', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def l2err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


def relerr(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


_ORACLE = {}


def _oracle_step(pkg):
    """The oracle's step on bench.py's problem, computed once for both engines (~25 s of host time)."""
    if not _ORACLE:
        bench = _bench()
        m, w = bench.default_configs()
        B, T, S = 8, 6656, 109
        model = pkg.model.VQVAE(m, w, S, device='cuda', seed=0)              # bench.py's weights
        x, spk = bench.synthetic_batch(B, T, S, 1234, 'cuda')                # bench.py's rank-0 batch
        P0 = {k: v.cpu() for k, v in model.named_parameters().items()}
        P = {k: v.clone() for k, v in P0.items()}
        del model
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
        st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
        out, grads = M.train_step(x.cpu().unsqueeze(-1), spk.cpu(), P, m, w, st, 0)
        keep = ('q', 'labels', 'z_e', 'logits', 'reconstruction_loss', 'vq_loss', 'loss')
        _ORACLE.update(m=m, w=w, x=x, spk=spk, P0=P0, P=P, grads=grads, out={k: out[k].detach() for k in keep})
    return _ORACLE


@pytest.mark.parametrize('engine', ['f16x3', 'f16x3_deferred', 'fp32'])
def test_bench_workload_full_step_matches_oracle(pkg, monkeypatch, engine):
    """engine: the default (fp16x3 with range guards: what bench.py times), the same with the range flag read one step late
    (model.defer_guard, as bench.py and train.py run it) and the fp32-MFMA engine (VQW_ENGINE=fp32)."""
    deferred = engine.endswith('_deferred')
    engine = engine.replace('_deferred', '')
    monkeypatch.setenv('VQW_ENGINE', engine)
    o = _oracle_step(pkg)
    m, w, x, spk, P, grads, out = o['m'], o['w'], o['x'], o['spk'], o['P'], o['grads'], o['out']
    model = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
    model.load_named(o['P0'])
    assert model.x3_guard == (engine == 'f16x3')
    ws = model.forward(x, spk, compute_grad_seed=False)
    assert torch.equal(ws['idx'].cpu(), out['q']), 'VQ indices differ'
    assert torch.equal(ws['labels'].cpu().reshape(-1), out['labels']), 'mu-law labels differ'
    assert relerr(ws['z_e'].permute(0, 2, 1), out['z_e']) < 2e-4
    logits = ws['logits'].permute(0, 2, 1).reshape(-1, model.Q)
    assert relerr(logits, out['logits']) < 5e-4
    del logits
    model.defer_guard = deferred
    ws = model.train_step(x, spk)
    if deferred:
        assert len(model._pending) == 1            # enqueued behind the device-side guard, not yet looked at by the host
        model.finish_steps()
        assert model.x3_steps == 1 and int(model.x3_void.item()) == 0
    assert bool(ws['x3_used']) == (engine == 'f16x3') and model.x3_fallbacks == 0
    loss, recon, vq, commit = model.losses(ws)
    np.testing.assert_allclose(recon, out['reconstruction_loss'].item(), rtol=2e-5)
    np.testing.assert_allclose(vq, out['vq_loss'].item(), rtol=2e-5)
    np.testing.assert_allclose(loss, out['loss'].item(), rtol=2e-5)
    got = model.named_gradients()
    worst = ('', 0.0)
    for name, gref in grads.items():
        e = l2err(got[name], gref)
        worst = max(worst, (name, e), key=lambda p: p[1])
        assert e < 5e-3, 'grad %s rel L2 err %.3e' % (name, e)
    # TF-Adam + EMA over the flat buffer at the full parameter count (model.py:116-128).  The first Adam step moves a
    # parameter by u = lr g / (|g| + eps): du/dg = lr eps / (|g| + eps)^2, so a relative gradient error d moves u by
    # less than lr d eps / |g|.  Elements with |g| > 1e-5 (eps = 1e-8) must therefore agree to 1e-2 lr (a few fp32 ulps of the
    # parameter itself: 1.8e-7 observed on values of 0.3); every element
    # stays within the step bound 2 lr of the oracle's (where g is at rounding level its sign is not determined).
    newp = model.named_parameters()
    lr = model.lr_at(0)
    for name, gref in grads.items():
        diff = (newp[name].cpu() - P[name]).abs()
        assert float(diff.max()) <= 2.001 * lr, 'param %s moved by more than the step bound' % name
        firm = gref.abs() > 1e-5
        if firm.any():
            assert float(diff[firm].max()) <= 1e-2 * lr, 'param %s after the step: %.3e' % (name, float(diff[firm].max()))
    print('bench-shape step (%s): loss %.6f (oracle %.6f), worst grad %s %.2e' % (engine, loss, out['loss'].item(), *worst))
    del model
    torch.cuda.empty_cache()


def test_batch_16_runs_on_the_fp16x3_engine(pkg, monkeypatch, capfd):
    """Round 2 turned the whole engine off for B*T > 69 905 (32-bit offsets over the all-layers gated planes) and the step ran
    silently on the fp32 engine.  The planes are now addressed through one buffer resource per plane, based at the
    contraction's first chunk, and a skip contraction longer than 2 GiB per plane is cut into layer groups:
    * reference widths, B = 16, T = 1024 with the skip contraction FORCED into 3 layer groups: one step against the oracle at
      the bars of the bench-shape test;
    * `-batch 16 -length 6656` (B*T = 106 496, planes of 3.3 GB): the step runs on the fp16x3 engine, no range fallback, and
      agrees with the same step on the fp32-MFMA engine (losses 2e-5, gradients 5e-3 relative L2);
    * a shape the engine cannot take says so on stderr, once."""
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P0 = M.init_params(m, w, 109, seed=4, randomize_all=True)
    x, spk, _ = M.synthetic_batch(16, 1024, 109, 99)
    P = {k: v.clone() for k, v in P0.items()}
    out, grads = M.train_step(x, spk, P, m, w, {'t': 0, 'm': {}, 'v': {}, 'ema': {}}, 0)
    monkeypatch.setenv('VQW_SKIP_GROUPS', '3')
    model = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
    model.load_named(P0)
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    ws = model.forward(xd, sd, compute_grad_seed=False)
    assert ws['skip_groups'] == 3 and ws['x3_used']
    assert torch.equal(ws['idx'].cpu(), out['q']) and torch.equal(ws['labels'].cpu().reshape(-1), out['labels'])
    assert relerr(ws['logits'].permute(0, 2, 1).reshape(-1, model.Q), out['logits']) < 5e-4
    ws = model.train_step(xd, sd)
    assert ws['x3_used'] and model.x3_fallbacks == 0
    np.testing.assert_allclose(model.losses(ws)[0], out['loss'].item(), rtol=2e-5)
    got = model.named_gradients()
    for name, gref in grads.items():
        assert l2err(got[name], gref) < 5e-3, 'grad %s rel L2 err %.3e' % (name, l2err(got[name], gref))
    monkeypatch.delenv('VQW_SKIP_GROUPS')
    del model, ws, got
    torch.cuda.empty_cache()
    # ---- batch 16 x 6656 on both engines
    bench = _bench()
    mb, wb = bench.default_configs()
    xb, sb = bench.synthetic_batch(16, 6656, 109, 4321, 'cuda')
    res = {}
    for engine in ('f16x3', 'fp32'):
        monkeypatch.setenv('VQW_ENGINE', engine)
        mdl = pkg.model.VQVAE(mb, wb, 109, device='cuda', seed=0)
        wsb = mdl.train_step(xb, sb)
        assert bool(wsb['x3_used']) == (engine == 'f16x3') and mdl.x3_fallbacks == 0
        if engine == 'f16x3':
            assert wsb['skip_groups'] == 1          # 1.6 GB per plane: one contraction
        res[engine] = (mdl.losses(wsb), {k: v.cpu() for k, v in mdl.named_gradients().items()}, wsb['idx'].cpu())
        del mdl, wsb
        torch.cuda.empty_cache()
    assert torch.equal(res['f16x3'][2], res['fp32'][2])
    np.testing.assert_allclose(res['f16x3'][0], res['fp32'][0], rtol=2e-5)
    for name, gref in res['fp32'][1].items():
        assert l2err(res['f16x3'][1][name], gref) < 5e-3, 'batch 16: grad %s differs between the engines by %.3e' % (name, l2err(res['f16x3'][1][name], gref))
    # ---- a shape the engine cannot take is reported, once
    monkeypatch.setenv('VQW_ENGINE', 'f16x3')
    mdl = pkg.model.VQVAE(mb, wb, 109, device='cuda', seed=0)
    capfd.readouterr()
    xs, ss = bench.synthetic_batch(1, 6656 - 128, 109, 1, 'cuda')       # 6528 = 64 * 102: not a multiple of 256
    for _ in range(2):
        wss = mdl.train_step(xs, ss)
    assert not wss['x3_used']
    err = capfd.readouterr().err
    assert err.count('runs on the fp32-MFMA engine') == 1 and 'not a multiple of 256' in err


@pytest.mark.parametrize('persistent', ['1', '0'])
def test_fast_generation_reference_width_rings_wrap(pkg, monkeypatch, persistent):
    monkeypatch.setenv('VQW_AR_PERSISTENT', persistent)
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=3, randomize_all=True)
    model = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
    model.load_named(P)
    n, ratio = 2200, 64
    Tz = -(-n // ratio)
    enc = torch.randn(1, model.Cc, Tz, generator=torch.Generator().manual_seed(5)) * 0.5     # [B][Cc][Tz]
    gen = pkg.generator.FastGenerator(model, batch=1)
    n1 = 1300
    u = torch.rand(1, n1, generator=torch.Generator().manual_seed(7))
    # first leg sampled with supplied uniforms (a varied history fills the rings), second leg greedy, continuing
    # the same run: the queue state carries over
    a1, i1 = gen.generate(enc.cuda(), n1, ratio=ratio, mode='sample', uniforms=u.cuda())
    a2, i2, probs = gen.generate(enc.cuda(), n - n1, ratio=ratio, return_probs=True)
    gen.close()
    got = torch.cat([i1, i2], 1).cpu().numpy()
    ga = torch.cat([a1, a2], 1).cpu().numpy()
    np.testing.assert_allclose(ga, M.R.mu_law_decode_np(got.astype(np.float32)), rtol=1e-5, atol=1e-6)
    assert len(np.unique(got[0, :n1])) > 16, 'degenerate run: the rings would hold one value'
    g = M.FastGenerator(P, w, 1)
    a = np.zeros([1, 1], np.float32)
    with torch.no_grad():
        for i in range(n):
            pr = g.step(torch.from_numpy(a), enc[:, :, i // ratio]).numpy()
            if i < n1:      # utils.py:13-27: searchsorted(cumsum(pdf), u), side='left'
                cdf = np.cumsum(pr[0])
                want = int(cdf.searchsorted(u[0, i].item()))
                if want != got[0, i]:   # only acceptable when u sits on a cdf edge (fp32 noise)
                    assert np.abs(cdf - u[0, i].item()).min() < 2e-6, 'step %d: %d vs oracle %d' % (i, got[0, i], want)
            else:
                assert pr[0].max() - pr[0, got[0, i]] <= 2e-6, 'step %d: GPU chose %d (p=%.8f), oracle argmax %d (p=%.8f)' % (
                    i, got[0, i], pr[0, got[0, i]], pr[0].argmax(), pr[0].max())
            a = ga[:, i:i + 1]
    np.testing.assert_allclose(probs.cpu().numpy(), pr, rtol=2e-4, atol=1e-7)

"""GPU parity of the assembled model (encoder + VQ + WaveNet decoder, forward + backward +
Adam/EMA) against the CPU oracle on identical inputs and weights.

Bars (the numbers the asserts below use): VQ indices and mu-law labels bit-exact; z_e 2e-4 and logits 5e-4 of the tensor
max; losses rtol 2e-5; gradients within 2e-3 of the per-tensor max on the tiny configuration, 5e-3 in relative L2 at
the reference widths (one relu mask flipped by summation-order noise moves every gradient by up to ~1e-2 at
B*T = 1024: DESIGN 6, tools/race_diag3.py); parameters / EMA after the Adam step 1e-4."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref_model as M

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from flips import describe, relu_flips  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def tiny_cfg():
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLD, 'make_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.tiny_cfg()


def build(pkg, m, w, S, P):
    model = pkg.model.VQVAE(m, w, S, device='cuda', seed=0)
    model.load_named(P)
    return model


def relerr(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


def l2err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


def run_parity(pkg, m, w, S, B, T, seed, steps=1, grad_tol=2e-3, err=relerr, check_params=True, resync=False, flip_tol=None):
    """flip_tol: relative-L2 bar that replaces `grad_tol` for a step in which a relu mask of the device differs from the
    oracle's at an input within summation-order noise of zero (tests/flips.py) -- and only for such a step.  Whatever the
    outcome, a failing gradient check names the flipped elements (or says that there are none)."""
    P = M.init_params(m, w, S, seed=seed, randomize_all=True)
    x, spk, _ = M.synthetic_batch(B, T, S, 1234)
    model = build(pkg, m, w, S, P)
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
    flipped_steps = []
    for step in range(steps):
        if resync and step:       # start every step from the oracle's parameters (the trajectories of two fp32 evaluations part)
            model.load_named(P, also_ema=False)
        col = {}
        out, grads = M.train_step(x, spk, P, m, w, st, step, collect=col)
        ws = model.forward(xd, sd, compute_grad_seed=False)
        logits = ws['logits'].permute(0, 2, 1).reshape(-1, model.Q)
        if out['q'] is not None:          # (use_vq = false: no indices)
            assert torch.equal(ws['idx'].cpu(), out['q']), 'VQ indices differ at step %d' % step
        assert torch.equal(ws['labels'].cpu().reshape(-1), out['labels']), 'mu-law labels differ'
        assert relerr(ws['z_e'].permute(0, 2, 1), out['z_e']) < 2e-4
        assert relerr(logits, out['logits']) < 5e-4
        snap = {}

        def relu_inputs(ws_):             # of the forward pass whose gradients are checked (backward overwrites them)
            snap['skip_sum'] = ws_['skip'].permute(0, 2, 1).clone()
            snap['post1_pre'] = ws_['h1'].permute(0, 2, 1).clone()
        model.train_step(xd, sd, on_forward=relu_inputs)
        pairs = {k: (snap[k], col[k]) for k in snap}
        if model.enc == '64':
            pairs.update({'enc_relu_%d' % i: (ws['r'][i].permute(0, 2, 1), col['enc_relu_%d' % i]) for i in range(6)})
        flips = relu_flips(pairs)
        loss, recon, vq, commit = model.losses(ws)
        np.testing.assert_allclose(recon, out['reconstruction_loss'].item(), rtol=2e-5)
        np.testing.assert_allclose(vq, out['vq_loss'].item() if 'vq_loss' in out else 0.0, rtol=2e-5)
        np.testing.assert_allclose(loss, out['loss'].item(), rtol=2e-5)
        got = model.named_gradients()
        worst = ('', 0.0)
        benign = bool(flips) and all(f['benign'] for f in flips)
        if benign and flip_tol is not None:
            flipped_steps.append(step)
            print('step %d: relu mask flipped inside summation-order noise: %s' % (step, describe(flips)))
        for name, gref in grads.items():
            if benign and flip_tol is not None:
                e, tol = l2err(got[name], gref), flip_tol
            else:
                e, tol = err(got[name], gref), grad_tol
            if e > worst[1]:
                worst = (name, e)
            assert e < tol, 'grad %s err %.3e (bar %.1e) at step %d; %s' % (name, e, tol, step, describe(flips))
        if not check_params:
            continue
        newp = model.named_parameters()
        for name, pref in P.items():
            assert err(newp[name], pref) < 1e-4, 'param %s after step %d' % (name, step)
        ema = model.named_parameters(ema=True)
        for name in grads:
            assert err(ema[name], st['ema'][name]) < 1e-4, 'ema %s' % name
    run_parity.flipped_steps = flipped_steps
    return worst


def test_tiny_model_two_steps(pkg):
    m, w = tiny_cfg()
    worst = run_parity(pkg, m, w, 10, 2, 512, seed=11, steps=2)
    print('worst grad', worst)


def test_tiny_model_matches_golden_fixture(pkg):
    m, w = tiny_cfg()
    fx = np.load(os.path.join(GOLD, 'tiny_model.npz'))
    P = M.init_params(m, w, 10, seed=11, randomize_all=True)
    model = build(pkg, m, w, 10, P)
    x = torch.from_numpy(fx['x'])[:, :, 0].contiguous().cuda()
    spk = torch.from_numpy(fx['spk']).cuda()
    ws = model.forward(x, spk, compute_grad_seed=False)
    assert relerr(ws['z_e'].permute(0, 2, 1), torch.from_numpy(fx['z_e'])) < 2e-4
    head = ws['logits'].permute(0, 2, 1).reshape(-1, model.Q)[:64]
    assert relerr(head, torch.from_numpy(fx['logits_head'])) < 5e-4
    ws = model.train_step(x, spk)
    assert np.array_equal(ws['idx'].cpu().numpy(), fx['q'])
    assert np.array_equal(ws['labels'].cpu().numpy().reshape(-1), fx['labels'])
    loss, recon, vq, commit = model.losses(ws)
    np.testing.assert_allclose(loss, fx['loss'], rtol=2e-5)
    np.testing.assert_allclose(commit, fx['commit'], rtol=2e-5)
    got = model.named_gradients()
    newp = model.named_parameters()
    for key in fx.files:
        if key.startswith('grad:'):
            assert relerr(got[key[5:]], torch.from_numpy(fx[key])) < 2e-3, key
        if key.startswith('new:'):
            assert relerr(newp[key[4:]], torch.from_numpy(fx[key])) < 1e-4, key


@pytest.mark.parametrize('persistent', ['1', '0'])
def test_generation_matches_golden_fixture(pkg, monkeypatch, persistent):
    """The committed 48 greedy steps of the oracle generator (tests/golden/tiny_model.npz: gen_idx, gen_audio;
    generate.py:103-113 with the pre-step weights): the device generator reproduces the indices (a differing index is
    accepted only where the oracle's decision is a near-tie, after which the comparison stops: the runs diverge)."""
    monkeypatch.setenv('VQW_AR_PERSISTENT', persistent)
    m, w = tiny_cfg()
    fx = np.load(os.path.join(GOLD, 'tiny_model.npz'))
    P = M.init_params(m, w, 10, seed=11, randomize_all=True)
    model = build(pkg, m, w, 10, P)
    x = torch.from_numpy(fx['x'])[:, :, 0].contiguous().cuda()
    enc = model.encode(x, torch.from_numpy(fx['spk']).cuda())
    n = fx['gen_idx'].shape[1]
    gen = pkg.generator.FastGenerator(model, batch=2)
    audio, idx = gen.generate(enc, n, ratio=n // enc.shape[2])           # generate.py:107: ratio = L // T_z
    gen.close()
    got, ga = idx.cpu().numpy(), audio.cpu().numpy()
    for b in range(2):
        same = got[b] == fx['gen_idx'][b]
        upto = n if same.all() else int(np.argmin(same))
        assert upto >= n - 8, 'row %d leaves the fixture at step %d' % (b, upto)   # tiny model: no near-ties seen
        np.testing.assert_allclose(ga[b, :upto], fx['gen_audio'][b, :upto], rtol=1e-5, atol=1e-6)


def test_default_width_short_segment(pkg):
    """Reference widths (768 / 256 / 512, K=512, 30 layers, dilations to 512) on B=1, T=1024.
    Gradients are compared in relative L2 norm: with ~1.3 M relu inputs a pre-activation
    within one fp32 ulp of zero can flip its mask between CPU and GPU (observed: skip value
    9e-7), which moves single gradient elements by their full value but not the norm."""
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    worst = run_parity(pkg, m, w, 109, 1, 1024, seed=3, steps=1, grad_tol=5e-3, err=l2err, check_params=False)   # Adam step: kernel test + tiny-model tests
    print('worst grad', worst)


def test_default_width_with_the_round3_paths_switched_off(pkg, monkeypatch):
    """The paths round 3 replaced stay in the code behind switches (and under shapes the new ones do not take): the condition
    projections on the conv engine (VQW_COND_PROJ=0), the short encoder layers without split-K (VQW_SCONV_SPLIT=0), the weight
    gradients one layer per launch from fp32 operands (VQW_WGRAD_BATCH=0).  Same problem and bars as the test above."""
    for k in ('VQW_COND_PROJ', 'VQW_SCONV_SPLIT', 'VQW_WGRAD_BATCH'):
        monkeypatch.setenv(k, '0')
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    worst = run_parity(pkg, m, w, 109, 1, 1024, seed=3, steps=1, grad_tol=5e-3, err=l2err, check_params=False)
    print('worst grad', worst)


@pytest.mark.parametrize('mode', ['1', '2', '3', '4', '5'])
def test_default_width_gate_f16x3(pkg, monkeypatch, mode):
    """The same parity run on the development ladder of the fp16x3 engine (VQW_GATE_F16X3=1..5: fixed scales, no guards):
    the oracle comparison holds at the SAME tolerances as the fp32-MFMA engine (VQ indices bit-exact, logits
    5e-4, losses 2e-5, gradients 5e-3 in relative L2)."""
    monkeypatch.setenv('VQW_GATE_F16X3', mode)   # 1: gate convs, 2: + 1x1 skip/residual convs, 3: skip path as one contraction, 4: + the gate convs' input gradient, 5: + gate backward
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    probe = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
    assert probe.gate_f16x3 and probe.out_f16x3 == (mode in '2345') and probe.skip_f16x3 == (mode in '345') and probe.dgrad_f16x3 == (mode in '45')
    del probe
    worst = run_parity(pkg, m, w, 109, 1, 1024, seed=3, steps=1, grad_tol=5e-3, err=l2err, check_params=False)
    print('worst grad', worst)


def _guarded_model(pkg, monkeypatch, P, m, w):
    monkeypatch.setenv('VQW_ENGINE', 'f16x3')
    monkeypatch.delenv('VQW_GATE_F16X3', raising=False)
    model = build(pkg, m, w, 109, P)
    assert model.x3_guard and model.gbwd_f16x3
    return model


def _check_step(model, out, grads, ws, grad_tol=5e-3):
    loss, recon, vq, commit = model.losses(ws)
    np.testing.assert_allclose(loss, out['loss'].item(), rtol=2e-5)
    np.testing.assert_allclose(recon, out['reconstruction_loss'].item(), rtol=2e-5)
    assert torch.equal(ws['idx'].cpu(), out['q'])
    got = model.named_gradients()
    for name, gref in grads.items():
        assert l2err(got[name], gref) < grad_tol, 'grad %s rel L2 err %.3e' % (name, l2err(got[name], gref))


@pytest.mark.parametrize('half', ['0', '1'], ids=['blocks256', 'blocks128'])
def test_guarded_f16x3_engine_three_steps(pkg, monkeypatch, half):
    """VQW_ENGINE=f16x3: the decoder's contractions as 3-term fp16-plane products with device-side range
    guards, against the oracle at the fp32 engine's tolerances for three consecutive steps: step 1 runs with the
    start-up scales, steps 2 and 3 with scales derived on the device from the max-abs values of the step before."""
    monkeypatch.setenv('VQW_X3_HALF', half)       # both block heights of the conv kernels, forward and backward
    for site in ('_SKIP', '_BWD', '_DGRAD'):
        monkeypatch.setenv('VQW_X3_HALF' + site, half)
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=3, randomize_all=True)
    x, spk, _ = M.synthetic_batch(1, 1024, 109, 1234)
    model = _guarded_model(pkg, monkeypatch, P, m, w)
    assert model.x3_mode_fwd == model.x3_mode_skip == model.x3_mode_bwd == model.x3_mode_dgrad == (2 if half == '1' else 0)
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
    for step in range(3):
        # every step starts from the oracle's parameters (two fp32 evaluations part ways after one Adam step: the sign of a
        # gradient at rounding level decides the direction of that parameter's first move); the guard state carries over
        model.load_named(P, also_ema=False)
        out, grads = M.train_step(x, spk, P, m, w, st, step)
        ws = model.train_step(xd, sd)
        assert ws['x3_used']
        _check_step(model, out, grads, ws)
    assert model.x3_steps == 3 and model.x3_fallbacks == 0
    sc = model.x3_scale.cpu()
    assert torch.equal(sc, torch.exp2(torch.round(torch.log2(sc)))), 'scales must be powers of two'
    amax_scaled = 0.06 * sc[model.SL['WG']]          # |gate kernels| <= sqrt(3 / 768) = 0.0625 (+ nothing: biases are separate)
    assert 2 ** 12 <= float(amax_scaled) < 2 ** 15


@pytest.mark.parametrize('what', ['activations', 'weights', 'gradients', 'nan'])
def test_guarded_f16x3_engine_adversarial_ranges(pkg, monkeypatch, what):
    """Operands the fixed scales of the development ladder could not hold: |net| > 65504 (preprocess kernel x 3e5),
    |w| > 255 (one gate kernel x 1e4), gradients above 2^-20 * 65504 (both head kernels x 8), a NaN weight.
    Weights always get an exact scale; activations / gradients trip the range flag on the first step, the step is repeated
    on the fp32 engine (results = the oracle's), and the following step runs on the fp16x3 engine again with measured
    scales -- still at the fp32 engine's tolerances."""
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=3, randomize_all=True)
    if what == 'activations':      # a residual stream of ~3e5 with gate kernels small enough to keep the gates unsaturated
        P['decoder/preprocess/kernel'] *= 3e5
        for name in P:
            if name.endswith('/gated/kernel'):
                P[name] *= 1e-5
    elif what == 'weights':
        P['decoder/cycle_2/layer_3/gated/kernel'] *= 1e4
    elif what == 'gradients':
        P['decoder/postprocess2/kernel'] *= 8.0
        P['decoder/postprocess1/kernel'] *= 8.0
    x, spk, _ = M.synthetic_batch(1, 1024, 109, 1234)
    model = _guarded_model(pkg, monkeypatch, P, m, w)
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    monkeypatch.setenv('VQW_ENGINE', 'fp32')
    plain = build(pkg, m, w, 109, P)                                     # the fp32-MFMA engine on the same problem
    assert not plain.x3_guard and not plain.gate_f16x3
    if what == 'nan':
        for mdl in (model, plain):
            mdl.P['out_w'][7, 3, 5] = float('nan')
        ws = model.train_step(xd, sd)
        assert model.x3_fallbacks == 1 and not ws['x3_used']          # seen by the guards, repeated on the fp32 engine:
        wp = plain.train_step(xd, sd)                                   # the result is whatever that engine makes of it
        a, b = model.losses(ws)[0], plain.losses(wp)[0]
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-5 * abs(b)
        return
    if what == 'activations':      # a bare forward pass with start-up scales would hand back garbage: forward_checked repeats it
        wc = model.forward_checked(xd, sd)
        wq = plain.forward(xd, sd, compute_grad_seed=False)
        assert not wc['x3_used'] and relerr(wc['logits'], wq['logits']) < 1e-5
    st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
    used = []
    for step in range(3):
        model.load_named(P, also_ema=False)          # as in test_guarded_f16x3_engine_three_steps
        plain.load_named(P, also_ema=False)
        out, grads = M.train_step(x, spk, P, m, w, st, step)
        snap, snap_p = {}, {}
        ws = model.train_step(xd, sd, on_forward=lambda w_: snap.update(skip=w_['skip'].clone(), h1=w_['h1'].clone()))
        used.append(bool(ws['x3_used']))
        loss = model.losses(ws)[0]
        # Against the fp32-MFMA engine on the same operands: a step the guard repeated ran the SAME kernels as
        # `plain`, so it must agree to rounding (1e-5 relative L2; not bitwise: split-K atomics); a step on the fp16x3 engine
        # must hold the engine's 5e-3 against it -- unless a relu mask differs between the two passes at an input within noise of
        # zero (shown in the message), where only the 1e-1 bar below applies.
        wp_ = plain.train_step(xd, sd, on_forward=lambda w_: snap_p.update(skip=w_['skip'].clone(), h1=w_['h1'].clone()))
        flips = relu_flips({k: (snap[k], snap_p[k]) for k in snap})
        gp_, gm_ = plain.named_gradients(), model.named_gradients()
        bar = 1e-1 if flips else (5e-3 if used[-1] else 1e-5)
        np.testing.assert_allclose(loss, plain.losses(wp_)[0], rtol=1e-6 if not used[-1] else 2e-5)
        for name in gp_:
            e = l2err(gm_[name], gp_[name])
            assert e < bar, 'step %d (%s) grad %s differs from the fp32 engine by %.3e (bar %.0e); %s' % (
                step, 'fp16x3' if used[-1] else 'fp32 repeat', name, e, bar, describe(flips))
        # Bars of this test: a plane that left fp16's range would show as errors of order 1 (inf / NaN / garbage).  The
        # extreme operands make single relu / saturation decisions flip between ANY two fp32 evaluations (tools/race_diag3.py:
        # one flipped mask element moves every gradient by ~1e-2 at B*T = 1024; 1.7e-2 observed), so the distances are bounded at 1e-1 here;
        # the fp32 engine's tolerances are held on ordinary operands by test_guarded_f16x3_engine_three_steps and
        # tests/test_bench_shape_gpu.py.
        np.testing.assert_allclose(loss, out['loss'].item(), rtol=5e-4)
        assert torch.equal(ws['idx'].cpu(), out['q'])
        got = model.named_gradients()
        for name, gref in grads.items():
            assert torch.isfinite(got[name]).all(), name
            assert l2err(got[name], gref) < 1e-1, 'step %d (%s) grad %s: %.3e' % (
                step, 'fp16x3' if used[-1] else 'fp32 repeat', name, l2err(got[name], gref))
    if what == 'weights':
        assert model.x3_fallbacks == 0 and used == [True, True, True]
    else:
        assert model.x3_fallbacks >= 1 and not used[0] and used[-1], (model.x3_fallbacks, used)


def test_deferred_guard_matches_immediate(pkg, monkeypatch):
    """model.defer_guard (bench.py / train.py): the range flag of step k is read after step k + 1 has been enqueued, the optimiser
    runs behind a device-side guard (vqw_adam_ema_step_guarded).  Six steps on different batches with two flagged steps -- the
    second and the fourth: a layer-input scale pushed 2^24 up behind the engine's back, so that a speculative step sits behind
    each flagged one; the last step is resolved by finish_steps -- against the same six steps in the immediate mode: the same
    steps are repeated on the fp32 engine, step counter / engine counters are equal, parameters, Adam slots, EMA shadows and
    plane scales agree as closely as two runs of the immediate mode do."""
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=3, randomize_all=True)
    batches = []
    for i in range(6):
        x, spk, _ = M.synthetic_batch(1, 1024, 109, 4321 + i)
        batches.append((x[:, :, 0].contiguous().cuda(), spk.cuda()))
    models = []
    for defer in (False, True):
        model = _guarded_model(pkg, monkeypatch, P, m, w)
        model.defer_guard = defer
        trace = []
        for i, (xd, sd) in enumerate(batches):
            if i in (1, 3):
                model.x3_scale[model.SL['X'] + 2] *= 2.0 ** 24
            model.train_step(xd, sd)
            trace.append((model.global_step, model.x3_fallbacks))
        if defer:
            assert model._pending, 'the last deferred step should still be unresolved here'
        model.finish_steps()
        assert not model._pending and int(model.x3_void.item()) == 0
        models.append((model, trace))
    (a, ta), (b, tb) = models
    assert a.x3_fallbacks == 2 and b.x3_fallbacks == 2, (a.x3_fallbacks, b.x3_fallbacks)
    assert a.global_step == b.global_step == 6 and a.x3_steps == b.x3_steps == 4
    assert ta[-1] == (6, 2) and tb != ta          # (the deferred run learns of a flagged step one call late)
    # Bars: two runs of the IMMEDIATE mode already differ from each other by 3e-4 (parameters) / 6e-2 (Adam's first moments) rel. L2
    # after these six steps and by a factor of two in one or two plane scales (tools/defer_diag.py: the fp32 engine's atomics and
    # the bias sums' are summation-order noise, and Adam turns noise on a near-zero gradient into a step of +-lr); the deferred
    # mode is held to that distance, a voided step that leaked into the state would show as errors of order one.
    ratio = a.x3_scale / b.x3_scale
    assert float(ratio.max()) <= 4.0 and float(ratio.min()) >= 0.25, 'plane scales: immediate %s deferred %s' % (a.x3_scale.tolist(), b.x3_scale.tolist())
    for name, bar in (('flat', 3e-3), ('ema', 3e-3), ('adam_v', 2e-1), ('adam_m', 3e-1)):
        ta_, tb_ = getattr(a, name), getattr(b, name)
        assert torch.isfinite(tb_).all(), name
        assert l2err(tb_, ta_) < bar, '%s: deferred vs immediate rel. L2 %.3e' % (name, l2err(tb_, ta_))
    # a caller's own forward pass (evaluation between training steps) settles the pending step first
    b.train_step(*batches[0])
    assert len(b._pending) == 1
    b.forward(*batches[1], compute_grad_seed=False)
    assert not b._pending and b.global_step == 7


@pytest.mark.parametrize('B,T', [(1, 1280), (2, 6400)])
def test_bf16_engine_encoder_2019(pkg, monkeypatch, B, T):
    """BASELINE.json configs[4]: the '2019' encoder (MFCC front end, T % 320 == 0) with the decoder's contractions on the
    bf16 engine (VQW_DTYPE=bf16: one bf16 plane per operand, v_mfma_f32_32x32x16_bf16, fp32 accumulate; master weights,
    residual stream, encoder, VQ and losses fp32) against the FP32 oracle.  bf16 carries 8 significand bits (unit
    roundoff 2^-9 = 2e-3 per operand) through 30 residual layers, hence the stated bf16 bars: VQ indices and mu-law labels
    still bit-exact (the encoder and the codebook search are fp32), logits 3e-2 of the tensor max, losses 5e-3,
    gradients 1.5e-1 in relative L2 (observed 9e-2 at B*T = 1280, where single relu decisions weigh most)."""
    monkeypatch.setenv('VQW_DTYPE', 'bf16')
    m, w = dict(M.DEFAULT_MODEL, encoder='2019'), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 251, seed=5, randomize_all=True)            # LibriSpeech: 251 speakers
    x, spk, _ = M.synthetic_batch(B, T, 251, 77)
    model = pkg.model.VQVAE(m, w, 251, device='cuda', seed=0)
    model.load_named(P)
    assert model.bf16 and model.x3_mode == 1 and not model.x3_guard
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
    out, grads = M.train_step(x, spk, P, m, w, st, 0)
    ws = model.forward(xd, sd, compute_grad_seed=False)
    assert ws['x3_used'] and ws['Tz'] == T // 320
    assert torch.equal(ws['idx'].cpu(), out['q']) and torch.equal(ws['labels'].cpu().reshape(-1), out['labels'])
    e_logits = relerr(ws['logits'].permute(0, 2, 1).reshape(-1, model.Q), out['logits'])
    assert e_logits < 3e-2, e_logits
    ws = model.train_step(xd, sd)
    loss, recon, vq, _ = model.losses(ws)
    np.testing.assert_allclose(loss, out['loss'].item(), rtol=5e-3)
    np.testing.assert_allclose(vq, out['vq_loss'].item(), rtol=2e-5)     # fp32 path
    got = model.named_gradients()
    worst = max(((l2err(got[n], g), n) for n, g in grads.items()))
    print('bf16 engine: logits %.2e, loss %.6f vs %.6f, worst grad %.2e (%s)' % (e_logits, loss, out['loss'].item(), *worst))
    assert worst[0] < 1.5e-1, worst
    assert model.x3_steps == 1


def test_data_parallel_shards_sum_to_full_batch(pkg):
    """Two 'ranks' with half the batch each: mean of their flat gradients == full-batch
    gradient (what the RCCL all-reduce + 1/world scaling computes)."""
    m, w = tiny_cfg()
    P = M.init_params(m, w, 10, seed=5, randomize_all=True)
    x, spk, _ = M.synthetic_batch(4, 256, 10, 9)
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    full = build(pkg, m, w, 10, P)
    ws = full.forward(xd, sd); full.backward(xd, sd, ws)
    parts = []
    for sl in (slice(0, 2), slice(2, 4)):
        mdl = build(pkg, m, w, 10, P)
        ws = mdl.forward(xd[sl].contiguous(), sd[sl].contiguous()); mdl.backward(xd[sl].contiguous(), sd[sl].contiguous(), ws)
        parts.append(mdl.grad.clone())
    avg = (parts[0] + parts[1]) / 2
    assert relerr(avg, full.grad) < 2e-3


@pytest.mark.parametrize('persistent', ['1', '0'])
def test_fast_generation_matches_oracle(pkg, monkeypatch, persistent):
    """Both generator back ends (persistent kernel; VQW_AR_PERSISTENT=0: the launch-per-phase path of ar_decode.hip).
    vqw_ar_decode (ring buffers + on-device decode) vs the oracle's FIFO-queue generator
    (wavenet_ops.py:147-267, generate.py:103-113): greedy indices and sampling with supplied
    uniforms; a mismatch is only accepted where the oracle's own decision is a near-tie."""
    monkeypatch.setenv('VQW_AR_PERSISTENT', persistent)
    m, w = tiny_cfg()
    P = M.init_params(m, w, 10, seed=11, randomize_all=True)
    model = build(pkg, m, w, 10, P)
    x, spk, _ = M.synthetic_batch(2, 512, 10, 1234)
    xd, sd = x[:, :, 0].contiguous().cuda(), spk.cuda()
    with torch.no_grad():
        enc_ref = M.forward(x, spk, P, m, w)['local_condition']          # [B,Tz,Cc]
    enc = model.encode(xd, sd)
    assert relerr(enc.permute(0, 2, 1), enc_ref) < 1e-5
    n = 160                                                                # crosses two condition frames
    gen = pkg.generator.FastGenerator(model, batch=2)
    audio, idx, probs = gen.generate(enc, n, return_probs=True)
    got, ga = idx.cpu().numpy(), audio.cpu().numpy()
    np.testing.assert_allclose(ga, M.R.mu_law_decode_np(got.astype(np.float32)), rtol=1e-5, atol=1e-6)
    # teacher-force the ORACLE with the GPU's own samples: every GPU decision must be the
    # oracle's argmax (or tie with it within fp32 noise), and the last probabilities must agree
    g = M.FastGenerator(P, w, 2)
    a = np.zeros([2, 1], np.float32)
    with torch.no_grad():
        for i in range(n):
            pr = g.step(torch.from_numpy(a), enc_ref[:, i // 64]).numpy()
            for b in range(2):
                assert pr[b].max() - pr[b, got[b, i]] <= 2e-6, \
                    'step %d row %d: GPU chose %d (p=%.8f), oracle argmax %d (p=%.8f)' % (
                        i, b, got[b, i], pr[b, got[b, i]], pr[b].argmax(), pr[b].max())
            a = ga[:, i:i + 1]
    np.testing.assert_allclose(probs.cpu().numpy(), pr, rtol=2e-4, atol=1e-7)
    # continuing a run == one long run (ring-buffer state carries over)
    gen.reset()
    a1, i1 = gen.generate(enc, 100)
    a2, i2 = gen.generate(enc, 60)
    assert torch.equal(torch.cat([i1, i2], 1), idx)
    # sampling with supplied uniforms (utils.py:13-27)
    u = torch.rand(2, n, generator=torch.Generator().manual_seed(0))
    gen.reset()
    sa, si = gen.generate(enc, n, mode='sample', uniforms=u.cuda())
    si, sa = si.cpu().numpy(), sa.cpu().numpy()
    g = M.FastGenerator(P, w, 2)
    a = np.zeros([2, 1], np.float32)
    with torch.no_grad():
        for i in range(n):
            pr = g.step(torch.from_numpy(a), enc_ref[:, i // 64]).numpy()
            cdf = np.cumsum(pr, axis=1)
            for b in range(2):
                want = int(cdf[b].searchsorted(u[b, i].item()))
                if want != si[b, i]:   # only acceptable when u sits on a cdf edge (fp32 noise)
                    assert np.abs(cdf[b] - u[b, i].item()).min() < 2e-6, 'step %d row %d: %d vs %d' % (i, b, si[b, i], want)
            a = sa[:, i:i + 1]
    gen.close()
    with pytest.raises(NotImplementedError):
        pkg.generator.FastGenerator(model, batch=1).generate(enc[:1].contiguous(), 4, mode='beam')


def test_fast_generation_reference_width(pkg):
    """The persistent generator at the reference widths (R=256, S=512, 30 layers, dilations up to 512; the
    kernel that bench.py times), teacher-forced against the oracle's FIFO-queue generator: every GPU decision is
    the oracle's argmax within 2e-6 in probability, the final probabilities agree to rtol 2e-4.  The condition
    frame changes inside the run (ratio 24) and the run is continued once (queue state carries over)."""
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=3, randomize_all=True)
    model = build(pkg, m, w, 109, P)
    n, ratio = 56, 24
    enc = torch.randn(1, model.Cc, 3, generator=torch.Generator().manual_seed(5)) * 0.5     # [B][Cc][Tz]
    gen = pkg.generator.FastGenerator(model, batch=1)
    a1, i1 = gen.generate(enc.cuda(), 30, ratio=ratio)
    a2, i2, probs = gen.generate(enc.cuda(), n - 30, ratio=ratio, return_probs=True)
    gen.close()
    got = torch.cat([i1, i2], 1).cpu().numpy()
    ga = torch.cat([a1, a2], 1).cpu().numpy()
    np.testing.assert_allclose(ga, M.R.mu_law_decode_np(got.astype(np.float32)), rtol=1e-5, atol=1e-6)
    g = M.FastGenerator(P, w, 1)
    a = np.zeros([1, 1], np.float32)
    with torch.no_grad():
        for i in range(n):
            pr = g.step(torch.from_numpy(a), enc[:, :, min(i // ratio, 2)]).numpy()
            assert pr[0].max() - pr[0, got[0, i]] <= 2e-6, 'step %d: GPU chose %d (p=%.8f), oracle argmax %d (p=%.8f)' % (
                i, got[0, i], pr[0, got[0, i]], pr[0].argmax(), pr[0].max())
            a = ga[:, i:i + 1]
    np.testing.assert_allclose(probs.cpu().numpy(), pr, rtol=2e-4, atol=1e-7)


def test_fast_generation_batch_split(pkg):
    """Batches above 4 rows run as several persistent handles side by side (rows never interact,
    generate.py:40): every row of a 6-row run (six one-row handles in ONE launch: generator.pick_layout) equals the same row generated in a 2-row run, bit for bit; and so does a run forced back to
    handles of three rows."""
    m, w = tiny_cfg()
    P = M.init_params(m, w, 10, seed=11, randomize_all=True)
    model = build(pkg, m, w, 10, P)
    x, spk, _ = M.synthetic_batch(6, 512, 10, 77)
    enc = model.encode(x[:, :, 0].contiguous().cuda(), spk.cuda())
    u = torch.rand(6, 96, generator=torch.Generator().manual_seed(1)).cuda()
    big = pkg.generator.FastGenerator(model, batch=6)
    assert big._parts == [1] * 6 and big._waves == [list(range(6))]
    ab, ib = big.generate(enc, 96, mode='sample', uniforms=u)
    big.close()
    os.environ['VQW_AR_ROWS'] = '3'
    try:
        three = pkg.generator.FastGenerator(model, batch=6)
    finally:
        del os.environ['VQW_AR_ROWS']
    assert three._parts == [3, 3]
    a3, i3 = three.generate(enc, 96, mode='sample', uniforms=u)
    three.close()
    assert torch.equal(i3, ib) and torch.equal(a3, ab)
    for r0 in (0, 2, 4):
        small = pkg.generator.FastGenerator(model, batch=2)
        a_s, i_s = small.generate(enc[r0:r0 + 2].contiguous(), 96, mode='sample', uniforms=u[r0:r0 + 2].contiguous())
        small.close()
        assert torch.equal(i_s, ib[r0:r0 + 2]) and torch.equal(a_s, ab[r0:r0 + 2])


def test_fast_generation_more_rows_than_one_launch(pkg):
    """20 rows = five 4-row handles; the chip holds only some of them at once (one resident workgroup per CU), so they
    run as waves of co-resident handles (generator.CU_MARGIN) instead of half-resident grids spinning into their
    timeout.  Every row equals the same row generated alone."""
    m, w = tiny_cfg()
    P = M.init_params(m, w, 10, seed=11, randomize_all=True)
    model = build(pkg, m, w, 10, P)
    x, spk, _ = M.synthetic_batch(20, 512, 10, 78)
    enc = model.encode(x[:, :, 0].contiguous().cuda(), spk.cuda())
    big = pkg.generator.FastGenerator(model, batch=20)
    assert sum(big._parts) == 20 and len(big._parts) == 5 and sorted(i for wv in big._waves for i in wv) == list(range(5))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    nwg = pkg._lib.lib().vqw_ar_decode_workgroups(big._hs[0])
    assert nwg > 0 and all(len(wv) * nwg <= cus for wv in big._waves)
    ab, ib = big.generate(enc, 80)
    big.close()
    for r0 in (0, 6, 18):
        small = pkg.generator.FastGenerator(model, batch=2)
        a_s, i_s = small.generate(enc[r0:r0 + 2].contiguous(), 80)
        small.close()
        assert torch.equal(i_s, ib[r0:r0 + 2]) and torch.equal(a_s, ab[r0:r0 + 2])


def test_encode_one_utterance_many_speakers_and_small_workspace(pkg):
    """generate.py:40 repeats one utterance per speaker: encode() runs encoder + VQ once and tiles only the speaker rows;
    its workspace is the encoder's, not the training step's (136 KB per audio sample)."""
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    model = pkg.model.VQVAE(m, w, 109, device='cuda', seed=0)
    T = 16000 * 4                                                      # a 4 s clip
    x, _, _ = M.synthetic_batch(1, T, 109, 5)
    xd = x[:, :, 0].contiguous().cuda()
    spk = torch.tensor([3, 17, 108, 0, 44, 45, 46, 47], dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    enc = model.encode(xd, spk)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert peak < 8 * 1024 * T, 'encode() of a %d-sample clip allocated %.1f MB' % (T, peak / 2 ** 20)   # < 8 KB per sample
    rep = model.encode(xd.repeat(8, 1).contiguous(), spk)       # what generate.py:40 does: same values, 8x the work
    D = model.D                                                 # (the batch size picks another tiling: not bit-equal)
    assert torch.equal(enc[:, D:], rep[:, D:]) and float((enc[:, :D] - rep[:, :D]).abs().max()) < 1e-5
    assert not any(k[2] for k in model._ws), 'encode() built a training workspace'


@pytest.mark.parametrize('variant', ['no_vq', 'one_hot_speaker', 'both'])
def test_config_variants_use_vq_false_and_one_hot_speakers(pkg, variant):
    """model_parameters.json variants of the reference: use_vq=false (z_q = e_k = z_e, reconstruction loss only,
    no codebook variable: model.py:137-141) and speaker_embedding=0 (the one-hot speaker vector itself is concatenated
    as the global condition: model.py:19-27, decoder_ops.py:39-43; 10 speakers -> a 26-channel condition that the
    kernels pad to 32).  Two train steps against the oracle, then generation through the padded condition.

    Round 2 saw [one_hot_speaker] fail twice in ~14 runs at 5e-3 and widened the bar to 2e-2.  Cause (round 3): in that variant
    the oracle's skip sum -- the input of the relu before postprocess1, wavenet.py:79-80 -- has an element at 7.45e-07 with a
    tensor max of 5.5 (1.4e-7 of max = one fp32 ulp of the sum; measured with M.train_step(collect=...)), and this build's forward
    pass is not bit-reproducible from run to run (the tiny configuration's short encoder layers are split over K with fp32
    atomics), so that element's relu mask comes out either way; one flipped element moves every gradient by up to ~1e-2 at
    B*T = 1024.  Not a state leak: the whole GPU suite ran once with VQW_POISON=1 (every workspace buffer NaN at allocation and
    at the start of every step, _alloc.py) and this test stayed finite and green.  The bar is 5e-3 again; the 2e-2 relative-L2
    form applies only to a step in which run_parity demonstrates a mask flip within 2e-6 of the tensor max of zero, and it
    prints the element."""
    m, w = tiny_cfg()
    if variant in ('no_vq', 'both'):
        m = dict(m, use_vq=False)
    if variant in ('one_hot_speaker', 'both'):
        m = dict(m, speaker_embedding=0)
    worst = run_parity(pkg, m, w, 10, 2, 512, seed=31, steps=2, grad_tol=5e-3, resync=True, check_params=False, flip_tol=2e-2)
    P = M.init_params(m, w, 10, seed=31, randomize_all=True)
    assert ('embedding/embedding' in P) == m['use_vq'] and ('speaker_embedding' in P) == (m['speaker_embedding'] > 0)
    model = build(pkg, m, w, 10, P)
    assert set(model.named_parameters()) == set(P)
    if m['speaker_embedding'] == 0:
        assert model.Cc == 16 + 16 and tuple(P['decoder/postprocess1/local_condition/kernel'].shape) == (1, 26, 64)
    x, spk, _ = M.synthetic_batch(2, 512, 10, 1234)
    with torch.no_grad():
        enc_ref = M.forward(x, spk, P, m, w)['local_condition']
    enc = model.encode(x[:, :, 0].contiguous().cuda(), spk.cuda())
    assert relerr(enc[:, :enc_ref.shape[2]].permute(0, 2, 1), enc_ref) < 1e-5 and float(enc[:, enc_ref.shape[2]:].abs().max() if enc.shape[1] > enc_ref.shape[2] else 0.0) == 0.0
    gen = pkg.generator.FastGenerator(model, batch=2)
    audio, idx = gen.generate(enc, 40)
    gen.close()
    want_idx, want_audio = M.generate(P, w, enc_ref[:, :1].expand(-1, 40, -1), 40, 'greedy')    # frame 0 for all 40 steps (ratio 64)
    assert (idx.cpu().numpy() == want_idx).mean() > 0.9          # greedy decisions (a near-tie may part the runs)
    print('variant', variant, 'worst grad', worst)


def test_magenta_encoder_model_parity(pkg):
    """Encoder_Magenta (encoder.py:29-63) wired into the same VQ + decoder: two full train steps."""
    m, w = tiny_cfg()
    m = dict(m, encoder='Magenta')
    worst = run_parity(pkg, m, w, 10, 2, 512, seed=21, steps=2)
    print('worst grad', worst)


def test_encoder_2019_model_parity(pkg):
    """Encoder_2019 (encoder.py:66-98): MFCC-13 front end + 768-wide conv stack, T % 320 == 0,
    one latent per 320 samples (so the decoder's condition ratio is 320, not 64)."""
    m, w = tiny_cfg()
    m = dict(m, encoder='2019')
    worst = run_parity(pkg, m, w, 10, 2, 1280, seed=23, steps=2, grad_tol=3e-3, check_params=False)  # Adam: kernel test
    print('worst grad', worst)


def test_mfcc_kernel_matches_oracle(pkg):
    from oracle import ref_ops as R
    K = pkg.kernels
    x, _, _ = M.synthetic_batch(3, 6400 + 160 * 3 + 57, 10, 5)      # ragged tail: zero padded frame
    xb = x[:, :, 0].contiguous().cuda()
    T = xb.shape[1]
    frames = -(-T // 160)
    out = torch.full((3, 16, frames), float('nan'), device='cuda')
    mel = pkg.encoders.mel_weight_matrix().cuda()
    assert torch.equal(mel.cpu(), torch.from_numpy(R.linear_to_mel_weight_matrix()))
    K.mfcc(xb, mel, out)
    want = R.mfcc(x[:, :, 0])                                        # [B,frames,13]
    got = out[:, :13].permute(0, 2, 1).cpu()
    assert float((got - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))      # observed 5.8e-7 of max (tools/mfcc_err.py)
    assert float(out[:, 13:].abs().max()) == 0.0
    assert torch.equal(pkg.ops.mfcc(xb).cpu(), got)                  # the reference-named op (encoder_ops.py:14)
    assert pkg.encoders.Encoder_2019 is pkg.encoders.Encoder2019 and pkg.encoders.Encoder_Magenta is pkg.encoders.EncoderMagenta

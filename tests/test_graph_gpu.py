"""The reference's builder surface (graph.py: implicit variables under variable scopes, classes Encoder_64 / Wavenet /
WavenetDecoder, per-sample fast_* queue ops) against the oracle on the tiny configuration: same variable names as the
reference, logits 2e-4 of the tensor max, labels bit-exact, per-sample probabilities 1e-5."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import ref_model as M

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def tiny_cfg():
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLD, 'make_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.tiny_cfg()


def relerr(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


def test_builder_classes_match_oracle(pkg):
    G = pkg.graph
    m, w = tiny_cfg()
    P = M.init_params(m, w, 10, seed=11, randomize_all=True)
    x, spk, _ = M.synthetic_batch(2, 512, 10, 1234)
    with torch.no_grad():
        ref = M.forward(x, spk, P, m, w)
    store = G.VariableStore({k: v.clone() for k, v in P.items()}, device='cuda')
    with store.use():
        with G.variable_scope('encoder'):
            z_e = G.Encoder_64(m['latent_dim'], filters=m['encoder_filters']).build(x.cuda())
        assert relerr(z_e, ref['z_e']) < 2e-4
        h = P['speaker_embedding'][spk].unsqueeze(1).cuda()                 # model.py:22-27
        dec = G.WavenetDecoder(w)
        with G.variable_scope('decoder'):
            logits, labels = dec.build(x.cuda(), ref['z_q'].cuda(), h)
    assert torch.equal(labels.cpu(), ref['labels'])
    assert relerr(logits, ref['logits']) < 2e-4
    assert set(store.vars) == set(P), 'the graph created variables the reference does not have: %s' % (set(store.vars) ^ set(P))
    assert dec.wavenet.receptive_field == sum(w['dilation_rates']) * 2 + 1 + 31     # wavenet.py:16-17
    # implicit creation with the reference's initialisers: names and shapes of Appendix B
    fresh = G.VariableStore(device='cuda', seed=1)
    with fresh.use(), G.variable_scope('decoder'):
        G.Wavenet(w).build(x.cuda(), ref['local_condition'].cuda())
    want = {k: tuple(v.shape) for k, v in P.items() if k.startswith('decoder/')}
    assert {k: tuple(v.shape) for k, v in fresh.vars.items()} == want
    with pytest.raises(RuntimeError):
        G.conv1d_v2(x.cuda(), 8, 3)                                        # no store active


def test_fast_ops_one_sample_at_a_time(pkg):
    """Wavenet.build_generator + push_ops (generate.py:103-113) for 40 samples of two rows, fed with the oracle's own
    greedy samples: the probabilities agree with the oracle's FIFO-queue generator at every step."""
    G = pkg.graph
    m, w = tiny_cfg()
    P = M.init_params(m, w, 10, seed=11, randomize_all=True)
    x, spk, _ = M.synthetic_batch(2, 512, 10, 1234)
    with torch.no_grad():
        enc = M.forward(x, spk, P, m, w)['local_condition']                # [B, Tz, Cc]
    ref = M.FastGenerator(P, w, 2)
    store = G.VariableStore({k: v.clone() for k, v in P.items()}, device='cuda')
    wn = G.Wavenet(w)
    a = np.zeros([2, 1], np.float32)
    with store.use(), torch.no_grad():
        for i in range(40):
            cond = enc[:, i // 16]
            pr = ref.step(torch.from_numpy(a), cond).numpy()
            with G.variable_scope('decoder'):
                got = wn.build_generator(torch.from_numpy(a).cuda(), cond.cuda(), None, 2)
            if i == 0:
                assert len(wn.init_ops) == len(wn.push_ops) == 31 + 2 * len(w['dilation_rates'])
            for op in wn.push_ops:
                op()
            np.testing.assert_allclose(got.cpu().numpy(), pr, rtol=2e-4, atol=1e-6)
            a = M.R.mu_law_decode_np(pr.argmax(-1).astype(np.float32)).reshape(2, 1)
        for op in wn.init_ops:                                             # generate.py:105: queues back to zeros
            op()
        ref.reset()
        pr = ref.step(torch.zeros(2, 1), enc[:, 0]).numpy()
        with G.variable_scope('decoder'):
            got = wn.build_generator(torch.zeros(2, 1).cuda(), enc[:, 0].cuda(), None, 2)
        np.testing.assert_allclose(got.cpu().numpy(), pr, rtol=2e-4, atol=1e-6)


def test_global_condition_is_a_second_add_condition(pkg):
    """wavenet_ops.py:109-110,232-233 and wavenet.py:89-91,160-162: a non-None global_condition ([B, 1, Cg], e.g. a speaker
    embedding that is NOT concatenated) is projected by its own 1x1 under the scope 'global_condition' and added like the local
    one.  The reference's decoder never does this (decoder.py:34-36 concatenates and passes None), but the op surface has it:
    training graph and per-sample graph against the oracle, and the variables the graph creates."""
    G = pkg.graph
    m, w = tiny_cfg()
    P = M.init_params(m, w, 10, seed=13, randomize_all=True)
    g = torch.Generator().manual_seed(5)
    Cg, R2, S = 24, 2 * w['dilation_filters'], w['skip_filters']
    ncl = w['num_cycle_layers']
    for i in range(len(w['dilation_rates'])):
        P[M.layer_scope(i, ncl) + '/gated/global_condition/kernel'] = (torch.rand(1, Cg, R2, generator=g) - 0.5) * 0.3
    P['decoder/postprocess1/global_condition/kernel'] = (torch.rand(1, Cg, S, generator=g) - 0.5) * 0.3
    x, spk, _ = M.synthetic_batch(2, 512, 10, 1234)
    glob = torch.randn(2, 1, Cg, generator=g)
    with torch.no_grad():
        local = M.forward(x, spk, P, m, w)['local_condition']
        want, labels = M.wavenet_build(x, local, P, w, global_condition=glob)
        base, _ = M.wavenet_build(x, local, P, w)
    assert relerr(want, base) > 1e-2                                       # the second condition matters in this set-up
    store = G.VariableStore({k: v.clone() for k, v in P.items()}, device='cuda')
    wn = G.Wavenet(w)
    with store.use(), G.variable_scope('decoder'):
        logits, lab = wn.build(x.cuda(), local.cuda(), glob.cuda())
    assert torch.equal(lab.cpu(), labels) and relerr(logits, want) < 2e-4
    assert set(store.vars) == set(P)
    # a global condition with its own frame rate: Tg = 4 frames against Tz = 8 (the epilogue operand is their sum at 8)
    glob4 = torch.randn(2, 4, Cg, generator=g)
    with torch.no_grad():
        want4, _ = M.wavenet_build(x, local, P, w, global_condition=glob4)
    with store.use(), G.variable_scope('decoder'):
        logits4, _ = wn.build(x.cuda(), local.cuda(), glob4.cuda())
    assert relerr(logits4, want4) < 2e-4
    # implicit creation: names and shapes
    fresh = G.VariableStore(device='cuda', seed=2)
    with fresh.use(), G.variable_scope('decoder'):
        G.Wavenet(w).build(x.cuda(), local.cuda(), glob.cuda())
    assert {k: tuple(v.shape) for k, v in fresh.vars.items()} == {k: tuple(v.shape) for k, v in P.items() if k.startswith('decoder/')}
    # per-sample graph
    ref = M.FastGenerator(P, w, 2)
    a = np.zeros([2, 1], np.float32)
    with store.use(), torch.no_grad():
        for i in range(12):
            cond = local[:, i // 64]
            pr = ref.step(torch.from_numpy(a), cond, glob[:, 0]).numpy()
            with G.variable_scope('decoder'):
                got = wn.build_generator(torch.from_numpy(a).cuda(), cond.cuda(), glob[:, 0].cuda(), 2)
            for op in wn.push_ops:
                op()
            np.testing.assert_allclose(got.cpu().numpy(), pr, rtol=2e-4, atol=1e-6)
            a = M.R.mu_law_decode_np(pr.argmax(-1).astype(np.float32)).reshape(2, 1)

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

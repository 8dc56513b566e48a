"""Relu-mask comparison for the model parity tests.

Two fp32 evaluations of the same forward pass differ by summation order (~1e-7 of a tensor's max); the forward pass of this
build is also not bit-reproducible from run to run where a short conv is split over K with fp32 atomics.  A relu input that
sits within that noise of zero can therefore have a different sign on the device than in the oracle ("a flipped mask"): the
element's whole backward contribution appears or disappears, which moves EVERY gradient by a fraction of a percent at
B*T = 1024 although nothing is wrong.  `relu_flips` finds such elements so that a test can (a) show them in its failure
message and (b) widen its gradient bar only when a flip is demonstrated AND the flipped input is within `noise` of zero on
both sides."""
import torch


def relu_flips(pairs, noise=2e-6):
    """pairs: {name: (device tensor, oracle tensor)} holding relu INPUTS (or outputs: a relu output is positive exactly where
    its input is) in the same element order.  -> list of dicts {tensor, index, device, oracle, max, benign}; benign = both
    values within noise * max|oracle tensor| of zero."""
    out = []
    for name, (got, ref) in pairs.items():
        got, ref = got.detach().cpu().float().reshape(-1), ref.detach().cpu().float().reshape(-1)
        if got.numel() != ref.numel():
            raise ValueError('%s: %d device elements against %d oracle elements' % (name, got.numel(), ref.numel()))
        mx = float(ref.abs().max())
        for i in torch.nonzero((got > 0) != (ref > 0)).reshape(-1).tolist():
            g, r = float(got[i]), float(ref[i])
            out.append({'tensor': name, 'index': i, 'device': g, 'oracle': r, 'max': mx,
                        'benign': max(abs(g), abs(r)) <= noise * mx})
    return out


def describe(flips, limit=6):
    if not flips:
        return 'no relu mask differs from the oracle'
    return '; '.join('%s[%d]: device %.3e, oracle %.3e (tensor max %.2f)%s' % (f['tensor'], f['index'], f['device'], f['oracle'],
                                                                              f['max'], '' if f['benign'] else ' NOT within noise')
                     for f in flips[:limit]) + (' ... %d in all' % len(flips) if len(flips) > limit else '')

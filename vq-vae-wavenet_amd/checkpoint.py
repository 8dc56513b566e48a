"""Checkpoint interoperability: the model state under the REFERENCE's TensorFlow variable names.

The reference saves TF-1.x checkpoints (`tf.train.Saver`, train.py:81-84,123) and generate.py:88-90 restores the
ExponentialMovingAverage shadows into the live variables (`model.ema.variables_to_restore()`).  TensorFlow is not
available here, so the wire format is a `safetensors` file (or an `.npz`, which is what three lines of numpy on a TF
machine produce from a TF checkpoint: INTEGRATION.md) keyed exactly like the TF checkpoint:

    <name>                                 live variable, reference shape, e.g. decoder/skip/kernel [1,256,512]
    <name>/ExponentialMovingAverage        EMA shadow of every trainable variable (ema.average_name)
    <name>/Adam, <name>/Adam_1             Adam first / second moment slots
    global_step                            int64 scalar

Slot scopes of a real TF checkpoint (`optimiser/...`, `OptimizeLoss/...` prefixes) are not pinned -- parity unpinned, as
for everything TF-internal; `load` therefore also accepts keys that END in the names above.
"""
import numpy as np
import torch

EMA, M1, M2 = '/ExponentialMovingAverage', '/Adam', '/Adam_1'


def _views(model):
    live = model._named(model.P)
    ema = model._named(model.E, bn_stats=False)
    m = model._named(model._views(model.adam_m), bn_stats=False)
    v = model._named(model._views(model.adam_v), bn_stats=False)
    return live, ema, m, v


def named_state(model):
    """name -> CPU tensor (copies) of everything a TF checkpoint of the reference holds."""
    live, ema, m, v = _views(model)
    out = {k: t.detach().cpu().contiguous() for k, t in live.items()}
    for sfx, group in ((EMA, ema), (M1, m), (M2, v)):
        out.update({k + sfx: t.detach().cpu().contiguous() for k, t in group.items()})
    out['global_step'] = torch.tensor(model.global_step, dtype=torch.int64)
    return out


def save(model, path):
    """`.safetensors` (default) or `.npz`."""
    state = named_state(model)
    if path.endswith('.npz'):
        np.savez(path, **{k: t.numpy() for k, t in state.items()})
    else:
        from safetensors.torch import save_file
        save_file(state, path, metadata={'format': 'vq-vae-wavenet TF variable names', 'num_speakers': str(model.S_spk)})
    return path


def _read(path):
    if path.endswith('.npz'):
        with np.load(path, allow_pickle=False) as z:
            return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
    from safetensors.torch import load_file
    return load_file(path, device='cpu')


def load(model, path, ema_to_live=False, strict=True):
    """Fill the model from a file written by `save` (or converted from a TF checkpoint).  ema_to_live: what generate.py:88-90
    does -- every trainable variable takes the value of its EMA shadow.  Missing EMA / Adam entries keep their current
    values unless `strict`.  Returns the list of keys that were used."""
    state = _read(path)

    def find(name):
        if name in state:
            return name
        hits = [k for k in state if k.endswith('/' + name)]        # tolerate slot-scope prefixes of a real TF checkpoint
        return hits[0] if len(hits) == 1 else None
    live, ema, m, v = _views(model)
    used = []
    for sfx, group, required in (('', live, True), (EMA, ema, strict), (M1, m, False), (M2, v, False)):
        for name, view in group.items():
            key = find(name + sfx)
            if key is None:
                if required:
                    raise KeyError('checkpoint %s has no variable %s' % (path, name + sfx))
                continue
            t = state[key]
            if tuple(t.shape) != tuple(view.shape):
                raise ValueError('%s: shape %s in the file, %s in the model' % (key, tuple(t.shape), tuple(view.shape)))
            view.copy_(t.to(view.device, dtype=view.dtype))
            used.append(key)
    key = find('global_step')
    if key is not None:
        model.global_step = int(state[key])
        used.append(key)
    if ema_to_live:
        model.use_ema_weights()
    return used

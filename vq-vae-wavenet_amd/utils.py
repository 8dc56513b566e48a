"""Host-side sampling helpers with the call surface of the reference's utils.py:13-46 (`sample`, `decode`) and
mu_law_ops.py:26-31 (`mu_law_decode_np`), for callers that hold the predicted probabilities on the host.

generate.py does NOT go through these: the persistent generator samples on the device with the same semantics
(sequential fp32 cumsum, first cdf entry >= u, argmax = first maximum).  Everything here is vectorised over the batch.
"""
import numpy as np

_MU_LEVELS = {}


def _decode_table(quantization_channels):
    """All decode levels at once: index i -> sign(y) ((1 + mu)^|y| - 1) / mu with y = 2 i / mu - 1, in numpy fp32
    (mu_law_ops.py:26-31); one extra entry for the index `quantization_channels` that sampling can return when
    u > cdf[-1]."""
    tab = _MU_LEVELS.get(quantization_channels)
    if tab is None:
        mu = np.float32(quantization_channels - 1)
        y = np.float32(2) * np.arange(quantization_channels + 1, dtype=np.float32) / mu - np.float32(1)
        tab = (np.sign(y) * (np.power(np.float32(1) + mu, np.abs(y)) - np.float32(1)) / mu).astype(np.float32)
        _MU_LEVELS[quantization_channels] = tab
    return tab


def mu_law_decode_np(output, quantization_channels=256):
    """Integer-valued mu-law indices (any float or int dtype, any shape) -> float32 amplitudes in [-1, 1]."""
    idx = np.asarray(output)
    as_int = idx.astype(np.int64)
    if idx.dtype.kind == 'f' and not np.array_equal(as_int, idx):     # fractional input: evaluate the formula itself
        mu = np.float32(quantization_channels - 1)
        y = np.float32(2) * idx.astype(np.float32) / mu - np.float32(1)
        return (np.sign(y) * (np.power(np.float32(1) + mu, np.abs(y)) - np.float32(1)) / mu).astype(np.float32)
    return _decode_table(quantization_channels)[np.clip(as_int, 0, quantization_channels)]


def sample_indices(pdf, uniforms=None):
    """Inverse-CDF draw per row: the number of cdf entries strictly below u (= searchsorted side='left')."""
    cdf = np.cumsum(np.asarray(pdf), axis=1)
    u = np.random.rand(cdf.shape[0]) if uniforms is None else np.asarray(uniforms)   # float64 against the fp32 cdf, as searchsorted
    return (cdf < u[:, None]).sum(axis=1)


def sample(pdf, quantization_channels=256, uniforms=None):
    """pdf [b, Q] -> decoded samples [b]; `uniforms` [b] replaces the reference's unseeded np.random.rand."""
    return mu_law_decode_np(sample_indices(pdf, uniforms), quantization_channels)


def decode(predictions, mode='sample', quantization_channels=256):
    if mode == 'greedy':
        return mu_law_decode_np(np.argmax(predictions, axis=-1), quantization_channels)
    if mode == 'sample':
        return sample(predictions, quantization_channels)
    raise NotImplementedError("decode mode %s not implemented" % mode)

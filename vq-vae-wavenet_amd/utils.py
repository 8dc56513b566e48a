"""Host-side mirror of utils.py:13-46 (sample / decode) for callers that hold the probabilities on the
host.  generate.py does NOT use these: the persistent generator samples on the device with the same
semantics (sequential fp32 cumsum, searchsorted side='left', argmax = first maximum)."""
import numpy as np


def mu_law_decode_np(output, quantization_channels=256):
    """mu_law_ops.py:26-31: y = 2 f32(idx)/mu - 1;  x = sign(y) ((1+mu)^|y| - 1) / mu, numpy fp32."""
    mu = np.asarray(quantization_channels - 1, dtype=np.float32)
    y = (2 * np.asarray(output, dtype=np.float32) / mu) - 1
    x = np.sign(y) * ((1 + mu) ** abs(y) - 1) / mu
    return x.astype(np.float32)


def sample(pdf, quantization_channels=256, uniforms=None):
    """utils.py:13-27: pdf [b, Q] -> decoded samples [b] in [-1, 1].  `uniforms` replaces np.random.rand."""
    cdf = np.cumsum(pdf, axis=1)
    batch_size = cdf.shape[0]
    sample_prob = np.random.rand(batch_size) if uniforms is None else np.asarray(uniforms)
    pred = np.zeros(batch_size, dtype=np.float32)
    for i, prob in enumerate(sample_prob):
        pred[i] = cdf[i].searchsorted(prob)
    return mu_law_decode_np(pred, quantization_channels=quantization_channels)


def decode(predictions, mode='sample', quantization_channels=256):
    """utils.py:30-46."""
    if mode == 'sample':
        return sample(predictions)
    elif mode == 'greedy':
        pred = np.argmax(predictions, axis=-1)
        return mu_law_decode_np(pred, quantization_channels=quantization_channels)
    raise NotImplementedError("decode mode %s not implemented" % mode)

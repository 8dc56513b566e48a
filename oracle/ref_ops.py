"""Oracle op library -- TEST INFRASTRUCTURE, never imported by the product.

Restates, on torch-CPU fp32 tensors in the reference's channels-last [B,T,C]
layout, the operator surface of the reference:

  mu_law_ops.py:5-31, Decoder/WaveNet/wavenet_ops.py:9-14,59-138,147-267,
  Encoder/encoder_ops.py:46-70, Decoder/decoder_ops.py:39-43, utils.py:13-46.

Variables are explicit arguments (the reference creates them implicitly through
tf.get_variable); kernels keep the reference shape [k, Cin, Cout].
PARITY UNPINNED against real TensorFlow (see oracle/__init__.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

MU = 255.0


# ----------------------------------------------------------------------------
# mu-law  (mu_law_ops.py:5-31)
# ----------------------------------------------------------------------------
def mu_law_encode_np(x, quantization_channels=256, to_int=False):
    """numpy fp32 evaluation of mu_law_ops.py:5-15 (every op rounded to fp32)."""
    mu = np.float32(quantization_channels - 1)
    x = np.clip(np.asarray(x, dtype=np.float32), np.float32(-1.0), np.float32(1.0))
    y = np.sign(x) * np.log1p(mu * np.abs(x)) / np.log1p(mu)
    y = y.astype(np.float32)
    if to_int:
        # tf.cast(float->int32) truncates toward zero (mu_law_ops.py:11)
        v = (y + np.float32(1.0)) / np.float32(2.0) * mu + np.float32(0.5)
        return v.astype(np.float32).astype(np.int32)
    return y


def mu_law_encode(x, quantization_channels=256, to_int=False):
    """torch fp32 twin (differentiable for the float form). mu_law_ops.py:5-15."""
    mu = float(quantization_channels - 1)
    x = torch.clamp(x, -1.0, 1.0)
    y = torch.sign(x) * torch.log1p(mu * torch.abs(x)) / math.log1p(mu)
    if to_int:
        return ((y + 1.0) / 2.0 * mu + 0.5).to(torch.int32)
    return y


def mu_law_decode_np(y, quantization_channels=256):
    """mu_law_ops.py:26-31, numpy fp32."""
    mu = np.asarray(quantization_channels - 1, dtype=np.float32)
    y = (2 * np.asarray(y, dtype=np.float32) / mu) - 1
    x = np.sign(y) * ((1 + mu) ** abs(y) - 1) / mu
    return x.astype(np.float32)


# ----------------------------------------------------------------------------
# wavenet_ops.py (training graph)
# ----------------------------------------------------------------------------
def shift_right(x):
    """wavenet_ops.py:9-14: x[t] <- x[t-1], x[0] <- 0 on [B,T,C]."""
    return F.pad(x, (0, 0, 1, 0))[:, :-1, :]


def conv1d_v2(net, kernel, bias=None, dilations=1, stride=1):
    """wavenet_ops.py:59-90.  net [B,T,Cin], kernel [k,Cin,Cout].

    Always left-pads d*(k-1) zeros (line 81), then a VALID conv2d with the given
    stride / dilation (lines 83-86), then + bias (88-89).
    """
    k = kernel.shape[0]
    x = net.transpose(1, 2)                       # [B,Cin,T]
    x = F.pad(x, (dilations * (k - 1), 0))
    w = kernel.permute(2, 1, 0).contiguous()      # [Cout,Cin,k]
    y = F.conv1d(x, w, None, stride=stride, dilation=dilations)
    y = y.transpose(1, 2)
    if bias is not None:
        y = y + bias
    return y


def add_condition(net, condition, cond_kernel):
    """wavenet_ops.py:93-101: 1x1 (no bias) on the condition, nearest-upsample add."""
    if condition is None:
        return net
    B, net_len, out_channels = net.shape
    T = condition.shape[1]
    encoding = conv1d_v2(condition, cond_kernel, None)
    net = net.reshape(B, T, net_len // T, out_channels)
    net = net + encoding.unsqueeze(2)
    return net.reshape(B, net_len, out_channels)


def gated_cnn(net, p, dilation_filters, dilations, local_condition, global_condition=None):
    """wavenet_ops.py:104-114.  p: dict with gated/{kernel,bias,local_condition/kernel[,global_condition/kernel]}."""
    net = conv1d_v2(net, p['gated/kernel'], p['gated/bias'], dilations)
    net = add_condition(net, local_condition, p.get('gated/local_condition/kernel'))
    net = add_condition(net, global_condition, p.get('gated/global_condition/kernel'))      # :109-110
    net_filter, net_gate = net[:, :, :dilation_filters], net[:, :, dilation_filters:]
    return torch.tanh(net_filter) * torch.sigmoid(net_gate)


def residual_stack(net, p, dilation_filters, dilations, local_condition, global_condition=None):
    """wavenet_ops.py:117-138 -> (skip_connection, residual_connection)."""
    gated = gated_cnn(net, p, dilation_filters, dilations, local_condition, global_condition)
    skip = conv1d_v2(gated, p['skip/kernel'], p['skip/bias'])
    res = conv1d_v2(gated, p['residual/kernel'], p['residual/bias'])
    return skip, res


# ----------------------------------------------------------------------------
# wavenet_ops.py (fast generation).  FIFO queues are explicit python deques of
# tensors; semantics follow fast_conv1d (163-195): queue i yields x[t-i*d].
# ----------------------------------------------------------------------------
def linear(net, kernel, bias=None):
    """wavenet_ops.py:147-160. net [B,Cin], kernel [1,Cin,Cout]."""
    y = net @ kernel[0]
    return y + bias if bias is not None else y


class FastConvState:
    """The k-1 FIFO queues of one fast_conv1d (wavenet_ops.py:178-193)."""

    def __init__(self, kernel_size, dilations, batch, channels):
        from collections import deque
        self.queues = [deque([torch.zeros(batch, channels) for _ in range(dilations)])
                       for _ in range(kernel_size - 1)]

    def step(self, current, kernel, bias):
        k = kernel.shape[0]
        new_state = current @ kernel[k - 1] + bias
        for i in range(1, k):
            q = self.queues[i - 1]
            past = q.popleft()
            q.append(current)
            current = past
            new_state = new_state + past @ kernel[k - i - 1]
        return new_state


# ----------------------------------------------------------------------------
# Encoder ops (Keras Conv1D 'same' semantics; encoder_ops.py:46-70, encoder.py:15-25)
# ----------------------------------------------------------------------------
def same_pads(n, k, s):
    """TF SAME: out=ceil(n/s); total=max((out-1)s+k-n,0); left=floor(total/2)."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def keras_conv1d(net, kernel, bias, stride=1, padding='same', relu=False):
    """tkl.Conv1D on [B,T,C]; kernel [k,Cin,Cout]."""
    k = kernel.shape[0]
    x = net.transpose(1, 2)
    if padding == 'same':
        pl, pr = same_pads(x.shape[-1], k, stride)
        x = F.pad(x, (pl, pr))
    y = F.conv1d(x, kernel.permute(2, 1, 0).contiguous(), bias, stride=stride)
    y = y.transpose(1, 2)
    return torch.relu(y) if relu else y


def batch_norm_inference(net, gamma, beta, mean, var, eps=1e-3):
    """tkl.BatchNormalization() called without `training` (encoder.py:20,25):
    TF-1.x Keras learning phase 0 => moving statistics."""
    return gamma * (net - mean) / torch.sqrt(var + eps) + beta


# ----------------------------------------------------------------------------
# MFCC front end of Encoder_2019 (encoder_ops.py:14-43), restating the tf.contrib.signal ops
# ----------------------------------------------------------------------------
def linear_to_mel_weight_matrix(num_mel_bins=80, num_spectrogram_bins=201, sample_rate=16000,
                                lower_edge_hertz=20.0, upper_edge_hertz=8000.0):
    """tf.contrib.signal.linear_to_mel_weight_matrix (HTK mel scale, DC bin zeroed)."""
    def hz_to_mel(f):
        return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)
    nyquist = sample_rate / 2.0
    linear = np.linspace(0.0, nyquist, num_spectrogram_bins)[1:]
    spec_mel = hz_to_mel(linear)[:, None]
    edges = np.linspace(hz_to_mel(lower_edge_hertz), hz_to_mel(upper_edge_hertz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None], edges[1:-1][None], edges[2:][None]
    w = np.maximum(0.0, np.minimum((spec_mel - lower) / (center - lower), (upper - spec_mel) / (upper - center)))
    return np.pad(w, [[1, 0], [0, 0]]).astype(np.float32)


def mfcc(batch_wav):
    """encoder_ops.py:14-43: batch_wav [B,T] -> [B, ceil(T/160), 13]."""
    frame_length, frame_step, num_mel = 400, 160, 80
    B, T = batch_wav.shape
    frames = -(-T // frame_step)
    pad = (frames - 1) * frame_step + frame_length - T                  # stft(pad_end=True)
    xp = F.pad(batch_wav, (0, max(pad, 0)))
    fr = xp.unfold(1, frame_length, frame_step)[:, :frames]              # [B,frames,400]
    n = torch.arange(frame_length, dtype=torch.float32)
    window = 0.5 - 0.5 * torch.cos(2 * math.pi * n / frame_length)      # hann_window(periodic=True)
    stft = torch.abs(torch.fft.rfft(fr * window, n=frame_length))        # [B,frames,201]
    feature = stft @ torch.from_numpy(linear_to_mel_weight_matrix(num_mel, stft.shape[-1]))
    feature = torch.log(feature + 1e-6)
    k = torch.arange(num_mel, dtype=torch.float32)
    basis = torch.cos(math.pi * k[None, :] * (2 * torch.arange(num_mel, dtype=torch.float32)[:, None] + 1) / (2 * num_mel))
    dct2 = 2.0 * (feature @ basis)                                       # tf.signal.dct(type=2)
    return (dct2 * (1.0 / math.sqrt(2.0 * num_mel)))[..., :13]           # mfccs_from_log_mel_spectrograms


def concat(net, global_condition):
    """decoder_ops.py:39-43."""
    g = global_condition.expand(-1, net.shape[1], -1)
    return torch.cat([net, g], dim=-1)


# ----------------------------------------------------------------------------
# utils.py:13-46 (numpy)
# ----------------------------------------------------------------------------
def sample_with_uniforms(pdf, u):
    """utils.py:13-27 with the uniforms supplied (np.random.rand in the reference)."""
    cdf = np.cumsum(pdf, axis=1)
    pred = np.zeros(cdf.shape[0], dtype=np.float32)
    for i, prob in enumerate(u):
        pred[i] = cdf[i].searchsorted(prob)
    return pred, mu_law_decode_np(pred)


def decode_greedy(predictions):
    """utils.py:42-44."""
    pred = np.argmax(predictions, axis=-1)
    return pred, mu_law_decode_np(pred)

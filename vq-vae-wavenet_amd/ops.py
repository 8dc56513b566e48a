"""The reference's operator surface (same names, same argument meaning) on GPU tensors.

Reference: mu_law_ops.py:5-31, Decoder/WaveNet/wavenet_ops.py:9-14,59-160,
Encoder/encoder_ops.py:46-70, Decoder/decoder_ops.py:39-43.  Tensors are channels-last
[B,T,C] like the reference's; each op transposes to the kernels' (batch, channel, time) layout,
calls libvqwave and transposes back (the assembled model in model.py stays in (B,C,T)
throughout).  Where the reference creates variables implicitly (tf.get_variable) the kernel /
bias tensors are explicit arguments with the reference's shapes: kernel [k, Cin, Cout].
The reference's own signatures (variables created implicitly under a variable scope), the
classes Encoder_64 / Wavenet / WavenetDecoder and the per-sample `fast_*` queue ops
(wavenet_ops.py:163-267) live in graph.py on top of these; the fused generator is
generator.FastGenerator (vqw_ar_decode_*).
"""
import torch

from . import _lib as L
from . import kernels as K


def _bct(x):
    return x.transpose(1, 2).contiguous()


def _btc(x):
    return x.transpose(1, 2).contiguous()


def mu_law_encode(x, quantization_channels=256, to_int=False, one_hot=False):
    """mu_law_ops.py:5-15."""
    if quantization_channels != 256:
        raise NotImplementedError('quantization_channels must be 256 (threshold table)')
    x = x.contiguous()
    if to_int or one_hot:
        y = K.mu_law_encode_i32(x)
        if one_hot:
            return torch.nn.functional.one_hot(y.long(), quantization_channels).float().squeeze(-2)
        return y
    return K.mu_law_encode_f32(x)


def mu_law_decode(y, quantization_channels=256):
    """mu_law_ops.py:18-31 (TF and numpy twins)."""
    if quantization_channels != 256:
        raise NotImplementedError('quantization_channels must be 256')
    return K.mu_law_decode_f32(y.float().contiguous())


mu_law_decode_np = mu_law_decode


def shift_right(x):
    """wavenet_ops.py:9-14."""
    return torch.nn.functional.pad(x, (0, 0, 1, 0))[:, :-1, :].contiguous()


class _CausalConv1d(torch.autograd.Function):
    """conv1d_v2 with its TF gradients (Conv2DBackpropInput / Conv2DBackpropFilter / BiasAddGrad) through the C ABI:
    vqw_causal_conv1d_{fwd,dgrad,wgrad}.  Stride 1, Cin a multiple of 16 (the shapes the reference differentiates)."""

    @staticmethod
    def forward(ctx, net, kernel, bias, dilations):
        B, T, Cin = net.shape
        k, _, Cout = kernel.shape
        x = _bct(net)
        y = torch.empty(B, Cout, T, device=net.device)
        L.check(L.lib().vqw_causal_conv1d_fwd(L.ptr(x), L.ptr(kernel.contiguous()), L.ptr(bias), L.ptr(y), B, Cin, Cout,
                                              T, k, dilations, 1, L.stream()))
        ctx.save_for_backward(x, kernel)
        ctx.dil, ctx.has_bias = dilations, bias is not None
        return _btc(y)

    @staticmethod
    def backward(ctx, dy_btc):
        x, kernel = ctx.saved_tensors
        B, Cin, T = x.shape
        k, _, Cout = kernel.shape
        dy = _bct(dy_btc)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wT = torch.empty(k, Cout, Cin, device=x.device)
            K.transpose(kernel.contiguous(), wT, k, Cin, Cout)
            dxc = torch.empty(B, Cin, T, device=x.device)
            L.check(L.lib().vqw_causal_conv1d_dgrad(L.ptr(dy), L.ptr(wT), L.ptr(dxc), B, Cin, Cout, T, k, ctx.dil, L.stream()))
            dx = _btc(dxc)
        if ctx.needs_input_grad[1]:
            dw = torch.zeros(k, Cin, Cout, device=x.device)
            L.check(L.lib().vqw_causal_conv1d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dw), B, Cin, Cout, T, k, ctx.dil, L.stream()))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(Cout, device=x.device)
            K.rowsum(dy, total=db)
        return dx, dw, db, None


def conv1d_v2(net, kernel, bias=None, padding='CAUSAL', dilations=1, stride=1):
    """wavenet_ops.py:59-90: always left-pads d*(k-1); the `padding` argument of the reference
    only selects VALID/SAME for the conv that follows the pad and every caller passes
    CAUSAL/VALID.  Differentiable (torch.autograd) for stride 1 and Cin % 16 == 0."""
    if padding.upper() not in ('CAUSAL', 'VALID'):
        raise NotImplementedError("padding %s not used by the reference's callers" % padding)
    B, T, Cin = net.shape
    k, _, Cout = kernel.shape
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (net, kernel, bias))
    if needs_grad:
        if stride != 1 or Cin % 16 != 0 or Cout % 16 != 0:
            raise NotImplementedError('conv1d_v2 is differentiable for stride 1 and channel counts that are multiples of 16')
        return _CausalConv1d.apply(net, kernel, bias, dilations)
    if Cin == 1:
        if dilations != 1:
            raise NotImplementedError('Cin == 1 convs are only used with dilation 1 by the reference')
        To = -(-T // stride)
        out = torch.empty(B, Cout, To, device=net.device)
        K.conv_cin1_fwd(net[:, :, 0].contiguous(), kernel.reshape(k, Cout).contiguous(), bias, out, k=k,
                        stride=stride, offset=-(k - 1))
        return _btc(out)
    x = _bct(net)
    if Cin % 16:        # the conv engine contracts over channel blocks of 16: zero channels against zero kernel rows
        pad = 16 - Cin % 16
        x = torch.nn.functional.pad(x, (0, 0, 0, pad)).contiguous()
        kernel = torch.nn.functional.pad(kernel, (0, 0, 0, pad))
        Cin += pad
    To = -(-T // stride)
    y = torch.empty(B, Cout, To, device=net.device)
    L.check(L.lib().vqw_causal_conv1d_fwd(L.ptr(x), L.ptr(kernel.contiguous()), L.ptr(bias), L.ptr(y), B, Cin, Cout,
                                          T, k, dilations, stride, L.stream()))
    return _btc(y)


def linear(net, kernel, bias=None):
    """wavenet_ops.py:147-160: one stride of a 1x1 conv on [B, Cin]."""
    return conv1d_v2(net.unsqueeze(1), kernel, bias).squeeze(1)


def add_condition(net, condition, kernel):
    """wavenet_ops.py:93-101: 1x1 (no bias) on the condition + nearest-neighbour upsample-add."""
    if condition is None:
        return net
    B, T, C = net.shape
    Tz = condition.shape[1]
    enc = _condition_projection(condition, kernel, C)
    return (net.reshape(B, Tz, T // Tz, C) + _btc(enc).unsqueeze(2)).reshape(B, T, C)


def _condition_projection(condition, cond_kernel, C):
    """The 1x1 (no bias) of add_condition (wavenet_ops.py:97) -> [B, C, T_cond] in the kernels' layout."""
    B, Tc, Cc = condition.shape
    enc = torch.empty(B, C, Tc, device=condition.device)
    x, w = _bct(condition), cond_kernel.reshape(Cc, C)
    if Cc % 16:         # the conv engine contracts over channel blocks of 16: zero channels against zero kernel rows
        pad = 16 - Cc % 16
        x = torch.nn.functional.pad(x, (0, 0, 0, pad))
        w = torch.nn.functional.pad(w, (0, 0, 0, pad))
        Cc += pad
    K.conv_gemm(x0=x.contiguous(), w=w.contiguous(), out0=enc, B=B, T_in=Tc, T_out=Tc, M=C, C0=Cc, taps=[0])
    return enc


def _merged_condition(net_len, C, conditions):
    """Projections of several conditions (local [B, Tz, Cc], global [B, Tg, Cg], ...) as ONE [B, C, Tc] operand of the conv
    epilogue's upsampled add: each is repeated up to the common frame count Tc = lcm(Tz, Tg) and they are summed, which is
    what the consecutive add_condition calls of wavenet_ops.py:107-110 amount to (net_len % Tc == 0 as their reshapes need)."""
    import math
    encs = [_condition_projection(c, k, C) for c, k in conditions if c is not None]
    if not encs:
        return None
    Tc = 1
    for e in encs:
        Tc = Tc * e.shape[2] // math.gcd(Tc, e.shape[2])
    if net_len % Tc != 0:
        raise ValueError('condition lengths %s do not divide the sequence length %d' % ([e.shape[2] for e in encs], net_len))
    out = None
    for e in encs:
        e = e if e.shape[2] == Tc else e.repeat_interleave(Tc // e.shape[2], dim=2)
        out = e if out is None else out + e
    return out.contiguous()


def gated_cnn(net, kernel, bias, dilations, local_condition=None, cond_kernel=None, global_condition=None, global_kernel=None):
    """wavenet_ops.py:104-114: causal dilated conv -> +local condition -> +global condition -> tanh(first half)*sigmoid(second)."""
    B, T, Cin = net.shape
    k, _, C2 = kernel.shape
    H = C2 // 2
    out = torch.empty(B, H, T, device=net.device)
    kw = {}
    enc = _merged_condition(T, C2, [(local_condition, cond_kernel), (global_condition, global_kernel)])
    if enc is not None:
        kw = dict(cond=enc, cond_T=enc.shape[2])
    K.conv_gemm(x0=_bct(net), w=kernel.contiguous(), bias=bias, out0=out, B=B, T_in=T, T_out=T, M=C2, C0=Cin,
                taps=[-(k - 1 - j) * dilations for j in range(k)], epilogue=K.EPI_GATE, **kw)
    return _btc(out)


def residual_stack(net, params, dilations, local_condition=None, global_condition=None):
    """wavenet_ops.py:117-138 -> (skip_connection, residual_connection).
    params: gated/{kernel,bias}, gated/local_condition/kernel, gated/global_condition/kernel, skip/{kernel,bias},
    residual/{kernel,bias}."""
    gated = gated_cnn(net, params['gated/kernel'], params['gated/bias'], dilations, local_condition,
                      params.get('gated/local_condition/kernel'), global_condition, params.get('gated/global_condition/kernel'))
    skip = conv1d_v2(gated, params['skip/kernel'], params['skip/bias'])
    res = conv1d_v2(gated, params['residual/kernel'], params['residual/bias'])
    return skip, res


def _keras_conv1d(net, kernel, bias, stride, relu):
    B, T, Cin = net.shape
    k, _, Cout = kernel.shape
    To = -(-T // stride)
    total = max((To - 1) * stride + k - T, 0)       # TF 'SAME' 
    y = torch.empty(B, Cout, To, device=net.device)
    if Cin == 1:        # the first encoder layer (encoder.py:15 on raw audio): VALU kernel, HBM-bound by its output
        K.conv_cin1_fwd(net[:, :, 0].contiguous(), kernel.reshape(k, Cout).contiguous(), bias, y, k=k, stride=stride,
                        offset=-(total // 2), relu=relu)
        return _btc(y)
    x = _bct(net)
    L.check(L.lib().vqw_conv1d_same_fwd(L.ptr(x), L.ptr(kernel.contiguous()), L.ptr(bias), L.ptr(y), B, Cin, Cout, T,
                                        To, k, stride, total // 2, int(relu), L.stream()))
    return _btc(y)


def keras_conv1d(net, kernel, bias, stride=1, relu=False):
    """tf.keras.layers.Conv1D(padding='same' -- or 'valid' with k=1) as used by Encoder/encoder.py:15-25."""
    return _keras_conv1d(net, kernel, bias, stride, relu)


def conv_3_768(net, kernel, bias, relu='relu'):
    """encoder_ops.py:46-52 (Keras Conv1D k=3, 'same', relu)."""
    return _keras_conv1d(net, kernel, bias, 1, relu == 'relu')


def strided_conv_4_768(net, kernel, bias, relu='relu'):
    """encoder_ops.py:55-61 (k=4, stride 2, 'same')."""
    return _keras_conv1d(net, kernel, bias, 2, relu == 'relu')


def linear_64(net, kernel, bias):
    """encoder_ops.py:64-70 (1x1, no activation)."""
    return _keras_conv1d(net, kernel, bias, 1, False)


def mfcc(audio, mel=None):
    """encoder_ops.py:14-43: audio [B,T] -> [B, ceil(T/160), 13] (STFT 400/160 hann, pad_end, 201 bins ->
    80 mel bins 20-8000 Hz -> log(. + 1e-6) -> DCT-II -> first 13), one vqw_mfcc launch."""
    from .encoders import mel_weight_matrix
    audio = audio.contiguous()
    B, T = audio.shape
    if mel is None:
        mel = mel_weight_matrix().to(audio.device)
    frames = -(-T // 160)
    out = torch.empty(B, 13, frames, device=audio.device)
    K.mfcc(audio, mel.contiguous(), out, n_keep=13)
    return _btc(out)


def concat(net, global_condition):
    """decoder_ops.py:39-43."""
    return torch.cat([net, global_condition.expand(-1, net.shape[1], -1)], dim=-1)

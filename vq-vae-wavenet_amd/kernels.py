"""Typed Python front end of the C ABI on (batch, channel, time) torch tensors.

Every function validates operand sizes on the host before launching (a mis-sized operand
would otherwise fault on the GPU) and then calls libvqwave; nothing here computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import EPI_ACCUM_SPLIT, EPI_GATE, EPI_GATE_BWD, EPI_MASK, EPI_STORE  # noqa: F401


def _need(t, n, what):
    if t is None:
        raise ValueError('%s is required' % what)
    if t.dtype != torch.float32:
        raise ValueError('%s must be float32, got %s' % (what, t.dtype))
    if t.numel() < n:
        raise ValueError('%s has %d elements, kernel needs %d' % (what, t.numel(), n))
    L.require_cuda(t)


def conv_gemm(*, x0, w, out0, B, T_in, T_out, M, C0, taps, x1=None, C1=0, in_stride=1, ldw=None,
              in_relu=False, epilogue=EPI_STORE, out_relu=False, M0=0, out_tstride=1, out_toffset=0,
              T_store=0, cond=None, cond_T=0, cond_bstride=0, w_tap_stride=0, bias=None, scale=None,
              shift=None, aux0=None, aux1=None, out1=None, save0=None, save1=None, tile=0, split_k=0):
    """vqw_conv_gemm (include/vqwave.h).  `taps` = list of input shifts per tap."""
    ldw = M if ldw is None else ldw
    Ts = T_store if T_store > 0 else out_tstride * T_out
    ntaps = len(taps)
    wts = w_tap_stride if w_tap_stride > 0 else (C0 + C1) * ldw
    _need(x0, B * C0 * T_in, 'x0')
    if C1:
        _need(x1, B * C1 * T_in, 'x1')
    _need(w, (ntaps - 1) * wts + (C0 + C1 - 1) * ldw + M, 'w')
    H = M // 2 if epilogue == EPI_GATE else M
    if epilogue == EPI_GATE:
        _need(out0, B * H * Ts, 'out0')
        for t, nm in ((save0, 'save0'), (save1, 'save1')):
            if t is not None:
                _need(t, B * H * Ts, nm)
        if bias is not None:
            _need(bias, M, 'bias')
    elif epilogue == EPI_GATE_BWD:
        _need(out0, B * 2 * M * Ts, 'out0')
        _need(aux0, B * M * Ts, 'aux0')
        _need(aux1, B * M * Ts, 'aux1')
    elif epilogue == EPI_MASK:
        _need(out0, B * M * Ts, 'out0')
        _need(aux0, B * M * Ts, 'aux0')
    elif epilogue == EPI_ACCUM_SPLIT:
        m0 = min(max(M0, 0), M)
        if m0:
            _need(out0, B * m0 * Ts, 'out0')
        if m0 < M:
            _need(out1, B * (M - m0) * Ts, 'out1')
            _need(aux1, B * (M - m0) * Ts, 'aux1')
    else:
        m0 = M0 if (0 < M0 <= M and out1 is not None) else M
        _need(out0, B * m0 * Ts, 'out0')
        if m0 < M:
            _need(out1, B * (M - m0) * Ts, 'out1')
        if save0 is not None:
            _need(save0, B * M * Ts, 'save0')
        if aux1 is not None:
            _need(aux1, B * M * Ts, 'aux1')
    if bias is not None and epilogue != EPI_GATE:
        _need(bias, M, 'bias')
    if scale is not None:
        _need(scale, M, 'scale')
        if epilogue == EPI_STORE:
            _need(shift, M, 'shift')
    if cond_T:
        rows = M
        bs = cond_bstride if cond_bstride else rows * cond_T
        _need(cond, (B - 1) * bs + rows * cond_T, 'cond')
        cond_bstride = bs
    d = L.ConvDesc()
    d.B, d.T_out, d.T_in, d.M, d.C0, d.C1 = B, T_out, T_in, M, C0, C1
    d.ntaps, d.in_stride = ntaps, in_stride
    for j, s in enumerate(taps):
        d.tap_shift[j] = int(s)
    d.ldw, d.in_relu, d.epilogue, d.out_relu, d.M0 = ldw, int(in_relu), epilogue, int(out_relu), M0
    d.out_tstride, d.out_toffset, d.T_store = out_tstride, out_toffset, T_store
    d.cond_T, d.tile, d.cond_bstride, d.w_tap_stride = cond_T, tile, cond_bstride, w_tap_stride
    d.split_k = split_k
    for name, t in (('x0', x0), ('x1', x1), ('w', w), ('bias', bias), ('cond', cond), ('scale', scale),
                    ('shift', shift), ('aux0', aux0), ('aux1', aux1), ('out0', out0), ('out1', out1),
                    ('save0', save0), ('save1', save1)):
        setattr(d, name, None if t is None else t.data_ptr())
    L.check(L.lib().vqw_conv_gemm(C.byref(d), L.stream()))


def wgrad_gemm(*, p, q0, dw, B, T_q, T_p, Cp, Q0, taps, q1=None, Q1=0, p_stride=1, p_relu=False,
               lddw=None, dw_tap_stride=None, splits=0):
    """vqw_wgrad_gemm: dw[j][c][o] += sum_{b,t} p[b][c][p_stride*t+taps[j]] * q[b][o][t]."""
    lddw = (Q0 + Q1) if lddw is None else lddw
    dw_tap_stride = Cp * lddw if dw_tap_stride is None else dw_tap_stride
    _need(p, B * Cp * T_p, 'p')
    _need(q0, B * Q0 * T_q, 'q0')
    if Q1:
        _need(q1, B * Q1 * T_q, 'q1')
    _need(dw, (len(taps) - 1) * dw_tap_stride + (Cp - 1) * lddw + Q0 + Q1, 'dw')
    d = L.WgradDesc()
    d.B, d.T_q, d.T_p, d.Cp, d.Q0, d.Q1 = B, T_q, T_p, Cp, Q0, Q1
    d.ntaps, d.p_stride = len(taps), p_stride
    for j, s in enumerate(taps):
        d.tap_shift[j] = int(s)
    d.p_relu, d.lddw, d.splits, d.dw_tap_stride = int(p_relu), lddw, splits, dw_tap_stride
    d.p, d.q0, d.q1, d.dw = p.data_ptr(), q0.data_ptr(), (None if q1 is None else q1.data_ptr()), dw.data_ptr()
    L.check(L.lib().vqw_wgrad_gemm(C.byref(d), L.stream()))


def mu_law_encode_f32(x, out=None):
    L.require_cuda(x)
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().vqw_mu_law_encode_f32(L.ptr(x), L.ptr(out), x.numel(), L.stream()))
    return out


def mu_law_encode_i32(x, out=None):
    L.require_cuda(x)
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device) if out is None else out
    L.check(L.lib().vqw_mu_law_encode_i32(L.ptr(x), L.ptr(out), x.numel(), L.stream()))
    return out


def mu_law_decode_f32(idx, out=None):
    L.require_cuda(idx)
    out = torch.empty_like(idx) if out is None else out
    L.check(L.lib().vqw_mu_law_decode_f32(L.ptr(idx), L.ptr(out), idx.numel(), L.stream()))
    return out


def wavenet_inputs(x, inputs, labels):
    """x [B][T] -> inputs f32 [B][T] (mu-law of shift_right), labels int32 [B][T]."""
    B, T = x.shape
    L.require_cuda(x, inputs, labels)
    L.check(L.lib().vqw_wavenet_inputs(L.ptr(x), L.ptr(inputs), L.ptr(labels), B, T, L.stream()))


def conv_cin1_fwd(x, w, bias, out, *, k, stride, offset, relu=False, scale=None, shift=None, save_r=None):
    B, T_in = x.shape
    _, F, T_out = out.shape
    _need(w, k * F, 'w')
    L.require_cuda(x, out, bias, scale, shift, save_r)
    L.check(L.lib().vqw_conv_cin1_fwd(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(scale), L.ptr(shift), L.ptr(out),
                                      L.ptr(save_r), B, T_in, T_out, F, k, stride, offset, int(relu), L.stream()))


def conv_cin1_wgrad(x, dout, dw, *, k, stride, offset):
    B, T_in = x.shape
    _, F, T_out = dout.shape
    _need(dw, k * F, 'dw')
    L.require_cuda(x, dout)
    L.check(L.lib().vqw_conv_cin1_wgrad(L.ptr(x), L.ptr(dout), L.ptr(dw), B, T_in, T_out, F, k, stride, offset,
                                        L.stream()))


def rowsum(x, *, y=None, seg_out=None, total=None, alpha=1.0, seg=0):
    """x [B][C][T]: seg_out[b][c][t/seg] (optional), total[c] += alpha*sum (optional)."""
    B, Cc, T = x.shape
    L.require_cuda(x, y, seg_out, total)
    if y is not None and y.numel() != x.numel():
        raise ValueError('rowsum: y size mismatch')
    if seg_out is not None:
        _need(seg_out, B * Cc * (T // seg), 'seg_out')
    if total is not None:
        _need(total, Cc, 'total')
    L.check(L.lib().vqw_rowsum(L.ptr(x), L.ptr(y), L.ptr(seg_out), L.ptr(total), float(alpha), B, Cc, T, seg,
                               L.stream()))


def bn_relu_bwd(dx, r, scale, dz):
    B, Cc, T = dx.shape
    L.require_cuda(dx, r, scale, dz)
    _need(scale, Cc, 'scale')
    _need(dz, dx.numel(), 'dz')
    if r is not None:
        _need(r, dx.numel(), 'r')
    L.check(L.lib().vqw_bn_relu_bwd(L.ptr(dx), L.ptr(r), L.ptr(scale), L.ptr(dz), B, Cc, T, L.stream()))


def bn_relu_bwd_sums(dx, y, r, scale, dz, *, dscale=None, dbeta=None, dbias=None):
    """dz = dx * scale[c] * (r > 0) (r None: no mask) and, in the same pass, dscale[c] += sum dx * y, dbeta[c] += sum dx,
    dbias[c] += sum dz (each optional)."""
    B, Cc, T = dx.shape
    L.require_cuda(dx, scale, dz)
    _need(scale, Cc, 'scale')
    _need(dz, dx.numel(), 'dz')
    for name, t in (('y', y), ('r', r)):
        if t is not None:
            _need(t, dx.numel(), name)
    for name, t in (('dscale', dscale), ('dbeta', dbeta), ('dbias', dbias)):
        if t is not None:
            _need(t, Cc, name)
    L.check(L.lib().vqw_bn_relu_bwd_sums(L.ptr(dx), L.ptr(y), L.ptr(r), L.ptr(scale), L.ptr(dz), L.ptr(dscale), L.ptr(dbeta),
                                         L.ptr(dbias), B, Cc, T, L.stream()))


def relu_bn_fwd(x, r, scale, shift):
    """In place: r = relu(x) (optional), x = scale[c]*relu(x) + shift[c]."""
    B, Cc, T = x.shape
    L.require_cuda(x, r, scale, shift)
    if r is not None:
        _need(r, x.numel(), 'r')
    if scale is not None:
        _need(scale, Cc, 'scale')
        _need(shift, Cc, 'shift')
    L.check(L.lib().vqw_relu_bn_fwd(L.ptr(x), L.ptr(r), L.ptr(scale), L.ptr(shift), B, Cc, T, L.stream()))


# ---- the fp16x3 / bf16 plane engine (include/vqwave.h)
def _need_planes(t, n_halves, what):
    if t is None or t.dtype != torch.float16 or t.numel() < n_halves:
        raise ValueError('%s must be a float16 tensor with at least %d elements' % (what, n_halves))
    L.require_cuda(t)


X3_BF16, X3_HALF_BLOCKS, X3_S2D = 1, 2, 4      # VQW_X3_* mode bits of include/vqwave.h


def x3_mode(mode=None, bf16=False):
    """mode bits of the vqw_f16x3_* calls; None: fp16x3 (or bf16) with the block height chosen by VQW_X3_HALF (default: see
    model.py / DESIGN 3.3)."""
    if mode is not None:
        return mode
    import os
    return (X3_BF16 if bf16 else 0) | (X3_HALF_BLOCKS if os.environ.get('VQW_X3_HALF', DEFAULT_X3_HALF) == '1' else 0)


DEFAULT_X3_HALF = '0'


def _slot(t, what, dtype=torch.float32):
    """Device scalar (a 1-element view of the guard state) or None."""
    if t is None:
        return None
    if t.dtype != dtype or t.numel() < 1 or not t.is_cuda:
        raise ValueError('%s must be a %s device scalar' % (what, dtype))
    return t.data_ptr()


def f16x3_amax(x, amax, *, rows=None, cols=None, ld=None, mstride=0, count=1, flag=None):
    """amax[i] (int32 holding the bits of a non-negative float) = max(amax[i], max |x_i|), `count` strided matrices."""
    if rows is None:
        rows, cols, ld = 1, x.numel(), x.numel()
    _need(x, (count - 1) * mstride + (rows - 1) * ld + cols, 'x')
    if amax.dtype != torch.int32 or amax.numel() < count:
        raise ValueError('amax must be int32 [count]')
    L.require_cuda(x, amax)
    L.check(L.lib().vqw_f16x3_amax(L.ptr(x), rows, cols, ld, mstride, count, L.ptr(amax), _slot(flag, 'flag', torch.int32), L.stream()))


def f16x3_update_scales(amax, scale, *, target_exp, reset=True, flag=None, skip=None):
    """scale[i] = the power of two that puts amax[i] into [2^(target_exp-1), 2^target_exp) (slots that saw nothing keep their scale);
    reset: amax[i] = 0 afterwards; a non-finite amax raises `flag`; skip (device int32): non-zero when the kernel runs = nothing is touched."""
    n = amax.numel()
    if amax.dtype != torch.int32 or scale.dtype != torch.float32 or scale.numel() != n:
        raise ValueError('amax int32 [n], scale float32 [n]')
    L.require_cuda(amax, scale)
    L.check(L.lib().vqw_f16x3_update_scales_guarded(L.ptr(amax), L.ptr(scale), n, target_exp, int(reset), _slot(flag, 'flag', torch.int32),
                                                    _slot(skip, 'skip', torch.int32), L.stream()))


def f16x3_split_activations(x, planes, B, Cc, T, scale=1.0, kc0=0, KC=0, scale_dev=None, amax=None, flag=None, mode=None):
    """scale * scale_dev * x [B][C][T] fp32 -> planes [2][KC or C/8][B*T][8] fp16, chunks kc0..."""
    mode = x3_mode(mode)
    _need(x, B * Cc * T, 'x')
    _need_planes(planes, 2 * B * T * (KC * 8 if KC else Cc), 'planes')
    L.check(L.lib().vqw_f16x3_split_activations(L.ptr(x), L.ptr(planes), B, Cc, T, float(scale), kc0, KC, _slot(scale_dev, 'scale_dev'),
                                                _slot(amax, 'amax', torch.int32), _slot(flag, 'flag', torch.int32), mode, L.stream()))


def f16x3_pack_gate_weights(w, planes, ks, R, ldw, scale, count=1, scale_dev=None, mode=None):
    """`count` layers back to back in `w` ([count][ks][R][ldw]) and in `planes`."""
    mode = x3_mode(mode)
    _need(w, (count - 1) * ks * R * ldw + (ks * R - 1) * ldw + 2 * R, 'w')
    _need_planes(planes, count * 2 * ks * R * 2 * R, 'planes')
    L.check(L.lib().vqw_f16x3_pack_gate_weights(L.ptr(w), L.ptr(planes), ks, R, ldw, float(scale), count, _slot(scale_dev, 'scale_dev'), mode, L.stream()))


def f16x3_pack_weights(w, planes, Kd, M, ldw, scale, count=1, scale_dev=None, mode=None):
    """w [count][K][ldw] fp32 -> planes [count][2][K/8][M][8] fp16 of scale * scale_dev * w."""
    mode = x3_mode(mode)
    _need(w, (count - 1) * Kd * ldw + (Kd - 1) * ldw + M, 'w')
    _need_planes(planes, count * 2 * Kd * M, 'planes')
    L.check(L.lib().vqw_f16x3_pack_weights(L.ptr(w), L.ptr(planes), Kd, M, ldw, float(scale), count, _slot(scale_dev, 'scale_dev'), mode, L.stream()))


def f16x3_pack_weights_t(w, planes, Kd, M, k_inner, ld_src, blk_stride, scale, count=1, scale_dev=None, mode=None):
    """Planes of the matrix W'[k][m] = w[(k // k_inner) * blk_stride + m * ld_src + k % k_inner]: the kernel of an input-gradient GEMM
    straight from the forward kernel (no transposed fp32 copy); `count` matrices (Kd // k_inner) * blk_stride floats apart."""
    mode = x3_mode(mode)
    nblk = Kd // k_inner
    _need(w, (count - 1) * nblk * blk_stride + (nblk - 1) * blk_stride + (M - 1) * ld_src + k_inner, 'w')
    _need_planes(planes, count * 2 * Kd * M, 'planes')
    L.check(L.lib().vqw_f16x3_pack_weights_t(L.ptr(w), L.ptr(planes), Kd, M, k_inner, ld_src, blk_stride, float(scale), count,
                                             _slot(scale_dev, 'scale_dev'), mode, L.stream()))


def f16x3_out_conv(*, xp, wp, B, T, R, S, w_scale_inv, skip=None, net_in=None, net_out=None, bias=None,
                   net_out_planes=None, Cin=0, xp_kc0=0, xp_KC=0, ks=1, dilation=1, direction=1,
                   planes_kc0=0, planes_KC=0, plane_scale=0.0, epi=0, aux0=None, aux1=None, x_scale=None, w_scale=None,
                   out_scale=None, out_amax=None, flag=None, mode=None, cond=None, cond_T=0, cond_bstride=0, relu_planes=False,
                   aux0_is_gated=False, aux0_planes=None, aux0_KC=0, aux0_kc0=0):
    """vqw_f16x3_out_conv; epi 0: skip rows += / residual rows = net_in + W x + b; epi 1: gate backward; epi 2: the 1x1 convs
    around the stack: net_out = (aux0 > 0) * (net_in + W x + bias + cond), planes of net_out or relu(net_out)."""
    mode = x3_mode(mode)
    cin = Cin if Cin > 0 else R
    kc_all = xp_KC if xp_KC > 0 else cin // 8
    _need_planes(xp, 2 * kc_all * 8 * B * T, 'xp')
    _need_planes(wp, 2 * ks * cin * (S + R), 'wp')
    if S:
        _need(skip, B * S * T, 'skip')
    if R:
        if net_in is not None:
            _need(net_in, B * R * T, 'net_in')
        if not (epi == 1 and net_out is None and net_out_planes is not None):     # (gate backward may write planes only)
            _need(net_out, B * R * T * (2 if epi == 1 else 1), 'net_out')
    if epi == 1:
        if aux0_planes is not None:      # the gated planes of the forward pass instead of fp32 tanh / gated
            _need_planes(aux0_planes, (1 if mode & X3_BF16 else 2) * (aux0_KC or R // 8) * 8 * B * T, 'aux0_planes')
        else:
            _need(aux0, B * R * T, 'aux0')
        _need(aux1, B * R * T, 'aux1')
    if epi == 2 and aux0 is not None:
        _need(aux0, B * R * T, 'aux0')
    if bias is not None:
        _need(bias, S + R, 'bias')
    if net_out_planes is not None:
        _need_planes(net_out_planes, 2 * B * T * (planes_KC * 8 if planes_KC else R * (2 if epi == 1 else 1)), 'net_out_planes')
    d = L.F16x3OutDesc()
    d.xp, d.wp = xp.data_ptr(), wp.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    d.skip = None if skip is None else skip.data_ptr()
    d.net_in = None if net_in is None else net_in.data_ptr()
    d.net_out = None if net_out is None else net_out.data_ptr()
    d.net_out_planes = None if net_out_planes is None else net_out_planes.data_ptr()
    d.B, d.T, d.R, d.S = B, T, R, S
    d.Cin, d.xp_kc0, d.xp_KC = Cin, xp_kc0, xp_KC
    d.ks, d.dilation, d.dir = ks, dilation, direction
    d.planes_kc0, d.planes_KC, d.plane_scale, d.epi = planes_kc0, planes_KC, float(plane_scale), epi
    d.aux0 = None if aux0 is None else aux0.data_ptr()
    if aux0_planes is not None:
        d.aux0, d.aux0_KC, d.aux0_kc0 = aux0_planes.data_ptr(), aux0_KC, aux0_kc0
    d.aux1 = None if aux1 is None else aux1.data_ptr()
    d.w_scale_inv = float(w_scale_inv)
    d.x_scale, d.w_scale, d.out_scale = _slot(x_scale, 'x_scale'), _slot(w_scale, 'w_scale'), _slot(out_scale, 'out_scale')
    d.out_amax, d.flag = _slot(out_amax, 'out_amax', torch.int32), _slot(flag, 'flag', torch.int32)
    d.mode = mode
    if cond is not None:
        _need(cond, (B - 1) * cond_bstride + R * cond_T, 'cond')
        d.cond, d.cond_T, d.cond_bstride = cond.data_ptr(), cond_T, cond_bstride
    d.flags = (1 if relu_planes else 0) | (2 if aux0_is_gated else 0) | (4 if aux0_planes is not None else 0)
    L.check(L.lib().vqw_f16x3_out_conv(C.byref(d), L.stream()))


def f16x3_strided_conv(*, xp, wp, out, B, T, Cin, M, ks, pad_left, w_scale_inv=1.0, bias=None, bn_scale=None, bn_shift=None,
                       save_r=None, relu=False, dgrad=False, x_scale=None, w_scale=None, shape=0, split_slab=None,
                       split_counters=None, ksplit=0):
    """vqw_f16x3_strided_conv: a stride-2 conv (SAME padding, pad_left zeros in front) -> bias -> relu -> BatchNorm affine over
    space-to-depth planes of x [B][Cin][2T] (out [B][M][T]), or with dgrad its input gradient from the planes of dy [B][Cin][T]
    (out [B][M][2T]; wp = planes of the transposed kernel).  split_slab (fp32 scratch) + split_counters (int32, zero before the
    first launch): launches that would leave most of the chip idle cut the K steps of a tile over several blocks (ksplit 0 = auto,
    1 = off, n = n blocks per tile)."""
    _need_planes(xp, 2 * Cin * B * T * (1 if dgrad else 2), 'xp')
    _need_planes(wp, 2 * ks * Cin * M, 'wp')
    _need(out, B * M * T * (2 if dgrad else 1), 'out')
    for name, t in (('bias', bias), ('bn_scale', bn_scale), ('bn_shift', bn_shift)):
        if t is not None:
            _need(t, M, name)
    if save_r is not None:
        _need(save_r, B * M * T, 'save_r')
    d = L.F16x3SconvDesc()
    d.xp, d.wp, d.out = xp.data_ptr(), wp.data_ptr(), out.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    d.bn_scale = None if bn_scale is None else bn_scale.data_ptr()
    d.bn_shift = None if bn_shift is None else bn_shift.data_ptr()
    d.save_r = None if save_r is None else save_r.data_ptr()
    d.x_scale, d.w_scale, d.w_scale_inv = _slot(x_scale, 'x_scale'), _slot(w_scale, 'w_scale'), float(w_scale_inv)
    d.B, d.T, d.Cin, d.M, d.ks, d.pad_left, d.relu, d.dgrad = B, T, Cin, M, ks, pad_left, int(bool(relu)), int(bool(dgrad))
    d.shape = shape
    if split_slab is not None:
        if split_counters is None or split_counters.dtype != torch.int32 or split_slab.dtype != torch.float32:
            raise ValueError('f16x3_strided_conv: split_slab (float32) needs split_counters (int32)')
        d.split_slab, d.split_slab_floats = split_slab.data_ptr(), split_slab.numel()
        d.split_counters, d.split_counters_n = split_counters.data_ptr(), split_counters.numel()
    d.ksplit = ksplit
    L.check(L.lib().vqw_f16x3_strided_conv(C.byref(d), L.stream()))


def f16x3_gate_conv(*, xp, wp, out0, B, T, R, ks, dilation, w_scale_inv, bias=None, cond=None, cond_T=0,
                    cond_bstride=0, save0=None, save1=None, out_planes=None, out_planes_kc0=0, out_planes_KC=0, x_scale=None,
                    w_scale=None, mode=None):
    """vqw_f16x3_gate_conv: dilated causal conv over the layer input planes + bias + upsampled condition, tanh(filter) * sigmoid(gate)
    -> out0 (and as planes), tanh / sigmoid saved for the backward pass when save0 / save1 are given."""
    mode = x3_mode(mode)
    _need_planes(xp, 2 * B * R * T, 'xp')
    _need_planes(wp, 2 * ks * R * 2 * R, 'wp')
    if out0 is not None or out_planes is None or save1 is None:       # (out0 may be left out where every reader takes the planes)
        _need(out0, B * R * T, 'out0')
    for t, nm in ((save0, 'save0'), (save1, 'save1')):
        if t is not None:
            _need(t, B * R * T, nm)
    if bias is not None:
        _need(bias, 2 * R, 'bias')
    if cond is not None:
        bs = cond_bstride if cond_bstride else 2 * R * cond_T
        _need(cond, (B - 1) * bs + 2 * R * cond_T, 'cond')
        cond_bstride = bs
    d = L.F16x3GateDesc()
    d.xp, d.wp = xp.data_ptr(), wp.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    d.cond = None if cond is None else cond.data_ptr()
    d.out0 = None if out0 is None else out0.data_ptr()
    d.save0 = None if save0 is None else save0.data_ptr()
    d.save1 = None if save1 is None else save1.data_ptr()
    if out_planes is not None:
        _need_planes(out_planes, 2 * B * T * (out_planes_KC * 8 if out_planes_KC else R), 'out_planes')
    d.out_planes = None if out_planes is None else out_planes.data_ptr()
    d.out_planes_kc0, d.out_planes_KC = out_planes_kc0, out_planes_KC
    d.cond_bstride = cond_bstride
    d.B, d.T, d.R, d.ks, d.dilation, d.cond_T = B, T, R, ks, dilation, cond_T
    d.w_scale_inv = float(w_scale_inv)
    d.x_scale, d.w_scale = _slot(x_scale, 'x_scale'), _slot(w_scale, 'w_scale')
    d.mode = mode
    L.check(L.lib().vqw_f16x3_gate_conv(C.byref(d), L.stream()))


def _wgrad_desc(d, *, dw, slab, B, T, Cp, Q0, taps, p=None, q0=None, q1=None, Q1=0, lddw=None, dw_tap_stride=None, nsplit=0, p_scale=None,
                q0_scale=None, q1_scale=None, q_total=None, total_cols=None, q_seg=None, seg_T=0, seg_bstride=0, mode=None,
                p_stride=1, T_p=None, p_relu=False, q_planes=None, q_planes_KC=0, q_planes_kc0=0, q_planes_scale=0.0,
                p_planes=None, p_planes_KC=0, p_planes_kc0=0, p_planes_scale=0.0, p_tap_chunk=None):
    """q_planes: q0 as operand planes [planes][q_planes_KC or Q0/8 chunks][B*T][8] (scaled by q0_scale) instead of fp32."""
    mode = x3_mode(mode)
    T_p = T if T_p is None else T_p
    lddw = (Q0 + Q1) if lddw is None else lddw
    dw_tap_stride = Cp * lddw if dw_tap_stride is None else dw_tap_stride
    if p_planes is not None:
        _need_planes(p_planes, (1 if mode & X3_BF16 else 2) * (p_planes_KC or Cp // 8) * 8 * B * T, 'p_planes')
        d.p_planes, d.p_planes_KC, d.p_planes_kc0, d.p_planes_scale = p_planes.data_ptr(), p_planes_KC, p_planes_kc0, float(p_planes_scale)
        for j, kc in enumerate(p_tap_chunk or ()):       # (space-to-depth planes of a stride-2 conv's input: the tap's parity block)
            d.p_tap_chunk[j] = int(kc)
    else:
        _need(p, B * Cp * T_p, 'p')
    if q_planes is not None:
        _need_planes(q_planes, (1 if mode & X3_BF16 else 2) * (q_planes_KC or Q0 // 8) * 8 * B * T, 'q_planes')
        d.q_planes, d.q_planes_KC, d.q_planes_kc0, d.q_planes_scale = q_planes.data_ptr(), q_planes_KC, q_planes_kc0, float(q_planes_scale)
    else:
        _need(q0, B * Q0 * T, 'q0')
    if Q1:
        _need(q1, B * Q1 * T, 'q1')
    _need(dw, (len(taps) - 1) * dw_tap_stride + (Cp - 1) * lddw + Q0 + Q1, 'dw')
    _need(slab, 65536, 'slab')
    d.p_stride, d.Tp, d.p_relu = p_stride, T_p, int(bool(p_relu))
    d.p, d.q0, d.q1, d.dw, d.slab = (None if p is None else p.data_ptr()), (None if q0 is None else q0.data_ptr()), (None if q1 is None else q1.data_ptr()), dw.data_ptr(), slab.data_ptr()
    d.slab_floats = slab.numel()
    d.p_scale, d.q0_scale, d.q1_scale = _slot(p_scale, 'p_scale'), _slot(q0_scale, 'q0_scale'), _slot(q1_scale, 'q1_scale')
    d.B, d.T, d.Cp, d.Q0, d.Q1, d.ntaps = B, T, Cp, Q0, Q1, len(taps)
    for j, sh in enumerate(taps):
        d.tap_shift[j] = int(sh)
    d.lddw, d.nsplit, d.dw_tap_stride, d.mode = lddw, nsplit, dw_tap_stride, mode
    if q_total is not None:      # q_total[o] += sum_{b,t} q[b][o][t] for o in total_cols = (o0, o1)
        o0, o1 = total_cols if total_cols is not None else (0, Q0 + Q1)
        _need(q_total, o1, 'q_total')
        d.q_total, d.total_o0, d.total_o1 = q_total.data_ptr(), o0, o1
    if q_seg is not None:        # q_seg[b][o][t / (T / seg_T)] += q[b][o][t] (batch stride seg_bstride); zeroed by the caller
        seg_bstride = seg_bstride or (Q0 + Q1) * seg_T
        _need(q_seg, (B - 1) * seg_bstride + (Q0 + Q1) * seg_T, 'q_seg')
        d.q_seg, d.seg_T, d.seg_bstride = q_seg.data_ptr(), seg_T, seg_bstride


def f16x3_wgrad(**kw):
    """vqw_f16x3_wgrad: dw[j][c][o] += sum_{b,t} p[b][c][p_stride*t+taps[j]] * q[b][o][t] on the fp16x3 engine (slab = scratch;
    p_stride 2: p rows are T_p long, indices outside [0, T_p) are zero padding).  Arguments: _wgrad_desc."""
    kw.pop('xcd_group', None)
    d = L.F16x3WgradDesc()
    _wgrad_desc(d, **kw)
    L.check(L.lib().vqw_f16x3_wgrad(C.byref(d), L.stream()))


WGRAD_MAX_BATCH = 32


def f16x3_wgrad_batch(problems, **common):
    """vqw_f16x3_wgrad_batch: the weight gradients of several layers (one shape) in ONE launch.  `problems`: a list of dicts with
    what differs per layer (p, q0, q1, dw, taps, p_scale, q0_scale, q1_scale, q_total, q_seg); `common`: everything else."""
    n = len(problems)
    if not 1 <= n <= WGRAD_MAX_BATCH:
        raise ValueError('1..%d problems per launch (got %d)' % (WGRAD_MAX_BATCH, n))
    arr = (L.F16x3WgradDesc * n)()
    for d, pr in zip(arr, problems):
        _wgrad_desc(d, **dict(common, **pr))
    L.check(L.lib().vqw_f16x3_wgrad_batch(arr, n, L.stream()))


def mfcc(x, mel, out, *, n_keep=13):
    """x [B][T], mel [201][n_mel] -> out [B][C_out][ceil(T/160)] (channels >= n_keep zeroed)."""
    B, T = x.shape
    _, C_out, frames = out.shape
    L.require_cuda(x, mel, out)
    if mel.shape[0] != 201:
        raise ValueError('mel must be [201][n_mel]')
    L.check(L.lib().vqw_mfcc(L.ptr(x), L.ptr(mel), L.ptr(out), B, T, frames, mel.shape[1], n_keep, C_out, L.stream()))


def transpose(src, dst, batch, rows, cols):
    _need(src, batch * rows * cols, 'src')
    _need(dst, batch * rows * cols, 'dst')
    L.check(L.lib().vqw_transpose(L.ptr(src), L.ptr(dst), batch, rows, cols, L.stream()))


def vq_nearest_fwd(z_e, emb, *, idx, e_k=None, zq=None, zq_bstride=0, mind=None):
    B, D, Tz = z_e.shape
    K = emb.shape[0]
    L.require_cuda(z_e, emb, idx, e_k, zq, mind)
    if idx.dtype != torch.int64 or idx.numel() < B * Tz:
        raise ValueError('idx must be int64 [B][Tz]')
    if e_k is not None:
        _need(e_k, B * D * Tz, 'e_k')
    if zq is not None:
        zq_bstride = zq_bstride or D * Tz
        _need(zq, (B - 1) * zq_bstride + D * Tz, 'zq')
    L.check(L.lib().vqw_vq_nearest_fwd(L.ptr(z_e), L.ptr(emb), L.ptr(idx), L.ptr(e_k), L.ptr(zq), zq_bstride,
                                       L.ptr(mind), B, D, Tz, K, L.stream()))


def vq_nearest_bwd(z_e, e_k, idx, *, dzq, dzq_bstride, dz_e, demb, cscale, escale, K):
    B, D, Tz = z_e.shape
    L.require_cuda(z_e, e_k, idx, dzq, dz_e, demb)
    if dzq is not None:
        _need(dzq, (B - 1) * dzq_bstride + D * Tz, 'dzq')
    if demb is not None:
        _need(demb, K * D, 'demb')
    L.check(L.lib().vqw_vq_nearest_bwd(L.ptr(z_e), L.ptr(e_k), L.ptr(idx), L.ptr(dzq), dzq_bstride, L.ptr(dz_e),
                                       L.ptr(demb), float(cscale), float(escale), B, D, Tz, K, L.stream()))


def _need_speakers(spk, table, Cs):
    if spk.dtype != torch.int64:
        raise ValueError('speaker ids must be int64, got %s' % spk.dtype)
    if table.dtype != torch.float32 or table.numel() % Cs != 0 or table.numel() < Cs:
        raise ValueError('speaker table must be float32 [n_speakers][%d]' % Cs)
    return table.numel() // Cs


def speaker_tile_fwd(table, spk, cond, *, cond_bstride, row0, Cs, Tz):
    """Ids outside the table read row 0 on the device (model.py:22: one_hot -> argmax); hosts that still hold the ids
    as Python ints reject them earlier (data.py, generate.py)."""
    B = spk.numel()
    L.require_cuda(table, spk, cond)
    n_spk = _need_speakers(spk, table, Cs)
    _need(cond, (B - 1) * cond_bstride + (row0 + Cs) * Tz, 'cond')
    L.check(L.lib().vqw_speaker_tile_fwd(L.ptr(table), L.ptr(spk), L.ptr(cond), cond_bstride, row0, B, Cs, Tz, n_spk,
                                         L.stream()))


def speaker_tile_bwd(dcond, spk, dtable, *, dcond_bstride, row0, Cs, Tz):
    B = spk.numel()
    L.require_cuda(dcond, spk, dtable)
    n_spk = _need_speakers(spk, dtable, Cs)
    _need(dcond, (B - 1) * dcond_bstride + (row0 + Cs) * Tz, 'dcond')
    L.check(L.lib().vqw_speaker_tile_bwd(L.ptr(dcond), dcond_bstride, row0, L.ptr(spk), L.ptr(dtable), B, Cs, Tz, n_spk,
                                         L.stream()))


def softmax_xent(logits, labels, *, loss_sum, dlogits=None, probs=None, grad_scale=1.0):
    B, Q, T = logits.shape
    L.require_cuda(logits, labels, loss_sum, dlogits, probs)
    if labels.dtype != torch.int32 or labels.numel() != B * T:
        raise ValueError('labels must be int32 [B][T]')
    L.check(L.lib().vqw_softmax_xent(L.ptr(logits), L.ptr(labels), L.ptr(dlogits), L.ptr(probs), L.ptr(loss_sum),
                                     float(grad_scale), B, Q, T, L.stream()))


def softmax_xent_fwd(logits, labels, *, loss_sum, probs=None):
    """vqw_softmax_xent_fwd: loss_sum[0] += sum of the cross-entropies (model.py:91-94); probs optional."""
    B, Q, T = logits.shape
    L.require_cuda(logits, labels, loss_sum, probs)
    if labels.dtype != torch.int32 or labels.numel() != B * T:
        raise ValueError('labels must be int32 [B][T]')
    L.check(L.lib().vqw_softmax_xent_fwd(L.ptr(logits), L.ptr(labels), L.ptr(probs), L.ptr(loss_sum), B, Q, T, L.stream()))


def softmax_xent_bwd(logits, labels, *, dlogits, grad_scale=1.0):
    """vqw_softmax_xent_bwd: dlogits = (softmax(logits) - onehot(labels)) * grad_scale (may alias logits)."""
    B, Q, T = logits.shape
    L.require_cuda(logits, labels, dlogits)
    if labels.dtype != torch.int32 or labels.numel() != B * T:
        raise ValueError('labels must be int32 [B][T]')
    _need(dlogits, B * Q * T, 'dlogits')
    L.check(L.lib().vqw_softmax_xent_bwd(L.ptr(logits), L.ptr(labels), L.ptr(dlogits), float(grad_scale), B, Q, T, L.stream()))


def cond_proj_fwd(cond, w, out, *, B, Cc, Mall, Tz):
    """out[b][m][t] = sum_c w[c][m] cond[b][c][t]: every add_condition projection of the decoder in one launch (wavenet_ops.py:93-101)."""
    _need(cond, B * Cc * Tz, 'cond')
    _need(w, Cc * Mall, 'w')
    _need(out, B * Mall * Tz, 'out')
    L.check(L.lib().vqw_cond_proj_fwd(L.ptr(cond), L.ptr(w), L.ptr(out), B, Cc, Mall, Tz, L.stream()))


def cond_proj_wgrad(cond, dce, dw, *, B, Cc, Mall, Tz):
    """dw[c][m] += sum_{b,t} cond[b][c][t] dce[b][m][t] (dw: the zeroed gradient buffer)."""
    _need(cond, B * Cc * Tz, 'cond')
    _need(dce, B * Mall * Tz, 'dce')
    _need(dw, Cc * Mall, 'dw')
    L.check(L.lib().vqw_cond_proj_wgrad(L.ptr(cond), L.ptr(dce), L.ptr(dw), B, Cc, Mall, Tz, L.stream()))


def cond_proj_dgrad_scratch(B, Cc, Mall, Tz):
    n = C.c_int64(0)
    L.check(L.lib().vqw_cond_proj_dgrad_scratch_floats(B, Cc, Mall, Tz, C.byref(n)))
    return int(n.value)


def cond_proj_dgrad(w, dce, dcond, scratch, *, B, Cc, Mall, Tz):
    """dcond[b][c][t] = sum_m w[c][m] dce[b][m][t]; scratch: cond_proj_dgrad_scratch(...) floats."""
    _need(w, Cc * Mall, 'w')
    _need(dce, B * Mall * Tz, 'dce')
    _need(dcond, B * Cc * Tz, 'dcond')
    _need(scratch, cond_proj_dgrad_scratch(B, Cc, Mall, Tz), 'scratch')
    L.check(L.lib().vqw_cond_proj_dgrad(L.ptr(w), L.ptr(dce), L.ptr(dcond), L.ptr(scratch), scratch.numel(), B, Cc, Mall, Tz, L.stream()))


def adam_ema_step(param, grad, m, v, ema, *, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, decay=0.999,
                  grad_scale=1.0, skip=None):
    """skip: device int32 read when the kernel runs; non-zero = the step changes nothing (vqw_adam_ema_step_guarded)."""
    n = param.numel()
    for t, nm in ((grad, 'grad'), (m, 'm'), (v, 'v'), (ema, 'ema')):
        _need(t, n, nm)
    if skip is not None and (skip.dtype != torch.int32 or skip.numel() < 1 or not skip.is_cuda):
        raise ValueError('adam_ema_step: skip must be a device int32')
    L.check(L.lib().vqw_adam_ema_step_guarded(L.ptr(param), L.ptr(grad), L.ptr(m), L.ptr(v), L.ptr(ema), n, float(lr_t),
                                              float(beta1), float(beta2), float(eps), float(decay), float(grad_scale),
                                              L.ptr(skip), L.stream()))

"""ctypes binding of libvqwave.so (include/vqwave.h).  No CPU fallback: a missing library
or a failing call raises."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', os.environ.get('VQW_LIB_NAME', 'libvqwave.so'))
MAX_TAPS = 8

EPI_STORE, EPI_ACCUM_SPLIT, EPI_GATE, EPI_GATE_BWD, EPI_MASK = range(5)

_fp = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [
        ('B', C.c_int32), ('T_out', C.c_int32), ('T_in', C.c_int32), ('M', C.c_int32),
        ('C0', C.c_int32), ('C1', C.c_int32), ('ntaps', C.c_int32), ('in_stride', C.c_int32),
        ('tap_shift', C.c_int32 * MAX_TAPS), ('ldw', C.c_int32), ('in_relu', C.c_int32),
        ('epilogue', C.c_int32), ('out_relu', C.c_int32), ('M0', C.c_int32),
        ('out_tstride', C.c_int32), ('out_toffset', C.c_int32), ('T_store', C.c_int32),
        ('cond_T', C.c_int32), ('tile', C.c_int32), ('split_k', C.c_int32), ('cond_bstride', C.c_int64),
        ('w_tap_stride', C.c_int64),
        ('x0', _fp), ('x1', _fp), ('w', _fp), ('bias', _fp), ('cond', _fp), ('scale', _fp),
        ('shift', _fp), ('aux0', _fp), ('aux1', _fp),
        ('out0', _fp), ('out1', _fp), ('save0', _fp), ('save1', _fp),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ('B', C.c_int32), ('T_q', C.c_int32), ('T_p', C.c_int32), ('Cp', C.c_int32),
        ('Q0', C.c_int32), ('Q1', C.c_int32), ('ntaps', C.c_int32), ('p_stride', C.c_int32),
        ('tap_shift', C.c_int32 * MAX_TAPS), ('p_relu', C.c_int32), ('lddw', C.c_int32),
        ('splits', C.c_int32), ('dw_tap_stride', C.c_int64),
        ('p', _fp), ('q0', _fp), ('q1', _fp), ('dw', _fp),
    ]


class ArWeights(C.Structure):
    _fields_ = [
        ('n_layers', C.c_int32), ('kernel_size', C.c_int32), ('R', C.c_int32), ('S', C.c_int32),
        ('Q', C.c_int32), ('Cc', C.c_int32), ('pre_k', C.c_int32),
        ('dilations', C.POINTER(C.c_int32)),
        ('pre_w', _fp), ('pre_b', _fp), ('skip0_w', _fp), ('skip0_b', _fp),
        ('gated_w', C.POINTER(_fp)), ('gated_b', C.POINTER(_fp)), ('cond_w', C.POINTER(_fp)),
        ('out_w', C.POINTER(_fp)), ('out_b', C.POINTER(_fp)),
        ('cond_ld', C.c_int32), ('out_ld', C.c_int32),
        ('post1_w', _fp), ('post1_b', _fp), ('post1_cond_w', _fp), ('post1_cond_ld', C.c_int32),
        ('post2_w', _fp), ('post2_b', _fp),
    ]


class F16x3GateDesc(C.Structure):
    _fields_ = [
        ('xp', _fp), ('wp', _fp), ('bias', _fp), ('cond', _fp), ('out0', _fp), ('save0', _fp), ('save1', _fp),
        ('out_planes', _fp), ('cond_bstride', C.c_int64),
        ('B', C.c_int32), ('T', C.c_int32), ('R', C.c_int32), ('ks', C.c_int32), ('dilation', C.c_int32),
        ('cond_T', C.c_int32), ('w_scale_inv', C.c_float), ('out_planes_kc0', C.c_int32), ('out_planes_KC', C.c_int32),
        ('x_scale', _fp), ('w_scale', _fp), ('mode', C.c_int32),
    ]


class F16x3OutDesc(C.Structure):
    _fields_ = [
        ('xp', _fp), ('wp', _fp), ('bias', _fp), ('skip', _fp), ('net_in', _fp), ('net_out', _fp),
        ('net_out_planes', _fp),
        ('B', C.c_int32), ('T', C.c_int32), ('R', C.c_int32), ('S', C.c_int32), ('w_scale_inv', C.c_float),
        ('Cin', C.c_int32), ('xp_kc0', C.c_int32), ('xp_KC', C.c_int32),
        ('ks', C.c_int32), ('dilation', C.c_int32), ('dir', C.c_int32),
        ('planes_kc0', C.c_int32), ('planes_KC', C.c_int32), ('plane_scale', C.c_float), ('epi', C.c_int32),
        ('aux0', _fp), ('aux1', _fp),
        ('x_scale', _fp), ('w_scale', _fp), ('out_scale', _fp), ('out_amax', _fp), ('flag', _fp), ('mode', C.c_int32),
        ('cond', _fp), ('cond_bstride', C.c_int64), ('cond_T', C.c_int32), ('flags', C.c_int32),
        ('aux0_KC', C.c_int32), ('aux0_kc0', C.c_int32),
    ]


class F16x3SconvDesc(C.Structure):
    _fields_ = [
        ('xp', _fp), ('wp', _fp), ('bias', _fp), ('bn_scale', _fp), ('bn_shift', _fp), ('out', _fp), ('save_r', _fp),
        ('x_scale', _fp), ('w_scale', _fp), ('w_scale_inv', C.c_float),
        ('B', C.c_int32), ('T', C.c_int32), ('Cin', C.c_int32), ('M', C.c_int32), ('ks', C.c_int32), ('pad_left', C.c_int32),
        ('relu', C.c_int32), ('dgrad', C.c_int32), ('shape', C.c_int32),
        ('split_slab', _fp), ('split_slab_floats', C.c_int64), ('split_counters', _fp), ('split_counters_n', C.c_int32),
        ('ksplit', C.c_int32),
    ]


class F16x3WgradDesc(C.Structure):
    _fields_ = [
        ('p', _fp), ('q0', _fp), ('q1', _fp), ('dw', _fp), ('slab', _fp), ('slab_floats', C.c_int64),
        ('p_scale', _fp), ('q0_scale', _fp), ('q1_scale', _fp),
        ('B', C.c_int32), ('T', C.c_int32), ('Cp', C.c_int32), ('Q0', C.c_int32), ('Q1', C.c_int32), ('ntaps', C.c_int32),
        ('tap_shift', C.c_int32 * MAX_TAPS), ('lddw', C.c_int32), ('nsplit', C.c_int32), ('dw_tap_stride', C.c_int64),
        ('q_total', _fp), ('q_seg', _fp), ('seg_bstride', C.c_int64), ('seg_T', C.c_int32), ('total_o0', C.c_int32),
        ('total_o1', C.c_int32), ('mode', C.c_int32), ('p_relu', C.c_int32), ('p_stride', C.c_int32), ('Tp', C.c_int32), ('xcd_group', C.c_int32),
        ('q_planes', _fp), ('q_planes_KC', C.c_int32), ('q_planes_kc0', C.c_int32), ('q_planes_scale', C.c_float),
        ('p_planes', _fp), ('p_planes_KC', C.c_int32), ('p_planes_kc0', C.c_int32), ('p_planes_scale', C.c_float),
        ('p_tap_chunk', C.c_int32 * MAX_TAPS),
    ]


_i, _f, _sz, _i64 = C.c_int, C.c_float, C.c_size_t, C.c_int64
SIGNATURES = {
    'vqw_last_error': (C.c_char_p, []),
    'vqw_abi_version': (_i, []),
    'vqw_mu_law_encode_f32': (_i, [_fp, _fp, _sz, _fp]),
    'vqw_mu_law_encode_i32': (_i, [_fp, _fp, _sz, _fp]),
    'vqw_mu_law_decode_f32': (_i, [_fp, _fp, _sz, _fp]),
    'vqw_wavenet_inputs': (_i, [_fp, _fp, _fp, _i, _i, _fp]),
    'vqw_conv_gemm': (_i, [C.POINTER(ConvDesc), _fp]),
    'vqw_wgrad_gemm': (_i, [C.POINTER(WgradDesc), _fp]),
    'vqw_causal_conv1d_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_causal_conv1d_dgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_causal_conv1d_wgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_conv1d_same_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_conv1d_same_dgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_conv1d_same_wgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_pointwise_gemm_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_pointwise_gemm_dgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'vqw_pointwise_gemm_wgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _fp]),
    'vqw_conv_cin1_fwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_conv_cin1_wgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_rowsum': (_i, [_fp, _fp, _fp, _fp, _f, _i, _i, _i, _i, _fp]),
    'vqw_bn_relu_bwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _fp]),
    'vqw_bn_relu_bwd_sums': (_i, [_fp] * 8 + [_i, _i, _i, _fp]),
    'vqw_relu_bn_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _fp]),
    'vqw_mfcc': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'vqw_transpose': (_i, [_fp, _fp, _i, _i, _i, _fp]),
    'vqw_vq_nearest_fwd': (_i, [_fp, _fp, _fp, _fp, _fp, _i64, _fp, _i, _i, _i, _i, _fp]),
    'vqw_vq_nearest_bwd': (_i, [_fp, _fp, _fp, _fp, _i64, _fp, _fp, _f, _f, _i, _i, _i, _i, _fp]),
    'vqw_speaker_tile_fwd': (_i, [_fp, _fp, _fp, _i64, _i, _i, _i, _i, _i, _fp]),
    'vqw_speaker_tile_bwd': (_i, [_fp, _i64, _i, _fp, _fp, _i, _i, _i, _i, _fp]),
    'vqw_softmax_xent': (_i, [_fp, _fp, _fp, _fp, _fp, _f, _i, _i, _i, _fp]),
    'vqw_softmax_xent_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _fp]),
    'vqw_softmax_xent_bwd': (_i, [_fp, _fp, _fp, _f, _i, _i, _i, _fp]),
    'vqw_cond_proj_fwd': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'vqw_cond_proj_wgrad': (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'vqw_cond_proj_dgrad': (_i, [_fp, _fp, _fp, _fp, C.c_int64, _i, _i, _i, _i, _fp]),
    'vqw_cond_proj_dgrad_scratch_floats': (_i, [_i, _i, _i, _i, C.POINTER(C.c_int64)]),
    'vqw_adam_ema_step': (_i, [_fp, _fp, _fp, _fp, _fp, _sz, _f, _f, _f, _f, _f, _f, _fp]),
    'vqw_adam_ema_step_guarded': (_i, [_fp, _fp, _fp, _fp, _fp, _sz, _f, _f, _f, _f, _f, _f, _fp, _fp]),
    'vqw_ar_decode_create': (_i, [C.POINTER(_fp), C.POINTER(ArWeights), _i]),
    'vqw_ar_decode_create_ex': (_i, [C.POINTER(_fp), C.POINTER(ArWeights), _i, _i]),
    'vqw_ar_decode_reset': (_i, [_fp, _fp]),
    'vqw_ar_decode_run': (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    'vqw_ar_decode_run_async': (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp]),
    'vqw_ar_decode_wait': (_i, [_fp]),
    'vqw_ar_decode_run_group_async': (_i, [C.POINTER(_fp), _i, C.POINTER(_fp), _i, _i, _i, _i, C.POINTER(_fp), C.POINTER(_fp),
                                           C.POINTER(_fp), C.POINTER(_fp), _fp]),
    'vqw_ar_decode_workgroups': (_i, [_fp]),
    'vqw_ar_decode_destroy': (_i, [_fp]),
    'vqw_f16x3_amax': (_i, [_fp, _i64, _i, _i64, _i64, _i, _fp, _fp, _fp]),
    'vqw_f16x3_update_scales': (_i, [_fp, _fp, _i, _i, _i, _fp, _fp]),
    'vqw_f16x3_update_scales_guarded': (_i, [_fp, _fp, _i, _i, _i, _fp, _fp, _fp]),
    'vqw_f16x3_split_activations': (_i, [_fp, _fp, _i, _i, _i, _f, _i, _i, _fp, _fp, _fp, _i, _fp]),
    'vqw_f16x3_pack_gate_weights': (_i, [_fp, _fp, _i, _i, _i, _f, _i, _fp, _i, _fp]),
    'vqw_f16x3_gate_conv': (_i, [C.POINTER(F16x3GateDesc), _fp]),
    'vqw_f16x3_pack_weights': (_i, [_fp, _fp, _i, _i, _i, _f, _i, _fp, _i, _fp]),
    'vqw_f16x3_pack_weights_t': (_i, [_fp, _fp, _i, _i, _i, _i, C.c_int64, _f, _i, _fp, _i, _fp]),
    'vqw_f16x3_out_conv': (_i, [C.POINTER(F16x3OutDesc), _fp]),
    'vqw_f16x3_wgrad': (_i, [C.POINTER(F16x3WgradDesc), _fp]),
    'vqw_f16x3_wgrad_batch': (_i, [C.POINTER(F16x3WgradDesc), _i, _fp]),
    'vqw_f16x3_strided_conv': (_i, [C.POINTER(F16x3SconvDesc), _fp]),
}

_lib = None


def lib():
    """Load libvqwave.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libvqwave.so not found at %s -- build it with `python vq-vae-wavenet_amd/build.py` '
                '(there is no CPU fallback)' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError('libvqwave: ' + lib().vqw_last_error().decode())


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    """hipStream_t of torch's current stream."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError('libvqwave ops need contiguous tensors on the GPU '
                             '(got device=%s contiguous=%s)' % (t.device, t.is_contiguous()))

"""VQ-VAE-WaveNet on MI355X: host-side mirror of the reference's model assembly.

Mirrors model.py:7-159 (class VQVAE), Encoder/encoder.py:8-26 (Encoder_64),
Decoder/decoder.py:12-62 (WavenetDecoder) and Decoder/WaveNet/wavenet.py:24-172 (Wavenet) of
the reference.  The reference builds a TF graph and lets TF autodiff + the TF runtime do the
work; here one training step is an explicit forward + backward sequence of libvqwave kernels
on (batch, channel, time) fp32 tensors.  PyTorch only owns device memory and streams.

Parameters live in ONE flat fp32 buffer (so gradients are one contiguous all-reduce payload
and Adam+EMA is one kernel); `named_parameters()` exposes them under the reference's TF
variable names and shapes.
"""
import json
import math
import os
import time
import sys
from collections import OrderedDict

import torch

from . import _alloc as A
from . import kernels as K
from .encoders import Encoder2019, EncoderMagenta

BN_EPS = 1e-3  # Keras BatchNormalization default epsilon
DEFAULT_ENGINE = 'f16x3'


def load_configs(model_json='model_parameters.json', wavenet_json=None):
    """Same two JSON files as the reference (train.py:54-61, wavenet.py:10-13)."""
    with open(model_json) as f:
        m = json.load(f)
    with open(wavenet_json or m['wavenet_parameters']) as f:
        w = json.load(f)
    assert len(w['dilation_rates']) == w['num_cycles'] * w['num_cycle_layers']  # wavenet.py:13
    return m, w


def same_pads(n, k, s):
    """TF 'SAME' padding (left, right) -- SURVEY.md Appendix A-4."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def layer_scope(i, num_cycle_layers):
    return 'decoder/cycle_%d/layer_%d' % (1 + i // num_cycle_layers, 1 + i % num_cycle_layers)


def _suffix(i):
    return '' if i == 0 else '_%d' % i


class VQVAE:
    """model.py:7-159.  encoder '64' + VQ + speaker embedding + WaveNet decoder."""

    def __init__(self, model_cfg, wavenet_cfg, num_speakers, device='cuda', seed=0):
        self.enc = model_cfg.get('encoder', '64')
        if self.enc not in ('64', 'Magenta', '2019'):
            raise NotImplementedError('encoder %s not implemented' % self.enc)
        self.m, self.w = model_cfg, wavenet_cfg
        self.dev = torch.device(device)
        self.S_spk = num_speakers
        self.F = model_cfg.get('encoder_filters', 768)
        self.D = model_cfg['latent_dim']
        self.Kc = model_cfg['k']
        self.Cs = model_cfg['speaker_embedding']
        self.beta = float(model_cfg['beta'])
        self.use_vq = bool(model_cfg.get('use_vq', True))     # false: z_q = e_k = z_e, reconstruction loss only (model.py:139-141)
        # speaker_embedding = 0: the one-hot speaker vector itself is the global condition (model.py:19-27 leaves self.h
        # as [B, 1, num_speakers]); its width is padded to a multiple of 16 channels for the conv engine (the extra
        # condition rows are always zero, their kernel rows never receive a gradient)
        self.spk_table = self.Cs > 0
        self.Cs_eff = self.Cs if self.spk_table else (num_speakers + 15) // 16 * 16
        self.Cc_ref = self.D + (self.Cs if self.spk_table else num_speakers)      # the reference's condition width
        self.Cc = self.D + self.Cs_eff
        self.magenta = {'Magenta': EncoderMagenta, '2019': Encoder2019}[self.enc](self.D) if self.enc != '64' else None
        w = wavenet_cfg
        self.dil = list(w['dilation_rates'])
        self.L = len(self.dil)
        self.ks = w['kernel_size']
        self.R, self.S, self.Q = w['residual_filters'], w['skip_filters'], w['quantization_channels']
        if w['dilation_filters'] != self.R or w['preprocess']['filters'] != self.R:
            raise NotImplementedError('dilation_filters / preprocess filters must equal residual_filters '
                                      '(the reference adds them: wavenet.py:73)')
        self.pre_k = w['preprocess']['kernel_size']
        self.Mall = self.L * 2 * self.R + self.S  # all local-condition projections side by side
        self.receptive_field = sum(self.dil) * (self.ks - 1) + 1 + self.pre_k - 1  # wavenet.py:16-17
        self.schedule = [(int(k), float(v)) for k, v in model_cfg['learning_rate_schedule'].items()]
        self.global_step = 0
        self.grad_sync = None   # parallel.GradAllReduce when training data-parallel
        self.overlap_wgrad = os.environ.get('VQW_OVERLAP', '1') != '0'   # decoder backward on two streams
        # Engine of the decoder's contractions.  VQW_ENGINE=f16x3: fp32 operands as two fp16 planes, three
        # MFMA terms on the fp16 matrix pipe, fp32 accumulate, with device-side range guards (per-tensor power-of-two
        # scales from measured max-abs values; a step whose planes would leave fp16's range is repeated on the fp32
        # engine).  VQW_ENGINE=fp32: the fp32-MFMA engine everywhere.  VQW_GATE_F16X3=1..5 (development ladder, fixed
        # scales, no guards): 1 gate convs; 2 + the 1x1 skip/residual convs; 3 skip path as ONE contraction; 4 + the gate
        # convs' input gradient; 5 + gate backward.
        engine = os.environ.get('VQW_ENGINE', DEFAULT_ENGINE)
        if engine not in ('fp32', 'f16x3'):
            raise ValueError("VQW_ENGINE must be 'fp32' or 'f16x3' (got %r)" % engine)
        ladder = os.environ.get('VQW_GATE_F16X3', '0')
        # VQW_DTYPE=bf16 (BASELINE.json configs[4]: bf16 storage + fp32 accumulate): the same kernels with ONE bf16 plane per
        # operand and one bf16 MFMA per product; master weights, optimiser state and the residual stream stay fp32
        self.bf16 = os.environ.get('VQW_DTYPE', model_cfg.get('dtype', 'f32')) == 'bf16'
        # mode bits of the plane engine: bf16, and the block height of its conv kernels per call site -- 128-row blocks (two
        # per CU) for the forward gate conv + residual 1x1 (decoder forward loop 6.05 vs 7.13 ms, tools/x3_chain.py) and for
        # gate backward, 256-row blocks for the K = 7680 skip contraction and the input gradient (whole step, same box:
        # 34.1 ms against 35.0 with 256-row blocks everywhere): VQW_X3_HALF, _SKIP, _BWD, _DGRAD
        self.x3_mode = K.X3_BF16 if self.bf16 else 0
        half = lambda name, dflt: self.x3_mode | (K.X3_HALF_BLOCKS if os.environ.get(name, dflt) == '1' else 0)  # noqa: E731
        self.x3_mode_fwd = half('VQW_X3_HALF', '1')                                        # gate conv + residual 1x1
        self.x3_mode_skip = half('VQW_X3_HALF_SKIP', '0')                                  # the all-layers skip contraction
        self.x3_mode_bwd = half('VQW_X3_HALF_BWD', '1')                                    # gate backward
        self.x3_mode_dgrad = half('VQW_X3_HALF_DGRAD', '0')                                # input gradient
        self.x3_mode_head = half('VQW_X3_HALF_HEAD', '1')                                  # postprocess1 / 2 and their input gradients
        self.x3_guard = engine == 'f16x3' and ladder == '0' and not self.bf16
        self.x3_all = self.x3_guard or self.bf16          # the plane engine carries every decoder contraction, or none
        if self.x3_all:
            ladder = '5'
        self.gate_f16x3 = ladder in ('1', '2', '3', '4', '5')
        self.out_f16x3 = ladder in ('2', '3', '4', '5')
        self.skip_f16x3 = ladder in ('3', '4', '5')
        self.dgrad_f16x3 = ladder in ('4', '5')
        self.gbwd_f16x3 = ladder == '5'
        self.wg_planes = os.environ.get('VQW_WGRAD_PP', '1') != '0'      # weight gradients read p from operand planes too
        self._x3_active = True         # False while a step is being repeated on the fp32 engine
        self._x3_warned = set()        # (B, T) shapes already reported as running on the fp32 engine
        self.x3_fallbacks = 0          # steps repeated on the fp32 engine because a plane left fp16's range
        self.x3_steps = 0              # steps that ran on the fp16x3 engine
        self._side = None
        self._build_layout()
        self._init_params(seed)
        self._ws = {}
        self.loss_buf = torch.zeros(4, device=self.dev)  # [CE sum, sum of min distances, -, -]
        # per-call-site tile choices of the conv engine (10*MT+NT; 0 = library heuristic), from
        # tools/bench_kernels.py on MI355X at B=8, T=6656: block counts are 13*2^k, so the tile
        # that balances best over 256 CUs differs per GEMM shape
        self.tiles = {'out': 12, 'gate_bwd': 12, 'dgrad': 12}
        # guard state of the fp16x3 engine: one power-of-two scale + one max-abs collector per tensor kind
        #   WG all gate kernels | WO all 1x1 skip/residual kernels | G = dskip and every dnet (they share one contraction)
        #   X[l] input planes of layer l (l = 0..L) | DP[l] d pre-activation of layer l
        L = self.L
        #   SK relu(skip) planes | H1 relu(postprocess1) planes | DH d postprocess1 | WH the three kernels around the stack
        n = 3 + 2 * L + 1
        self.SL = {'WG': 0, 'WO': 1, 'G': 2, 'X': 3, 'DP': 3 + L + 1, 'SK': n, 'H1': n + 1, 'DH': n + 2, 'WH': n + 3, 'N': n + 4}
        self.x3_scale = torch.ones(self.SL['N'], device=self.dev)
        self.x3_scale[self.SL['G']] = 2.0 ** 20          # first step: |d loss / d logits| <= 1 / (B T)
        self.x3_scale[self.SL['DP']:self.SL['SK']] = 2.0 ** 20
        self.x3_scale[self.SL['DH']] = 2.0 ** 20
        self.head_x3 = os.environ.get('VQW_HEAD_X3', '1') != '0'
        self.x3_amax = torch.zeros(self.SL['N'], dtype=torch.int32, device=self.dev)
        self.x3_flag = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.x3_void = torch.zeros(1, dtype=torch.int32, device=self.dev)      # deferred mode: a flagged step is waiting to be repeated
        self.defer_guard = os.environ.get('VQW_DEFER_GUARD', '0') == '1'       # read the range flag one step late (train_step)
        self._pending, self._void_host, self._void_slot, self._in_step = [], None, 0, False
        # the condition projections (K = Cc ~ 80, rows of Tz ~ 104 frames) on their own fp32-MFMA kernels (csrc/cond_proj.hip)
        self.cond_proj = self.Cc <= 128 and self.Mall % 4 == 0 and os.environ.get('VQW_COND_PROJ', '1') != '0'
        self.host_enqueue_ms = None

    # ------------------------------------------------------------------ parameter layout
    def _build_layout(self):
        F, D, R, S, Q, L, ks, Cc = self.F, self.D, self.R, self.S, self.Q, self.L, self.ks, self.Cc
        seg = OrderedDict()  # internal (grouped) tensors of the flat buffer
        if self.spk_table:
            seg['speaker_embedding'] = (self.S_spk, self.Cs)
        if self.enc == '64':
            seg['enc_w0'] = (5, F)                 # conv1d/kernel [5,1,F]
            seg['enc_w'] = (5, 5, F, F)            # conv1d_1..5/kernel
            seg['enc_b'] = (6, F)
            seg['enc_w6'] = (F, D)                 # conv1d_6/kernel [1,F,D]
            seg['enc_b6'] = (D,)
            seg['bn_gamma'] = (6 * F + D,)
            seg['bn_beta'] = (6 * F + D,)
        else:
            seg.update(self.magenta.segments())
        seg['embedding'] = (self.Kc, D)
        seg['pre_w'] = (self.pre_k, R)
        seg['pre_b'] = (R,)
        seg['skip0_w'] = (R, S)
        seg['skip0_b'] = (S,)
        seg['gated_w'] = (L, ks, R, 2 * R)
        seg['gated_b'] = (L, 2 * R)
        seg['cond_w'] = (Cc, self.Mall)        # [layer0 | layer1 | ... | postprocess1]
        seg['out_w'] = (L, R, S + R)           # skip | residual 1x1 kernels side by side
        seg['out_b'] = (L, S + R)
        seg['post1_w'] = (S, S)
        seg['post1_b'] = (S,)
        seg['post2_w'] = (S, Q)
        seg['post2_b'] = (Q,)
        off = 0
        self.seg_off = {}
        for n, shp in seg.items():
            self.seg_off[n] = (off, shp)
            off += (math.prod(shp) + 3) // 4 * 4  # keep every segment 16-byte aligned
        self.n_flat = off
        self.seg_shapes = seg

    def _views(self, flat):
        return {n: flat[o:o + math.prod(shp)].view(shp) for n, (o, shp) in self.seg_off.items()}

    def _init_params(self, seed):
        dev = self.dev
        self.flat = torch.zeros(self.n_flat, device=dev)
        self.grad = torch.zeros(self.n_flat, device=dev)
        self.adam_m = torch.zeros(self.n_flat, device=dev)
        self.adam_v = torch.zeros(self.n_flat, device=dev)
        self.P = self._views(self.flat)
        self.G = self._views(self.grad)
        self.bn_mean = torch.zeros(6 * self.F + self.D, device=dev)   # moving stats (never updated:
        self.bn_var = torch.ones(6 * self.F + self.D, device=dev)     #  SURVEY.md Appendix A-3)
        g = torch.Generator().manual_seed(seed)

        def uus(shape, fan, factor):   # tf.uniform_unit_scaling_initializer
            lim = math.sqrt(3.0 / fan) * factor
            return (torch.rand(shape, generator=g) * 2 - 1) * lim

        def glorot(shape, k, cin, cout):  # Keras glorot_uniform
            lim = math.sqrt(6.0 / (k * cin + k * cout))
            return (torch.rand(shape, generator=g) * 2 - 1) * lim

        F, D, R, S, Q, L, ks, Cc = self.F, self.D, self.R, self.S, self.Q, self.L, self.ks, self.Cc
        P = self.P
        if self.spk_table:
            P['speaker_embedding'].copy_(uus((self.S_spk, self.Cs), self.S_spk, 2.0))   # model.py:23-26
        else:
            self.onehot = torch.eye(self.S_spk, self.Cs_eff, device=dev)              # row s = one_hot(s), zero padded
        if self.enc == '64':
            P['enc_w0'].copy_(glorot((5, F), 5, 1, F))
            P['enc_w'].copy_(glorot((5, 5, F, F), 5, F, F))
            P['enc_w6'].copy_(glorot((F, D), 1, F, D))
            P['bn_gamma'].fill_(1.0)
        else:
            self.magenta.init(P, uus if self.enc == 'Magenta' else glorot)
        P['embedding'].copy_(uus((self.Kc, D), self.Kc, 1.7))                       # model.py:47-49
        P['pre_w'].copy_(uus((self.pre_k, R), self.pre_k, 1.0))                     # wavenet_ops.py:69
        P['skip0_w'].copy_(uus((R, S), R, 1.0))
        P['gated_w'].copy_(uus((L, ks, R, 2 * R), ks * R, 1.0))
        P['cond_w'].copy_(uus((Cc, self.Mall), self.Cc_ref, 1.0))
        P['out_w'].copy_(uus((L, R, S + R), R, 1.0))
        P['post1_w'].copy_(uus((S, S), S, 1.0))
        P['post2_w'].copy_(uus((S, Q), S, 1.0))
        self.ema = self.flat.clone()     # ExponentialMovingAverage shadows start at the variables
        self.E = self._views(self.ema)
        # scratch: per-tap transposed kernels for the input-gradient GEMMs
        self._poison_T = []
        with A.record(self._poison_T):
            self._alloc_scratch()

    def _alloc_scratch(self):
        dev = self.dev
        F, D, R, S, Q, L, ks, Cc = self.F, self.D, self.R, self.S, self.Q, self.L, self.ks, self.Cc
        self.T = {
            'gated_w': A.empty(L, ks, 2 * R, R, device=dev),
            'out_w': A.empty(L, S + R, R, device=dev),
            'post1_w': A.empty(S, S, device=dev),
            'post2_w': A.empty(Q, S, device=dev),
            'skip0_w': A.empty(S, R, device=dev),
            'cond_w': A.empty(self.Mall, Cc, device=dev),
        }
        if self.enc == '64':
            self.T['enc_w'] = A.empty(5, 5, F, F, device=dev)
            self.T['enc_w6'] = A.empty(D, F, device=dev)
        else:
            self.T.update(self.magenta.scratch(dev))

    # ------------------------------------------------------------------ reference-name views
    def _named(self, V, bn_stats=True):
        """Reference TF variable name -> tensor (copy) with the reference shape."""
        F, D, R, S, L = self.F, self.D, self.R, self.S, self.L
        out = OrderedDict()
        if self.spk_table:
            out['speaker_embedding'] = V['speaker_embedding']
        if self.enc == '64':
            out['encoder/conv1d/kernel'] = V['enc_w0'].reshape(5, 1, F)
            for i in range(1, 6):
                out['encoder/conv1d_%d/kernel' % i] = V['enc_w'][i - 1]
            for i in range(6):
                out['encoder/conv1d%s/bias' % _suffix(i)] = V['enc_b'][i]
            out['encoder/conv1d_6/kernel'] = V['enc_w6'].reshape(1, F, D)
            out['encoder/conv1d_6/bias'] = V['enc_b6']
            for i in range(7):
                n = F if i < 6 else D
                sl = slice(i * F, i * F + n)
                out['encoder/batch_normalization%s/gamma' % _suffix(i)] = V['bn_gamma'][sl]
                out['encoder/batch_normalization%s/beta' % _suffix(i)] = V['bn_beta'][sl]
                if bn_stats:
                    out['encoder/batch_normalization%s/moving_mean' % _suffix(i)] = self.bn_mean[sl]
                    out['encoder/batch_normalization%s/moving_variance' % _suffix(i)] = self.bn_var[sl]
        else:
            self.magenta.named(V, out)
        if self.use_vq:          # (the reference only creates the codebook under use_vq, model.py:137-138)
            out['embedding/embedding'] = V['embedding']
        out['decoder/preprocess/kernel'] = V['pre_w'].reshape(self.pre_k, 1, R)
        out['decoder/preprocess/bias'] = V['pre_b']
        out['decoder/skip/kernel'] = V['skip0_w'].reshape(1, R, S)
        out['decoder/skip/bias'] = V['skip0_b']
        ncl = self.w['num_cycle_layers']
        for l in range(L):
            s = layer_scope(l, ncl)
            out[s + '/gated/kernel'] = V['gated_w'][l]
            out[s + '/gated/bias'] = V['gated_b'][l]
            out[s + '/gated/local_condition/kernel'] = V['cond_w'][:self.Cc_ref, l * 2 * R:(l + 1) * 2 * R].unsqueeze(0)
            out[s + '/skip/kernel'] = V['out_w'][l][:, :S].unsqueeze(0)
            out[s + '/skip/bias'] = V['out_b'][l][:S]
            out[s + '/residual/kernel'] = V['out_w'][l][:, S:].unsqueeze(0)
            out[s + '/residual/bias'] = V['out_b'][l][S:]
        out['decoder/postprocess1/kernel'] = V['post1_w'].reshape(1, S, S)
        out['decoder/postprocess1/bias'] = V['post1_b']
        out['decoder/postprocess1/local_condition/kernel'] = V['cond_w'][:self.Cc_ref, L * 2 * R:].unsqueeze(0)
        out['decoder/postprocess2/kernel'] = V['post2_w'].reshape(1, S, self.Q)
        out['decoder/postprocess2/bias'] = V['post2_b']
        return out

    def named_parameters(self, ema=False):
        """Copies of all variables under the reference's names (ema=True: the EMA shadows
        that generate.py:88-90 restores)."""
        self.finish_steps()
        return OrderedDict((k, v.detach().clone()) for k, v in self._named(self.E if ema else self.P).items())

    def named_gradients(self):
        self.finish_steps()
        return OrderedDict((k, v.detach().clone()) for k, v in self._named(self.G, bn_stats=False).items())

    def load_named(self, params, also_ema=True):
        """Load variables given under the reference's names (any device)."""
        self.finish_steps()
        dst = self._named(self.P)
        for name, view in dst.items():
            if name not in params:
                raise KeyError('missing variable ' + name)
            view.copy_(torch.as_tensor(params[name]).to(self.dev).reshape(view.shape))
        if also_ema:
            self.ema.copy_(self.flat)

    def use_ema_weights(self):
        """generate.py:88-90: restore the EMA shadow variables into the live variables."""
        self.flat.copy_(self.ema)

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, T, train=True):
        """Buffers of one (B, T) problem, cached.  train=False: only what the encoder + VQ forward pass writes
        (model.encoding of generate.py:92) -- about 3 KB per audio sample instead of the 136 KB per sample of the
        training workspace (saved decoder activations of 30 layers + backward buffers)."""
        key = (B, T, train)
        if key in self._ws:
            return self._ws[key]
        if not train and (B, T, True) in self._ws:
            return self._ws[(B, T, True)]
        rec = []
        with A.record(rec):
            ws = self._build_workspace(B, T, train)
        ws['_poison'] = rec          # (VQW_POISON=1: refilled with NaN at the start of every step)
        self._ws[key] = ws
        return ws

    def _build_workspace(self, B, T, train):
        if self.enc == '64':
            if T % 64 != 0:
                raise ValueError('length must be a multiple of 64 for Encoder_64 (got %d)' % T)
            Tz = T // 64
        else:
            Tz = self.magenta.latent_len(T)
        dev, F, D, R, S, Q, L = self.dev, self.F, self.D, self.R, self.S, self.Q, self.L
        e = lambda *s: A.empty(*s, device=dev)  # noqa: E731
        ws = {'B': B, 'T': T, 'Tz': Tz, 'ratio': T // Tz}
        ws['Tl'] = [T // (2 ** (i + 1)) for i in range(6)]
        ws['inputs'] = e(B, T)
        ws['labels'] = A.empty(B, T, dtype=torch.int32, device=dev)
        if self.enc == '64':
            ws['X'] = [e(B, F, t) for t in ws['Tl']]      # BN outputs of encoder layers 0..5
            if self.x3_guard and not self.bf16 and F % 256 == 0:      # fp16x3 engine for layers 1..3 (_enc_x3_layers)
                ws['eplanes'] = A.empty(2 * B * F * ws['Tl'][0], dtype=torch.float16, device=dev)   # planes of one layer's operand
                ws['ewp'] = A.empty(5, 2 * 5 * F * F, dtype=torch.float16, device=dev)
                ws['enc_amax'] = torch.zeros(12, dtype=torch.int32, device=dev)
                ws['enc_scale'] = torch.ones(12, device=dev)
                if train:
                    ws['ewtp'] = A.empty(5, 2 * 5 * F * F, dtype=torch.float16, device=dev)
                    if self.wg_planes:     # the planes of each layer's input (space-to-depth) and output gradient are KEPT: the strided weight
                        # gradients read them (p_planes / q_planes) instead of fetching fp32 one float per request (240 MB at B = 8)
                        ws['esp'] = {i: A.empty(2 * B * F * ws['Tl'][i - 1], dtype=torch.float16, device=dev) for i in range(1, 6)}
                        ws['edp'] = {i: A.empty(2 * B * F * ws['Tl'][i], dtype=torch.float16, device=dev) for i in range(1, 6)}
            if train:
                ws['r'] = [e(B, F, t) for t in ws['Tl']]      # relu outputs
                ws['y6'] = e(B, D, Tz)
                ws['dX'] = [e(B, F, t) for t in ws['Tl']]
        else:
            self.magenta.workspace(ws, B, T, dev, train)
        ws['z_e'] = e(B, D, Tz)
        ws['idx'] = A.empty(B, Tz, dtype=torch.int64, device=dev)
        ws['e_k'] = e(B, D, Tz)
        ws['mind'] = e(B, Tz)
        ws['cond'] = e(B, self.Cc, Tz)
        ws['scale'] = e(6 * F + D)
        ws['shift'] = e(6 * F + D)
        if not train:
            return ws
        ws['condenc'] = e(B, self.Mall, Tz)
        ws['net'] = [e(B, R, T) for _ in range(L + 1)]
        ws['skip'] = e(B, S, T)
        ws['gated'] = [e(B, R, T) for _ in range(L)]
        ws['th'] = [e(B, R, T) for _ in range(L)]
        ws['sg'] = [e(B, R, T) for _ in range(L)]
        if self.gate_f16x3:
            ws['xp'] = A.empty(2 * B * R * T, dtype=torch.float16, device=dev)
            # the input planes of EVERY layer are kept (1.6 GB at B = 8) where the weight gradients read them (backward: p_planes)
            if self.x3_all and self.wg_planes:
                ws['xp_all'] = [ws['xp']] + [A.empty(2 * B * R * T, dtype=torch.float16, device=dev) for _ in range(L)]
            ws['wp_all'] = A.empty(L, 2 * self.ks * R * 2 * R, dtype=torch.float16, device=dev)
            ws['wp'] = [ws['wp_all'][l] for l in range(L)]
            ws['gp'] = A.empty(2 * B * R * T * (L if self.skip_f16x3 else 1), dtype=torch.float16, device=dev)
            if self.dgrad_f16x3:
                ws['dp'] = A.empty(2 * B * 2 * R * T, dtype=torch.float16, device=dev)
                ws['wdg'] = A.empty(L, 2 * self.ks * 2 * R * R, dtype=torch.float16, device=dev)
                if self.gbwd_f16x3:
                    ws['gr'] = A.empty(2 * B * (S + R) * T, dtype=torch.float16, device=dev)   # [dskip | dnet] lifted planes
                    ws['wgb'] = A.empty(L, 2 * (S + R) * R, dtype=torch.float16, device=dev)
                    ws['wgb_top'] = A.empty(2 * S * R, dtype=torch.float16, device=dev)
            if self.skip_f16x3:
                ws['wskip'] = A.empty(2 * L * R * S, dtype=torch.float16, device=dev)
                ws['wres'] = A.empty(L, 2 * R * R, dtype=torch.float16, device=dev)
                if (self.x3_guard or self.bf16) and self.gbwd_f16x3 and S % 256 == 0 and Q % 256 == 0:     # the convs around the stack on the engine too
                    ws['hp'] = A.empty(2 * B * S * T, dtype=torch.float16, device=dev)     # relu(skip) / d postprocess1 planes
                    ws['hp2'] = A.empty(2 * B * S * T, dtype=torch.float16, device=dev)    # relu(postprocess1) / d logits planes
                    for name, n_ in (('wskip0', R * S), ('wpost1', S * S), ('wpost2', S * Q)):
                        ws[name] = A.empty(2 * n_, dtype=torch.float16, device=dev)
                        ws[name + 't'] = A.empty(2 * n_, dtype=torch.float16, device=dev)
                    # |d loss / d logits| <= 1 / (B T): a fixed power-of-two scale, max-abs below 2^13
                    ws['dl_scale'] = torch.full((1,), 2.0 ** math.floor(math.log2(2.0 ** 13 * B * T)), device=dev)
            ws['wop_all'] = A.empty(L, 2 * R * (S + R), dtype=torch.float16, device=dev)
            ws['wop'] = [ws['wop_all'][l] for l in range(L)]
        ws['h1'] = e(B, S, T)
        ws['logits'] = e(B, Q, T)
        # backward
        ws['dnet'] = e(B, R, T)
        ws['dpre'] = e(B, 2 * R, T)
        ws['dnet_ring'] = [ws['dnet'], e(B, R, T), e(B, R, T)]      # ping-pong sets for the two-stream backward
        ws['dpre_ring'] = [ws['dpre'], e(B, 2 * R, T)]
        ws['dcondenc'] = e(B, self.Mall, Tz)
        ws['dcond'] = e(B, self.Cc, Tz)
        ws['dz'] = e(B, D, Tz)
        ws['bskip'] = e(S)
        ws['dscale'] = e(6 * F + D)
        return ws

    # ------------------------------------------------------------------ forward pieces
    def _bn_affine(self, ws):
        """Inference-mode BatchNorm as a per-channel affine (encoder.py:20,25; Appendix A-3)."""
        inv = torch.rsqrt(self.bn_var + BN_EPS)
        torch.mul(self.P['bn_gamma'], inv, out=ws['scale'])
        torch.addcmul(self.P['bn_beta'], self.bn_mean, ws['scale'], value=-1.0, out=ws['shift'])
        return inv

    def _encode(self, x, spk, ws, save=True):
        """encoder.py:13-26 + model.py:57-74 + decoder_ops.py:39-43 -> ws['cond'] [B][Cc][Tz]."""
        P, F, D, B, T = self.P, self.F, self.D, ws['B'], ws['T']
        if self.enc != '64':
            self.magenta.forward(x, ws, P, self.T, save)
            self._quantise(spk, ws)
            return
        self._bn_affine(ws)
        sc, sh = ws['scale'], ws['shift']
        pl, _ = same_pads(T, 5, 2)
        K.conv_cin1_fwd(x, P['enc_w0'], P['enc_b'][0], ws['X'][0], k=5, stride=2, offset=-pl, relu=True,
                        scale=sc[:F], shift=sh[:F], save_r=ws['r'][0] if save else None)
        Tin = ws['Tl'][0]
        # the long strided layers on the fp16x3 engine (space-to-depth planes): exact power-of-two scales from a
        # max-abs pass over this step's tensors; a non-finite value raises the step's range flag (-> fp32 repeat)
        ex3 = ws['enc_x3'] = self._enc_x3_layers(ws)
        if ex3:
            ea, es, flag = ws['enc_amax'], ws['enc_scale'], self.x3_flag
            K.f16x3_amax(P['enc_w'], ea[0:1], flag=flag)
            K.f16x3_update_scales(ea[0:1], es[0:1], target_exp=14, flag=flag)
            K.f16x3_pack_weights(P['enc_w'], ws['ewp'], 5 * F, F, F, 1.0, count=5, scale_dev=es[0:1], mode=0)
        for i in range(1, 6):
            Tout = ws['Tl'][i]
            pl, _ = same_pads(Tin, 5, 2)
            nsplit = self._short_layer_split(F, Tout, B)
            if i in ex3:
                K.f16x3_amax(ws['X'][i - 1], ea[i:i + 1], flag=flag)
                K.f16x3_update_scales(ea[i:i + 1], es[i:i + 1], target_exp=13, flag=flag)
                epl = ws['esp'][i] if (save and 'esp' in ws) else ws['eplanes']
                K.f16x3_split_activations(ws['X'][i - 1], epl, B, F, Tin, scale_dev=es[i:i + 1], mode=K.X3_S2D)
                K.f16x3_strided_conv(xp=epl, wp=ws['ewp'][i - 1], out=ws['X'][i], save_r=ws['r'][i] if save else None,
                                     B=B, T=Tout, Cin=F, M=F, ks=5, pad_left=pl, bias=P['enc_b'][i], bn_scale=sc[i * F:(i + 1) * F],
                                     bn_shift=sh[i * F:(i + 1) * F], relu=True, x_scale=es[i:i + 1], w_scale=es[0:1], **self._sconv_split(ws))
            elif nsplit > 1:
                # short layer (T_out down to 104): too few tiles for 256 CUs, but K = 5*768 is long: split-K into the
                # output buffer (plain STORE + atomics), then relu / save / BatchNorm affine as a second, tiny pass
                K.conv_gemm(x0=ws['X'][i - 1], w=P['enc_w'][i - 1], bias=P['enc_b'][i], out0=ws['X'][i], B=B, T_in=Tin,
                            T_out=Tout, M=F, C0=F, in_stride=2, taps=[j - pl for j in range(5)], tile=12, split_k=nsplit)
                K.relu_bn_fwd(ws['X'][i], ws['r'][i] if save else None, sc[i * F:(i + 1) * F], sh[i * F:(i + 1) * F])
            else:
                K.conv_gemm(x0=ws['X'][i - 1], w=P['enc_w'][i - 1], bias=P['enc_b'][i], out0=ws['X'][i],
                            save0=ws['r'][i] if save else None, scale=sc[i * F:(i + 1) * F], shift=sh[i * F:(i + 1) * F],
                            B=B, T_in=Tin, T_out=Tout, M=F, C0=F, in_stride=2, taps=[j - pl for j in range(5)],
                            out_relu=True)
            Tin = Tout
        Tz = ws['Tz']
        K.conv_gemm(x0=ws['X'][5], w=P['enc_w6'], bias=P['enc_b6'], out0=ws['z_e'], save0=ws['y6'] if save else None,
                    scale=sc[6 * F:], shift=sh[6 * F:], B=B, T_in=Tz, T_out=Tz, M=D, C0=F, taps=[0])
        self._quantise(spk, ws)

    def _enc_x3_layers(self, ws):
        """Encoder layers (1..5) whose conv and input gradient run on the fp16x3 engine this step: the guarded engine is active,
        channel blocks of 128 (the 256-column tiles over the flat (batch, time) rows may be partial: layers 4 and 5 have 1664 and
        832 columns at B = 8).  Slots of ws['enc_scale'] / ws['enc_amax']: 0 the kernels, i = 1..5 the input of layer i (X[i-1]),
        5 + i the gradient of layer i's conv output."""
        if not (self.x3_guard and self._x3_active and not self.bf16 and self.enc == '64' and self.F % 256 == 0
                and os.environ.get('VQW_ENC_X3', '1') != '0' and 'ewtp' in ws):      # (training workspaces only: the step's
            return ()                                                                   #  range flag is read by train_step)
        B, Tl = ws['B'], ws['Tl']
        return tuple(i for i in (1, 2, 3, 4, 5) if B * Tl[i] >= 256 and Tl[i - 1] == 2 * Tl[i])

    def _wslab(self, ws):
        """Slab of the engine's weight-gradient kernels: the partial 256x256 tiles of ONE launch.  The launcher cuts K so that
        tiles x K splits <= the device's CU count (vqw_device_cus), hence one 256x256 fp32 tile per CU."""
        if 'wslab' not in ws:
            cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
            ws['wslab'] = A.empty(cus * 65536, device=self.dev)
            ws['_poison'].append(ws['wslab'])
        return ws['wslab']

    def _sconv_split(self, ws):
        """Scratch of the strided convs' split-K launches (encoder layers 3-5: 24..78 tiles of 240 K steps on 256 CUs): partial tiles
        of one launch (at most two rounds of 128-row blocks) and the tiles' ticket counters, which every launch leaves at zero."""
        if os.environ.get('VQW_SCONV_SPLIT', '1') == '0':
            return {}
        if 'sslab' not in ws:
            cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
            ws['sslab'] = A.empty(cus * 65536, device=self.dev)
            ws['_poison'].append(ws['sslab'])
            ws['scount'] = torch.zeros(1024, dtype=torch.int32, device=self.dev)
        return {'split_slab': ws['sslab'], 'split_counters': ws['scount']}

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    @staticmethod
    def _short_layer_split(M, T_out, B, cus=256):
        """K slices for a conv whose 64x128 tiles (4 resident per CU) cannot fill the chip: aim at ~4 blocks per CU."""
        nb = -(-M // 64) * -(-T_out // 128) * B
        if nb * 2 > 4 * cus:
            return 1
        return max(1, min(8, (4 * cus) // nb))

    def _quantise(self, spk, ws):
        """model.py:57-74 (VQ) + model.py:22-27 / decoder_ops.py:39-43 (speaker embedding tiled over time)."""
        P, D, Tz = self.P, self.D, ws['Tz']
        if self.use_vq:
            K.vq_nearest_fwd(ws['z_e'], P['embedding'], idx=ws['idx'], e_k=ws['e_k'], zq=ws['cond'],
                             zq_bstride=self.Cc * Tz, mind=ws['mind'])
        else:                       # z_q = e_k = z_e (model.py:139-141): a copy, no quantisation losses
            ws['cond'][:, :D].copy_(ws['z_e'])
            ws['idx'].zero_()
            ws['mind'].zero_()
        K.speaker_tile_fwd(P['speaker_embedding'] if self.spk_table else self.onehot, spk, ws['cond'],
                           cond_bstride=self.Cc * Tz, row0=D, Cs=self.Cs_eff, Tz=Tz)

    def _decode_train(self, x, ws, save=True):
        """wavenet.py:24-100 -> ws['logits'] [B][Q][T], ws['labels']."""
        self._decode_layers(ws, self._decode_prologue(x, ws, save))

    def _decode_prologue(self, x, ws, save=True):
        """Everything of the decoder's forward pass that does not depend on the encoder: inputs / labels, the preprocess conv
        (wavenet.py:33-44), this step's guard scales and weight planes, the skip start (wavenet.py:53-54).  Returns the
        plan of the step (which engine carries what, the guard slots) for _decode_layers."""
        P, R, S, Q, L, B, T, Tz = self.P, self.R, self.S, self.Q, self.L, ws['B'], ws['T'], ws['Tz']
        K.wavenet_inputs(x, ws['inputs'], ws['labels'])                                   # wavenet.py:33-37
        net = ws['net']
        K.conv_cin1_fwd(ws['inputs'], P['pre_w'], P['pre_b'], net[0], k=self.pre_k, stride=1,
                        offset=-(self.pre_k - 1))                                         # wavenet.py:42-44
        # fp16x3 needs whole 256-step tiles inside a batch row, 128-channel blocks and one condition frame per 32 steps;
        # |w| < 255 and |net| < 65504 (fp16 range of the leading planes) are assumed, not checked
        f16x3 = self.gate_f16x3 and T % 256 == 0 and R % 128 == 0 and (T // Tz) % 32 == 0
        f16x3_skip = f16x3 and self.skip_f16x3 and R % 256 == 0 and S % 256 == 0
        f16x3_out = f16x3 and self.out_f16x3 and R % 256 == 0 and S % 256 == 0    # the 1x1 skip + residual conv too; it hands the next layer its planes
        if self.x3_all and not (f16x3_skip and self._x3_active):   # guarded / bf16 engine: all of it or none of it
            if self._x3_active and (B, T) not in self._x3_warned:
                self._x3_warned.add((B, T))
                why = ('length %d is not a multiple of 256' % T if T % 256 else
                       'residual / skip widths %d / %d are not multiples of 256' % (R, S) if (R % 256 or S % 256) else
                       '%d samples per condition frame is not a multiple of 32' % (T // Tz))
                print('[vqwave] batch %d x length %d runs on the fp32-MFMA engine (about half the speed of the %s engine): %s'
                      % (B, T, 'bf16' if self.bf16 else 'fp16x3', why), file=sys.stderr, flush=True)
            f16x3 = f16x3_skip = f16x3_out = False
        # the skip contraction over all layers reads L*R/8 chunks of the gated planes through 32-bit offsets: at most 2 GiB
        # per plane and launch, so a large batch cuts it into groups of layers (batch 16 x 6656: 2 groups; one up to B*T = 139 k)
        ngrp = max(1, int(os.environ.get('VQW_SKIP_GROUPS', '1')))          # (the variable forces groups at small shapes: tests)
        while f16x3_skip and (L % ngrp or 2 * (L // ngrp) * R * B * T >= (1 << 31)):
            ngrp += 1
        ws['skip_groups'] = ngrp if f16x3_skip else 0
        Lg = L // ngrp
        ws['x3_used'] = bool(f16x3_skip and self.x3_all)
        md = self.x3_mode_fwd
        gd = self.x3_guard and f16x3_skip
        sc = (lambda name, i=0: self.x3_scale[self.SL[name] + i:self.SL[name] + i + 1]) if gd else (lambda name, i=0: None)
        am = (lambda name, i=0: self.x3_amax[self.SL[name] + i:self.SL[name] + i + 1]) if gd else (lambda name, i=0: None)
        flag = self.x3_flag if gd else None
        WS = 1.0 if gd else 256.0          # guarded: the weight scale lives on the device (exact max-abs of this step's weights)
        head_x3 = ws['head_x3'] = bool((gd or (self.bf16 and f16x3_skip)) and self.head_x3 and 'hp' in ws and T % 32 == 0)
        # layer l's input planes: kept per layer for the weight gradients where the skip contraction runs on the engine (then
        # every layer hands its successor planes), else one buffer (slot L: the planes of net[L], which nothing reads)
        keep_xp = bool('xp_all' in ws and f16x3_skip)
        ws['xp_kept'] = keep_xp
        xpl = (lambda l: ws['xp_all'][l]) if keep_xp else (lambda l: ws['xp'])    # noqa: E731
        # the guarded engine's gate backward forms tanh = gated / sigmoid itself: tanh is not stored (54 MB less per layer and gate
        # conv, 166 -> 157 us); VQW_SAVE_TANH=1 stores it
        drop_th = ws['th_dropped'] = bool((gd or (self.bf16 and f16x3_skip)) and self.gbwd_f16x3 and os.environ.get('VQW_SAVE_TANH', '0') != '1')
        # ... and then nothing reads the fp32 gated output either (gate backward and the 1x1 kernels' weight gradients take the
        # gated planes): the gate conv does not write it (54 MB less per layer)
        drop_g = ws['gated_dropped'] = bool(gd and keep_xp and save and drop_th and os.environ.get('VQW_WGRAD_BATCH', '1') != '0'
                                            and os.environ.get('VQW_SAVE_GATED', '0') != '1')
        if gd:
            K.f16x3_amax(P['gated_w'], am('WG'), flag=flag)
            K.f16x3_amax(P['out_w'], am('WO'), flag=flag)
            K.f16x3_amax(net[0], am('X', 0), flag=flag)
            # weights and the first layer's input: exact scales (amax < 2^14 after scaling); the collectors restart
            K.f16x3_update_scales(self.x3_amax[:2], self.x3_scale[:2], target_exp=14, flag=flag)
            K.f16x3_update_scales(am('X', 0), sc('X', 0), target_exp=13, flag=flag)
        if head_x3:      # the three kernels around the stack share one scale (bf16 planes need none)
            if gd:
                for name in ('skip0_w', 'post1_w', 'post2_w'):
                    K.f16x3_amax(P[name], am('WH'), flag=flag)
                K.f16x3_update_scales(am('WH'), sc('WH'), target_exp=14, flag=flag)
            K.f16x3_pack_weights(P['skip0_w'], ws['wskip0'], R, S, S, 1.0, scale_dev=sc('WH'), mode=md)
            K.f16x3_pack_weights(P['post1_w'], ws['wpost1'], S, S, S, 1.0, scale_dev=sc('WH'), mode=md)
            K.f16x3_pack_weights(P['post2_w'], ws['wpost2'], S, Q, Q, 1.0, scale_dev=sc('WH'), mode=md)
            # wavenet.py:53-54 on the first layer's input planes
            K.f16x3_split_activations(net[0], xpl(0), B, R, T, scale_dev=sc('X', 0), flag=flag, mode=md)
            K.f16x3_out_conv(xp=xpl(0), Cin=R, wp=ws['wskip0'], bias=P['skip0_b'], net_out=ws['skip'], B=B, T=T, R=S, S=0,
                             w_scale_inv=1.0, x_scale=sc('X', 0), w_scale=sc('WH'), mode=md)
        else:
            K.conv_gemm(x0=net[0], w=P['skip0_w'], bias=P['skip0_b'], out0=ws['skip'], B=B, T_in=T, T_out=T, M=S,
                        C0=R, taps=[0])                                                   # wavenet.py:53-54
        if f16x3:      # this step's weights of all layers as fp16 planes, one launch per kind
            K.f16x3_pack_gate_weights(P['gated_w'], ws['wp_all'], self.ks, R, 2 * R, WS, count=L, scale_dev=sc('WG'), mode=md)
            if f16x3_skip:     # [L*R][S] skip kernels of all layers as one K = L*R operand (per layer group); the residual kernels per layer
                K.f16x3_pack_weights(P['out_w'], ws['wskip'], Lg * R, S, S + R, WS, count=ngrp, scale_dev=sc('WO'), mode=md)
                K.f16x3_pack_weights(P['out_w'].view(-1)[S:], ws['wres'], R, R, S + R, WS, count=L, scale_dev=sc('WO'), mode=md)
            elif f16x3_out:
                K.f16x3_pack_weights(P['out_w'], ws['wop_all'], R, S + R, S + R, WS, count=L, mode=md)
        return dict(f16x3=f16x3, f16x3_skip=f16x3_skip, f16x3_out=f16x3_out, ngrp=ngrp, Lg=Lg, md=md, gd=gd, sc=sc, am=am, flag=flag, WS=WS,
                    head_x3=head_x3, xpl=xpl, drop_th=drop_th, drop_g=drop_g, save=save)

    def _decode_layers(self, ws, plan):
        """The condition projections, the residual stack and the convs behind it (wavenet.py:58-100; wavenet_ops.py:93-138)."""
        P, R, S, Q, L, B, T, Tz = self.P, self.R, self.S, self.Q, self.L, ws['B'], ws['T'], ws['Tz']
        f16x3, f16x3_skip, f16x3_out, ngrp, Lg, md, gd = (plan[k] for k in ('f16x3', 'f16x3_skip', 'f16x3_out', 'ngrp', 'Lg', 'md', 'gd'))
        sc, am, flag, WS, head_x3, xpl, drop_th, drop_g, save = (plan[k] for k in ('sc', 'am', 'flag', 'WS', 'head_x3', 'xpl', 'drop_th',
                                                                                  'drop_g', 'save'))
        net = ws['net']
        if self.cond_proj:                                                                # all add_condition 1x1s
            K.cond_proj_fwd(ws['cond'], P['cond_w'], ws['condenc'], B=B, Cc=self.Cc, Mall=self.Mall, Tz=Tz)
        else:
            K.conv_gemm(x0=ws['cond'], w=P['cond_w'], out0=ws['condenc'], B=B, T_in=Tz, T_out=Tz, M=self.Mall,
                        C0=self.Cc, taps=[0])
        cbs = self.Mall * Tz
        ce_flat = ws['condenc'].view(-1)   # layer l's rows start at l*2R*Tz inside every batch block
        for l, d in enumerate(self.dil):
            if f16x3:
                if (l == 0 and not head_x3) or not f16x3_out:
                    K.f16x3_split_activations(net[l], xpl(l), B, R, T, scale_dev=sc('X', 0), flag=flag, mode=md)
                K.f16x3_gate_conv(xp=xpl(l), wp=ws['wp'][l], out0=None if drop_g else ws['gated'][l],
                                  save0=ws['th'][l] if (save and not drop_th) else None,
                                  save1=ws['sg'][l] if save else None, bias=P['gated_b'][l],
                                  cond=ce_flat[l * 2 * R * Tz:], cond_T=Tz, cond_bstride=cbs, B=B, T=T, R=R, ks=self.ks,
                                  dilation=d, w_scale_inv=1.0 / WS, out_planes=ws['gp'] if f16x3_out else None,
                                  out_planes_kc0=l * (R // 8) if f16x3_skip else 0, out_planes_KC=L * (R // 8) if f16x3_skip else 0,
                                  x_scale=sc('X', l), w_scale=sc('WG'), mode=md)
                if f16x3_skip:   # residual half now; the skip half of all layers after the loop
                    if l == L - 1:
                        continue     # the top layer's residual output feeds nothing (wavenet.py:58-78: TF prunes the op from the graph)
                    K.f16x3_out_conv(xp=ws['gp'], xp_kc0=l * (R // 8), xp_KC=L * (R // 8), Cin=R, wp=ws['wres'][l],
                                     bias=P['out_b'][l][S:], net_in=net[l], net_out=net[l + 1], net_out_planes=xpl(l + 1),
                                     B=B, T=T, R=R, S=0, w_scale_inv=1.0 / WS, w_scale=sc('WO'), out_scale=sc('X', l + 1),
                                     out_amax=am('X', l + 1), flag=flag, mode=md)
                elif f16x3_out:
                    K.f16x3_out_conv(xp=ws['gp'], wp=ws['wop'][l], bias=P['out_b'][l], skip=ws['skip'], net_in=net[l],
                                     net_out=net[l + 1], net_out_planes=xpl(l + 1), B=B, T=T, R=R, S=S, w_scale_inv=1.0 / 256.0, mode=md)
                else:
                    K.conv_gemm(x0=ws['gated'][l], w=P['out_w'][l], bias=P['out_b'][l], out0=ws['skip'], out1=net[l + 1],
                                aux1=net[l], B=B, T_in=T, T_out=T, M=S + R, M0=S, C0=R, taps=[0],
                                epilogue=K.EPI_ACCUM_SPLIT, tile=self.tiles['out'])
                continue
            K.conv_gemm(x0=net[l], w=P['gated_w'][l], bias=P['gated_b'][l], out0=ws['gated'][l],
                        save0=ws['th'][l] if save else None, save1=ws['sg'][l] if save else None,
                        cond=ce_flat[l * 2 * R * Tz:], cond_T=Tz, cond_bstride=cbs, B=B, T_in=T, T_out=T,
                        M=2 * R, C0=R, taps=[-(self.ks - 1 - j) * d for j in range(self.ks)],
                        epilogue=K.EPI_GATE)                                              # wavenet_ops.py:104-114
            K.conv_gemm(x0=ws['gated'][l], w=P['out_w'][l], bias=P['out_b'][l], out0=ws['skip'], out1=net[l + 1],
                        aux1=net[l], B=B, T_in=T, T_out=T, M=S + R, M0=S, C0=R, taps=[0],
                        epilogue=K.EPI_ACCUM_SPLIT, tile=self.tiles['out'])                                       # :132-136, wavenet.py:72-73
        wsk = ws['wskip'].view(ngrp, -1) if f16x3_skip else None
        if head_x3:      # the same contraction hands relu(skip) over as planes; wavenet.py:80-96 on planes
            for gi in range(ngrp):
                last = gi == ngrp - 1
                K.f16x3_out_conv(epi=2, xp=ws['gp'], Cin=Lg * R, xp_kc0=gi * Lg * (R // 8), xp_KC=L * (R // 8), wp=wsk[gi],
                                 bias=P['out_b'][:, :S].sum(0) if gi == 0 else None, net_in=ws['skip'], net_out=ws['skip'],
                                 net_out_planes=ws['hp'] if last else None, relu_planes=last, B=B, T=T, R=S, S=0,
                                 w_scale_inv=1.0 / WS, w_scale=sc('WO'), out_scale=sc('SK') if last else None,
                                 out_amax=am('SK') if last else None, flag=flag if last else None, mode=self.x3_mode_skip)
            K.f16x3_out_conv(epi=2, xp=ws['hp'], Cin=S, wp=ws['wpost1'], bias=P['post1_b'], cond=ce_flat[L * 2 * R * Tz:], cond_T=Tz,
                             cond_bstride=cbs, net_out=ws['h1'], net_out_planes=ws['hp2'], relu_planes=True, B=B, T=T, R=S, S=0,
                             w_scale_inv=1.0, x_scale=sc('SK'), w_scale=sc('WH'), out_scale=sc('H1'), out_amax=am('H1'), flag=flag,
                             mode=self.x3_mode_head)
            K.f16x3_out_conv(epi=2, xp=ws['hp2'], Cin=S, wp=ws['wpost2'], bias=P['post2_b'], net_out=ws['logits'], B=B, T=T, R=Q,
                             S=0, w_scale_inv=1.0, x_scale=sc('H1'), w_scale=sc('WH'), mode=self.x3_mode_head)
            return
        for gi in range(ngrp if f16x3_skip else 0):   # skip = skip0 + sum_l (W_s,l g_l + b_s,l)   (wavenet.py:72 summed over the layers)
            K.f16x3_out_conv(xp=ws['gp'], Cin=Lg * R, xp_kc0=gi * Lg * (R // 8), xp_KC=L * (R // 8), wp=wsk[gi],
                             bias=P['out_b'][:, :S].sum(0) if gi == 0 else None, skip=ws['skip'], B=B, T=T, R=0, S=S,
                             w_scale_inv=1.0 / WS, w_scale=sc('WO'), mode=self.x3_mode_skip)
        K.conv_gemm(x0=ws['skip'], in_relu=True, w=P['post1_w'], bias=P['post1_b'], out0=ws['h1'],
                    cond=ce_flat[L * 2 * R * Tz:], cond_T=Tz, cond_bstride=cbs, B=B, T_in=T, T_out=T, M=S,
                    C0=S, taps=[0])                                                       # wavenet.py:80-88
        K.conv_gemm(x0=ws['h1'], in_relu=True, w=P['post2_w'], bias=P['post2_b'], out0=ws['logits'], B=B, T_in=T,
                    T_out=T, M=Q, C0=S, taps=[0])                                         # wavenet.py:94-96

    def forward(self, x, spk, compute_grad_seed=True):
        """model.py:145-151 up to the losses.  x [B][T] raw audio in [-1,1], spk int64 [B].
        Leaves logits (or d loss/d logits when compute_grad_seed) in the workspace and the
        loss sums in self.loss_buf (no host sync)."""
        if self._pending and not self._in_step:      # a caller's own forward pass between deferred steps: settle those first
            self.finish_steps()
        B, T = x.shape
        ws = self._workspace(B, T)
        if A.POISON:             # debug: every workspace buffer and the transposed-kernel scratch start the step as NaN
            A.repoison(ws['_poison'] + self._poison_T)
        # (the decoder's encoder-independent prologue on the side stream under the encoder's short layers measured null, 27.0 vs
        # 27.0 ms: the step is not bound by those ~35 small launches)
        plan = self._decode_prologue(x, ws)
        self._encode(x, spk, ws)
        self._decode_layers(ws, plan)
        self.loss_buf.zero_()
        N = B * T
        K.softmax_xent(ws['logits'], ws['labels'], loss_sum=self.loss_buf[0:1],
                       dlogits=ws['logits'] if compute_grad_seed else None, grad_scale=1.0 / N)  # model.py:91-94
        K.rowsum(ws['mind'].view(1, 1, -1), total=self.loss_buf[1:2])
        return ws

    def forward_checked(self, x, spk, compute_grad_seed=False):
        """forward() for callers that do not go on to train_step (evaluation, summaries of a held-out batch).  On the guarded
        fp16x3 engine the activation planes are scaled with the PREVIOUS training step's max-abs values (1.0 on a fresh model):
        train_step reads the range flag and repeats a flagged step on the fp32 engine; a bare forward() does not.  This one
        does: the flag is zeroed, read after the pass (one host sync) and a flagged pass is repeated on the fp32 engine."""
        self.finish_steps()
        if not self.x3_guard:
            return self.forward(x, spk, compute_grad_seed)
        self.x3_flag.zero_()
        ws = self.forward(x, spk, compute_grad_seed)
        if (ws.get('x3_used') or ws.get('enc_x3')) and int(self.x3_flag.item()) != 0:
            self._x3_active = False
            try:
                ws = self.forward(x, spk, compute_grad_seed)
            finally:
                self._x3_active = True
        return ws

    def losses(self, ws):
        """(loss, reconstruction, vq, commitment) as python floats (synchronises)."""
        self.finish_steps()
        v = self.loss_buf.tolist()
        recon = v[0] / (ws['B'] * ws['T'])
        vq = v[1] / (ws['B'] * ws['Tz'] * self.D) if self.use_vq else 0.0   # model.py:100
        commit = self.beta * vq                             # model.py:103 (same forward value)
        return recon + vq + commit, recon, vq, commit

    def summaries(self, ws, bins=30):
        """What the reference hands to TensorBoard every `-interval` steps (model.py:28-69,95-104; train.py:104-109), as plain
        numbers: for every histogram tag its min / max / mean / std and `bins` equal-width counts, plus the loss scalars.
        `_u` / `_v` are the mean / variance over the last (feature) axis, as tf.nn.moments(x, [-1]).  The reference's
        'distances' histogram is over the [B, Tz, K] tensor this build never materialises; 'distances_min' (the distance to
        the chosen code) stands in for it.  Synchronises."""
        self.finish_steps()
        def hist(t):
            t = t.detach().float().reshape(-1)
            lo, hi = float(t.min()), float(t.max())
            counts = torch.histc(t, bins=bins, min=lo, max=hi if hi > lo else lo + 1.0)
            return {'min': lo, 'max': hi, 'mean': float(t.mean()), 'std': float(t.std()) if t.numel() > 1 else 0.0,
                    'counts': [int(c) for c in counts.tolist()]}

        def moments(t):          # features on the last axis
            return t.mean(-1), t.var(-1, unbiased=False)
        out = {}
        z_e = ws['z_e'].permute(0, 2, 1)                    # [B, Tz, D] as the reference holds it
        tags = {'z_e': z_e}
        if self.spk_table:
            tags['speaker_embedding'] = self.P['speaker_embedding']
        if self.use_vq:
            tags['embedding'] = self.P['embedding']
            tags['e_k'] = ws['e_k'].permute(0, 2, 1)
            out['q(z|x)'] = hist(ws['idx'])
            out['distances_min'] = hist(ws['mind'])
        for name, t in tags.items():
            out[name] = hist(t)
            if name != 'e_k':
                u, v = moments(t)
                out[name + '_u'], out[name + '_v'] = hist(u), hist(v)
        loss, recon, vq, commit = self.losses(ws)
        out['reconstruction_loss'] = recon
        if self.use_vq:
            out['vq_loss'], out['commitment_loss'] = vq, commit
        return out

    # ------------------------------------------------------------------ backward
    def _tt(self, name):
        """The transposed fp32 copy of a kernel for the fp32 engine's input-gradient GEMMs, made on first use in this backward pass
        (the fp16x3 engine packs its planes straight from the parameters, vqw_f16x3_pack_weights_t: on the default path only the
        condition projection and the encoder's last 1x1 layer still need a copy -- six transposes per step fewer)."""
        if name not in self._tt_done:
            L, ks, R, S, F = self.L, self.ks, self.R, self.S, self.F
            shape = {'gated_w': (L * ks, R, 2 * R), 'out_w': (L, R, S + R), 'post1_w': (1, S, S), 'post2_w': (1, S, self.Q),
                     'skip0_w': (1, R, S), 'cond_w': (1, self.Cc, self.Mall), 'enc_w': (25, F, F), 'enc_w6': (1, F, self.D)}[name]
            K.transpose(self.P[name], self.T[name], *shape)
            self._tt_done.add(name)
        return self.T[name]

    def backward(self, x, spk, ws):
        """Gradients of loss = CE + vq + commitment (model.py:90-106) w.r.t. every trainable
        variable, accumulated into self.grad (zeroed here)."""
        P, G, Tt = self.P, self.G, self.T
        R, S, Q, L, F, D, ks = self.R, self.S, self.Q, self.L, self.F, self.D, self.ks
        B, T, Tz, ratio = ws['B'], ws['T'], ws['Tz'], ws['ratio']
        self.grad.zero_()
        self._tt_done = set()
        if self.enc != '64':
            self.magenta.transpose(P, Tt)
        dlog, h1, skip = ws['logits'], ws['h1'], ws['skip']
        cbs = self.Mall * Tz
        dce = ws['dcondenc']
        head_x3 = bool(ws.get('head_x3')) and bool(ws.get('x3_used')) and (self.x3_guard or self.bf16)
        GSh = 1.0 if self.x3_guard else float(2 ** 20)      # the gradient planes' lift where no device scale carries it (bf16 engine)
        if head_x3:
            # the convs around the stack on the fp16x3 engine (forward: _decode_train): d logits as planes with a fixed scale
            # (|d logits| <= 1 / (B T)), each input gradient hands the next one its planes, the weight gradients split their fp32
            # operands in registers (p = relu of the forward tensor), bias and condition sums ride along
            hsc = (lambda name: self.x3_scale[self.SL[name]:self.SL[name] + 1]) if self.x3_guard else (lambda name: None)      # noqa: E731
            ham = (lambda name: self.x3_amax[self.SL[name]:self.SL[name] + 1]) if self.x3_guard else (lambda name: None)       # noqa: E731
            mdh, dl, hflag = self.x3_mode_bwd, ws['dl_scale'], (self.x3_flag if self.x3_guard else None)
            self._wslab(ws)
            K.f16x3_pack_weights_t(P['post2_w'], ws['wpost2t'], Q, S, Q, Q, 0, 1.0, scale_dev=hsc('WH'), mode=mdh)
            K.f16x3_pack_weights_t(P['post1_w'], ws['wpost1t'], S, S, S, S, 0, 1.0, scale_dev=hsc('WH'), mode=mdh)
            K.f16x3_pack_weights_t(P['skip0_w'], ws['wskip0t'], S, R, S, S, 0, 1.0, scale_dev=hsc('WH'), mode=mdh)
            dce.zero_()
            # ---- postprocess2 (wavenet.py:93-96)
            K.f16x3_split_activations(dlog, ws['hp2'], B, Q, T, scale_dev=dl, mode=mdh)
            K.f16x3_wgrad(p=h1, p_relu=True, q0=dlog, dw=G['post2_w'], slab=ws['wslab'], B=B, T=T, Cp=S, Q0=Q, taps=[0],
                          p_scale=hsc('H1'), q0_scale=dl, q_total=G['post2_b'], mode=mdh)
            K.f16x3_out_conv(epi=2, xp=ws['hp2'], Cin=Q, wp=ws['wpost2t'], net_out=h1, aux0=h1, net_out_planes=ws['hp'], B=B, T=T,
                             R=S, S=0, w_scale_inv=1.0, x_scale=dl, w_scale=hsc('WH'), out_scale=hsc('DH'), out_amax=ham('DH'),
                             flag=hflag, mode=self.x3_mode_head)   # h1 := d h1 (pre-relu)
            # ---- postprocess1 (wavenet.py:79-88)
            K.f16x3_wgrad(p=skip, p_relu=True, q0=h1, dw=G['post1_w'], slab=ws['wslab'], B=B, T=T, Cp=S, Q0=S, taps=[0],
                          p_scale=hsc('SK'), q0_scale=hsc('DH'), q_total=G['post1_b'], q_seg=dce.view(-1)[L * 2 * R * Tz:], seg_T=Tz,
                          seg_bstride=cbs, mode=mdh)
            K.f16x3_out_conv(epi=2, xp=ws['hp'], Cin=S, wp=ws['wpost1t'], net_out=skip, aux0=skip, net_out_planes=ws['gr'],
                             planes_kc0=0, planes_KC=(S + R) // 8, plane_scale=GSh, B=B, T=T, R=S, S=0, w_scale_inv=1.0, x_scale=hsc('DH'),
                             w_scale=hsc('WH'), out_scale=hsc('G'), out_amax=ham('G'), flag=hflag, mode=self.x3_mode_head)   # skip := d skip, and its planes
        else:
            # ---- postprocess2 (wavenet.py:93-96)
            K.wgrad_gemm(p=h1, p_relu=True, q0=dlog, dw=G['post2_w'], B=B, T_q=T, T_p=T, Cp=S, Q0=Q, taps=[0])
            K.rowsum(dlog, total=G['post2_b'])
            K.conv_gemm(x0=dlog, w=self._tt('post2_w'), out0=h1, aux0=h1, B=B, T_in=T, T_out=T, M=S, C0=Q, taps=[0],
                        epilogue=K.EPI_MASK)                       # h1 := d h1 (pre-relu)
            if self.x3_guard and not self._x3_active:              # fp32 repeat of a step: what the planes would have held
                K.f16x3_amax(h1, self.x3_amax[self.SL['DH']:self.SL['DH'] + 1])
            # ---- postprocess1 (wavenet.py:79-88)
            K.wgrad_gemm(p=skip, p_relu=True, q0=h1, dw=G['post1_w'], B=B, T_q=T, T_p=T, Cp=S, Q0=S, taps=[0])
            seg_p1 = A.empty(B, S, Tz, device=self.dev)
            K.rowsum(h1, seg_out=seg_p1, total=G['post1_b'], seg=ratio)
            dce[:, L * 2 * R:].copy_(seg_p1)
            K.conv_gemm(x0=h1, w=self._tt('post1_w'), out0=skip, aux0=skip, B=B, T_in=T, T_out=T, M=S, C0=S, taps=[0],
                        epilogue=K.EPI_MASK)                       # skip := d skip (same for every layer)
        dskip = skip
        # ---- residual stack, top layer first (wavenet.py:63-74)
        net = ws['net']
        seg_l = A.empty(B, 2 * R, Tz, device=self.dev)
        # Two streams: the chain gate-backward -> input gradient -> next layer stays on the current stream, the
        # weight gradients and bias/condition sums of a layer (which nothing downstream waits for) run on a side
        # stream, so one kernel's thin last round is filled by the other's blocks.  dnet / dpre are rings (3 / 2
        # buffers): the chain may run two layers ahead of the side stream before it has to wait for it.
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_wgrad else main
        dnet_ring, dpre_ring = ws['dnet_ring'], ws['dpre_ring']
        # Gate backward and the gate convs' input gradient on the plane engine.  Guarded engine: the gradient planes carry
        # device-side power-of-two scales (slots G and DP[l]); development ladder (VQW_GATE_F16X3=4,5): a fixed lift by 2^20
        # (d(loss)/d(logits) is bounded by 1 / (B T); |dpre| < 0.06 assumed there, not checked).
        dgrad_x3 = self.dgrad_f16x3 and T % 256 == 0 and R % 256 == 0
        gbwd_x3 = dgrad_x3 and self.gbwd_f16x3 and S % 256 == 0
        full = bool(ws.get('x3_used'))
        gd = full and self.x3_guard
        md = self.x3_mode_bwd
        if self.x3_all and not full:
            dgrad_x3 = gbwd_x3 = False
        calib = self.x3_guard and not self._x3_active     # fp32 repeat of a step: measure what the planes would have held
        sc = (lambda name, i=0: self.x3_scale[self.SL[name] + i:self.SL[name] + i + 1]) if gd else (lambda name, i=0: None)
        am = (lambda name, i=0: self.x3_amax[self.SL[name] + i:self.SL[name] + i + 1]) if (gd or calib) else (lambda name, i=0: None)
        flag = self.x3_flag if gd else None
        GS = 1.0 if gd else float(2 ** 20)      # guarded: the gradient scales live on the device
        WS = 1.0 if gd else 256.0
        wg_x3 = full and gbwd_x3 and T % 32 == 0 and os.environ.get('VQW_WGRAD_X3', '1') != '0'
        th_dropped = bool(ws.get('th_dropped'))
        if th_dropped and not gbwd_x3:
            raise RuntimeError('the forward pass did not store tanh but gate backward is not on the fp16x3 engine')
        if wg_x3:
            self._wslab(ws)
        if dgrad_x3:
            K.f16x3_pack_weights_t(P['gated_w'], ws['wdg'], ks * 2 * R, R, 2 * R, 2 * R, R * 2 * R, WS, count=L, scale_dev=sc('WG'), mode=md)
        if gbwd_x3:
            K.f16x3_pack_weights_t(P['out_w'], ws['wgb'], S + R, R, S + R, S + R, R * (S + R), WS, count=L, scale_dev=sc('WO'), mode=md)
            K.f16x3_pack_weights_t(P['out_w'][L - 1], ws['wgb_top'], S, R, S, S + R, 0, WS, scale_dev=sc('WO'), mode=md)       # the top layer has no dnet
            if not head_x3:      # (the input gradient of postprocess1 wrote them otherwise)
                K.f16x3_split_activations(dskip, ws['gr'], B, S, T, scale=GS, kc0=0, KC=(S + R) // 8, scale_dev=sc('G'),
                                          amax=am('G'), flag=flag, mode=md)   # one tensor for all layers
        if calib:
            K.f16x3_amax(dskip, am('G'))
        if wg_x3 and not head_x3:      # the weight-gradient kernels add the per-frame sums of dpre into the condition gradient
            dce[:, :L * 2 * R].zero_()
        # Batched weight gradients (the engine's default): dpre and dnet of EVERY layer are kept (4.9 GB at B = 8 instead of
        # rings of 2 / 3 buffers) and the weight gradients of several layers go out as ONE launch (vqw_f16x3_wgrad_batch):
        #   * the skip halves of all layers' 1x1 kernels, dW_s[l] = gated[l] (x) dskip -- dskip is the same tensor for every
        #     layer -- at once, before the layer loop: 60 tiles x 4 K splits instead of 30 x (2 tiles x 80 splits);
        #   * the gate kernels of up to 6 layers (VQW_WG_GATE_BATCH) (layers whose tap shifts are all multiples of 4 and the others --
        #     dilations 1 and 2 -- in separate batches: the latter need the kernel's slower unaligned-window variant);
        #   * the residual halves, dW_r[l] = gated[l] (x) dnet[l+1], of up to 29 layers (VQW_WG_RES_BATCH).
        # A launch writes tiles x splits <= CUs partial 256x256 tiles to the slab whatever its batch size: per layer the slab
        # traffic (2 x 61 MB per single launch, 7.3 GB per step) falls with the batch size, dskip is read once per XCD instead
        # of once per layer, and 68 reductions become ~10.
        batched = bool(wg_x3) and os.environ.get('VQW_WGRAD_BATCH', '1') != '0'
        # ... and dpre is kept as the operand PLANES gate backward writes for the input gradient anyway: the gate kernels' weight
        # gradient reads its q operand from them (transposed LDS reads, vqw_f16x3_wgrad q_planes) and gate backward no longer
        # writes fp32 dpre at all (109 of its 312 MB per layer)
        qp = batched and os.environ.get('VQW_WGRAD_QP', '1') != '0'
        # ... and so do p = the layer's input planes (kept per layer by the forward pass) and p = the gated planes of all layers
        pp = batched and bool(ws.get('xp_kept'))
        g_dropped = bool(ws.get('gated_dropped'))
        if g_dropped and not (pp and gbwd_x3):
            raise RuntimeError('the forward pass did not store the fp32 gated output but the backward pass needs it')
        gated_p = (lambda i: dict(p_planes=ws['gp'], p_planes_kc0=i * (R // 8), p_planes_KC=L * (R // 8))) if pp else \
            (lambda i: dict(p=ws['gated'][i]))
        if batched and 'dnet_all' not in ws:
            ws['dnet_all'] = [A.empty(B, R, T, device=self.dev) for _ in range(L)]       # dnet_all[l] = d loss / d net[l]
            ws['_poison'] += ws['dnet_all']
        if batched and ('dp_all' if qp else 'dpre_all') not in ws:
            if qp:
                ws['dp_all'] = [A.empty(2 * B * 2 * R * T, dtype=torch.float16, device=self.dev) for _ in range(L)]
            else:
                ws['dpre_all'] = [A.empty(B, 2 * R, T, device=self.dev) for _ in range(L)]
            ws['_poison'] += ws['dp_all' if qp else 'dpre_all']
        gate_batch = max(1, min(K.WGRAD_MAX_BATCH, int(os.environ.get('VQW_WG_GATE_BATCH', '6'))))      # 36 tiles x 7 K splits = 252 blocks (tools/wg_batch_sweep.sh)
        res_batch = max(1, min(K.WGRAD_MAX_BATCH, int(os.environ.get('VQW_WG_RES_BATCH', '29'))))
        pend_gate, pend_res = {False: [], True: []}, []

        def on_side(launch):           # weight gradients: nothing downstream waits for them
            if side is main:
                return launch()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                launch()

        def flush_gate(odd):
            layers, pend_gate[odd] = pend_gate[odd], []
            if layers:
                probs = [dict(dw=G['gated_w'][i], taps=[-(ks - 1 - j) * self.dil[i] for j in range(ks)],
                              p_scale=sc('X', i), q0_scale=sc('DP', i), q_seg=dce.view(-1)[i * 2 * R * Tz:],
                              **(dict(p_planes=ws['xp_all'][i]) if pp else dict(p=net[i])),
                              **(dict(q_planes=ws['dp_all'][i], q_planes_scale=GS) if qp else dict(q0=ws['dpre_all'][i]))) for i in layers]
                on_side(lambda: K.f16x3_wgrad_batch(probs, slab=ws['wslab'], B=B, T=T, Cp=R, Q0=2 * R, seg_T=Tz, seg_bstride=cbs, mode=md))

        def flush_res():
            nonlocal pend_res
            layers, pend_res = pend_res, []
            if layers:         # (the top layer has no dnet: its residual kernel gets no gradient)
                probs = [dict(q0=ws['dnet_all'][i + 1], dw=G['out_w'][i].view(-1)[S:], q_total=G['out_b'][i][S:], **gated_p(i))
                         for i in layers]
                on_side(lambda: K.f16x3_wgrad_batch(probs, slab=ws['wslab'], B=B, T=T, Cp=R, Q0=R, lddw=S + R, taps=[0],
                                                    q0_scale=sc('G'), total_cols=(0, R), mode=md))

        if batched:      # the skip halves of all layers (dskip: the first S / 8 chunks of the gradient planes)
            qsk = dict(q_planes=ws['gr'], q_planes_KC=(S + R) // 8, q_planes_scale=GS) if (qp and gbwd_x3) else dict(q0=dskip)
            for i0 in range(0, L, K.WGRAD_MAX_BATCH):
                probs = [dict(dw=G['out_w'][i].view(-1), **gated_p(i)) for i in range(i0, min(L, i0 + K.WGRAD_MAX_BATCH))]
                on_side(lambda: K.f16x3_wgrad_batch(probs, slab=ws['wslab'], B=B, T=T, Cp=R, Q0=S, lddw=S + R, taps=[0],
                                                    q0_scale=sc('G'), mode=md, **qsk))
        side_done = {}
        dnet = dnet_ring[(L - 1) % 3]
        for l in range(L - 1, -1, -1):
            d = self.dil[l]
            top = (l == L - 1)       # net[L] is unused by the graph: its gradient is zero
            dpre = (None if qp else ws['dpre_all'][l]) if batched else dpre_ring[l % 2]
            dplanes = ws['dp_all'][l] if qp else ws.get('dp')
            dnet_next = ws['dnet_all'][l] if batched else dnet_ring[(l - 1) % 3]
            if not batched and side is not main and (l + 2) in side_done:
                main.wait_event(side_done[l + 2])            # dpre[l % 2] and dnet[(l - 1) % 3] are free again
            if gbwd_x3:
                K.f16x3_out_conv(epi=1, xp=ws['gr'], Cin=S if top else S + R, xp_KC=(S + R) // 8,
                                 wp=ws['wgb_top'] if top else ws['wgb'][l], aux0=None if g_dropped else (ws['gated'][l] if th_dropped else ws['th'][l]),
                                 aux0_planes=ws['gp'] if g_dropped else None, aux0_KC=L * (R // 8), aux0_kc0=l * (R // 8),
                                 aux0_is_gated=th_dropped, aux1=ws['sg'][l], net_out=dpre,
                                 net_out_planes=dplanes, plane_scale=GS, B=B, T=T, R=R, S=0, w_scale_inv=1.0 / (WS * GS),
                                 x_scale=sc('G'), w_scale=sc('WO'), out_scale=sc('DP', l), out_amax=am('DP', l), flag=flag, mode=md)
            else:
                K.conv_gemm(x0=dskip, x1=None if top else dnet, w=self._tt('out_w')[l], out0=dpre, aux0=ws['th'][l],
                            aux1=ws['sg'][l], B=B, T_in=T, T_out=T, M=R, C0=S, C1=0 if top else R, taps=[0],
                            epilogue=K.EPI_GATE_BWD, tile=self.tiles['gate_bwd'])
            if calib:
                K.f16x3_amax(dpre, am('DP', l))
            if side is not main:
                ready = torch.cuda.Event()
                ready.record(main)
            taps_b = [(ks - 1 - j) * d for j in range(ks)]
            if dgrad_x3:
                if not gbwd_x3:      # (gate backward hands dpre over as planes)
                    K.f16x3_split_activations(dpre, ws['dp'], B, 2 * R, T, scale=GS, mode=md)
                K.f16x3_out_conv(xp=dplanes, Cin=2 * R, ks=ks, dilation=d, direction=-1, wp=ws['wdg'][l],
                                 net_in=None if top else dnet, net_out=dnet_next, B=B, T=T, R=R, S=0,
                                 w_scale_inv=1.0 / (WS * GS),
                                 net_out_planes=ws['gr'] if gbwd_x3 else None, planes_kc0=S // 8 if gbwd_x3 else 0,
                                 planes_KC=(S + R) // 8 if gbwd_x3 else 0, plane_scale=GS if gbwd_x3 else 0.0,
                                 x_scale=sc('DP', l), w_scale=sc('WG'), out_scale=sc('G'), out_amax=am('G'), flag=flag, mode=self.x3_mode_dgrad)
            elif top:
                K.conv_gemm(x0=dpre, w=self._tt('gated_w')[l], out0=dnet_next, B=B, T_in=T, T_out=T, M=R, C0=2 * R, taps=taps_b,
                            tile=self.tiles['dgrad'])
            else:
                K.conv_gemm(x0=dpre, w=self._tt('gated_w')[l], out1=dnet_next, aux1=dnet, out0=dnet_next, B=B, T_in=T, T_out=T,
                            M=R, M0=0, C0=2 * R, taps=taps_b, epilogue=K.EPI_ACCUM_SPLIT, tile=self.tiles['dgrad'])
            if calib:
                K.f16x3_amax(dnet_next, am('G'))
            if batched:
                odd = any(((ks - 1 - j) * d) % 4 for j in range(ks))
                pend_gate[odd].append(l)
                if len(pend_gate[odd]) >= gate_batch:
                    flush_gate(odd)
                if not top:                  # layer l + 1 wrote dnet[l + 1] in the iteration before this one
                    pend_res.append(l)
                    if len(pend_res) >= res_batch:
                        flush_res()
                dnet = dnet_next
                continue
            with torch.cuda.stream(side):
                if side is not main:
                    side.wait_event(ready)
                if wg_x3:    # weight gradients on the fp16 pipe too: operands split in registers with the planes' guard scales;
                    # the bias sums and the condition gradient (sums of dpre per condition frame) come out of the same kernels
                    K.f16x3_wgrad(p=ws['gated'][l], q0=dskip, q1=None if top else dnet, dw=G['out_w'][l], slab=ws['wslab'], B=B,
                                  T=T, Cp=R, Q0=S, Q1=0 if top else R, lddw=S + R, taps=[0], q0_scale=sc('G'), q1_scale=sc('G'),
                                  q_total=None if top else G['out_b'][l], total_cols=(S, S + R), mode=md)
                    K.f16x3_wgrad(p=net[l], q0=dpre, dw=G['gated_w'][l], slab=ws['wslab'], B=B, T=T, Cp=R, Q0=2 * R,
                                  taps=[-(ks - 1 - j) * d for j in range(ks)], p_scale=sc('X', l), q0_scale=sc('DP', l),
                                  q_seg=dce.view(-1)[l * 2 * R * Tz:], seg_T=Tz, seg_bstride=cbs, mode=md)
                else:
                    K.wgrad_gemm(p=ws['gated'][l], q0=dskip, q1=None if top else dnet, dw=G['out_w'][l], B=B, T_q=T, T_p=T,
                                 Cp=R, Q0=S, Q1=0 if top else R, lddw=S + R, taps=[0])
                    if not top:
                        K.rowsum(dnet, total=G['out_b'][l][S:])
                    K.wgrad_gemm(p=net[l], q0=dpre, dw=G['gated_w'][l], B=B, T_q=T, T_p=T, Cp=R, Q0=2 * R,
                                 taps=[-(ks - 1 - j) * d for j in range(ks)])
                    K.rowsum(dpre, seg_out=seg_l, total=G['gated_b'][l], seg=ratio)
                    dce[:, l * 2 * R:(l + 1) * 2 * R].copy_(seg_l)
                if side is not main:
                    side_done[l] = torch.cuda.Event()
                    side_done[l].record(side)
            dnet = dnet_next
        if batched:
            flush_gate(False)
            flush_gate(True)
            flush_res()
        if side is not main:
            main.wait_stream(side)
        if wg_x3:      # gated biases: the condition gradient summed over batch and frames, all layers in one launch
            tot = torch.zeros(self.Mall, device=self.dev)
            K.rowsum(dce, total=tot)
            G['gated_b'].view(-1).add_(tot[:L * 2 * R])
        # ---- skip start + preprocess (wavenet.py:42-55)
        if head_x3 and gbwd_x3:      # dnet += W_skip0^T dskip over dskip's planes (chunks 0..S/8 of the gradient planes)
            K.f16x3_out_conv(xp=ws['gr'], xp_KC=(S + R) // 8, Cin=S, wp=ws['wskip0t'], net_in=dnet, net_out=dnet, B=B, T=T, R=R, S=0,
                             w_scale_inv=1.0 / GS, x_scale=sc('G'), w_scale=sc('WH'), mode=md)
        else:
            K.conv_gemm(x0=dskip, w=self._tt('skip0_w'), out1=dnet, aux1=dnet, out0=dnet, B=B, T_in=T, T_out=T, M=R, M0=0,
                        C0=S, taps=[0], epilogue=K.EPI_ACCUM_SPLIT)
        ws['bskip'].zero_()      # sum of dskip over batch and time: the bias gradient of skip0 and of every layer's skip half
        if wg_x3:      # both operands already have guard scales (layer-0 input planes, gradient planes); the sum rides along
            K.f16x3_wgrad(p=net[0], q0=dskip, dw=G['skip0_w'], slab=ws['wslab'], B=B, T=T, Cp=R, Q0=S, taps=[0],
                          p_scale=sc('X', 0), q0_scale=sc('G'), q_total=ws['bskip'], mode=md)
        else:
            K.wgrad_gemm(p=net[0], q0=dskip, dw=G['skip0_w'], B=B, T_q=T, T_p=T, Cp=R, Q0=S, taps=[0])
            K.rowsum(dskip, total=ws['bskip'])
        G['out_b'][:, :S] += ws['bskip']
        G['skip0_b'] += ws['bskip']
        K.conv_cin1_wgrad(ws['inputs'], dnet, G['pre_w'], k=self.pre_k, stride=1, offset=-(self.pre_k - 1))
        K.rowsum(dnet, total=G['pre_b'])
        # ---- local condition (wavenet_ops.py:93-101) -> d cond
        if self.cond_proj and Tz % 4 == 0 and P['cond_w'].data_ptr() % 16 == 0:
            if 'cp_scratch' not in ws:
                ws['cp_scratch'] = A.empty(K.cond_proj_dgrad_scratch(B, self.Cc, self.Mall, Tz), device=self.dev)
                ws['_poison'].append(ws['cp_scratch'])
            K.cond_proj_wgrad(ws['cond'], dce, G['cond_w'], B=B, Cc=self.Cc, Mall=self.Mall, Tz=Tz)
            K.cond_proj_dgrad(P['cond_w'], dce, ws['dcond'], ws['cp_scratch'], B=B, Cc=self.Cc, Mall=self.Mall, Tz=Tz)
        else:
            K.wgrad_gemm(p=ws['cond'], q0=dce, dw=G['cond_w'], B=B, T_q=Tz, T_p=Tz, Cp=self.Cc, Q0=self.Mall, taps=[0])
            K.conv_gemm(x0=dce, w=self._tt('cond_w'), out0=ws['dcond'], B=B, T_in=Tz, T_out=Tz, M=self.Cc, C0=self.Mall, taps=[0])
        if self.grad_sync is not None:      # decoder gradients are final: exchange them under the encoder backward
            self.grad_sync.bucket_ready(self.seg_off['pre_w'][0], self.n_flat)
        # ---- speaker embedding + VQ (model.py:22-27, 57-74, 99-106)
        if self.spk_table:
            K.speaker_tile_bwd(ws['dcond'], spk, G['speaker_embedding'], dcond_bstride=self.Cc * Tz, row0=D, Cs=self.Cs, Tz=Tz)
        nd = float(B * Tz * D)
        if self.use_vq:
            K.vq_nearest_bwd(ws['z_e'], ws['e_k'], ws['idx'], dzq=ws['dcond'], dzq_bstride=self.Cc * Tz, dz_e=ws['dz'],
                             demb=G['embedding'], cscale=2.0 * self.beta / nd, escale=2.0 / nd, K=self.Kc)
        else:
            ws['dz'].copy_(ws['dcond'][:, :D])
        if self.enc != '64':
            self.magenta.backward(x, ws, P, G, Tt)
            if self.grad_sync is not None:
                self.grad_sync.bucket_ready(0, self.seg_off['pre_w'][0])
            return
        # ---- encoder (encoder.py:13-26), BN in inference mode
        sc, dsc = ws['scale'], ws['dscale']
        dsc.zero_()
        dz = ws['dz']
        # BatchNorm / relu backward with its three sums (d scale, d beta, the conv's bias gradient) in one pass per layer
        K.bn_relu_bwd_sums(dz, ws['y6'], None, sc[6 * F:], dz, dscale=dsc[6 * F:], dbeta=G['bn_beta'][6 * F:],
                           dbias=G['enc_b6'])                        # dz := d(conv6 output)
        K.wgrad_gemm(p=ws['X'][5], q0=dz, dw=G['enc_w6'], B=B, T_q=Tz, T_p=Tz, Cp=F, Q0=D, taps=[0])
        K.conv_gemm(x0=dz, w=self._tt('enc_w6'), out0=ws['dX'][5], B=B, T_in=Tz, T_out=Tz, M=F, C0=D, taps=[0])
        main = torch.cuda.current_stream()
        side = self._side_stream() if self.overlap_wgrad else main
        # layers 1..5 on the fp16x3 engine: the input gradient where the forward conv ran there (ws['enc_x3']), the weight
        # gradient (operands split in registers, bias sums riding along) whenever the decoder's ran there this step.  Scales:
        # exact powers of two from max-abs passes over THIS step's tensors (slots: _enc_x3_layers)
        ex3 = ws.get('enc_x3', ())
        wg3 = ('wslab' in ws and bool(ws.get('x3_used')) and self.x3_guard and not self.bf16 and F % 256 == 0
               and os.environ.get('VQW_ENC_WGRAD_X3', '1') != '0')
        ea, es, flag = ws.get('enc_amax'), ws.get('enc_scale'), self.x3_flag
        if ex3:
            K.f16x3_pack_weights_t(P['enc_w'], ws['ewtp'], 5 * F, F, F, F, F * F, 1.0, count=5, scale_dev=es[0:1], mode=0)
        for i in range(5, -1, -1):
            dX, r = ws['dX'][i], ws['r'][i]
            Ti = ws['Tl'][i]
            Tin = ws['Tl'][i - 1] if i > 0 else T
            pl, _ = same_pads(Tin, 5, 2)
            on_c = i in ex3
            on_w = wg3 and 1 <= i <= 5 and Ti % 4 == 0 and B * Ti >= 256 and Tin == 2 * Ti and B * F * Tin * 4 < (1 << 31)
            K.bn_relu_bwd_sums(dX, r, r, sc[i * F:(i + 1) * F], dX, dscale=dsc[i * F:(i + 1) * F],
                               dbeta=G['bn_beta'][i * F:(i + 1) * F],
                               dbias=None if on_w else G['enc_b'][i])   # dX := d(conv_i output); (the engine's wgrad sums the bias itself)
            if on_c or on_w:
                K.f16x3_amax(dX, ea[5 + i:6 + i], flag=flag)
                K.f16x3_update_scales(ea[5 + i:6 + i], es[5 + i:6 + i], target_exp=13, flag=flag)
                if not on_c:                                         # (the forward pass measured X[i-1] otherwise)
                    K.f16x3_amax(ws['X'][i - 1], ea[i:i + 1], flag=flag)
                    K.f16x3_update_scales(ea[i:i + 1], es[i:i + 1], target_exp=13, flag=flag)
            if side is not main:
                ready = torch.cuda.Event()
                ready.record(main)
            # both operands as planes where the forward conv ran on the engine (its space-to-depth input planes were kept):
            # tap j, e = j - pad_left, is parity block e & 1 at row offset e >> 1
            w_planes = on_w and on_c and 'esp' in ws
            if w_planes:     # (the split the input gradient needs anyway, into this layer's own buffer, BEFORE the weight gradient)
                K.f16x3_split_activations(dX, ws['edp'][i], B, F, Ti, scale_dev=es[5 + i:6 + i], mode=0)
                if side is not main:
                    ready = torch.cuda.Event()
                    ready.record(main)
            with torch.cuda.stream(side):                            # weight / bias gradients: nothing downstream waits
                if side is not main:
                    side.wait_event(ready)
                if w_planes:
                    K.f16x3_wgrad(p_planes=ws['esp'][i], p_planes_KC=2 * F // 8, p_tap_chunk=[((j - pl) & 1) * (F // 8) for j in range(5)],
                                  q_planes=ws['edp'][i], dw=G['enc_w'][i - 1], slab=ws['wslab'], B=B, T=Ti, Cp=F, Q0=F,
                                  taps=[(j - pl) >> 1 for j in range(5)], p_scale=es[i:i + 1], q0_scale=es[5 + i:6 + i],
                                  q_total=G['enc_b'][i], mode=0)
                elif on_w:
                    K.f16x3_wgrad(p=ws['X'][i - 1], q0=dX, dw=G['enc_w'][i - 1], slab=ws['wslab'], B=B, T=Ti, Cp=F, Q0=F,
                                  taps=[j - pl for j in range(5)], p_scale=es[i:i + 1], q0_scale=es[5 + i:6 + i], p_stride=2, T_p=Tin,
                                  q_total=G['enc_b'][i], mode=0)
                else:
                    if i == 0:
                        K.conv_cin1_wgrad(x, dX, G['enc_w0'], k=5, stride=2, offset=-pl)
                    else:
                        K.wgrad_gemm(p=ws['X'][i - 1], q0=dX, dw=G['enc_w'][i - 1], B=B, T_q=Ti, T_p=Tin, Cp=F, Q0=F, p_stride=2,
                                     taps=[j - pl for j in range(5)])
                if self.grad_sync is not None and i >= 2:      # this layer's kernel gradient (11.8 MB) is final: exchange it under
                    ew, ne = self.seg_off['enc_w'][0], 5 * F * F   # the rest of the encoder backward
                    self.grad_sync.bucket_ready(ew + (i - 1) * ne, ew + i * ne)
            if i == 0:
                break
            if on_c:
                if not w_planes:
                    K.f16x3_split_activations(dX, ws['eplanes'], B, F, Ti, scale_dev=es[5 + i:6 + i], mode=0)
                K.f16x3_strided_conv(xp=ws['edp'][i] if w_planes else ws['eplanes'], wp=ws['ewtp'][i - 1], out=ws['dX'][i - 1], B=B, T=Ti, Cin=F, M=F, ks=5,
                                     pad_left=pl, dgrad=True, x_scale=es[5 + i:6 + i], w_scale=es[0:1], **self._sconv_split(ws))
                continue
            # transposed conv: output times tau = 2u+p get taps j with j = p + pad_left (mod 2)
            nsplit = self._short_layer_split(F, (Tin + 1) // 2, B)
            if nsplit > 1:
                ws['dX'][i - 1].zero_()        # both parity launches add their K slices into it (split_k < 0)
            for p in (0, 1):
                j0 = (p + pl) % 2
                js = list(range(j0, 5, 2))
                K.conv_gemm(x0=dX, w=self._tt('enc_w')[i - 1][j0:], w_tap_stride=2 * F * F, out0=ws['dX'][i - 1], B=B,
                            T_in=Ti, T_out=(Tin - p + 1) // 2, M=F, C0=F, taps=[(p + pl - j) // 2 for j in js],
                            out_tstride=2, out_toffset=p, T_store=Tin, tile=12 if nsplit > 1 else 0,
                            split_k=-nsplit if nsplit > 1 else 0)
        if side is not main:
            main.wait_stream(side)
        dsc.addcmul_(self.bn_mean, G['bn_beta'], value=-1.0)         # shift = beta - mean*scale
        torch.mul(dsc, torch.rsqrt(self.bn_var + BN_EPS), out=dsc)
        G['bn_gamma'] += dsc
        if self.grad_sync is not None:      # what the per-layer buckets above left: the first layers' kernels, biases, BatchNorm, codebook
            ew, ne = self.seg_off['enc_w'][0], 5 * F * F
            self.grad_sync.bucket_ready(0, ew + ne)
            self.grad_sync.bucket_ready(ew + 5 * ne, self.seg_off['pre_w'][0])

    # ------------------------------------------------------------------ optimiser
    def lr_at(self, step):
        """Piecewise-constant schedule of model.py:111-114."""
        lr = self.schedule[0][1]
        for key, value in self.schedule:
            if not (step < key):
                lr = value
        return lr

    def apply_gradients(self, grad_scale=1.0, skip=None):
        """TF-1.x Adam + EMA(0.999) (model.py:116-128).  skip: device int32, non-zero when the kernel runs = nothing changes."""
        t = self.global_step + 1
        lr = self.lr_at(self.global_step)
        lr_t = lr * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        K.adam_ema_step(self.flat, self.grad, self.adam_m, self.adam_v, self.ema, lr_t=lr_t, grad_scale=grad_scale, skip=skip)
        self.global_step = t
        return lr

    def train_step(self, x, spk, on_forward=None):
        """One sess.run(train_op) (train.py:104-114).  on_forward(ws): called between the forward and the backward pass of the
        step that is kept (tests snapshot the relu inputs there: backward overwrites them in place).  With self.grad_sync set (data parallel)
        the flat gradient is sum-all-reduced over RCCL in buckets that overlap the backward pass, and averaged.
        Guarded fp16x3 engine: a step whose planes left fp16's range is repeated on the fp32 engine, which also measures the
        max-abs values the next step's scales come from.  The step's range flag is read before the optimiser runs (one host sync
        per step) -- or, with self.defer_guard (bench.py, train.py), one step LATE: see _train_step_deferred."""
        if self.defer_guard and self.x3_guard and on_forward is None:
            return self._train_step_deferred(x, spk)
        self.finish_steps()
        return self._train_step_now(x, spk, on_forward)

    def _train_step_now(self, x, spk, on_forward=None, known_flagged=False):
        """The step with its range flag read on the spot.  known_flagged: the fp16x3 attempt of this step already ran and
        raised the flag (deferred mode): go straight to the repeat."""
        guarded, world, ws = self.x3_guard and known_flagged, 1, None
        if not known_flagged:
            if self.x3_guard:
                self.x3_flag.zero_()
            ws = self.forward(x, spk)
            if on_forward is not None:
                on_forward(ws)
            self.backward(x, spk, ws)
            world = self.grad_sync.finish() if self.grad_sync is not None else 1
            if self.bf16 and ws.get('x3_used'):
                self.x3_steps += 1
            guarded = bool(self.x3_guard and (ws.get('x3_used') or ws.get('enc_x3')))
        if guarded:
            if known_flagged or self._x3_overflowed():
                self.x3_fallbacks += 1
                self._x3_active = False
                try:
                    self.x3_amax.zero_()
                    ws = self.forward(x, spk)
                    for l in range(1, self.L):       # what the layer-input planes would have held (net[L] feeds nothing)
                        K.f16x3_amax(ws['net'][l], self.x3_amax[self.SL['X'] + l:self.SL['X'] + l + 1])
                    K.f16x3_amax(ws['skip'], self.x3_amax[self.SL['SK']:self.SL['SK'] + 1])
                    K.f16x3_amax(ws['h1'], self.x3_amax[self.SL['H1']:self.SL['H1'] + 1])
                    if on_forward is not None:
                        on_forward(ws)
                    self.backward(x, spk, ws)
                    world = self.grad_sync.finish() if self.grad_sync is not None else 1
                finally:
                    self._x3_active = True
            else:
                self.x3_steps += 1
            # next step's scales of the planes written inside the kernels (max-abs * scale in [2^12, 2^13): 8x headroom)
            n0 = self.SL['G']
            K.f16x3_update_scales(self.x3_amax[n0:], self.x3_scale[n0:], target_exp=13)
        self.apply_gradients(1.0 / world)
        return ws

    def _train_step_deferred(self, x, spk):
        """The guarded step without a host sync on its path.  Reading the range flag before the optimiser drains the launch queue
        once per step, and the host then needs ~1.2 ms of the next step to get ahead of the GPU again (bench.py's shape: 26.6 ->
        25.5 ms).  Here the verdict stays on the device: the flag is latched into the sticky x3_void, the optimiser is enqueued
        behind that guard (vqw_adam_ema_step_guarded, vqw_f16x3_update_scales_guarded: a voided step changes no parameter, Adam slot,
        EMA shadow or plane scale), x3_void is
        copied to pinned host memory, and the host looks at the copy of step k only after it has enqueued step k + 1.  If step k
        was flagged, it and step k + 1 (enqueued behind it, voided by the sticky guard) have changed nothing: the guard is
        cleared, the step counter rewound, step k is repeated on the fp32 engine and step k + 1 is run again -- parameters after
        every step are bit-identical to the immediate mode (tests/test_model_gpu.py::test_deferred_guard_matches_immediate).
        Callers keep x / spk unchanged until the step is resolved (finish_steps(), or the next-but-one train_step) and call
        finish_steps() before reading what a step left behind (losses, gradients, parameters; state_dict() / encode() do)."""
        t_host = time.perf_counter()
        self.x3_flag.zero_()
        self._in_step = True
        try:
            ws = self.forward(x, spk)
            self.backward(x, spk, ws)
        finally:
            self._in_step = False
        world = self.grad_sync.finish() if self.grad_sync is not None else 1
        if not (ws.get('x3_used') or ws.get('enc_x3')):       # nothing on the guarded engine in this workspace
            self.apply_gradients(1.0 / world)
            return ws
        if self.grad_sync is not None and self.grad_sync.active:
            self.grad_sync.all_reduce_max(self.x3_flag)       # every rank takes the same branch
        torch.maximum(self.x3_void, self.x3_flag, out=self.x3_void)
        n0 = self.SL['G']
        K.f16x3_update_scales(self.x3_amax[n0:], self.x3_scale[n0:], target_exp=13, skip=self.x3_void)
        gs0 = self.global_step
        self.apply_gradients(1.0 / world, skip=self.x3_void)
        if self._void_host is None:
            self._void_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        slot = self._void_slot
        self._void_slot = (slot + 1) % 4
        self._void_host[slot:slot + 1].copy_(self.x3_void, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append({'x': x, 'spk': spk, 'gs0': gs0, 'ev': ev, 'slot': slot})
        self.host_enqueue_ms = (time.perf_counter() - t_host) * 1e3        # what the host needs to enqueue one step (it must stay below the step's GPU time)
        while len(self._pending) > 1:
            self._resolve_oldest()
        return ws

    def _resolve_oldest(self):
        p = self._pending[0]
        p['ev'].synchronize()
        if int(self._void_host[p['slot']]) == 0:
            self._pending.pop(0)
            self.x3_steps += 1
            return
        pend, self._pending = self._pending, []           # p was flagged: it and everything behind it changed nothing
        self.x3_void.zero_()
        self.global_step = p['gs0']
        for n, q in enumerate(pend):
            self._train_step_now(q['x'], q['spk'], known_flagged=(n == 0))

    def finish_steps(self):
        """Resolve the deferred steps (defer_guard): afterwards parameters, gradients and losses are those of the last step."""
        while self._pending:
            self._resolve_oldest()

    def _x3_overflowed(self):
        """Range flag of this step (max over the data-parallel ranks: every rank must take the same branch)."""
        flag = self.x3_flag
        if self.grad_sync is not None and self.grad_sync.active:
            self.x3_own_flag = int(flag.item())      # (this rank's own verdict, kept for diagnostics)
            self.grad_sync.all_reduce_max(flag)
        return int(flag.item()) != 0

    # ------------------------------------------------------------------ generation
    def encode(self, x, spk):
        """model.encoding of generate.py:92: [B][Cc][Tz] (channel-major).  x [B][T], or ONE utterance [1][T] with
        B speaker ids (generate.py:40 repeats the utterance per speaker: the encoder and VQ then run once and only the
        speaker rows of the condition differ)."""
        self.finish_steps()
        Bx, T = x.shape
        B = spk.numel()
        if Bx != B and Bx != 1:
            raise ValueError('encode: %d utterances for %d speaker ids' % (Bx, B))
        ws = self._workspace(Bx, T, train=False)
        self._encode(x, spk[:Bx].contiguous(), ws, save=False)
        if Bx == B:
            return ws['cond'].clone()
        cond = ws['cond'].repeat(B, 1, 1)
        K.speaker_tile_fwd(self.P['speaker_embedding'] if self.spk_table else self.onehot, spk, cond,
                           cond_bstride=self.Cc * ws['Tz'], row0=self.D, Cs=self.Cs_eff, Tz=ws['Tz'])
        return cond

    def free_workspaces(self):
        """Drop the cached (B, T) workspaces (a long-running host changing shapes would otherwise keep them all)."""
        self._ws.clear()

    def state_dict(self):
        self.finish_steps()
        # x3_scale: the guarded engine's power-of-two plane scales (measured by the last step, used by the next): with them a
        # resumed run continues exactly as the uninterrupted one would (without them its first step runs on the start-up scales)
        return {'flat': self.flat, 'ema': self.ema, 'adam_m': self.adam_m, 'adam_v': self.adam_v,
                'bn_mean': self.bn_mean, 'bn_var': self.bn_var, 'x3_scale': self.x3_scale,
                'global_step': torch.tensor(self.global_step, dtype=torch.int64)}

    def load_state_dict(self, sd):
        self.finish_steps()
        for k in ('flat', 'ema', 'adam_m', 'adam_v', 'bn_mean', 'bn_var'):
            getattr(self, k).copy_(sd[k].to(self.dev))
        if 'x3_scale' in sd and tuple(sd['x3_scale'].shape) == tuple(self.x3_scale.shape):      # (absent in round-2 files)
            self.x3_scale.copy_(sd['x3_scale'].to(self.dev))
        self.global_step = int(sd['global_step'])

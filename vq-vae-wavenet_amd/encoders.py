"""Alternative encoders of the reference on the libvqwave engines.

`EncoderMagenta` mirrors Encoder/encoder.py:29-63 (class Encoder_Magenta): shift + mu-law,
causal k=5 conv, then 6 x [1x1 stride-2 conv -> gate / filter k=5 dilated (1,2,4,8,16,16)
causal convs -> tanh(gate) * sigmoid(filter) -> d + 1x1], then a 1x1 to latent_dim.  Every conv is
`conv1d_v2` (wavenet_ops.py:59-90), i.e. the same implicit-GEMM engine as the decoder: the
gate/filter pair is ONE launch with the GATE epilogue (kernels stored side by side), the
residual add is the ACCUM_SPLIT epilogue, the stride-2 1x1 uses in_stride=2 staging.
(`Encoder_64` lives in model.py.)
"""
import math
from collections import OrderedDict

import torch

from . import _alloc as A
from . import kernels as K


class EncoderMagenta:
    FILTERS = 128
    KS = 5
    DIL = [1, 2, 4, 8, 16, 16]     # encoder.py:33

    def __init__(self, latent_dim):
        self.D = latent_dim

    @staticmethod
    def latent_len(T):
        if T % 64 != 0:
            raise ValueError('length must be a multiple of 64 for Encoder_Magenta (got %d)' % T)
        return T // 64

    # ------------------------------------------------------------------ parameters
    def segments(self):
        F, D, L, k = self.FILTERS, self.D, len(self.DIL), self.KS
        seg = OrderedDict()
        seg['mag_pre_w'] = (k, F)            # encoder/preprocess/kernel [5,1,F]
        seg['mag_pre_b'] = (F,)
        seg['mag_d_w'] = (L, F, F)           # .../dilated/kernel [1,F,F] (stride 2)
        seg['mag_d_b'] = (L, F)
        seg['mag_gf_w'] = (L, k, F, 2 * F)   # gate | filter kernels side by side
        seg['mag_gf_b'] = (L, 2 * F)
        seg['mag_r_w'] = (L, F, F)           # .../residual/kernel
        seg['mag_r_b'] = (L, F)
        seg['mag_post_w'] = (F, D)           # encoder/postprocess/kernel
        seg['mag_post_b'] = (D,)
        return seg

    def init(self, P, uus):
        F, D, L, k = self.FILTERS, self.D, len(self.DIL), self.KS
        P['mag_pre_w'].copy_(uus((k, F), k, 1.0))
        P['mag_d_w'].copy_(uus((L, F, F), F, 1.0))
        P['mag_gf_w'].copy_(uus((L, k, F, 2 * F), k * F, 1.0))
        P['mag_r_w'].copy_(uus((L, F, F), F, 1.0))
        P['mag_post_w'].copy_(uus((F, D), F, 1.0))

    def named(self, V, out):
        F, D, k = self.FILTERS, self.D, self.KS
        out['encoder/preprocess/kernel'] = V['mag_pre_w'].reshape(k, 1, F)
        out['encoder/preprocess/bias'] = V['mag_pre_b']
        for i in range(len(self.DIL)):
            s = 'encoder/cycle_1/layer_%d' % (i + 1)
            out[s + '/dilated/kernel'] = V['mag_d_w'][i].unsqueeze(0)
            out[s + '/dilated/bias'] = V['mag_d_b'][i]
            out[s + '/gate/kernel'] = V['mag_gf_w'][i][:, :, :F]
            out[s + '/gate/bias'] = V['mag_gf_b'][i][:F]
            out[s + '/filter/kernel'] = V['mag_gf_w'][i][:, :, F:]
            out[s + '/filter/bias'] = V['mag_gf_b'][i][F:]
            out[s + '/residual/kernel'] = V['mag_r_w'][i].unsqueeze(0)
            out[s + '/residual/bias'] = V['mag_r_b'][i]
        out['encoder/postprocess/kernel'] = V['mag_post_w'].unsqueeze(0)
        out['encoder/postprocess/bias'] = V['mag_post_b']

    def scratch(self, dev):
        F, D, L, k = self.FILTERS, self.D, len(self.DIL), self.KS
        return {'mag_d_w': A.empty(L, F, F, device=dev), 'mag_gf_w': A.empty(L, k, 2 * F, F, device=dev),
                'mag_r_w': A.empty(L, F, F, device=dev), 'mag_post_w': A.empty(D, F, device=dev)}

    def transpose(self, P, Tt):
        F, D, L, k = self.FILTERS, self.D, len(self.DIL), self.KS
        K.transpose(P['mag_d_w'], Tt['mag_d_w'], L, F, F)
        K.transpose(P['mag_gf_w'], Tt['mag_gf_w'], L * k, F, 2 * F)
        K.transpose(P['mag_r_w'], Tt['mag_r_w'], L, F, F)
        K.transpose(P['mag_post_w'], Tt['mag_post_w'], 1, F, D)

    def workspace(self, ws, B, T, dev, train=True):
        F, L = self.FILTERS, len(self.DIL)
        e = lambda *s: A.empty(*s, device=dev)  # noqa: E731
        Tl = [T // (2 ** (i + 1)) for i in range(L)]
        ws['m_Tl'] = Tl
        ws['m_en'] = [e(B, F, T)] + [e(B, F, t) for t in Tl]     # en_0 .. en_6
        ws['m_dd'] = [e(B, F, t) for t in Tl]
        ws['m_gated'] = [e(B, F, t) for t in Tl]
        if not train:       # forward only (model.encode): nothing is saved for a backward pass
            ws['m_th'] = ws['m_sg'] = [None] * L
            return
        ws['m_th'] = [e(B, F, t) for t in Tl]
        ws['m_sg'] = [e(B, F, t) for t in Tl]
        ws['m_den'] = [e(B, F, T)] + [e(B, F, t) for t in Tl]    # gradients w.r.t. en_i
        ws['m_dpre'] = [e(B, 2 * F, t) for t in Tl]
        ws['m_ddd'] = [e(B, F, t) for t in Tl]

    # ------------------------------------------------------------------ forward / backward
    def forward(self, x, ws, P, Tt, save=True):
        """x [B][T] -> ws['z_e'] [B][D][T/64] (encoder.py:38-63)."""
        F, D, k, B, T = self.FILTERS, self.D, self.KS, ws['B'], ws['T']
        K.wavenet_inputs(x, ws['inputs'], ws['labels'])                       # shift_right + mu_law_encode
        en = ws['m_en']
        K.conv_cin1_fwd(ws['inputs'], P['mag_pre_w'], P['mag_pre_b'], en[0], k=k, stride=1, offset=-(k - 1))
        Tin = T
        for i, d in enumerate(self.DIL):
            To = ws['m_Tl'][i]
            dd = ws['m_dd'][i]
            K.conv_gemm(x0=en[i], w=P['mag_d_w'][i], bias=P['mag_d_b'][i], out0=dd, B=B, T_in=Tin, T_out=To, M=F,
                        C0=F, in_stride=2, taps=[0])                                               # 'dilated'
            K.conv_gemm(x0=dd, w=P['mag_gf_w'][i], bias=P['mag_gf_b'][i], out0=ws['m_gated'][i],
                        save0=ws['m_th'][i] if save else None, save1=ws['m_sg'][i] if save else None, B=B, T_in=To,
                        T_out=To, M=2 * F, C0=F, taps=[-(k - 1 - j) * d for j in range(k)], epilogue=K.EPI_GATE)
            K.conv_gemm(x0=ws['m_gated'][i], w=P['mag_r_w'][i], bias=P['mag_r_b'][i], out0=en[i + 1], out1=en[i + 1],
                        aux1=dd, B=B, T_in=To, T_out=To, M=F, M0=0, C0=F, taps=[0], epilogue=K.EPI_ACCUM_SPLIT)
            Tin = To
        K.conv_gemm(x0=en[-1], w=P['mag_post_w'], bias=P['mag_post_b'], out0=ws['z_e'], B=B, T_in=Tin, T_out=Tin, M=D,
                    C0=F, taps=[0])

    def backward(self, x, ws, P, G, Tt):
        """ws['dz'] = d loss / d z_e -> parameter gradients (accumulated into G)."""
        F, D, k, B, T = self.FILTERS, self.D, self.KS, ws['B'], ws['T']
        L = len(self.DIL)
        en, den = ws['m_en'], ws['m_den']
        dz, Tz = ws['dz'], ws['Tz']
        K.wgrad_gemm(p=en[L], q0=dz, dw=G['mag_post_w'], B=B, T_q=Tz, T_p=Tz, Cp=F, Q0=D, taps=[0])
        K.rowsum(dz, total=G['mag_post_b'])
        K.conv_gemm(x0=dz, w=Tt['mag_post_w'], out0=den[L], B=B, T_in=Tz, T_out=Tz, M=F, C0=D, taps=[0])
        for i in range(L - 1, -1, -1):
            d, To = self.DIL[i], ws['m_Tl'][i]
            Tin = ws['m_Tl'][i - 1] if i > 0 else T
            g_out, dpre, ddd = den[i + 1], ws['m_dpre'][i], ws['m_ddd'][i]
            # en_{i+1} = dd + conv1x1(gated)
            K.wgrad_gemm(p=ws['m_gated'][i], q0=g_out, dw=G['mag_r_w'][i], B=B, T_q=To, T_p=To, Cp=F, Q0=F, taps=[0])
            K.rowsum(g_out, total=G['mag_r_b'][i])
            K.conv_gemm(x0=g_out, w=Tt['mag_r_w'][i], out0=dpre, aux0=ws['m_th'][i], aux1=ws['m_sg'][i], B=B, T_in=To,
                        T_out=To, M=F, C0=F, taps=[0], epilogue=K.EPI_GATE_BWD)
            K.wgrad_gemm(p=ws['m_dd'][i], q0=dpre, dw=G['mag_gf_w'][i], B=B, T_q=To, T_p=To, Cp=F, Q0=2 * F,
                         taps=[-(k - 1 - j) * d for j in range(k)])
            K.rowsum(dpre, total=G['mag_gf_b'][i])
            K.conv_gemm(x0=dpre, w=Tt['mag_gf_w'][i], out0=ddd, out1=ddd, aux1=g_out, B=B, T_in=To, T_out=To, M=F,
                        M0=0, C0=2 * F, taps=[(k - 1 - j) * d for j in range(k)], epilogue=K.EPI_ACCUM_SPLIT)
            # dd = conv1x1(en_i, stride 2)
            K.wgrad_gemm(p=en[i], q0=ddd, dw=G['mag_d_w'][i], B=B, T_q=To, T_p=Tin, Cp=F, Q0=F, p_stride=2, taps=[0])
            K.rowsum(ddd, total=G['mag_d_b'][i])
            den[i].zero_()                                       # odd time steps receive no gradient
            K.conv_gemm(x0=ddd, w=Tt['mag_d_w'][i], out0=den[i], B=B, T_in=To, T_out=To, M=F, C0=F, taps=[0],
                        out_tstride=2, out_toffset=0, T_store=Tin)
        K.conv_cin1_wgrad(ws['inputs'], den[0], G['mag_pre_w'], k=k, stride=1, offset=-(k - 1))
        K.rowsum(den[0], total=G['mag_pre_b'])


def same_pads(n, k, s):
    """TF 'SAME' padding (left, right)."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def mel_weight_matrix(num_mel_bins=80, num_spectrogram_bins=201, sample_rate=16000, lower_edge_hertz=20.0,
                      upper_edge_hertz=8000.0):
    """tf.contrib.signal.linear_to_mel_weight_matrix as called by encoder_ops.py:30-36 (HTK mel
    scale, triangular filters, DC bin row zero) -> float32 [num_spectrogram_bins][num_mel_bins]."""
    import numpy as np
    mel = lambda f: 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)  # noqa: E731
    linear = np.linspace(0.0, sample_rate / 2.0, num_spectrogram_bins)[1:]
    spec = mel(linear)[:, None]
    edges = np.linspace(mel(lower_edge_hertz), mel(upper_edge_hertz), num_mel_bins + 2)
    lo, ce, up = edges[:-2][None], edges[1:-1][None], edges[2:][None]
    w = np.maximum(0.0, np.minimum((spec - lo) / (ce - lo), (up - spec) / (up - ce)))
    return torch.from_numpy(np.pad(w, [[1, 0], [0, 0]]).astype(np.float32))


class Encoder2019:
    """Encoder/encoder.py:66-98 (class Encoder_2019) + encoder_ops.py:14-70: MFCC-13 front end
    (vqw_mfcc), conv_3_768 -> conv_3_768 + residual -> strided_conv_4_768 -> 2 x (conv + residual)
    -> 4 x (`relu + relu`, i.e. 2*relu(conv): encoder.py:91-93) -> linear_64.  Keras layer names
    conv1d, conv1d_1 .. conv1d_9 in creation order.  The 13 MFCC channels are padded to 16 (zero
    channels / zero kernel rows) for the conv engine.  Needs T % 320 == 0."""
    F = 768
    CPAD = 16
    K3 = {1: 0, 3: 1, 4: 2, 5: 3, 6: 4, 7: 5, 8: 6}      # conv index -> slot in the stacked k=3 kernels

    def __init__(self, latent_dim):
        self.D = latent_dim

    def segments(self):
        F, D = self.F, self.D
        seg = OrderedDict()
        seg['e19_w0'] = (3, self.CPAD, F)        # conv1d/kernel [3,13,768] (+3 zero rows)
        seg['e19_wk3'] = (7, 3, F, F)            # conv1d_{1,3,4,5,6,7,8}/kernel
        seg['e19_w2'] = (4, F, F)                # conv1d_2/kernel (stride 2)
        seg['e19_b'] = (9, F)
        seg['e19_w9'] = (F, D)                   # conv1d_9/kernel [1,768,D]
        seg['e19_b9'] = (D,)
        return seg

    def init(self, P, glorot):
        F, D = self.F, self.D
        P['e19_w0'].zero_()
        P['e19_w0'][:, :13, :].copy_(glorot((3, 13, F), 3, 13, F))
        P['e19_wk3'].copy_(glorot((7, 3, F, F), 3, F, F))
        P['e19_w2'].copy_(glorot((4, F, F), 4, F, F))
        P['e19_w9'].copy_(glorot((F, D), 1, F, D))

    def named(self, V, out):
        F, D = self.F, self.D
        sfx = lambda i: '' if i == 0 else '_%d' % i  # noqa: E731
        out['encoder/conv1d/kernel'] = V['e19_w0'][:, :13, :]
        for i, slot in self.K3.items():
            out['encoder/conv1d_%d/kernel' % i] = V['e19_wk3'][slot]
        out['encoder/conv1d_2/kernel'] = V['e19_w2']
        for i in range(9):
            out['encoder/conv1d%s/bias' % sfx(i)] = V['e19_b'][i]
        out['encoder/conv1d_9/kernel'] = V['e19_w9'].unsqueeze(0)
        out['encoder/conv1d_9/bias'] = V['e19_b9']

    def scratch(self, dev):
        F, D = self.F, self.D
        return {'e19_wk3': A.empty(7, 3, F, F, device=dev), 'e19_w2': A.empty(4, F, F, device=dev),
                'e19_w9': A.empty(D, F, device=dev), 'e19_mel': mel_weight_matrix().to(dev),
                'e19_ones': torch.ones(F, device=dev), 'e19_twos': torch.full((F,), 2.0, device=dev),
                'e19_zeros': torch.zeros(F, device=dev)}

    def transpose(self, P, Tt):
        F, D = self.F, self.D
        K.transpose(P['e19_wk3'], Tt['e19_wk3'], 21, F, F)
        K.transpose(P['e19_w2'], Tt['e19_w2'], 4, F, F)
        K.transpose(P['e19_w9'], Tt['e19_w9'], 1, F, D)

    @staticmethod
    def latent_len(T):
        if T % 320 != 0:
            raise ValueError('length must be a multiple of 320 for Encoder_2019 (got %d)' % T)
        return T // 320

    def workspace(self, ws, B, T, dev, train=True):
        F = self.F
        e = lambda *s: A.empty(*s, device=dev)  # noqa: E731
        Fr = T // 160
        Tz = Fr // 2
        ws['e_Fr'] = Fr
        ws['e_mf'] = e(B, self.CPAD, Fr)
        Ts = [Fr, Fr] + [Tz] * 7
        ws['e_T'] = Ts
        ws['e_a'] = [e(B, F, t) for t in Ts]          # layer outputs a_0 .. a_8
        if not train:       # forward only (model.encode)
            ws['e_r'] = [None] * len(Ts)
            return
        ws['e_r'] = [e(B, F, t) for t in Ts]          # relu outputs before residual / doubling
        ws['e_da'] = [e(B, F, t) for t in Ts]         # gradients w.r.t. a_i
        ws['e_dc'] = [e(B, F, Fr), e(B, F, Tz)]       # masked conv-output gradients (per resolution)

    def forward(self, x, ws, P, Tt, save=True):
        F, D, B = self.F, self.D, ws['B']
        Fr, Tz = ws['e_Fr'], ws['Tz']
        a, r = ws['e_a'], ws['e_r']
        K.mfcc(x, Tt['e19_mel'], ws['e_mf'])                                        # encoder_ops.py:14-43
        k3 = [-1, 0, 1]                                                            # 'same', k=3: pads (1,1)
        K.conv_gemm(x0=ws['e_mf'], w=P['e19_w0'], bias=P['e19_b'][0], out0=a[0], B=B, T_in=Fr, T_out=Fr, M=F,
                    C0=self.CPAD, taps=k3, out_relu=True)
        K.conv_gemm(x0=a[0], w=P['e19_wk3'][0], bias=P['e19_b'][1], out0=a[1], save0=r[1] if save else None, aux1=a[0], B=B, T_in=Fr,
                    T_out=Fr, M=F, C0=F, taps=k3, out_relu=True)
        pl, _ = same_pads(Fr, 4, 2)
        K.conv_gemm(x0=a[1], w=P['e19_w2'], bias=P['e19_b'][2], out0=a[2], B=B, T_in=Fr, T_out=Tz, M=F, C0=F,
                    in_stride=2, taps=[j - pl for j in range(4)], out_relu=True)
        for i in (3, 4):
            K.conv_gemm(x0=a[i - 1], w=P['e19_wk3'][self.K3[i]], bias=P['e19_b'][i], out0=a[i], save0=r[i] if save else None,
                        aux1=a[i - 1], B=B, T_in=Tz, T_out=Tz, M=F, C0=F, taps=k3, out_relu=True)
        for i in (5, 6, 7, 8):                                                     # net = relu + relu
            K.conv_gemm(x0=a[i - 1], w=P['e19_wk3'][self.K3[i]], bias=P['e19_b'][i], out0=a[i], save0=r[i] if save else None,
                        scale=Tt['e19_twos'], shift=Tt['e19_zeros'], B=B, T_in=Tz, T_out=Tz, M=F, C0=F, taps=k3,
                        out_relu=True)
        K.conv_gemm(x0=a[8], w=P['e19_w9'], bias=P['e19_b9'], out0=ws['z_e'], B=B, T_in=Tz, T_out=Tz, M=D, C0=F,
                    taps=[0])

    def backward(self, x, ws, P, G, Tt):
        F, D, B = self.F, self.D, ws['B']
        Fr, Tz = ws['e_Fr'], ws['Tz']
        a, r, da, dc = ws['e_a'], ws['e_r'], ws['e_da'], ws['e_dc']
        dz = ws['dz']
        k3, k3b = [-1, 0, 1], [1, 0, -1]
        ones, twos = Tt['e19_ones'], Tt['e19_twos']
        K.wgrad_gemm(p=a[8], q0=dz, dw=G['e19_w9'], B=B, T_q=Tz, T_p=Tz, Cp=F, Q0=D, taps=[0])
        K.rowsum(dz, total=G['e19_b9'])
        K.conv_gemm(x0=dz, w=Tt['e19_w9'], out0=da[8], B=B, T_in=Tz, T_out=Tz, M=F, C0=D, taps=[0])
        for i in (8, 7, 6, 5):                       # a_i = 2*relu(conv_i(a_{i-1}))
            K.bn_relu_bwd(da[i], r[i], twos, da[i])
            K.rowsum(da[i], total=G['e19_b'][i])
            K.wgrad_gemm(p=a[i - 1], q0=da[i], dw=G['e19_wk3'][self.K3[i]], B=B, T_q=Tz, T_p=Tz, Cp=F, Q0=F, taps=k3)
            K.conv_gemm(x0=da[i], w=Tt['e19_wk3'][self.K3[i]], out0=da[i - 1], B=B, T_in=Tz, T_out=Tz, M=F, C0=F,
                        taps=k3b)
        for i in (4, 3):                             # a_i = relu(conv_i(a_{i-1})) + a_{i-1}
            K.bn_relu_bwd(da[i], r[i], ones, dc[1])
            K.rowsum(dc[1], total=G['e19_b'][i])
            K.wgrad_gemm(p=a[i - 1], q0=dc[1], dw=G['e19_wk3'][self.K3[i]], B=B, T_q=Tz, T_p=Tz, Cp=F, Q0=F, taps=k3)
            K.conv_gemm(x0=dc[1], w=Tt['e19_wk3'][self.K3[i]], out0=da[i - 1], out1=da[i - 1], aux1=da[i], B=B,
                        T_in=Tz, T_out=Tz, M=F, M0=0, C0=F, taps=k3b, epilogue=K.EPI_ACCUM_SPLIT)
        # a_2 = relu(strided_conv_4(a_1))
        K.bn_relu_bwd(da[2], a[2], ones, da[2])
        K.rowsum(da[2], total=G['e19_b'][2])
        pl, _ = same_pads(Fr, 4, 2)
        K.wgrad_gemm(p=a[1], q0=da[2], dw=G['e19_w2'], B=B, T_q=Tz, T_p=Fr, Cp=F, Q0=F, p_stride=2,
                     taps=[j - pl for j in range(4)])
        for p in (0, 1):
            j0 = (p + pl) % 2
            js = list(range(j0, 4, 2))
            K.conv_gemm(x0=da[2], w=Tt['e19_w2'][j0:], w_tap_stride=2 * F * F, out0=da[1], B=B, T_in=Tz,
                        T_out=(Fr - p + 1) // 2, M=F, C0=F, taps=[(p + pl - j) // 2 for j in js], out_tstride=2,
                        out_toffset=p, T_store=Fr)
        # a_1 = relu(conv_1(a_0)) + a_0
        K.bn_relu_bwd(da[1], r[1], ones, dc[0])
        K.rowsum(dc[0], total=G['e19_b'][1])
        K.wgrad_gemm(p=a[0], q0=dc[0], dw=G['e19_wk3'][0], B=B, T_q=Fr, T_p=Fr, Cp=F, Q0=F, taps=k3)
        K.conv_gemm(x0=dc[0], w=Tt['e19_wk3'][0], out0=da[0], out1=da[0], aux1=da[1], B=B, T_in=Fr, T_out=Fr, M=F,
                    M0=0, C0=F, taps=k3b, epilogue=K.EPI_ACCUM_SPLIT)
        # a_0 = relu(conv_0(mfcc))
        K.bn_relu_bwd(da[0], a[0], ones, da[0])
        K.rowsum(da[0], total=G['e19_b'][0])
        K.wgrad_gemm(p=ws['e_mf'], q0=da[0], dw=G['e19_w0'], B=B, T_q=Fr, T_p=Fr, Cp=self.CPAD, Q0=F, taps=k3)


# the reference's class names (Encoder/encoder.py:29,66)
Encoder_Magenta = EncoderMagenta
Encoder_2019 = Encoder2019

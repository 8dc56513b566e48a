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

from . import kernels as K


class EncoderMagenta:
    FILTERS = 128
    KS = 5
    DIL = [1, 2, 4, 8, 16, 16]     # encoder.py:33

    def __init__(self, latent_dim):
        self.D = latent_dim

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
        return {'mag_d_w': torch.empty(L, F, F, device=dev), 'mag_gf_w': torch.empty(L, k, 2 * F, F, device=dev),
                'mag_r_w': torch.empty(L, F, F, device=dev), 'mag_post_w': torch.empty(D, F, device=dev)}

    def transpose(self, P, Tt):
        F, D, L, k = self.FILTERS, self.D, len(self.DIL), self.KS
        K.transpose(P['mag_d_w'], Tt['mag_d_w'], L, F, F)
        K.transpose(P['mag_gf_w'], Tt['mag_gf_w'], L * k, F, 2 * F)
        K.transpose(P['mag_r_w'], Tt['mag_r_w'], L, F, F)
        K.transpose(P['mag_post_w'], Tt['mag_post_w'], 1, F, D)

    def workspace(self, ws, B, T, dev):
        F, L = self.FILTERS, len(self.DIL)
        e = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
        Tl = [T // (2 ** (i + 1)) for i in range(L)]
        ws['m_Tl'] = Tl
        ws['m_en'] = [e(B, F, T)] + [e(B, F, t) for t in Tl]     # en_0 .. en_6
        ws['m_dd'] = [e(B, F, t) for t in Tl]
        ws['m_gated'] = [e(B, F, t) for t in Tl]
        ws['m_th'] = [e(B, F, t) for t in Tl]
        ws['m_sg'] = [e(B, F, t) for t in Tl]
        ws['m_den'] = [e(B, F, T)] + [e(B, F, t) for t in Tl]    # gradients w.r.t. en_i
        ws['m_dpre'] = [e(B, 2 * F, t) for t in Tl]
        ws['m_ddd'] = [e(B, F, t) for t in Tl]

    # ------------------------------------------------------------------ forward / backward
    def forward(self, x, ws, P, save=True):
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

/* vqwave.h -- C ABI of libvqwave.so: the MI355X (gfx950) hot path of VQ-VAE-WaveNet.
 *
 * The reference (StanislavParovoy/VQ-VAE-WaveNet) is pure Python on TensorFlow 1.x and
 * has no FFI of its own; every entry point below replaces the TensorFlow-runtime work
 * behind one reference call site, cited as `file:line` of /root/reference.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + explicit sizes + a hipStream_t passed as void*.
 *   - every function returns 0 on success, non-zero on error (never throws, never
 *     aborts); vqw_last_error() returns the message of the calling thread's last error.
 *   - asynchronous on the given stream, no hidden synchronisation, no allocation;
 *     the caller owns every buffer (except the opaque AR-decoder handle).
 *   - activations are (batch, channel, time) fp32, time contiguous: x[b][c][t].
 *     The reference's op surface is channels-last [B,T,C]; the Python mirror
 *     transposes at that surface only.
 *   - conv kernels keep the reference's variable layout kernel[k][Cin][Cout]
 *     (wavenet_ops.py:66-69), row-major, i.e. one [Cin][Cout] matrix per tap.
 */
#ifndef VQWAVE_H
#define VQWAVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vqw_stream_t; /* hipStream_t */

#define VQW_ABI_VERSION 1
#define VQW_MAX_TAPS 8

const char* vqw_last_error(void);
int vqw_abi_version(void);

/* ------------------------------------------------------------------------------------
 * mu-law companding -- reference mu_law_ops.py:5-31, wavenet_ops.py:9-14
 * ---------------------------------------------------------------------------------- */
/* y = sign(x) log1p(255|x|)/log1p(255), x clipped to [-1,1]   (mu_law_ops.py:6-8) */
int vqw_mu_law_encode_f32(const float* x, float* y, size_t n, vqw_stream_t s);
/* integer labels 0..255, bit-exact vs the fp32 formula via a threshold table
 * (mu_law_ops.py:11) */
int vqw_mu_law_encode_i32(const float* x, int32_t* y, size_t n, vqw_stream_t s);
/* idx (float holding 0..256) -> sample in [-1,1]   (mu_law_ops.py:26-31) */
int vqw_mu_law_decode_f32(const float* idx, float* x, size_t n, vqw_stream_t s);
/* Fused decoder front end (wavenet.py:33-37): labels = encode_i32(x),
 * inputs = encode_f32(shift_right(x)); x is [B][T]. Either output may be NULL. */
int vqw_wavenet_inputs(const float* x, float* inputs, int32_t* labels, int B, int T,
                       vqw_stream_t s);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution engine (fp32 MFMA) -- replaces tf.pad + tf.nn.conv2d + bias
 * (wavenet_ops.py:81-89), add_condition (93-101), the gate (112-113), the 1x1 skip /
 * residual convs + accumulation (132-136, wavenet.py:72-73), Keras Conv1D + BatchNorm
 * (encoder.py:15-25) and their gradients (TF autodiff: Conv2DBackpropInput).
 *
 *   acc[b][m][t] = sum_{j<ntaps} sum_{c<C0+C1} w[j][c][m] * x[b][c][in_stride*t + tap_shift[j]]
 *   (x = x0 for c < C0, x1 for c >= C0; reads outside [0,T_in) are zero)
 * followed by one of the epilogues below.
 * ---------------------------------------------------------------------------------- */
enum {
    VQW_EPI_STORE = 0,       /* v=acc+bias+cond; r=out_relu?max(v,0):v; save0=r;
                                out=(scale?scale*r+shift:r) + (aux1?aux1:0)              */
    VQW_EPI_ACCUM_SPLIT = 1, /* v=acc+bias; rows<M0: out0+=v; rows>=M0: out1=aux1+v      */
    VQW_EPI_GATE = 2,        /* M=2H rows (f,g pairs); out0=tanh(vf)*sigmoid(vg) [H rows];
                                save0=tanh, save1=sigmoid                                */
    VQW_EPI_GATE_BWD = 3,    /* M=H rows; v=acc (d gated); aux0=tanh, aux1=sigmoid;
                                out0[row]=v*s*(1-t^2), out0[H+row]=v*t*s*(1-s)          */
    VQW_EPI_MASK = 4         /* v=acc*(scale?scale[row]:1); out0 = aux0>0 ? v : 0        */
};

typedef struct vqw_conv_desc {
    int32_t B, T_out, T_in;
    int32_t M;        /* GEMM rows (output channels; 2H for GATE)                        */
    int32_t C0, C1;   /* input channels taken from x0 / x1 (multiples of 16; C1 may be 0) */
    int32_t ntaps, in_stride;
    int32_t tap_shift[VQW_MAX_TAPS];
    int32_t ldw;      /* row length of w (>= M, multiple of 4)                           */
    int32_t in_relu;  /* relu applied to x while it is staged                            */
    int32_t epilogue;
    int32_t out_relu;
    int32_t M0;       /* rows [0,M0) go to out0, [M0,M) to out1 (STORE / ACCUM_SPLIT)     */
    int32_t out_tstride, out_toffset, T_store; /* store index = out_tstride*t+out_toffset,
                                                  output rows have T_store elements      */
    int32_t cond_T;   /* 0 = none; cond[b][row][t / (T_out/cond_T)] is added             */
    int32_t tile;     /* 0 = auto; else 10*MT+NT (per-wave 32x32 tile counts), optionally
                         + 10000*n: the last n main-tile columns of every row are computed
                         by tiles half as wide at the end of the same grid (load balance;
                         chosen automatically when n is not given)                        */
    int32_t split_k;  /* 0 = auto, 1 = none, n > 1: every tile's K range is cut over n blocks that
                         meet by fp32 atomics (plain STORE epilogue only; out0 is zeroed first);
                         -n: the same, but the CALLER has zeroed out0 (strided outputs written by
                         several launches)                                                       */
    int64_t cond_bstride; /* batch stride of cond in floats                              */
    int64_t w_tap_stride; /* floats between consecutive taps of w; 0 = (C0+C1)*ldw       */
    const float *x0, *x1, *w, *bias, *cond, *scale, *shift, *aux0, *aux1;
    float *out0, *out1, *save0, *save1;
} vqw_conv_desc;

int vqw_conv_gemm(const vqw_conv_desc* d, vqw_stream_t s);

/* Weight-gradient engine (TF autodiff: Conv2DBackpropFilter), split-K with fp32 atomics:
 *   dw[j][c][o] += sum_{b,t<T_q} p[b][c][p_stride*t + tap_shift[j]] * q[b][o][t]
 * (q = q0 for o < Q0, q1 above).  dw must be pre-zeroed / holds the running sum. */
typedef struct vqw_wgrad_desc {
    int32_t B, T_q, T_p;
    int32_t Cp;       /* rows of dw (input channels)                                     */
    int32_t Q0, Q1;   /* columns of dw taken from q0 / q1                                */
    int32_t ntaps, p_stride;
    int32_t tap_shift[VQW_MAX_TAPS];
    int32_t p_relu;   /* relu applied to p while it is staged                            */
    int32_t lddw;
    int32_t splits;   /* 0 = auto: time chunks per batch element                         */
    int64_t dw_tap_stride;
    const float *p, *q0, *q1;
    float* dw;
} vqw_wgrad_desc;

int vqw_wgrad_gemm(const vqw_wgrad_desc* d, vqw_stream_t s);

/* Named wrappers with the reference's argument meaning --------------------------------
 * conv1d_v2 (wavenet_ops.py:59-90): causal left pad d*(k-1), stride, dilation, +bias.
 * x [B][Cin][T], w [k][Cin][Cout], y [B][Cout][ceil(T/stride)].                       */
int vqw_causal_conv1d_fwd(const float* x, const float* w, const float* bias, float* y,
                          int B, int Cin, int Cout, int T, int k, int dilation, int stride,
                          vqw_stream_t s);
/* dx[b][c][t] = sum_j sum_o wT[j][o][c] * dy[b][o][t + (k-1-j)*dilation]; wT = per-tap
 * transposed kernel [k][Cout][Cin] (see vqw_transpose).  stride 1 only.               */
int vqw_causal_conv1d_dgrad(const float* dy, const float* wT, float* dx, int B, int Cin,
                            int Cout, int T, int k, int dilation, vqw_stream_t s);
int vqw_causal_conv1d_wgrad(const float* x, const float* dy, float* dw, int B, int Cin,
                            int Cout, int T, int k, int dilation, vqw_stream_t s);
/* Keras Conv1D (encoder.py:15-19, encoder_ops.py:46-70) with explicit TF-SAME pads:
 * y[b][o][t] = bias[o] + sum_j sum_c w[j][c][o] x[b][c][stride*t + j - pad_left].     */
int vqw_conv1d_same_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                        int Cin, int Cout, int T_in, int T_out, int k, int stride,
                        int pad_left, int relu, vqw_stream_t s);
int vqw_conv1d_same_dgrad(const float* dy, const float* wT, float* dx, int B, int Cin,
                          int Cout, int T_in, int T_out, int k, int stride, int pad_left,
                          vqw_stream_t s);
int vqw_conv1d_same_wgrad(const float* x, const float* dy, float* dw, int B, int Cin,
                          int Cout, int T_in, int T_out, int k, int stride, int pad_left,
                          vqw_stream_t s);
/* 1x1 conv == channel-mixing GEMM (wavenet_ops.py:132-136, `linear` 147-160).
 * accumulate != 0: y += result.                                                        */
int vqw_pointwise_gemm_fwd(const float* x, const float* w, const float* bias, float* y,
                           int B, int Cin, int Cout, int T, int relu_in, int accumulate,
                           vqw_stream_t s);
int vqw_pointwise_gemm_dgrad(const float* dy, const float* wT, float* dx, int B, int Cin,
                             int Cout, int T, vqw_stream_t s);
int vqw_pointwise_gemm_wgrad(const float* x, const float* dy, float* dw, int B, int Cin,
                             int Cout, int T, int relu_in, vqw_stream_t s);

/* ------------------------------------------------------------------------------------
 * Small convs with Cin == 1 (decoder preprocess wavenet.py:42-44, encoder.py:15 layer 1):
 *   v = bias[f] + sum_j w[j][f] * x[b][stride*t + j + offset];   r = relu ? max(v,0) : v
 *   save_r (optional) = r;  out = scale ? scale[f]*r + shift[f] : r
 * x [B][T_in], w [k][F], out [B][F][T_out].                                            */
int vqw_conv_cin1_fwd(const float* x, const float* w, const float* bias, const float* scale,
                      const float* shift, float* out, float* save_r, int B, int T_in,
                      int T_out, int F, int k, int stride, int offset, int relu,
                      vqw_stream_t s);
/* dw[j][f] += sum_{b,t} x[b][stride*t+j+offset] * dout[b][f][t] */
int vqw_conv_cin1_wgrad(const float* x, const float* dout, float* dw, int B, int T_in,
                        int T_out, int F, int k, int stride, int offset, vqw_stream_t s);

/* ------------------------------------------------------------------------------------
 * Row reductions over (b, t) of a [B][C][T] tensor (bias / BN gradients, and the
 * transpose of add_condition's nearest-neighbour upsampling, wavenet_ops.py:98-100):
 *   seg_out[b][c][t/seg] = sum over the segment of x*(y?y:1)   (optional, seg | T)
 *   total[c]            += alpha * sum_{b,t} x*(y?y:1)         (optional)               */
int vqw_rowsum(const float* x, const float* y, float* seg_out, float* total, float alpha,
               int B, int C, int T, int seg, vqw_stream_t s);

/* Backward of relu -> BatchNorm(inference affine) (encoder.py:19-20):
 *   dz[b][c][t] = dx[b][c][t] * scale[c] * (r ? (r[b][c][t] > 0) : 1);  dz may alias dx.    */
int vqw_bn_relu_bwd(const float* dx, const float* r, const float* scale, float* dz, int B,
                    int C, int T, vqw_stream_t s);
/* The same in one pass with the sums an encoder layer's backward needs (each optional, accumulated):
 *   dscale[c] += sum_{b,t} dx * y (y = r, or the BatchNorm input of a layer without relu; NULL = 1),
 *   dbeta[c] += sum_{b,t} dx,  dbias[c] += sum_{b,t} dz.                                                   */
int vqw_bn_relu_bwd_sums(const float* dx, const float* y, const float* r, const float* scale, float* dz, float* dscale,
                         float* dbeta, float* dbias, int B, int C, int T, vqw_stream_t s);

/* relu -> BatchNorm(inference affine) in place (encoder.py:15-20), the second pass of a split-K layer:
 *   r[b][c][t] = relu(x) (optional);  x := scale[c]*relu(x) + shift[c]  (scale NULL: x := relu(x))     */
int vqw_relu_bn_fwd(float* x, float* r, const float* scale, const float* shift, int B, int C, int T,
                    vqw_stream_t s);

/* MFCC front end of Encoder_2019 (encoder_ops.py:14-43): STFT (frame 400, step 160, periodic hann,
 * zero pad at the end: frames = ceil(T/160)) -> magnitude (201 bins) -> mel [201][n_mel] (device
 * matrix, tf linear_to_mel_weight_matrix) -> log(. + 1e-6) -> DCT-II * 2 / sqrt(2 n_mel) -> first n_keep
 * coefficients.  x [B][T] -> out [B][C_out][frames], channels >= n_keep are written as zero (padding
 * to the conv engine's multiple-of-16 channel count). */
int vqw_mfcc(const float* x, const float* mel, float* out, int B, int T, int frames, int n_mel,
             int n_keep, int C_out, vqw_stream_t s);

/* dst[b][c][r] = src[b][r][c]  (batched 2-D transpose; per-tap kernel transposes) */
int vqw_transpose(const float* src, float* dst, int batch, int rows, int cols,
                  vqw_stream_t s);

/* ------------------------------------------------------------------------------------
 * VQ nearest codebook entry -- model.py:57-74.
 * z_e [B][D][Tz]; emb [K][D].  dist = sum_{d=0..D-1} (z-e)^2 accumulated sequentially in
 * fp32 with separately rounded multiply and add; idx = lowest index of the minimum.
 * Outputs: idx int64 [B][Tz]; e_k [B][D][Tz]; z_q = z_e + (e_k - z_e) written to
 * zq[b*zq_bstride + d*Tz + t] (so it can land inside the decoder condition tensor);
 * mind [B][Tz] = winning distance.                                                     */
int vqw_vq_nearest_fwd(const float* z_e, const float* emb, int64_t* idx, float* e_k,
                       float* zq, int64_t zq_bstride, float* mind, int B, int D, int Tz,
                       int K, vqw_stream_t s);
/* Gradients of model.py:73,100,103 (Appendix A-5 of SURVEY.md):
 *   dz_e = dzq + cscale*(z_e - e_k);   demb[idx] += escale*(e_k - z_e)                  */
int vqw_vq_nearest_bwd(const float* z_e, const float* e_k, const int64_t* idx,
                       const float* dzq, int64_t dzq_bstride, float* dz_e, float* demb,
                       float cscale, float escale, int B, int D, int Tz, int K,
                       vqw_stream_t s);
/* Speaker path, model.py:22-27 + decoder_ops.py:39-43:
 * cond[b*cond_bstride + (row0+j)*Tz + t] = table[spk[b]][j] for all t.  table has n_speakers rows; an id
 * outside [0, n_speakers) reads row 0 (the reference's one_hot -> argmax of an all-zero row, model.py:22),
 * never memory outside the table.                                                        */
int vqw_speaker_tile_fwd(const float* table, const int64_t* spk, float* cond,
                         int64_t cond_bstride, int row0, int B, int Cs, int Tz, int n_speakers,
                         vqw_stream_t s);
/* dtable[spk[b]][j] += sum_t dcond[b][row0+j][t] */
int vqw_speaker_tile_bwd(const float* dcond, int64_t dcond_bstride, int row0,
                         const int64_t* spk, float* dtable, int B, int Cs, int Tz, int n_speakers,
                         vqw_stream_t s);

/* ------------------------------------------------------------------------------------
 * Softmax cross-entropy over the channel axis -- model.py:91-94.
 * logits [B][Q][T], labels int32 [B][T].  loss_sum[0] += sum_{b,t} CE;
 * dlogits (optional, may alias logits) = (softmax - onehot) * grad_scale;
 * probs (optional) = softmax.                                                          */
int vqw_softmax_xent(const float* logits, const int32_t* labels, float* dlogits,
                     float* probs, float* loss_sum, float grad_scale, int B, int Q, int T,
                     vqw_stream_t s);
/* The same op as the two entries SURVEY 8(b) names (tf.nn.sparse_softmax_cross_entropy_with_logits and its gradient,
 * model.py:91-94): _fwd adds the CE sum into loss_sum[0] (probs optional); _bwd writes (softmax - onehot) * grad_scale,
 * recomputed from the logits (dlogits may alias logits).                                */
int vqw_softmax_xent_fwd(const float* logits, const int32_t* labels, float* probs, float* loss_sum,
                         int B, int Q, int T, vqw_stream_t s);
int vqw_softmax_xent_bwd(const float* logits, const int32_t* labels, float* dlogits, float grad_scale,
                         int B, int Q, int T, vqw_stream_t s);

/* ------------------------------------------------------------------------------------
 * The decoder's local-condition projections -- add_condition, wavenet_ops.py:93-101: a 1x1 conv1d_v2 of the condition
 * per gated_cnn and for postprocess1 (wavenet.py:58-100).  All L + 1 kernels side by side are one matrix
 * w[Cc][Mall] (Mall = L * 2R + S); cond [B][Cc][Tz], out / dce [B][Mall][Tz], dcond [B][Cc][Tz].
 *   fwd:    out[b][m][t]   = sum_c w[c][m] cond[b][c][t]
 *   wgrad:  dw[c][m]      += sum_{b,t} cond[b][c][t] dce[b][m][t]     (TF Conv2DBackpropFilter); Cc <= 128, Tz % 4 == 0;
 *           dw is the zeroed gradient buffer (two batch ranges may meet in it by fp32 atomics)
 *   dgrad:  dcond[b][c][t] = sum_m w[c][m] dce[b][m][t]               (TF Conv2DBackpropInput); Cc <= 128, Mall % 4 == 0;
 *           scratch: vqw_cond_proj_dgrad_scratch_floats() floats (partial sums over ranges of m, added in a fixed order)
 * fp32 MFMA (v_mfma_f32_32x32x2_f32) fed straight from global memory, one tile per wave: a contraction of Cc ~ 80 over
 * 104-frame rows is too short for the conv engine's K loops and 128-column tiles.                                   */
int vqw_cond_proj_fwd(const float* cond, const float* w, float* out, int B, int Cc, int Mall, int Tz, vqw_stream_t s);
int vqw_cond_proj_wgrad(const float* cond, const float* dce, float* dw, int B, int Cc, int Mall, int Tz, vqw_stream_t s);
int vqw_cond_proj_dgrad(const float* w, const float* dce, float* dcond, float* scratch, int64_t scratch_floats,
                        int B, int Cc, int Mall, int Tz, vqw_stream_t s);
int vqw_cond_proj_dgrad_scratch_floats(int B, int Cc, int Mall, int Tz, int64_t* out);

/* ------------------------------------------------------------------------------------
 * Fused TF-1.x Adam + ExponentialMovingAverage step over a flat buffer --
 * model.py:116-128:
 *   g = grad*grad_scale; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   p -= lr_t * m / (sqrt(v) + eps);  ema -= (1-decay) * (ema - p)
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller.                           */
int vqw_adam_ema_step(float* param, const float* grad, float* m, float* v, float* ema,
                      size_t n, float lr_t, float beta1, float beta2, float eps,
                      float decay, float grad_scale, vqw_stream_t s);
/* The same step behind a device-side guard: with *skip != 0 (a device int32 read when the kernel RUNS) nothing is
 * changed.  The guarded fp16x3 engine enqueues its optimiser this way when the host reads the step's range flag one
 * step late (model.defer_guard): a flagged step and everything enqueued behind it leave parameters, Adam slots and EMA
 * shadows untouched and are then repeated in order.  skip NULL = vqw_adam_ema_step.                                */
int vqw_adam_ema_step_guarded(float* param, const float* grad, float* m, float* v, float* ema,
                              size_t n, float lr_t, float beta1, float beta2, float eps,
                              float decay, float grad_scale, const int32_t* skip, vqw_stream_t s);

/* ------------------------------------------------------------------------------------
 * Fast autoregressive generation -- wavenet.py:103-172, wavenet_ops.py:147-267,
 * generate.py:103-113, utils.py:13-46.  The FIFO queues become device ring buffers.
 * Weight layout: see vqw_ar_weights; all pointers are device pointers that must stay
 * valid for the lifetime of the handle.
 * ---------------------------------------------------------------------------------- */
typedef struct vqw_ar_weights {
    int32_t n_layers, kernel_size, R, S, Q, Cc, pre_k; /* residual(=dilation) filters R,
                                                          skip S, quantisation Q         */
    const int32_t* dilations;   /* HOST pointer, n_layers entries                        */
    const float* pre_w;         /* [pre_k][R]           decoder/preprocess/kernel        */
    const float* pre_b;         /* [R]                                                   */
    const float* skip0_w;       /* [R][S]               decoder/skip/kernel              */
    const float* skip0_b;       /* [S]                                                   */
    const float* const* gated_w;   /* HOST array of n_layers device ptrs [k][R][2R]      */
    const float* const* gated_b;   /* [2R]                                               */
    const float* const* cond_w;    /* [Cc][2R]  row stride cond_ld                       */
    const float* const* out_w;     /* [R][S+R]  skip|residual, row stride out_ld         */
    const float* const* out_b;     /* [S+R]                                              */
    int32_t cond_ld, out_ld;
    const float* post1_w;       /* [S][S]                                                */
    const float* post1_b;       /* [S]                                                   */
    const float* post1_cond_w;  /* [Cc][S] row stride post1_cond_ld                      */
    int32_t post1_cond_ld;
    const float* post2_w;       /* [S][Q]                                                */
    const float* post2_b;       /* [Q]                                                   */
} vqw_ar_weights;

typedef struct vqw_ar_decoder vqw_ar_decoder;

int vqw_ar_decode_create(vqw_ar_decoder** out, const vqw_ar_weights* w, int batch);
/* The same with the persistent kernel's decomposition chosen by the caller: channels_per_workgroup 4 (R/4 workgroups: the
 * fastest single handle) or 8 (R/8 workgroups: 8 handles x 32 workgroups hold a 256-CU chip, one handle per XCD), 0 = auto. */
int vqw_ar_decode_create_ex(vqw_ar_decoder** out, const vqw_ar_weights* w, int batch, int channels_per_workgroup);
/* zero the queues (generate.py:105 sess.run(init_ops)) and restart at sample 0 */
int vqw_ar_decode_reset(vqw_ar_decoder* h, vqw_stream_t s);
/* Generate `n_steps` samples.  encoding [B][Cc][Tz] (model.encoding, channel-major);
 * sample i uses encoding[:, :, (start+i)/ratio].  mode 0 = greedy (argmax), 1 = sample
 * with caller-supplied uniforms u[B][n_steps] (searchsorted(cumsum(p), u), utils.py:20-25).
 * Outputs: audio [B][n_steps] (mu-law decoded floats), indices int32 [B][n_steps]
 * (optional), probs_last [B][Q] (optional: probabilities of the final step).           */
int vqw_ar_decode_run(vqw_ar_decoder* h, const float* encoding, int Tz, int ratio,
                      int n_steps, int mode, const float* uniforms, float* audio,
                      int32_t* indices, float* probs_last, vqw_stream_t s);
/* The same, split: _run_async only enqueues (on the handle's own stream, ordered after the work already in `s`);
 * _wait blocks the host until the enqueued run is done -- only then may the outputs be used, on any stream -- and
 * reports a device-side failure (a spin-wait timeout of the persistent kernel).  vqw_ar_decode_run == _run_async
 * + _wait.  Several handles started with _run_async generate concurrently (batches above 4 rows are split this
 * way by the host code).                                                                                     */
int vqw_ar_decode_run_async(vqw_ar_decoder* h, const float* encoding, int Tz, int ratio,
                            int n_steps, int mode, const float* uniforms, float* audio,
                            int32_t* indices, float* probs_last, vqw_stream_t s);
int vqw_ar_decode_wait(vqw_ar_decoder* h);
/* Up to 8 handles (independent batch slices: rows never interact, generate.py:40,103-113) in ONE persistent launch
 * (workgroup g works for handle g % n: with n = 8 every handle sits on one XCD): they generate side by side whatever hardware queues the runtime maps streams to (two
 * _run_async calls on two streams overlap only when those streams land on different queues).  Every handle must be on
 * the persistent kernel with the same instantiation (vqw_ar_decode_workgroups > 0, equal batch class) and
 * n * workgroups must not exceed the CU count (one resident workgroup per CU); per-handle pointer arrays, `uniforms`,
 * `indices`, `probs_last` may be NULL as a whole.  Then vqw_ar_decode_wait on every handle.                      */
int vqw_ar_decode_run_group_async(vqw_ar_decoder* const* hs, int n, const float* const* encoding, int Tz,
                                  int ratio, int n_steps, int mode, const float* const* uniforms,
                                  float* const* audio, int32_t* const* indices, float* const* probs_last,
                                  vqw_stream_t s);
/* workgroups (= CUs held for the whole run) of the handle's persistent kernel; 0: launch-per-phase path */
int vqw_ar_decode_workgroups(const vqw_ar_decoder* h);
int vqw_ar_decode_destroy(vqw_ar_decoder* h);

/* ---- The fp16x3 engine (the default path of model.py for the decoder's residual stack): contractions
 * that are fp32-accurate on the fp16 matrix pipe.  Every operand is split into two fp16 planes (x = h1 + h2), products
 * h1 h1 + h1 h2 + h2 h1 in fp32 accumulators.  The gate conv (wavenet_ops.py:104-114, conv1d_v2 k taps + add_condition
 * + tanh*sigmoid), the 1x1 skip / residual convs, gate backward, the input gradient and the weight gradients.
 * Plane layout: [plane 0..1][channel chunk of 8][row][8 fp16] (16-byte entries); 2 * rows * channels bytes each.    */

/* `mode` (every vqw_f16x3_* entry point / descriptor) is a bit set:
 *   VQW_X3_BF16        the bf16 engine of BASELINE.json configs[4] -- bf16 storage + fp32 accumulate: ONE bf16 plane per
 *                      operand (planes buffers hold only the first plane), one v_mfma_f32_32x32x16_bf16 per product; no
 *                      range scales needed.  Operands written in one plane format must be consumed in the same.
 *   VQW_X3_HALF_BLOCKS conv kernels with 128-row blocks, two resident blocks per CU (one block's HBM-bound epilogue
 *                      overlaps the other's MFMA loop).  Results are the same; vqw_f16x3_pack_gate_weights must be given
 *                      the same bit as the gate conv that reads its planes (the filter / gate row order follows the
 *                      block height).  No effect on the other entry points.                                          */
#define VQW_X3_BF16 1
#define VQW_X3_HALF_BLOCKS 2
#define VQW_X3_S2D 4          /* vqw_f16x3_split_activations only: space-to-depth planes for vqw_f16x3_strided_conv */

/* Range guards.  The leading plane of scale * x must stay inside fp16 (|.| <= 65504).  Scales are powers of two held in
 * DEVICE memory, chosen from measured max-abs values with no host round trip:
 *   vqw_f16x3_amax           amax[i] = max(amax[i], bits(max |x_i|)) over `count` strided matrices [rows][cols]
 *                            (row stride ld, matrix stride mstride); raises *flag on inf / NaN;
 *   vqw_f16x3_update_scales  scale[i] = 2^k with amax[i] * scale[i] in [2^(target_exp-1), 2^target_exp); a slot whose
 *                            amax is 0 keeps its scale; reset != 0 zeroes amax[] for the next collection.
 * Kernels that WRITE planes take the scale from a device slot (`*_scale` / `scale_dev` below), report the max-abs of
 * their fp32 values into `out_amax` (the scale of the NEXT step comes from it) and raise *flag (|= 1) when an element
 * leaves fp16's range or is not finite: the host then repeats the step on the fp32 engine (model.py).  Every guard
 * pointer may be NULL (fixed host-side scales, no checks).                                                          */
int vqw_f16x3_amax(const float* x, int64_t rows, int cols, int64_t ld, int64_t mstride, int count, uint32_t* amax,
                   int32_t* flag, vqw_stream_t s);
int vqw_f16x3_update_scales(uint32_t* amax, float* scale, int n, int target_exp, int reset, int32_t* flag, vqw_stream_t s);
/* ... behind a device-side guard (as vqw_adam_ema_step_guarded): with *skip != 0 when the kernel runs, nothing is touched */
int vqw_f16x3_update_scales_guarded(uint32_t* amax, float* scale, int n, int target_exp, int reset, int32_t* flag,
                                    const int32_t* skip, vqw_stream_t s);

/* scale * scale_dev[0] * x [B][C][T] fp32 -> planes [2][C/8][B*T][8] fp16; C % 8 == 0.  scale: 1 for activations, a
 * power of two that lifts a gradient tensor into fp16's range (undone through w_scale_inv / x_scale of the consumer) */
int vqw_f16x3_split_activations(const float* x, void* planes, int B, int C, int T, float scale, int kc0, int KC,
                                const float* scale_dev, uint32_t* amax, int32_t* flag, int mode,
                                vqw_stream_t s);   /* planes hold KC chunks per plane, x goes to chunks kc0.. (0, 0: KC = C/8) */
/* w [ks][R][ldw] fp32 (kernel[k, Cin, Cout]: filter columns 0..R-1, gate columns R..2R-1), multiplied by `scale`
 * (a power of two that lifts the residual plane into fp16's normal range, e.g. 256) -> planes [2][ks*R/8][2R][8]
 * with the output channels in the kernel's block order; R % 128 == 0.  `count` layers stored back to back (w: ks*R*ldw
 * floats apart, planes: 2*ks*R*2R halves apart) are packed by one launch */
int vqw_f16x3_pack_gate_weights(const float* w, void* planes, int ks, int R, int ldw, float scale, int count,
                                const float* scale_dev, int mode, vqw_stream_t s);   /* planes hold scale * scale_dev[0] * w */

typedef struct vqw_f16x3_gate_desc {
    const void* xp;      /* activation planes of the layer input [B][R][T]                  */
    const void* wp;      /* weight planes (vqw_f16x3_pack_gate_weights)                     */
    const float* bias;   /* [2R] or NULL                                                    */
    const float* cond;   /* projected condition [B][2R][cond_T] (batch stride cond_bstride) or NULL */
    float* out0;         /* tanh(filter) * sigmoid(gate)  [B][R][T]                         */
    float* save0;        /* tanh    [B][R][T] or NULL (kept for the backward pass)          */
    float* save1;        /* sigmoid [B][R][T] or NULL                                       */
    void* out_planes;    /* out0 once more as fp16 planes [2][R/8][B*T][8] (input of vqw_f16x3_out_conv) or NULL */
    int64_t cond_bstride;
    int32_t B, T, R, ks, dilation, cond_T;
    float w_scale_inv;   /* 1 / scale of the weight planes                                  */
    int32_t out_planes_kc0, out_planes_KC;  /* out_planes holds KC chunks of 8 channels per plane, this layer's start at
                                             * chunk kc0 (several layers side by side); 0, 0 = exactly this layer's R/8 */
    const float* x_scale;  /* device scalars (or NULL = 1): the scales xp and wp were written with; the accumulators */
    const float* w_scale;  /* are multiplied by w_scale_inv / (x_scale * w_scale)                                    */
    int32_t mode;          /* VQW_X3_* bits (see below)                                                              */
} vqw_f16x3_gate_desc;
/* T % 256 == 0, R % 128 == 0, (T / cond_T) % 32 == 0; tap j reads x[t - (ks-1-j)*dilation], zero before t = 0 */
int vqw_f16x3_gate_conv(const vqw_f16x3_gate_desc* d, vqw_stream_t s);

/* w [K][ldw] fp32 (row k, column m), times `scale` -> planes [2][K/8][M][8]; K % 8 == 0; `count` matrices back to back
 * (w: K*ldw floats apart, planes: 2*K*M halves apart) */
int vqw_f16x3_pack_weights(const float* w, void* planes, int K, int M, int ldw, float scale, int count,
                           const float* scale_dev, int mode, vqw_stream_t s);
/* The same planes of the matrix W'[k][m] that is stored TRANSPOSED, block by block: k = jb * k_inner + ko is
 * src[jb * blk_stride + m * ld_src + ko] -- e.g. the kernel of a conv's input gradient straight from the forward kernel
 * w[tap][Cin][Cout] (TF Conv2DBackpropInput: k = (tap, cout), m = cin: k_inner = ld_src = Cout, blk_stride = Cin * Cout), with no
 * transposed fp32 copy in between.  k_inner % 8 == 0, K % k_inner == 0, rows 16-byte aligned; `count` matrices
 * (K / k_inner) * blk_stride floats apart.                                                                             */
int vqw_f16x3_pack_weights_t(const float* w, void* planes, int K, int M, int k_inner, int ld_src, int64_t blk_stride,
                             float scale, int count, const float* scale_dev, int mode, vqw_stream_t s);

/* The layer's 1x1 skip + residual conv (wavenet_ops.py:132-136, wavenet.py:72-73) on the gated planes:
 * skip[b][m][t] += (W g)[m] + bias[m] for m < S;  net_out[b][c][t] = net_in[b][c][t] + (W g)[S+c] + bias[S+c],
 * and net_out once more as planes for the next layer's gate conv.  T % 256 == 0, R % 256 == 0, S % 256 == 0.
 * With R = 0 and Cin = L*R_layer over the gated planes of all L layers side by side it is the whole skip path
 * as ONE contraction (skip += sum_l W_s,l g_l), with S = 0 the residual half alone.                                  */
typedef struct vqw_f16x3_out_desc {
    const void* xp;        /* gated planes [2][R/8][B*T][8]                                  */
    const void* wp;        /* vqw_f16x3_pack_weights(out_w [Cin][S+R], K = Cin, M = S+R)     */
    const float* bias;     /* [S+R] or NULL                                                  */
    float* skip;           /* [B][S][T], accumulated in place                                */
    const float* net_in;   /* [B][R][T]                                                      */
    float* net_out;        /* [B][R][T]                                                      */
    void* net_out_planes;  /* [2][R/8][B*T][8] or NULL                                       */
    int32_t B, T, R, S;    /* S skip rows first, then R residual rows; either may be 0       */
    float w_scale_inv;
    int32_t Cin;           /* contracted channels; 0 = R                                    */
    int32_t xp_kc0, xp_KC; /* xp holds KC chunks per plane, the contraction starts at chunk kc0; KC 0 = Cin/8 */
    int32_t ks, dilation;  /* taps (0 = 1): wp = [ks*Cin][S+R], tap j reads x[t - dir*(ks-1-j)*dilation]          */
    int32_t dir;           /* >= 0: causal conv; < 0: reads ahead = the input gradient of a causal conv with
                            * transposed kernels (net_in = the gradient arriving from the residual path or NULL) */
    int32_t planes_kc0, planes_KC; /* placement of net_out_planes (0, 0: R/8 chunks, or 2R/8 for epi 1)         */
    float plane_scale;     /* net_out_planes hold plane_scale * net_out (0 = 1): gradients are lifted by a power of two */
    int32_t epi;           /* 0: as described above.  1: gate backward -- S = 0, dg = W^T x over Cin gradient channels,
                            * net_out = dpre [B][2R][T] = {dg*sg*(1-th^2), dg*th*sg*(1-sg)} with aux0 = th, aux1 = sg [B][R][T] */
    const float* aux0;
    const float* aux1;
    const float* x_scale;  /* device scalars (or NULL = 1) of xp / wp: accumulators times w_scale_inv / (x_scale * w_scale) */
    const float* w_scale;
    const float* out_scale; /* device scalar (or NULL = 1): net_out_planes hold plane_scale * out_scale * net_out   */
    uint32_t* out_amax;    /* atomicMax of the bit pattern of max |net_out| (or NULL)                                */
    int32_t* flag;         /* |= 1 when plane_scale * out_scale * |net_out| > 65504 or net_out is not finite (or NULL) */
    int32_t mode;          /* VQW_X3_* bits                                                                          */
    /* epi 2 -- the 1x1 convs around the stack (wavenet.py:53-54, 80-96) and their input gradients; S = 0, R rows, fp16x3 mode:
     *   net_out = mask * (net_in + W x + bias + cond[b][m][t / (T / cond_T)]),  mask = (aux0 > 0) or 1 (aux0 NULL); net_in, bias,
     *   cond optional; net_out may alias net_in and aux0; net_out_planes get net_out, or relu(net_out) with flags bit 0        */
    const float* cond;
    int64_t cond_bstride;
    int32_t cond_T;
    int32_t flags;         /* bit 0 (epi 2): planes hold relu(net_out).  bit 1 (epi 1): aux0 is the layer's gated OUTPUT tanh * sigmoid instead
                            * of tanh (a forward pass that does not store tanh: vqw_f16x3_gate_conv with save0 = NULL writes 54 MB less
                            * per layer); the kernel forms tanh = aux0 / aux1, 0 where the sigmoid underflowed.  bit 2 (epi 1): that gated
                            * output is given as the PLANES vqw_f16x3_gate_conv wrote (aux0 = planes base, aux0_KC chunks per plane, this
                            * layer's R / 8 chunks from aux0_kc0; scale 1): vqw_f16x3_gate_conv then needs no fp32 out0 at all             */
    int32_t aux0_KC, aux0_kc0;
} vqw_f16x3_out_desc;
int vqw_f16x3_out_conv(const vqw_f16x3_out_desc* d, vqw_stream_t s);

/* The encoder's stride-2 convs on the fp16x3 engine (encoder.py:17-18: tf.layers.conv1d(768, 5, strides 2, 'same') -> relu ->
 * batch_normalization in inference mode, and TF's Conv2DBackpropInput of it).  Kernel w[ks][Cin][M] as planes
 * (vqw_f16x3_pack_weights, K = ks * Cin); for the input gradient the transposed kernel wt[ks][Cout][Cin] with Cin := Cout, M := Cin.
 *   dgrad = 0: out[b][m][t] = bn_scale[m] * relu(sum_j sum_c w[j][c][m] x[b][c][2t + j - pad_left] + bias[m]) + bn_shift[m],
 *              t < T; xp = vqw_f16x3_split_activations(x [B][Cin][2T], mode | VQW_X3_S2D); save_r (or NULL) gets the relu output
 *   dgrad = 1: out[b][m][u] (u < 2T) = sum_{j, t: 2t + j - pad_left = u} sum_o wt[j][o][m] dy[b][o][t]; xp = plain planes of dy [B][Cin][T]
 * Cin % 32 == 0, M % 128 == 0 (B * T need not be a multiple of the 256-column tiles).  Results are scaled by w_scale_inv / (x_scale[0] * w_scale[0]).           */
typedef struct vqw_f16x3_sconv_desc {
    const void* xp;
    const void* wp;
    const float* bias;      /* [M] or NULL (forward)                                                        */
    const float* bn_scale;  /* [M] or NULL (forward)                                                        */
    const float* bn_shift;
    float* out;
    float* save_r;          /* [B][M][T] or NULL (forward)                                                  */
    const float* x_scale;   /* device scalars (or NULL = 1): the scales the operand planes were made with   */
    const float* w_scale;
    float w_scale_inv;
    int32_t B, T, Cin, M, ks, pad_left, relu, dgrad;
    int32_t shape;          /* 0 = by how the launch fills the chip; 1: 128-row blocks, two per CU; 2: 128-row blocks, one per
                             * CU with deeper prefetch; 3: 256-row blocks (M % 256 == 0); 4: 192-row blocks (M % 192 == 0);
                             * 5: 64-row blocks                                                                               */
    /* Split-K for launches that would leave most of the chip idle (encoder layers 2-5: 24..156 tiles of 240 K steps on 256 CUs):
     * with scratch given, the K steps of a tile are cut over `ksplit` blocks (and the two output parities of an input gradient go
     * to separate blocks); the partial tiles meet in split_slab and the last block to arrive at a tile adds them in a fixed order
     * and runs the epilogue (results are bitwise reproducible).  ksplit: 0 = chosen by the launch's tile count, 1 = off, n = n
     * blocks per tile (the K steps of every parity must divide into n even parts).  split_slab: blocks * rows per block * 256
     * floats (CUs * 65536 always suffices); split_counters: one int per tile, ZERO before the first launch (left zero by every
     * launch).                                                                                                              */
    float* split_slab;
    int64_t split_slab_floats;
    int32_t* split_counters;
    int32_t split_counters_n;
    int32_t ksplit;
} vqw_f16x3_sconv_desc;
int vqw_f16x3_strided_conv(const vqw_f16x3_sconv_desc* d, vqw_stream_t s);

/* Weight gradient of conv1d_v2 (TF Conv2DBackpropFilter, wavenet_ops.py:83-86) on the fp16x3 engine:
 *   dw[j][c][o] += sum_{b,t} p[b][c][t + tap_shift[j]] * q[b][o][t],   q = [q0 (Q0 rows); q1 (Q1 rows)] along o,
 * p [B][Cp][T], q0 [B][Q0][T], q1 [B][Q1][T] fp32 (time contiguous = the contraction index: the operands are split into
 * fp16 planes inside the kernel, with the power-of-two guard scales *_scale (device scalars or NULL) of the same tensors'
 * planes, whose producers range-check them).  T % 32 == 0; Cp, Q0, Q1 multiples of 256; tap_shift <= 0 (reads outside
 * [0, T) are zero).  `slab` is scratch: tiles * nsplit * 65536 floats (tiles = ntaps * Cp/256 * (Q0+Q1)/256; nsplit 0 =
 * CUs / tiles): partial tiles are summed in a fixed order by a second launch -- dw is bitwise reproducible (q_total /
 * q_seg are met by fp32 atomics).                                                                                  */
typedef struct vqw_f16x3_wgrad_desc {
    const float* p;
    const float* q0;
    const float* q1;        /* or NULL with Q1 = 0                                                          */
    float* dw;              /* [ntaps][Cp][lddw] (tap stride dw_tap_stride), accumulated into               */
    float* slab;
    int64_t slab_floats;
    const float* p_scale;
    const float* q0_scale;
    const float* q1_scale;
    int32_t B, T, Cp, Q0, Q1, ntaps;
    int32_t tap_shift[VQW_MAX_TAPS];
    int32_t lddw;           /* 0 = Q0 + Q1                                                                  */
    int32_t nsplit;         /* 0 = one round of blocks                                                      */
    int64_t dw_tap_stride;  /* 0 = Cp * lddw                                                                */
    /* optional sums of q over time, formed from the registers that hold q anyway (no extra pass over q):           */
    float* q_total;         /* q_total[o] += sum_{b,t} q[b][o][t] for o in [total_o0, total_o1) (bias gradients)     */
    float* q_seg;           /* q_seg[b*seg_bstride + o*seg_T + t/(T/seg_T)] += q[b][o][t]: the gradient of add_condition's
                             * projected condition (wavenet_ops.py:98-100); (T/seg_T) % 32 == 0; the caller zeroes it    */
    int64_t seg_bstride;
    int32_t seg_T;
    int32_t total_o0, total_o1;   /* 0, 0 = all of [0, Q0 + Q1)                                              */
    int32_t mode;           /* VQW_X3_* bits                                                                 */
    int32_t p_relu;         /* p := max(p, 0) on the way in                                                   */
    int32_t p_stride;       /* 0 / 1, or 2: p is the input of a stride-2 conv (encoder.py:17-18), row length Tp, read at
                             * 2 t + tap_shift[j] (shifts of either sign; outside [0, Tp) = zero padding); fp16x3 mode only;
                             * T then only has to be a multiple of 4                                           */
    int32_t Tp;
    int32_t xcd_group;      /* ignored since round 3 (the block -> XCD placement is always K range, then tile, then tap) */
    /* q as the operand planes its producer wrote anyway (vqw_f16x3_out_conv epi 1 writes dpre as planes for the input gradient and,
     * with net_out = NULL, nothing else): [planes][q_planes_KC chunks][B*T rows][8] scaled by *q0_scale, this problem's Q0 rows from
     * chunk q_planes_kc0.  q0 may then be NULL; Q1 = 0, p_stride 1.  q_total / q_seg are formed from the planes (2^-22 relative). */
    const void* q_planes;
    int32_t q_planes_KC;    /* 0 = Q0 / 8 */
    int32_t q_planes_kc0;
    float q_planes_scale;   /* host-side factor of the planes' scale (the producer's plane_scale), 0 = 1: planes = q * q_planes_scale * *q0_scale */
    /* p likewise (the layer-input planes of vqw_f16x3_out_conv / the gated planes of vqw_f16x3_gate_conv): a tap's shift is then a row
     * offset of the planes, so shifts that are not multiples of 4 cost nothing extra.  p may be NULL; p_stride 1, no p_relu.        */
    const void* p_planes;
    int32_t p_planes_KC;    /* 0 = Cp / 8 */
    int32_t p_planes_kc0;
    float p_planes_scale;
    /* with p_planes: tap j reads rows t + tap_shift[j] (either sign; outside [0, T) = zero padding) of the chunks starting at
     * p_planes_kc0 + p_tap_chunk[j].  The weight gradient of a STRIDE-2 conv (encoder.py:17-18) over the space-to-depth planes of its
     * input (vqw_f16x3_split_activations VQW_X3_S2D, T = output length): tap j with e = j - pad_left has
     * p_tap_chunk[j] = (e & 1) * Cp / 8 and tap_shift[j] = e >> 1 -- no p_stride 2, no single-float requests.                    */
    int32_t p_tap_chunk[VQW_MAX_TAPS];
} vqw_f16x3_wgrad_desc;
int vqw_f16x3_wgrad(const vqw_f16x3_wgrad_desc* d, vqw_stream_t s);
/* `n` (<= 32) weight gradients of ONE shape in one launch -- the same kernels of several layers: d[i] may differ in p, q0, q1, dw,
 * the scales, q_total, q_seg and tap_shift only.  The K split is chosen for the tiles of all problems together, so the slab
 * traffic per problem falls with n (tiles * nsplit <= CUs partial tiles per LAUNCH).  Results per problem are bit-identical
 * for a given n (fixed summation order), not across different n (the K ranges differ).                                   */
int vqw_f16x3_wgrad_batch(const vqw_f16x3_wgrad_desc* d, int n, vqw_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* VQWAVE_H */

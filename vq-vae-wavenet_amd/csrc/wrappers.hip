// Named C-ABI wrappers carrying the reference's argument meaning (conv1d_v2, Keras Conv1D,
// `linear`); each one only fills a descriptor of the conv / wgrad engines.
#include <string.h>

#include "vqw_common.h"

namespace {
inline void zero(vqw_conv_desc& d) { memset(&d, 0, sizeof(d)); }
inline void zero(vqw_wgrad_desc& d) { memset(&d, 0, sizeof(d)); }
}  // namespace

// wavenet_ops.py:59-90
extern "C" int vqw_causal_conv1d_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                                     int Cin, int Cout, int T, int k, int dilation, int stride,
                                     vqw_stream_t s) {
    VQW_CHECK(k >= 1 && k <= VQW_MAX_TAPS, "vqw_causal_conv1d_fwd: k=%d out of range (1..%d)", k, VQW_MAX_TAPS);
    VQW_CHECK(stride == 1 || stride == 2, "vqw_causal_conv1d_fwd: stride must be 1 or 2");
    vqw_conv_desc d;
    zero(d);
    d.B = B; d.T_in = T; d.T_out = (T + stride - 1) / stride; d.M = Cout; d.C0 = Cin;
    d.ntaps = k; d.in_stride = stride; d.ldw = Cout;
    for (int j = 0; j < k; ++j) d.tap_shift[j] = -(k - 1 - j) * dilation;
    d.epilogue = VQW_EPI_STORE;
    d.x0 = x; d.w = w; d.bias = bias; d.out0 = y;
    return vqw_conv_gemm(&d, s);
}

extern "C" int vqw_causal_conv1d_dgrad(const float* dy, const float* wT, float* dx, int B, int Cin, int Cout,
                                       int T, int k, int dilation, vqw_stream_t s) {
    VQW_CHECK(k >= 1 && k <= VQW_MAX_TAPS, "vqw_causal_conv1d_dgrad: k=%d out of range", k);
    vqw_conv_desc d;
    zero(d);
    d.B = B; d.T_in = T; d.T_out = T; d.M = Cin; d.C0 = Cout;
    d.ntaps = k; d.in_stride = 1; d.ldw = Cin;
    for (int j = 0; j < k; ++j) d.tap_shift[j] = (k - 1 - j) * dilation;
    d.epilogue = VQW_EPI_STORE;
    d.x0 = dy; d.w = wT; d.out0 = dx;
    return vqw_conv_gemm(&d, s);
}

extern "C" int vqw_causal_conv1d_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout,
                                       int T, int k, int dilation, vqw_stream_t s) {
    VQW_CHECK(k >= 1 && k <= VQW_MAX_TAPS, "vqw_causal_conv1d_wgrad: k=%d out of range", k);
    vqw_wgrad_desc d;
    zero(d);
    d.B = B; d.T_q = T; d.T_p = T; d.Cp = Cin; d.Q0 = Cout; d.ntaps = k; d.p_stride = 1;
    for (int j = 0; j < k; ++j) d.tap_shift[j] = -(k - 1 - j) * dilation;
    d.lddw = Cout; d.dw_tap_stride = (int64_t)Cin * Cout;
    d.p = x; d.q0 = dy; d.dw = dw;
    return vqw_wgrad_gemm(&d, s);
}

// encoder.py:15-19, encoder_ops.py:46-70 (TF SAME padding passed explicitly)
extern "C" int vqw_conv1d_same_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                                   int Cout, int T_in, int T_out, int k, int stride, int pad_left, int relu,
                                   vqw_stream_t s) {
    VQW_CHECK(k >= 1 && k <= VQW_MAX_TAPS, "vqw_conv1d_same_fwd: k=%d out of range", k);
    VQW_CHECK(stride == 1 || stride == 2, "vqw_conv1d_same_fwd: stride must be 1 or 2");
    vqw_conv_desc d;
    zero(d);
    d.B = B; d.T_in = T_in; d.T_out = T_out; d.M = Cout; d.C0 = Cin;
    d.ntaps = k; d.in_stride = stride; d.ldw = Cout;
    for (int j = 0; j < k; ++j) d.tap_shift[j] = j - pad_left;
    d.epilogue = VQW_EPI_STORE; d.out_relu = relu;
    d.x0 = x; d.w = w; d.bias = bias; d.out0 = y;
    return vqw_conv_gemm(&d, s);
}

// dx[b][c][tau] = sum_j sum_o wT[j][o][c] * dy[b][o][(tau + pad_left - j)/stride]  (exact divisions only)
extern "C" int vqw_conv1d_same_dgrad(const float* dy, const float* wT, float* dx, int B, int Cin, int Cout,
                                     int T_in, int T_out, int k, int stride, int pad_left, vqw_stream_t s) {
    VQW_CHECK(k >= 1 && k <= VQW_MAX_TAPS, "vqw_conv1d_same_dgrad: k=%d out of range", k);
    VQW_CHECK(stride == 1 || stride == 2, "vqw_conv1d_same_dgrad: stride must be 1 or 2");
    for (int p = 0; p < stride; ++p) {
        vqw_conv_desc d;
        zero(d);
        // taps with j == p + pad_left (mod stride) reach output times tau = stride*u + p
        int j0 = ((p + pad_left) % stride + stride) % stride;
        int n = 0;
        for (int j = j0; j < k; j += stride) d.tap_shift[n++] = (p + pad_left - j) / stride;
        if (n == 0) continue;
        d.B = B; d.T_in = T_out; d.T_out = (T_in - p + stride - 1) / stride; d.M = Cin; d.C0 = Cout;
        d.ntaps = n; d.in_stride = 1; d.ldw = Cin;
        d.w_tap_stride = (int64_t)stride * Cout * Cin;
        d.epilogue = VQW_EPI_STORE;
        d.out_tstride = stride; d.out_toffset = p; d.T_store = T_in;
        d.x0 = dy; d.w = wT + (size_t)j0 * Cout * Cin; d.out0 = dx;
        if (d.T_out <= 0) continue;
        const int rc = vqw_conv_gemm(&d, s);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int vqw_conv1d_same_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout,
                                     int T_in, int T_out, int k, int stride, int pad_left, vqw_stream_t s) {
    VQW_CHECK(k >= 1 && k <= VQW_MAX_TAPS, "vqw_conv1d_same_wgrad: k=%d out of range", k);
    VQW_CHECK(stride == 1 || stride == 2, "vqw_conv1d_same_wgrad: stride must be 1 or 2");
    vqw_wgrad_desc d;
    zero(d);
    d.B = B; d.T_q = T_out; d.T_p = T_in; d.Cp = Cin; d.Q0 = Cout; d.ntaps = k; d.p_stride = stride;
    for (int j = 0; j < k; ++j) d.tap_shift[j] = j - pad_left;
    d.lddw = Cout; d.dw_tap_stride = (int64_t)Cin * Cout;
    d.p = x; d.q0 = dy; d.dw = dw;
    return vqw_wgrad_gemm(&d, s);
}

// wavenet_ops.py:132-136 / 147-160
extern "C" int vqw_pointwise_gemm_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                                      int Cin, int Cout, int T, int relu_in, int accumulate, vqw_stream_t s) {
    vqw_conv_desc d;
    zero(d);
    d.B = B; d.T_in = T; d.T_out = T; d.M = Cout; d.C0 = Cin; d.ntaps = 1; d.in_stride = 1; d.ldw = Cout;
    d.in_relu = relu_in;
    d.x0 = x; d.w = w; d.bias = bias; d.out0 = y;
    if (accumulate) { d.epilogue = VQW_EPI_ACCUM_SPLIT; d.M0 = Cout; } else { d.epilogue = VQW_EPI_STORE; }
    return vqw_conv_gemm(&d, s);
}

extern "C" int vqw_pointwise_gemm_dgrad(const float* dy, const float* wT, float* dx, int B, int Cin, int Cout,
                                        int T, vqw_stream_t s) {
    vqw_conv_desc d;
    zero(d);
    d.B = B; d.T_in = T; d.T_out = T; d.M = Cin; d.C0 = Cout; d.ntaps = 1; d.in_stride = 1; d.ldw = Cin;
    d.epilogue = VQW_EPI_STORE;
    d.x0 = dy; d.w = wT; d.out0 = dx;
    return vqw_conv_gemm(&d, s);
}

extern "C" int vqw_pointwise_gemm_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout,
                                        int T, int relu_in, vqw_stream_t s) {
    vqw_wgrad_desc d;
    zero(d);
    d.B = B; d.T_q = T; d.T_p = T; d.Cp = Cin; d.Q0 = Cout; d.ntaps = 1; d.p_stride = 1;
    d.p_relu = relu_in; d.lddw = Cout; d.dw_tap_stride = (int64_t)Cin * Cout;
    d.p = x; d.q0 = dy; d.dw = dw;
    return vqw_wgrad_gemm(&d, s);
}

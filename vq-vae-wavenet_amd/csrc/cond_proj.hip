// The decoder's local-condition projections: add_condition (wavenet_ops.py:93-101) is a 1x1 conv1d_v2 of the condition
// [B][Cc][Tz] per layer (wavenet.py:58-100 calls it for every gated_cnn and for postprocess1); all L + 1 kernels side by
// side are ONE matrix w[Cc][Mall], Mall = L * 2R + S, so the forward pass is one product  out[b][m][t] = sum_c w[c][m] cond[b][c][t]
// and the backward pass two:  dw[c][m] += sum_{b,t} cond[b][c][t] dce[b][m][t],  dcond[b][c][t] = sum_m w[c][m] dce[b][m][t].
//
// Shapes at the benchmark: Cc = 80, Mall = 15 872, B * Tz = 832 frames: 2.1 GFLOP each with 52 MB of output (forward) or input
// (backward).  The conv engine is built around long K loops and 128-column tiles; at K = 80 and 104-frame rows its blocks are all
// prologue and epilogue (92 / 108 / 81 us for the three).  Here every WAVE owns its tile and feeds v_mfma_f32_32x32x2_f32 (exact fp32
// products, fp32 accumulate: the arithmetic of the fp32 engine) straight from global memory -- lane (row, k half) of the MFMA is one
// coalesced 4-byte load per operand, or where the contraction index is the contiguous one a 16-byte load that serves four MFMA steps
// (the order of k inside a group is free as long as both operands use the same one) -- with no LDS, no barrier and enough
// independent waves per SIMD to cover the load latency.  Plain fp32 FMA kernels were tried first: 78 / 191 / 105 us, the vector
// ALU's peak alone is 30 us for 2.1 GFLOP.
#include "vqw_common.h"

namespace {

__device__ __forceinline__ f32x16 cp_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}

// out[b][m][t] = sum_c w[c][m] cond[b][c][t].  Wave = 32 rows m x up to 128 frames of one batch row; block = 4 waves (4 row tiles);
// grid (ceil(Mall / 128), B).
__global__ __launch_bounds__(256) void cond_proj_fwd_kernel(const float* __restrict__ cond, const float* __restrict__ w,
                                                            float* __restrict__ out, int Cc, int Mall, int Tz) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int b = blockIdx.y, m0 = (blockIdx.x * 4 + wv) * 32;
    if (m0 >= Mall) return;
    const int m = m0 + l31;
    const bool mv = m < Mall;
    const float* cb = cond + (size_t)b * Cc * Tz;
    const int ksteps = (Cc + 1) >> 1;
    for (int t0 = 0; t0 < Tz; t0 += 128) {
        f32x16 acc[4] = {cp_zero(), cp_zero(), cp_zero(), cp_zero()};
#pragma unroll 8
        for (int k2 = 0; k2 < ksteps; ++k2) {
            const int c = 2 * k2 + lhi;
            const bool cv = c < Cc;
            const float a = (cv && mv) ? w[(size_t)c * Mall + m] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + 32 * j + l31;
                const float bv = (cv && t < Tz) ? cb[(size_t)c * Tz + t] : 0.0f;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + 32 * j + l31;
            if (t >= Tz) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (r >> 2) * 8 + lhi * 4 + (r & 3);
                if (row < Mall) out[((size_t)b * Mall + row) * Tz + t] = acc[j][r];
            }
        }
    }
}

// dw[c][m] += sum_{b in this split} sum_t cond[b][c][t] dce[b][m][t].  Wave = all CT row tiles of c x one 32-column tile of m, one
// of KS batch ranges; the contraction index t is contiguous in both operands: lane (row, k half) loads t = 8 s + 4 lhi .. + 3 as 16
// bytes and feeds MFMA step j with element j.  KS > 1: the partial sums meet by fp32 atomics (two addends on a zeroed gradient commute).
template <int CT>
__global__ __launch_bounds__(256) void cond_proj_wgrad_kernel(const float* __restrict__ cond, const float* __restrict__ dce,
                                                              float* __restrict__ dw, int B, int Cc, int Mall, int Tz, int KS) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int m0 = (blockIdx.x * 4 + wv) * 32, ks = blockIdx.y;
    if (m0 >= Mall) return;
    const int m = m0 + l31;
    const bool mv = m < Mall;
    const int b_lo = (int)((long)ks * B / KS), b_hi = (int)((long)(ks + 1) * B / KS);
    f32x16 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = cp_zero();
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int b = b_lo; b < b_hi; ++b) {
        const float* qrow = dce + ((size_t)b * Mall + (mv ? m : 0)) * Tz;
        const float* cb = cond + (size_t)b * Cc * Tz;
#pragma unroll 2
        for (int t8 = 0; t8 < Tz; t8 += 8) {
            const int t = t8 + 4 * lhi;
            const bool tv = t < Tz;                     // (Tz % 4 == 0: a group of four is inside the row or behind it)
            const f32x4 q = (mv && tv) ? *reinterpret_cast<const f32x4*>(qrow + t) : z4;
            f32x4 p[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int c = 32 * i + l31;
                p[i] = (c < Cc && tv) ? *reinterpret_cast<const f32x4*>(cb + (size_t)c * Tz + t) : z4;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[i][j], q[j], acc[i], 0, 0, 0);
        }
    }
    if (!mv) return;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * i + (r >> 2) * 8 + lhi * 4 + (r & 3);
            if (c >= Cc) continue;
            float* o = dw + (size_t)c * Mall + m;
            if (KS > 1) unsafeAtomicAdd(o, acc[i][r]);
            else *o += acc[i][r];
        }
}

// part[chunk][b][c][t] = sum_{m in chunk} w[c][m] dce[b][m][t].  Wave = all CT row tiles of c x one 32-frame column tile of one
// batch row and one chunk of m; the contraction index m is contiguous in w (16-byte loads, four MFMA steps each) and a row stride
// in dce (four coalesced 4-byte loads).  grid (chunks, ceil(Tz / 32), B), one wave per block.
template <int CT>
__global__ __launch_bounds__(64) void cond_proj_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ dce,
                                                             float* __restrict__ part, int B, int Cc, int Mall, int Tz, int chunk_m) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const int chunk = blockIdx.x, t = blockIdx.y * 32 + l31, b = blockIdx.z;
    const int mlo = chunk * chunk_m, mhi = (mlo + chunk_m < Mall) ? mlo + chunk_m : Mall;
    const bool tv = t < Tz;
    f32x16 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = cp_zero();
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const float* db = dce + (size_t)b * Mall * Tz + (tv ? t : 0);
#pragma unroll 2
    for (int m8 = mlo; m8 < mhi; m8 += 8) {
        const int mm = m8 + 4 * lhi;
        const bool kv = mm < mhi;                        // (chunk_m and Mall are multiples of 4)
        float q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = (kv && tv) ? db[(size_t)(mm + j) * Tz] : 0.0f;
        f32x4 p[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            const int c = 32 * i + l31;
            p[i] = (c < Cc && kv) ? *reinterpret_cast<const f32x4*>(w + (size_t)c * Mall + mm) : z4;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(p[i][j], q[j], acc[i], 0, 0, 0);
    }
    if (!tv) return;
    float* pb = part + ((size_t)chunk * B + b) * Cc * Tz + t;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * i + (r >> 2) * 8 + lhi * 4 + (r & 3);
            if (c < Cc) pb[(size_t)c * Tz] = acc[i][r];
        }
}

// out[i] = sum_k part[k][i] in chunk order
__global__ void cond_proj_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nchunk, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
}

constexpr int CP_CHUNKS = 32;      // partial sums of dcond per element

inline int cp_chunk_m(int Mall) {
    int c = (Mall + CP_CHUNKS - 1) / CP_CHUNKS;
    return (c + 7) & ~7;
}

}  // namespace

extern "C" {

int vqw_cond_proj_fwd(const float* cond, const float* w, float* out, int B, int Cc, int Mall, int Tz, vqw_stream_t s) {
    VQW_CHECK(cond && w && out, "vqw_cond_proj_fwd: null pointer");
    VQW_CHECK(B > 0 && B <= 65535 && Cc > 0 && Mall > 0 && Tz > 0, "vqw_cond_proj_fwd: bad shape (B=%d Cc=%d Mall=%d Tz=%d)", B, Cc, Mall, Tz);
    hipLaunchKernelGGL(cond_proj_fwd_kernel, dim3(vqw_cdiv(Mall, 128), B), dim3(256), 0, (hipStream_t)s, cond, w, out, Cc, Mall, Tz);
    VQW_LAUNCH_CHECK("vqw_cond_proj_fwd");
    return 0;
}

int vqw_cond_proj_wgrad(const float* cond, const float* dce, float* dw, int B, int Cc, int Mall, int Tz, vqw_stream_t s) {
    VQW_CHECK(cond && dce && dw, "vqw_cond_proj_wgrad: null pointer");
    VQW_CHECK(B > 0 && Cc > 0 && Cc <= 128 && Mall > 0 && Tz > 0 && Tz % 4 == 0, "vqw_cond_proj_wgrad: needs Cc <= 128, Tz %% 4 == 0 (B=%d Cc=%d Mall=%d Tz=%d)", B, Cc, Mall, Tz);
    VQW_CHECK(((reinterpret_cast<uintptr_t>(cond) | reinterpret_cast<uintptr_t>(dce)) & 15) == 0, "vqw_cond_proj_wgrad: operands must be 16-byte aligned");
    const int ct = (Cc + 31) / 32;
    // two batch ranges per column tile when one would leave most SIMDs without a wave (their sums meet by atomics: the caller's dw
    // is the zeroed gradient buffer, and two addends commute)
    const int KS = (B >= 2 && vqw_cdiv(Mall, 32) < 4 * vqw_device_cus()) ? 2 : 1;
    typedef void (*kfn_t)(const float*, const float*, float*, int, int, int, int, int);
    const kfn_t kfn = ct == 1 ? cond_proj_wgrad_kernel<1> : (ct == 2 ? cond_proj_wgrad_kernel<2> : (ct == 3 ? cond_proj_wgrad_kernel<3> : cond_proj_wgrad_kernel<4>));
    hipLaunchKernelGGL(kfn, dim3(vqw_cdiv(Mall, 128), KS), dim3(256), 0, (hipStream_t)s, cond, dce, dw, B, Cc, Mall, Tz, KS);
    VQW_LAUNCH_CHECK("vqw_cond_proj_wgrad");
    return 0;
}

int vqw_cond_proj_dgrad(const float* w, const float* dce, float* dcond, float* scratch, int64_t scratch_floats, int B, int Cc, int Mall, int Tz,
                        vqw_stream_t s) {
    VQW_CHECK(w && dce && dcond && scratch, "vqw_cond_proj_dgrad: null pointer");
    VQW_CHECK(B > 0 && B <= 65535 && Cc > 0 && Cc <= 128 && Mall > 0 && Mall % 4 == 0 && Tz > 0, "vqw_cond_proj_dgrad: needs Cc <= 128, Mall %% 4 == 0 (B=%d Cc=%d Mall=%d Tz=%d)", B, Cc, Mall, Tz);
    VQW_CHECK((reinterpret_cast<uintptr_t>(w) & 15) == 0, "vqw_cond_proj_dgrad: w must be 16-byte aligned");
    const int chunk_m = cp_chunk_m(Mall), nchunk = vqw_cdiv(Mall, chunk_m);
    const size_t n = (size_t)B * Cc * Tz;
    VQW_CHECK((size_t)scratch_floats >= (size_t)nchunk * n, "vqw_cond_proj_dgrad: scratch needs %d * B * Cc * Tz = %zu floats", nchunk, (size_t)nchunk * n);
    const int ct = (Cc + 31) / 32;
    typedef void (*kfn_t)(const float*, const float*, float*, int, int, int, int, int);
    const kfn_t kfn = ct == 1 ? cond_proj_dgrad_kernel<1> : (ct == 2 ? cond_proj_dgrad_kernel<2> : (ct == 3 ? cond_proj_dgrad_kernel<3> : cond_proj_dgrad_kernel<4>));
    hipLaunchKernelGGL(kfn, dim3(nchunk, vqw_cdiv(Tz, 32), B), dim3(64), 0, (hipStream_t)s, w, dce, scratch, B, Cc, Mall, Tz, chunk_m);
    hipLaunchKernelGGL(cond_proj_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)s, (const float*)scratch, dcond, nchunk, n);
    VQW_LAUNCH_CHECK("vqw_cond_proj_dgrad");
    return 0;
}

int vqw_cond_proj_dgrad_scratch_floats(int B, int Cc, int Mall, int Tz, int64_t* out) {
    VQW_CHECK(out && B > 0 && Cc > 0 && Mall > 0 && Tz > 0, "vqw_cond_proj_dgrad_scratch_floats: bad arguments");
    *out = (int64_t)vqw_cdiv(Mall, cp_chunk_m(Mall)) * B * Cc * Tz;
    return 0;
}

}  // extern "C"

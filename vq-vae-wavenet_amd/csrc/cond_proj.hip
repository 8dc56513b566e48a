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
// independent waves per SIMD to cover the load latency: 36 / 38 / 53 us (tools/cond_bench.py).  What it took: the requests are
// buffer loads whose out-of-range lanes read zero (a conditional load splits the loop into blocks that wait for every single
// request: 54 / 46 / 66 us), the loops are hand-unrolled into double-buffered groups (hipcc leaves them rolled), and register use
// is kept low enough for three or more waves per SIMD.  Plain fp32 FMA kernels were tried first: 78 / 191 / 105 us -- the vector
// ALU's peak alone is 30 us for 2.1 GFLOP.
#include "vqw_common.h"

namespace {

// Loads go through buffer resources: a lane that has nothing to load passes an offset behind the range and receives zero -- no
// branch around the load (a conditional load splits the loop body into blocks that wait for every single request).
constexpr int CP_OOB = (int)0x80000000;
__device__ __forceinline__ float cp_ld1(__amdgpu_buffer_rsrc_t r, int off_bytes) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off_bytes, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cp_rsrc(const float* p, size_t floats) {
    const size_t bytes = floats * 4;
    return vqw_make_rsrc(p, bytes < 0x7fffffffull ? (unsigned)bytes : 0x7fffffffu);
}

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
    const __amdgpu_buffer_rsrc_t rw = cp_rsrc(w, (size_t)Cc * Mall), rc = cp_rsrc(cond + (size_t)b * Cc * Tz, (size_t)Cc * Tz);
    const int ksteps = (Cc + 1) >> 1;
    for (int t0 = 0; t0 < Tz; t0 += 128) {
        f32x16 acc[4] = {cp_zero(), cp_zero(), cp_zero(), cp_zero()};
        // groups of U contraction steps, double-buffered: the requests of the next group are in flight under the MFMAs of this one
        // (hipcc does not unroll this loop on its own, and a rolled loop waits for every request)
        // No masks: a row m >= Mall or a frame t >= Tz reads a neighbour's value and fills an output row / column that is never
        // stored; c >= Cc (odd Cc, the last group's overshoot) lies behind both buffer ranges and reads as zero.
        constexpr int U = 4;
        float ga[2][U], gb[2][U][4];
        const int aoff = (lhi * Mall + m) * 4, boff = (lhi * Tz + t0 + l31) * 4;
        auto load_group = [&](int g, float (&a_)[U], float (&b_)[U][4]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k2 = g * U + u;
                a_[u] = cp_ld1(rw, aoff + k2 * (2 * Mall * 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) b_[u][j] = cp_ld1(rc, boff + k2 * (2 * Tz * 4) + 128 * j);
            }
        };
        auto mfma_group = [&](const float (&a_)[U], const float (&b_)[U][4]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[u], b_[u][j], acc[j], 0, 0, 0);
        };
        const int ngroups = (ksteps + U - 1) / U;
        load_group(0, ga[0], gb[0]);
        for (int g = 0; g < ngroups; g += 2) {
            load_group(g + 1, ga[1], gb[1]);
            mfma_group(ga[0], gb[0]);
            load_group(g + 2, ga[0], gb[0]);
            if (g + 1 < ngroups) mfma_group(ga[1], gb[1]);
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
    const int m0 = (blockIdx.x * 4 + wv) * 32, ks = blockIdx.y, cbase = blockIdx.z * 32 * CT;
    if (m0 >= Mall) return;
    const int m = m0 + l31;
    const bool mv = m < Mall;
    const int b_lo = (int)((long)ks * B / KS), b_hi = (int)((long)(ks + 1) * B / KS);
    f32x16 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = cp_zero();
    for (int b = b_lo; b < b_hi; ++b) {
        // one resource per operand row block: the lane's own row of dce, the batch row of cond
        const __amdgpu_buffer_rsrc_t rq = cp_rsrc(dce + (size_t)b * Mall * Tz, (size_t)Mall * Tz), rp = cp_rsrc(cond + (size_t)b * Cc * Tz, (size_t)Cc * Tz);
        const int qoff = m * Tz * 4;
        int poff[CT];
        bool pv[CT];
#pragma unroll
        for (int i = 0; i < CT; ++i) { pv[i] = cbase + 32 * i + l31 < Cc; poff[i] = (cbase + 32 * i + l31) * Tz * 4; }
        // groups of U x 8 frames, double-buffered (see the forward kernel)
        constexpr int U = 4;
        f32x4 gq[2][U], gp[2][U][CT];
        auto load_group = [&](int g, f32x4 (&q_)[U], f32x4 (&p_)[U][CT]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = (g * U + u) * 8 + 4 * lhi;
                const bool tv = t < Tz;                     // (Tz % 4 == 0: a group of four is inside the row or behind it)
                q_[u] = vqw_buf_load4(rq, (mv && tv) ? qoff + t * 4 : CP_OOB, 0);
#pragma unroll
                for (int i = 0; i < CT; ++i) p_[u][i] = vqw_buf_load4(rp, (pv[i] && tv) ? poff[i] + t * 4 : CP_OOB, 0);
            }
        };
        auto mfma_group = [&](const f32x4 (&q_)[U], const f32x4 (&p_)[U][CT]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(p_[u][i][j], q_[u][j], acc[i], 0, 0, 0);
        };
        const int ngroups = (Tz + 8 * U - 1) / (8 * U);
        load_group(0, gq[0], gp[0]);
        for (int g = 0; g < ngroups; g += 2) {
            load_group(g + 1, gq[1], gp[1]);
            mfma_group(gq[0], gp[0]);
            load_group(g + 2, gq[0], gp[0]);
            if (g + 1 < ngroups) mfma_group(gq[1], gp[1]);
        }
    }
    if (!mv) return;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = cbase + 32 * i + (r >> 2) * 8 + lhi * 4 + (r & 3);
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
    const int nct = (Cc + 32 * CT - 1) / (32 * CT);
    const int chunk = blockIdx.x, t = (blockIdx.y / nct) * 32 + l31, b = blockIdx.z, cbase = (blockIdx.y % nct) * 32 * CT;
    const int mlo = chunk * chunk_m, mhi = (mlo + chunk_m < Mall) ? mlo + chunk_m : Mall;
    const bool tv = t < Tz;
    f32x16 acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = cp_zero();
    const __amdgpu_buffer_rsrc_t rq = cp_rsrc(dce + (size_t)b * Mall * Tz, (size_t)Mall * Tz), rp = cp_rsrc(w, (size_t)Cc * Mall);
    const int qoff = t * 4;
    int poff[CT];
    bool pv[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) { pv[i] = cbase + 32 * i + l31 < Cc; poff[i] = (cbase + 32 * i + l31) * Mall * 4; }
    // groups of U x 8 channels m, double-buffered (see the forward kernel)
    constexpr int U = 4;
    float gq[2][U][4];
    f32x4 gp[2][U][CT];
    auto load_group = [&](int g, float (&q_)[U][4], f32x4 (&p_)[U][CT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int mm = mlo + (g * U + u) * 8 + 4 * lhi;
            const bool kv = mm < mhi;                    // (chunk_m and Mall are multiples of 4: a group of four is inside the chunk or behind it)
#pragma unroll
            for (int j = 0; j < 4; ++j) q_[u][j] = cp_ld1(rq, (kv && tv) ? qoff + (mm + j) * Tz * 4 : CP_OOB);
#pragma unroll
            for (int i = 0; i < CT; ++i) p_[u][i] = vqw_buf_load4(rp, (kv && pv[i]) ? poff[i] + mm * 4 : CP_OOB, 0);
        }
    };
    auto mfma_group = [&](const float (&q_)[U][4], const f32x4 (&p_)[U][CT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < CT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(p_[u][i][j], q_[u][j], acc[i], 0, 0, 0);
    };
    const int ngroups = (mhi - mlo + 8 * U - 1) / (8 * U);
    load_group(0, gq[0], gp[0]);
    for (int g = 0; g < ngroups; g += 2) {
        load_group(g + 1, gq[1], gp[1]);
        mfma_group(gq[0], gp[0]);
        load_group(g + 2, gq[0], gp[0]);
        if (g + 1 < ngroups) mfma_group(gq[1], gp[1]);
    }
    if (!tv) return;
    float* pb = part + ((size_t)chunk * B + b) * Cc * Tz + t;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = cbase + 32 * i + (r >> 2) * 8 + lhi * 4 + (r & 3);
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
    const int KS = (B >= 2 && vqw_cdiv(Mall, 32) * ct < 8 * vqw_device_cus()) ? 2 : 1;
    typedef void (*kfn_t)(const float*, const float*, float*, int, int, int, int, int);
    // one 32-row tile of c per wave (grid.z): three times the waves of an all-rows-per-wave layout, each re-reading its dce rows
    // through L2 -- 44 -> 38 us: the kernel is bound by request latency, not by bytes
    const kfn_t kfn = cond_proj_wgrad_kernel<1>;
    hipLaunchKernelGGL(kfn, dim3(vqw_cdiv(Mall, 128), KS, ct), dim3(256), 0, (hipStream_t)s, cond, dce, dw, B, Cc, Mall, Tz, KS);
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
    const kfn_t kfn = cond_proj_dgrad_kernel<1>;
    hipLaunchKernelGGL(kfn, dim3(nchunk, vqw_cdiv(Tz, 32) * ct, B), dim3(64), 0, (hipStream_t)s, w, dce, scratch, B, Cc, Mall, Tz, chunk_m);
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

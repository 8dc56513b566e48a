// VQ nearest-codebook kernels for gfx950 -- reference model.py:57-74 (forward),
// model.py:73,100,103 (gradients), model.py:22-27 + decoder_ops.py:39-43 (speaker tiling).
//
// Bit-exactness contract: the distance is the DIRECT form
// sum_d (z-e)^2 accumulated for d = 0..D-1 in order, with the subtract, the multiply and the
// add each rounded to fp32 (no FMA contraction: explicit __f*_rn intrinsics), and the argmin
// returns the lowest index among equal minima -- identical to oracle/vqw_oracle.c.
#include "vqw_common.h"

namespace {

// One wave per latent row (b, t).  Lane l scans codes l, l+64, ... in increasing order
// (strict '<' keeps the lowest index inside a lane); the 64 per-lane candidates are then
// merged with wavefront xor-shuffles using (distance, index) lexicographic order.
__global__ __launch_bounds__(256) void vq_nearest_fwd_kernel(
    const float* __restrict__ z_e, const float* __restrict__ emb, int64_t* __restrict__ idx,
    float* __restrict__ e_k, float* __restrict__ zq, long zq_bstride, float* __restrict__ mind,
    int rows, int D, int Tz, int K) {
    extern __shared__ float zs_all[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wid;
    float* zs = zs_all + wid * D;
    const bool live = row < rows;
    const int b = live ? row / Tz : 0, t = live ? row % Tz : 0;
    const float* zrow = z_e + (size_t)b * D * Tz + t;
    if (live)
        for (int d = lane; d < D; d += 64) zs[d] = zrow[(size_t)d * Tz];
    __syncthreads();
    if (!live) return;

    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += 64) {
        const float* e = emb + (size_t)k * D;
        float acc = 0.0f;
        for (int d = 0; d < D; d += 4) {
            const f32x4 ev = *reinterpret_cast<const f32x4*>(e + d);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float diff = __fsub_rn(zs[d + u], ev[u]);
                acc = __fadd_rn(acc, __fmul_rn(diff, diff));
            }
        }
        if (acc < best) { best = acc; bi = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (bi == 0x7fffffff) bi = 0;  // all-NaN row: tf.argmin would return 0
    if (lane == 0) {
        idx[row] = bi;
        if (mind) mind[row] = best;
    }
    const float* e = emb + (size_t)bi * D;
    for (int d = lane; d < D; d += 64) {
        const float ek = e[d], z = zs[d];
        if (e_k) e_k[(size_t)b * D * Tz + (size_t)d * Tz + t] = ek;
        if (zq) zq[(size_t)b * zq_bstride + (size_t)d * Tz + t] = __fadd_rn(z, __fsub_rn(ek, z));  // model.py:73
    }
}

__global__ void vq_nearest_bwd_kernel(const float* __restrict__ z_e, const float* __restrict__ e_k,
                                      const int64_t* __restrict__ idx, const float* __restrict__ dzq,
                                      long dzq_bstride, float* __restrict__ dz_e,
                                      float* __restrict__ demb, float cscale, float escale, int B,
                                      int D, int Tz) {
    const size_t n = (size_t)B * D * Tz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % Tz);
        const int d = (int)((i / Tz) % D);
        const int b = (int)(i / ((size_t)Tz * D));
        const float z = z_e[i], e = e_k[i];
        const float g = dzq ? dzq[(size_t)b * dzq_bstride + (size_t)d * Tz + t] : 0.0f;
        if (dz_e) dz_e[i] = g + cscale * (z - e);
        if (demb) unsafeAtomicAdd(demb + (size_t)idx[(size_t)b * Tz + t] * D + d, escale * (e - z));
    }
}

// Codebook gradient through an LDS copy of the whole table: early in training most latents share a few codes
// (measured: 53 k global atomics on ~64 addresses = 308 us), LDS atomics absorb that contention; every block
// owns a slice of the (b, t) rows and flushes only the entries it touched.  dz_e as above.
__global__ __launch_bounds__(256) void vq_nearest_bwd_lds_kernel(const float* __restrict__ z_e, const float* __restrict__ e_k,
                                                                 const int64_t* __restrict__ idx, const float* __restrict__ dzq,
                                                                 long dzq_bstride, float* __restrict__ dz_e, float* __restrict__ demb,
                                                                 float cscale, float escale, int B, int D, int Tz, int K) {
    extern __shared__ float tabl[];   // [K][D]
    const int KD = K * D;
    for (int i = threadIdx.x; i < KD; i += blockDim.x) tabl[i] = 0.0f;
    __syncthreads();
    const size_t n = (size_t)B * D * Tz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % Tz);
        const int d = (int)((i / Tz) % D);
        const int b = (int)(i / ((size_t)Tz * D));
        const float z = z_e[i], e = e_k[i];
        const float g = dzq ? dzq[(size_t)b * dzq_bstride + (size_t)d * Tz + t] : 0.0f;
        if (dz_e) dz_e[i] = g + cscale * (z - e);
        unsafeAtomicAdd(&tabl[(size_t)idx[(size_t)b * Tz + t] * D + d], escale * (e - z));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KD; i += blockDim.x) {
        const float v = tabl[i];
        if (v != 0.0f) unsafeAtomicAdd(demb + i, v);
    }
}

__global__ void speaker_tile_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ spk,
                                        float* __restrict__ cond, long cond_bstride, int row0, int B,
                                        int Cs, int Tz, int n_speakers) {
    const size_t n = (size_t)B * Cs * Tz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % Tz);
        const int j = (int)((i / Tz) % Cs);
        const int b = (int)(i / ((size_t)Tz * Cs));
        // model.py:22: one_hot -> argmax; an id outside the table is an all-zero one-hot row, whose argmax is 0
        const int64_t id = spk[b];
        const size_t row = (id >= 0 && id < n_speakers) ? (size_t)id : 0;
        cond[(size_t)b * cond_bstride + (size_t)(row0 + j) * Tz + t] = table[row * Cs + j];
    }
}

__global__ void speaker_tile_bwd_kernel(const float* __restrict__ dcond, long dcond_bstride, int row0,
                                        const int64_t* __restrict__ spk, float* __restrict__ dtable,
                                        int B, int Cs, int Tz, int n_speakers) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Cs) return;
    const int b = i / Cs, j = i % Cs;
    const float* r = dcond + (size_t)b * dcond_bstride + (size_t)(row0 + j) * Tz;
    float s = 0.0f;
    for (int t = 0; t < Tz; ++t) s += r[t];
    const int64_t id = spk[b];
    const size_t row = (id >= 0 && id < n_speakers) ? (size_t)id : 0;     // same row the forward pass read
    unsafeAtomicAdd(dtable + row * Cs + j, s);
}

}  // namespace

extern "C" int vqw_vq_nearest_fwd(const float* z_e, const float* emb, int64_t* idx, float* e_k, float* zq,
                                  int64_t zq_bstride, float* mind, int B, int D, int Tz, int K,
                                  vqw_stream_t s) {
    VQW_CHECK(z_e && emb && idx, "vqw_vq_nearest_fwd: null pointer");
    VQW_CHECK(B > 0 && Tz > 0 && K > 0 && D > 0 && D % 4 == 0, "vqw_vq_nearest_fwd: D=%d must be a positive multiple of 4", D);
    VQW_CHECK((reinterpret_cast<uintptr_t>(emb) & 15u) == 0, "vqw_vq_nearest_fwd: emb must be 16-byte aligned");
    const int rows = B * Tz;
    hipLaunchKernelGGL(vq_nearest_fwd_kernel, dim3(vqw_cdiv(rows, 4)), dim3(256), 4 * D * sizeof(float),
                       (hipStream_t)s, z_e, emb, idx, e_k, zq, (long)zq_bstride, mind, rows, D, Tz, K);
    VQW_LAUNCH_CHECK("vqw_vq_nearest_fwd");
    return 0;
}

extern "C" int vqw_vq_nearest_bwd(const float* z_e, const float* e_k, const int64_t* idx, const float* dzq,
                                  int64_t dzq_bstride, float* dz_e, float* demb, float cscale,
                                  float escale, int B, int D, int Tz, int K, vqw_stream_t s) {
    VQW_CHECK(z_e && e_k && idx && (dz_e || demb), "vqw_vq_nearest_bwd: null pointer");
    const size_t n = (size_t)B * D * Tz;
    const size_t lds = (size_t)K * D * sizeof(float);
    if (demb && K > 0 && lds <= 144 * 1024) {
        // per call: the attribute belongs to the current device's copy of the kernel (no process-wide "done" flag)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(vq_nearest_bwd_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                144 * 1024) != hipSuccess)
            return vqw_set_error("vqw_vq_nearest_bwd: hipFuncSetAttribute failed");
        int g = (int)((n + 256 * 16 - 1) / (256 * 16));   // ~16 elements per thread: few table flushes
        if (g < 1) g = 1;
        if (g > 64) g = 64;
        hipLaunchKernelGGL(vq_nearest_bwd_lds_kernel, dim3(g), dim3(256), lds, (hipStream_t)s, z_e, e_k, idx, dzq,
                           (long)dzq_bstride, dz_e, demb, cscale, escale, B, D, Tz, K);
        VQW_LAUNCH_CHECK("vqw_vq_nearest_bwd");
        return 0;
    }
    int g = (int)((n + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(vq_nearest_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)s, z_e, e_k, idx, dzq,
                       (long)dzq_bstride, dz_e, demb, cscale, escale, B, D, Tz);
    VQW_LAUNCH_CHECK("vqw_vq_nearest_bwd");
    return 0;
}

extern "C" int vqw_speaker_tile_fwd(const float* table, const int64_t* spk, float* cond, int64_t cond_bstride,
                                    int row0, int B, int Cs, int Tz, int n_speakers, vqw_stream_t s) {
    VQW_CHECK(table && spk && cond && B > 0 && Cs > 0 && Tz > 0 && n_speakers > 0, "vqw_speaker_tile_fwd: bad arguments");
    const size_t n = (size_t)B * Cs * Tz;
    int g = (int)((n + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(speaker_tile_fwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)s, table, spk, cond,
                       (long)cond_bstride, row0, B, Cs, Tz, n_speakers);
    VQW_LAUNCH_CHECK("vqw_speaker_tile_fwd");
    return 0;
}

extern "C" int vqw_speaker_tile_bwd(const float* dcond, int64_t dcond_bstride, int row0, const int64_t* spk,
                                    float* dtable, int B, int Cs, int Tz, int n_speakers, vqw_stream_t s) {
    VQW_CHECK(dcond && spk && dtable && B > 0 && Cs > 0 && Tz > 0 && n_speakers > 0, "vqw_speaker_tile_bwd: bad arguments");
    hipLaunchKernelGGL(speaker_tile_bwd_kernel, dim3(vqw_cdiv(B * Cs, 256)), dim3(256), 0, (hipStream_t)s,
                       dcond, (long)dcond_bstride, row0, spk, dtable, B, Cs, Tz, n_speakers);
    VQW_LAUNCH_CHECK("vqw_speaker_tile_bwd");
    return 0;
}

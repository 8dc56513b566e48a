// The fp16x3 engine: every contraction of the decoder's residual stack -- the gate conv
// (wavenet_ops.py:104-114), the 1x1 residual conv (:135-136), the skip path of all layers as ONE contraction (:132-133,
// wavenet.py:72), gate backward, the gate conv's input gradient and both weight gradients -- as fp32-accurate
// contractions on the fp16 matrix pipe of gfx950: every fp32 operand is split into two fp16 pieces, x = h1 + h2,
// h2 = fp16(x - h1), and  a*b ~ a1 b1 + a1 b2 + a2 b1  (every term exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_f16; the dropped a2 b2 is 2^-22 per product).  Same bytes per operand element as fp32, three
// MFMAs of the fast pipe instead of eight fp32 ones per 32x32x16 block.  Device-side range guards (power-of-two scales
// from measured max-abs values, a range flag) keep the leading planes inside fp16; the same kernels instantiated with
// one bf16 plane and one bf16 MFMA per product are the bf16 engine of BASELINE.json configs[4].
//
// Operand planes ("chunk-major"): P[plane 0..1][channel chunk of 8][row][8 fp16]; a row is an output channel
// (weights) or a (batch, time) position (activations).  The 16-byte entries of 32 consecutive rows are contiguous:
// one 64-lane x 16-byte load fetches a 32-row x 16-k MFMA operand fragment in the lane order of the MFMA, and the
// dilation shift of a tap is a row offset (rows before the start of a batch row read as zero through the buffer
// range check: the causal left padding of conv1d_v2, wavenet_ops.py:81).
//
// Conv kernels: block = 256 (or 128) output rows x 256 time steps, four waves of all rows x 64 time steps (16 or 8
// accumulator tiles); K step = 16 input channels of one tap; operands through VGPRs into an LDS ring, one barrier per
// step, the MFMAs issued with the step's memory instructions interleaved behind them (f16x3_mainloop).  Weight-gradient
// kernel: the fp32 operands themselves, split in registers (wgrad_f16x3_kernel).
#include <string.h>

#include "vqw_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u16 f16_bits(_Float16 h) { return __builtin_bit_cast(u16, h); }
__device__ __forceinline__ u16 bf16_bits(float x) { return __builtin_bit_cast(u16, (__bf16)x); }   // round to nearest even
// BF (every template below): the bf16 engine -- ONE bf16 plane per operand, one v_mfma_f32_32x32x16_bf16 per product
// (bf16 storage + fp32 accumulate, BASELINE.json configs[4]) instead of two fp16 planes and three fp16 MFMA terms.
template <bool BF = false>
__device__ __forceinline__ void split8(const float (&x)[8], uint4& p0, uint4& p1) {
    u16 a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (BF) { a[e] = bf16_bits(x[e]); b[e] = 0; continue; }
        const _Float16 h1 = (_Float16)x[e];            // round to nearest even
        a[e] = f16_bits(h1);
        b[e] = f16_bits((_Float16)(x[e] - (float)h1)); // the difference is exact in fp32
    }
    p0 = make_uint4(a[0] | ((unsigned)a[1] << 16), a[2] | ((unsigned)a[3] << 16), a[4] | ((unsigned)a[5] << 16), a[6] | ((unsigned)a[7] << 16));
    p1 = make_uint4(b[0] | ((unsigned)b[1] << 16), b[2] | ((unsigned)b[3] << 16), b[4] | ((unsigned)b[5] << 16), b[6] | ((unsigned)b[7] << 16));
}

// ---- range guards: the leading plane of scale * x must stay inside fp16 (|.| <= 65504).  Scales are
// powers of two held in DEVICE memory (chosen from measured max-abs values, vqw_f16x3_update_scales); a kernel that
// writes planes reports the max-abs of its fp32 values (atomicMax on the bit pattern: monotonic for non-negative
// floats) and raises `flag` when an element leaves the range or is not finite -- the host then repeats the step on the
// fp32 engine.
constexpr float F16_MAX = 65504.0f;
__device__ __forceinline__ float dev_scale(const float* p) { return p ? *p : 1.0f; }
// 1 / (sa * sb) for powers of two (v_rcp_f32 is exact on them)
__device__ __forceinline__ float inv_scales(float host_inv, const float* sa, const float* sb) {
    return (sa || sb) ? host_inv * __builtin_amdgcn_rcpf(dev_scale(sa) * dev_scale(sb)) : host_inv;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// called by all lanes of a wave: amax of the wave's fp32 values (`m`, NaN-propagating through `bad`), range check of m * ps
__device__ __forceinline__ void guard_report(float m, bool bad, float ps, unsigned* amax, int* flag) {
    if (!amax && !flag) return;
    const float wm = wave_max(m);
    const bool wbad = __any(bad || !(m * ps <= F16_MAX));
    if ((threadIdx.x & 63) == 0) {
        // look before the atomic: tens of thousands of waves hitting ONE address serialise (53 k contended atomics cost
        // ~300 us); after the first few, a plain (L2-coherent) read shows a value that is already at least as large
        const unsigned bits = __float_as_uint(wm);
        if (amax && bits > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, bits);
        if (flag && wbad) atomicOr(flag, 1);
    }
}

// x [B][C][T] fp32 -> planes [2][C/8][B*T][8] fp16 of scale * scale_dev[0] * x
// s2d (the input of a stride-2 conv; T even): planes [2][2 C/8][B*T/2][8], sample t = 2 t' + r of channel c in chunk block r
// (chunks r C/8 ..), row b T/2 + t' -- the strided conv then reads whole consecutive rows per tap (LoopGeom, TAB)
template <bool BF>
__global__ void split_act_kernel(const float* __restrict__ x, uint4* __restrict__ planes, int B, int C, int T, float scale, int kc0, int KC,
                                 const float* __restrict__ scale_dev, unsigned* amax, int* flag, int s2d) {
    const size_t NB = (size_t)B * T;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < NB * (C / 8);
    const float sc = scale * dev_scale(scale_dev);
    float m = 0.0f;
    bool bad = false;
    if (live) {
        const size_t row = i % NB;
        const int kc = (int)(i / NB);
        const int b = (int)(row / T), t = (int)(row % T);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xv = x[((size_t)b * C + kc * 8 + e) * T + t];
            v[e] = xv * sc;
            bad |= !(fabsf(xv) <= 3.0e38f);
            m = fmaxf(m, fabsf(xv));
        }
        uint4 p0, p1;
        split8<BF>(v, p0, p1);
        if (s2d) {
            const size_t NH = NB / 2, r2 = (size_t)b * (T / 2) + t / 2;
            const int kcs = (t & 1) * (C / 8) + kc;
            planes[(size_t)kcs * NH + r2] = p0;
            if (!BF) planes[((size_t)(C / 4) + kcs) * NH + r2] = p1;
        } else {
            planes[(size_t)(kc0 + kc) * NB + row] = p0;
            if (!BF) planes[((size_t)KC + kc0 + kc) * NB + row] = p1;
        }
    }
    if (!amax && !flag) return;
    // one atomic per block (see guard_report: same-address atomics of tens of thousands of waves serialise)
    __shared__ float smax[4];
    __shared__ int sbad[4];
    const float wm = wave_max(m);
    const bool wbad = __any(bad || !(m * sc <= F16_MAX));
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = wm; sbad[threadIdx.x >> 6] = wbad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned bits = __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
        if (amax && bits > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, bits);
        if (flag && (sbad[0] | sbad[1] | sbad[2] | sbad[3])) atomicOr(flag, 1);
    }
}

// max |x| over `count` strided matrices [rows][cols] (row stride ld, matrix stride mstride) -> amax[blockIdx.y]
__global__ void amax_kernel(const float* __restrict__ x, long rows, int cols, long ld, long mstride, unsigned* amax, int* flag) {
    const float* xb = x + (size_t)blockIdx.y * mstride;
    const size_t n = (size_t)rows * cols;
    float m = 0.0f;
    bool bad = false;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if (ld == cols && (n & 3) == 0 && (reinterpret_cast<uintptr_t>(xb) & 15u) == 0) {      // contiguous: 16 bytes per lane, four requests in flight
        const f32x4* x4 = reinterpret_cast<const f32x4*>(xb);
        const size_t n4 = n >> 2;
        auto fold = [&](const f32x4 v) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = fabsf(v[e]);
                bad |= !(a <= 3.0e38f);
                m = fmaxf(m, a);
            }
        };
        size_t i = tid;
        for (; i + 7 * nthr < n4; i += 8 * nthr) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(x4 + i + u * nthr);
#pragma unroll
            for (int u = 0; u < 8; ++u) fold(v[u]);
        }
        for (; i < n4; i += nthr) fold(__builtin_nontemporal_load(x4 + i));
    } else {
        for (size_t i = tid; i < n; i += nthr) {
            const float v = xb[(i / cols) * ld + (i % cols)];
            bad |= !(fabsf(v) <= 3.0e38f);
            m = fmaxf(m, fabsf(v));
        }
    }
    // one atomic per BLOCK (thousands of waves hitting one address serialise in L2: 8 k wave-level atomics cost ~70 us)
    __shared__ float smax[4];
    __shared__ int sbad[4];
    const float wm = wave_max(m);
    const bool wbad = __any(bad);
    if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = wm; sbad[threadIdx.x >> 6] = wbad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned bits = __float_as_uint(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3])));
        unsigned* dst = amax + blockIdx.y;
        if (bits > __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dst, bits);
        if (flag && (sbad[0] | sbad[1] | sbad[2] | sbad[3])) atomicOr(flag, 1);
    }
}

// scale[i] = 2^(target_exp - 1 - floor(log2(amax[i]))): amax * scale in [2^(target_exp-1), 2^target_exp); a slot that saw
// nothing (amax == 0) keeps its scale; reset: amax[i] = 0 afterwards (the next step collects afresh)
__global__ void update_scales_kernel(unsigned* amax, float* scale, int n, int target_exp, int reset, int* flag, const int* skip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (skip && *skip)) return;      // (skip: a voided step of the deferred guard leaves scales and max-abs slots alone)
    const unsigned b = amax[i];
    if (b >= 0x7f800000u) {                  // inf / NaN was seen
        if (flag) atomicOr(flag, 1);
    } else if (b != 0) {
        int e = (int)(b >> 23) - 127;        // floor(log2(amax)); denormals: -127
        int se = target_exp - 1 - e;
        se = se > 100 ? 100 : (se < -100 ? -100 : se);
        scale[i] = __uint_as_float((unsigned)(se + 127) << 23);
    } else if (scale[i] == 0.0f) {
        scale[i] = 1.0f;
    }
    if (reset) amax[i] = 0;
}

// w [ks][R][ldw] (kernel[k, Cin, Cout], filter columns 0..R-1, gate columns R..2R-1) -> planes [2][ks*R/8][2R][8],
// rows in block order: row m' = 256 mt + i is filter channel 128 mt + i (i < 128) or gate channel 128 mt + i - 128
template <bool BF>
__global__ void pack_gate_w_kernel(const float* __restrict__ w, uint4* __restrict__ planes, int ks, int R, int ldw, float scale,
                                   const float* __restrict__ scale_dev, int hb) {     // hb: rows of a block (256 or 128)
    scale *= dev_scale(scale_dev);
    const int M = 2 * R, KC = ks * R / 8;
    w += (size_t)blockIdx.y * ks * R * ldw;          // one layer per grid row
    planes += (size_t)blockIdx.y * 2 * KC * M;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KC * M) return;
    const int mp = i % M, kc = i / M;
    const int mt = mp / hb, ii = mp % hb, hh = hb / 2;          // a block holds hh filter rows, then the hh matching gate rows
    const int col = ii < hh ? hh * mt + ii : R + hh * mt + (ii - hh);
    const int j = kc / (R / 8), c0 = (kc % (R / 8)) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[((size_t)j * R + c0 + e) * ldw + col] * scale;
    uint4 p0, p1;
    split8<BF>(v, p0, p1);
    planes[(size_t)kc * M + mp] = p0;
    if (!BF) planes[((size_t)KC + kc) * M + mp] = p1;
}

// w [K][ldw] fp32 (columns = output rows m) -> planes [2][K/8][M][8], natural row order
template <bool BF>
__global__ void pack_w_kernel(const float* __restrict__ w, uint4* __restrict__ planes, int K, int M, int ldw, float scale,
                              const float* __restrict__ scale_dev) {
    scale *= dev_scale(scale_dev);
    const int KC = K / 8;
    w += (size_t)blockIdx.y * K * ldw;
    planes += (size_t)blockIdx.y * 2 * KC * M;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KC * M) return;
    const int m = i % M, kc = i / M;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[(size_t)(kc * 8 + e) * ldw + m] * scale;
    uint4 p0, p1;
    split8<BF>(v, p0, p1);
    planes[(size_t)kc * M + m] = p0;
    if (!BF) planes[((size_t)KC + kc) * M + m] = p1;
}

// The same planes from the TRANSPOSED storage: logical row k = jb * Kin + ko, column m is src[jb * blk_stride + m * ld_src + ko]
// (blocks of Kin rows: the taps of a conv kernel [tap][m][Kin], transposed tap by tap) -- the kernels of the input-gradient GEMMs
// straight from the parameters, without a transposed fp32 copy in between.  A thread reads its 8 rows as 32 contiguous bytes.
template <bool BF>
__global__ void pack_w_t_kernel(const float* __restrict__ w, uint4* __restrict__ planes, int K, int M, int Kin, int ld_src, long blk_stride,
                                float scale, const float* __restrict__ scale_dev) {
    scale *= dev_scale(scale_dev);
    const int KC = K / 8;
    w += (size_t)blockIdx.y * (K / Kin) * blk_stride;
    planes += (size_t)blockIdx.y * 2 * KC * M;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KC * M) return;
    const int m = i % M, kc = i / M, k0 = kc * 8, jb = k0 / Kin, ko = k0 - jb * Kin;
    const float* src = w + (size_t)jb * blk_stride + (size_t)m * ld_src + ko;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e] * scale; v[4 + e] = b[e] * scale; }
    uint4 p0, p1;
    split8<BF>(v, p0, p1);
    planes[(size_t)kc * M + m] = p0;
    if (!BF) planes[((size_t)KC + kc) * M + m] = p1;
}

// ... through LDS, for blocks of k that are multiples of 64: a tile of 64 columns m x 64 rows k is READ with k fastest (eight
// threads = 256 contiguous bytes of a source row) and WRITTEN with m fastest (64 threads = 1 KiB of a plane), the 16-byte plane
// entries change hands in LDS.  The direct kernel reads 32-byte pieces 1-3 KB apart: 17.3 us per launch on the step's seven packs
// against 10.9 us (tools/pmc_step.sh).
template <bool BF>
__global__ __launch_bounds__(256) void pack_w_t_tile_kernel(const float* __restrict__ w, uint4* __restrict__ planes, int K, int M, int Kin, int ld_src,
                                                            long blk_stride, float scale, const float* __restrict__ scale_dev) {
    __shared__ uint4 lp[2][8][65];
    scale *= dev_scale(scale_dev);
    const int KC = K / 8, mt = (M + 63) / 64;
    const int m0 = (blockIdx.x % mt) * 64, kc0 = (blockIdx.x / mt) * 8, k0 = kc0 * 8, jb = k0 / Kin, ko0 = k0 - jb * Kin;
    w += (size_t)blockIdx.y * (K / Kin) * blk_stride + (size_t)jb * blk_stride + ko0;
    planes += (size_t)blockIdx.y * 2 * KC * M;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = threadIdx.x + 256 * r, mm = e >> 3, kq = e & 7;
        float v[8];
        if (m0 + mm < M) {
            const float* src = w + (size_t)(m0 + mm) * ld_src + kq * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = a[i] * scale; v[4 + i] = b[i] * scale; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.0f;
        }
        uint4 p0, p1;
        split8<BF>(v, p0, p1);
        lp[0][kq][mm] = p0;
        if (!BF) lp[1][kq][mm] = p1;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = threadIdx.x + 256 * r, kq = e >> 6, mm = e & 63;
        if (m0 + mm >= M) continue;
        planes[(size_t)(kc0 + kq) * M + m0 + mm] = lp[0][kq][mm];
        if (!BF) planes[((size_t)KC + kc0 + kq) * M + m0 + mm] = lp[1][kq][mm];
    }
}

struct GateArgs {
    vqw_f16x3_gate_desc d;
    int NB;        // B * T rows of the activation planes
    int ratio;     // T / cond_T
};
struct OutArgs {
    vqw_f16x3_out_desc d;
    int NB;
};

#ifndef VQW_WG_PAT_A
#define VQW_WG_PAT_A 4     // weight-gradient loop: VALU / LDS instructions per MFMA while a chunk is converted ...
#define VQW_WG_PAT_B 2      // ... and address computations / requests per MFMA in the second stage
#endif

#ifndef VQW_SCONV_64
#define VQW_SCONV_64 1        // strided conv: 64-row blocks for launches that would leave three quarters of the chip idle
#endif
#ifndef VQW_SCONV_192
#define VQW_SCONV_192 1       // strided conv: 192-row blocks where they fill more CUs than 256-row blocks (tools/sconv_bench.py)
#endif
#ifndef VQW_X3_TAP_MINOR
#define VQW_X3_TAP_MINOR 1      // K order of the conv main loop (0: taps outermost, as in round 1)
#endif
#ifndef VQW_X3_PAT_MEM
#define VQW_X3_PAT_MEM 1      // issue pattern of the conv main loop: LDS / global-memory instructions per MFMA ...
#define VQW_X3_PAT_ALU 2      // ... and address computations per MFMA (tools/x3_bench.py: 1/2 measured best)
#endif
constexpr int NSTG = 4, STG_BYTES = 32 * 1024, PIECES = 8;   // per wave and stage: 4 weight + 4 activation pieces of 1 KiB

// Geometry of one block's contraction: 256 weight rows from row m_row0 of planes with M rows, K = ks taps x Cin
// input channels, 256 activation rows from row n0 (time t0 of its batch row) of planes with NB rows.
struct LoopGeom {
    const void* wp;
    const void* xp;
    int M, Cin, ks, dilation, NB, m_row0, n0, t0;
    int xKC, xkc0;   // chunks per plane of xp and the first chunk of this contraction (planes may hold more channels)
    int dir, T;      // dir > 0: tap j reads x[t - (ks-1-j) d] (causal conv); dir < 0: x[t + (ks-1-j) d] (its input gradient)
    // TAB (stride-2 convs, encoder.py:17-18): the loop's tap i is tap j = tj0 + tjstep * i of a kernel with `wks` taps (the weight
    // planes hold all of them); with e = j - toff it reads activation row t - tsgn * (e >> 1) of chunk block (ts2d ? e & 1 : 0):
    //   forward over space-to-depth planes (chunk block = parity of the input sample): tsgn = -1, toff = pad_left, ts2d = 1;
    //   input gradient of output parity r: tj0 = (r + pad_left) & 1, tjstep = 2, toff = r + pad_left, tsgn = +1, ts2d = 0.
    // n0 may lie anywhere in the flat (batch, time) row space: rows outside the lane's own batch row read as zero.
    int wks, tj0, tjstep, toff, tsgn, ts2d;
    int sb, sn;      // TAB: this block's K steps [sb, sb + sn) of the ks * Cin / 16 (split-K of the short encoder layers); sn even
};

// acc[i][j] += W[m_row0 + 32 i .., :] X[:, n0 + 64 wv + 32 j ..]: MR x 2 accumulator tiles per wave, operands through
// VGPRs into an LDS ring, one barrier per K step of 16.
//   MR = 8: block = 256 rows x 256 columns, 16 accumulator tiles (256 AGPRs) per wave, one block per CU, 4 LDS stages of
//           32 KiB, two stages of requests in flight.
//   MR = 4: block = 128 rows x 256 columns, 8 accumulator tiles per wave, 256 registers per wave => TWO blocks per CU
//           (3 stages of 24 KiB each): while one block is in its HBM-bound epilogue (all blocks of a one-block-per-CU
//           grid reach it together: gate backward spent 64 % of its time there) the other block's MFMAs run; each weight
//           panel is read by twice as many blocks (all of the loop's data movement is 16 % of the MR = 8 kernel).
// Two blocks that share a CU start together and would reach their epilogues together; the blocks of every second
// batch of `CUs` block ids wait a fraction of a main loop first, so that one block stores while the other multiplies.
#ifndef VQW_X3_STAGGER
#define VQW_X3_STAGGER 0      // in units of 8128 cycles (~3.7 us)
#endif
__device__ __forceinline__ void x3_stagger(int n) {
    if (VQW_X3_STAGGER > 0 && ((blockIdx.x >> 8) & 1))
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
template <int MR> struct X3Shape {
    static constexpr int NSTAGE = MR == 8 ? 4 : 3;
    static constexpr int STAGE_BYTES = (MR * 2 + 16) * 1024;     // MR*2 weight pieces + 16 activation pieces of 1 KiB
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
    static constexpr int DEPTH = MR == 8 ? 2 : 1;                 // stages of requests in flight
};
template <bool BF, int MR, bool TAB = false, int DEPTH = X3Shape<MR>::DEPTH>
__device__ __forceinline__ void f16x3_mainloop(f32x16 (&acc)[MR][2], char* smem, const LoopGeom& g, int wv, int lane) {
    constexpr int NP = BF ? 1 : 2;        // planes per operand
    constexpr int NA = MR * NP / 4, NBP = 2 * NP;      // weight / activation pieces a wave moves per step
    if (MR == 4) x3_stagger(VQW_X3_STAGGER);
    constexpr int NSTG_ = DEPTH + 2, STGB = X3Shape<MR>::STAGE_BYTES, BOFF = MR * 2 * 1024;      // (DEPTH: stages of requests in flight)
    const int l31 = lane & 31, lhi = lane >> 5;
    const int KCA = (TAB ? g.wks : g.ks) * g.Cin / 8, KCB = g.Cin / 8, spt = g.Cin / 16;   // spt: K steps per tap
    const int nsteps = TAB ? g.sn : g.ks * spt;
    const __amdgpu_buffer_rsrc_t ra = vqw_make_rsrc(g.wp, (unsigned)((size_t)2 * KCA * g.M * 16));
    // One buffer resource per activation plane, based at the contraction's first chunk: 32-bit offsets then only span the
    // chunks this contraction reads (checked by the callers: Cin / 8 * NB * 16 < 2 GiB), not the planes tensor -- the gated planes
    // of all layers side by side are 3.3 GB at batch 16 and neither their size nor the distance between the two planes may
    // decide whether the engine can run.
    const size_t plane_bytes = (size_t)g.xKC * g.NB * 16, span = (size_t)(g.xKC - g.xkc0) * g.NB * 16;
    const char* xbase = reinterpret_cast<const char*>(g.xp) + (size_t)g.xkc0 * g.NB * 16;
    const unsigned xspan = span < 0x7fffffffu ? (unsigned)span : 0x7fffffffu;
    const __amdgpu_buffer_rsrc_t rb0 = vqw_make_rsrc(xbase, xspan);
    const __amdgpu_buffer_rsrc_t rb1 = vqw_make_rsrc(xbase + (NP > 1 ? plane_bytes : 0), xspan);
    // Stage image: MR*2 weight pieces (row tile i, plane p at (i * 2 + p) KiB), then 16 activation pieces.  lane = (k half, row)
    // as the MFMA wants it.
    int voffA[NA], pieceA[NA], voffB[NBP], trow[NBP], pieceB[NBP];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = wv * NA + i, tile = q / NP, p = q % NP;
        pieceA[i] = tile * 2 + p;
        voffA[i] = ((p * KCA + lhi) * g.M + g.m_row0 + tile * 32 + l31) * 16;
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int q = wv * NBP + i, tile = q / NP, p = q % NP;
        pieceB[i] = tile * 2 + p;
        voffB[i] = (lhi * g.NB + g.n0 + tile * 32 + l31) * 16;      // (plane p = i % NP: its own resource)
        trow[i] = TAB ? (g.n0 + tile * 32 + l31) % g.T : g.t0 + tile * 32 + l31;      // time of this lane's activation row
    }
    f32x4 rgA[NA + NBP], rgB[NA + NBP];
    auto rissue = [&](int s_, f32x4 (&rg)[NA + NBP]) {
        const int s = TAB ? s_ + g.sb : s_;
#if VQW_X3_TAP_MINOR
        // K order: channel chunk outermost, taps innermost -- the taps of one chunk read the same activation lines a few rows apart,
        // in consecutive steps instead of Cin / 16 steps apart, while they are still in L2 (HBM reads of the gate conv d=8: 172 -> 104 MB)
        const int ji = TAB ? s / spt : s % g.ks, kc = TAB ? (s - ji * spt) * 2 : (s / g.ks) * 2;
#else
        const int ji = s / spt, kc = (s - ji * spt) * 2;
#endif
        const int j = TAB ? g.tj0 + g.tjstep * ji : ji;
        const int e_ = TAB ? j - g.toff : 0;
        const int shift = TAB ? g.tsgn * (e_ >> 1) : (g.ks - 1 - j) * g.dilation * (g.dir < 0 ? -1 : 1);   // rows before / behind the batch row read as zero
        const int kcx = TAB ? kc + (g.ts2d ? (e_ & 1) * KCB : 0) : kc;
#pragma unroll
        for (int i = 0; i < NA; ++i) rg[i] = vqw_buf_load4(ra, voffA[i] + (j * KCB + kc) * g.M * 16, 0);
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int tr = trow[i] - shift;
            const int vb = (tr >= 0 && tr < g.T) ? voffB[i] + (kcx * g.NB - shift) * 16 : (int)0x80000000;   // out of range -> 0
            rg[NA + i] = vqw_buf_load4((i % NP) ? rb1 : rb0, vb, 0);      // (wv * NBP is even: plane = i % NP at compile time)
        }
    };
    auto rcommit = [&](int s, const f32x4 (&rg)[NA + NBP]) {
        char* dst = smem + (s % NSTG_) * STGB + lane * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<f32x4*>(dst + pieceA[i] * 1024) = rg[i];
#pragma unroll
        for (int i = 0; i < NBP; ++i) *reinterpret_cast<f32x4*>(dst + BOFF + pieceB[i] * 1024) = rg[NA + i];
    };
    // Fragments: ONE set of A fragments (MR row tiles x NP planes) that is refilled row tile by row tile -- right behind the
    // MFMAs of a row tile its fragments of the NEXT stage are fetched from LDS, a whole step before they are used -- and two
    // sets of this wave's B fragments.
    uint4 fa[MR][NP], fb[2][NP], fbn[2][NP];
    auto read_a = [&](int i, int s) {
        const char* st = smem + (s % NSTG_) * STGB + lane * 16;
#pragma unroll
        for (int p = 0; p < NP; ++p) fa[i][p] = *reinterpret_cast<const uint4*>(st + (i * 2 + p) * 1024);
    };
    auto read_b = [&](uint4 (&b)[2][NP], int s) {
        const char* st = smem + (s % NSTG_) * STGB + lane * 16 + BOFF;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < NP; ++p) b[j][p] = *reinterpret_cast<const uint4*>(st + ((wv * 2 + j) * 2 + p) * 1024);
    };
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Two row tiles x two column tiles at a time, the three terms outermost (small terms first): an MFMA that accumulates
    // into the same tile as its predecessor waits for that one's whole latency, so the four independent accumulators are
    // rotated through between two terms of the same tile.
    auto mfma_rows = [&](int i, const uint4 (&b)[2][NP]) {
        if constexpr (BF) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[i + (q >> 1)][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i + (q >> 1)][0]), __builtin_bit_cast(bf16x8, b[q & 1][0]), acc[i + (q >> 1)][q & 1], 0, 0, 0);
        } else {
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ii = i + (q >> 1), j = q & 1;
                    const int pa = term == 1 ? 1 : 0, pb = term == 0 ? 1 : 0;      // h1 h2, h2 h1, h1 h1
                    acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ii][pa]), __builtin_bit_cast(f16x8, b[j][pb]), acc[ii][j], 0, 0, 0);
                }
        }
    };

    rissue(0, rgA); rcommit(0, rgA);
    rissue(1, rgA); rcommit(1, rgA);
    rissue(2, rgA);
    if (DEPTH == 2) rissue(3, rgB);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < MR; ++i) read_a(i, 0);
    read_b(fb, 0);
    // One step s (one barrier): stage s + 1 is complete in LDS behind the barrier and the slot of stage s is free (its
    // fragments are in registers); the MFMAs of stage s run row-tile pair by row-tile pair with the LDS reads of stage s + 1
    // behind them; stage s + 2 (requested DEPTH steps ago) goes to LDS and stage s + 2 + DEPTH is requested.  nsteps is even
    // (Cin % 32 == 0 is checked by the callers).
    auto step = [&](const uint4 (&bc)[2][NP], uint4 (&bn)[2][NP], int s, f32x4 (&rg)[NA + NBP]) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int sn = s + 1 < nsteps ? s + 1 : s;          // (the last step re-reads its own stage: no branch in the body)
        // No conditionals in the body (they would cut it into scheduling regions): past the last stages the commit rewrites
        // stale registers into an LDS stage nobody reads again and the requests run past the end of the weight planes,
        // where raw buffer loads return zero.
        // (VQW_ABL_*: timing ablations, wrong results, never shipped -- which of the three data paths bounds the loop)
#ifndef VQW_ABL_NOLDSREAD
        read_b(bn, sn);
#endif
#ifndef VQW_ABL_NOCOMMIT
        rcommit(s + 2, rg);
#endif
#ifndef VQW_ABL_NOGLOBAL
        rissue(s + 2 + DEPTH, rg);
#endif
#pragma unroll
        for (int i = 0; i < MR; i += 2) {
            mfma_rows(i, bc);
#ifndef VQW_ABL_NOLDSREAD
            read_a(i, sn);
            read_a(i + 1, sn);
#endif
        }
        // issue order: one MFMA, then one LDS / global-memory instruction and a few address computations in its shadow
#pragma unroll
        for (int k_ = 0; k_ < (BF ? 2 : 6) * MR; ++k_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100 | 0x200 | 0x020, BF ? 2 : VQW_X3_PAT_MEM, 0);
            __builtin_amdgcn_sched_group_barrier(0x002 | 0x004, VQW_X3_PAT_ALU, 0);
        }
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(fb, fbn, s, rgA);
        step(fbn, fb, s + 1, DEPTH == 2 ? rgB : rgA);
    }
}

// Four consecutive channels (rows 4 lhi .. 4 lhi + 3 of chunk kc) of one (batch, time) row as the lane's 8-byte
// half of the 16-byte plane entries: the 64 lanes of a wave cover 32 rows x 16 bytes = 512 contiguous bytes per plane.
template <bool BF>
__device__ __forceinline__ void store_plane_quad(void* planes, int KC, int NB, int kc, int row, int lhi, const float (&x)[4], float scale = 1.0f) {
    u16 h1[4], h2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xs = x[e] * scale;
        if (BF) { h1[e] = bf16_bits(xs); h2[e] = 0; continue; }
        const _Float16 a = (_Float16)xs;
        h1[e] = f16_bits(a);
        h2[e] = f16_bits((_Float16)(xs - (float)a));
    }
    char* base = reinterpret_cast<char*>(planes) + ((size_t)kc * NB + row) * 16 + lhi * 8;
    *reinterpret_cast<uint2*>(base) = make_uint2(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16));
    if (!BF) *reinterpret_cast<uint2*>(base + (size_t)KC * NB * 16) = make_uint2(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16));
}

template <bool BF, int MR>
__global__ __launch_bounds__(256, MR == 8 ? 1 : 2) void gate_f16x3_kernel(const GateArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const vqw_f16x3_gate_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int R = d.R, T = d.T;
    constexpr int HB = 32 * MR, HH = HB / 2;                 // block rows: HH filter channels, then the HH matching gate channels
    const int n_mt = R / HH;
    const int bid = vqw_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid % n_mt, n0 = (bid / n_mt) * 256;     // 256 | T: a block never straddles two batch rows
    const int b = n0 / T, t0 = n0 - b * T;
    f32x16 acc[MR][2];
    {
        LoopGeom g;
        g.wp = d.wp; g.xp = d.xp; g.M = 2 * R; g.Cin = R; g.ks = d.ks; g.dilation = d.dilation; g.NB = a.NB;
        g.xKC = R / 8; g.xkc0 = 0; g.dir = 1; g.T = T;
        g.m_row0 = mt * HB; g.n0 = n0; g.t0 = t0;
        f16x3_mainloop<BF, MR>(acc, smem, g, wv, lane);
    }

    // ---- epilogue: + bias + upsampled condition (add_condition, wavenet_ops.py:93-101), tanh(filter) * sigmoid(gate).
    // With one block per CU nothing overlaps it, so it is kept lean: one 64-bit row offset per four channels, 32-bit
    // offsets inside, v_rcp_f32 instead of IEEE divisions (1 ulp; the quotients feed tanh/sigmoid values in [-1, 1]).
    const bool hb = d.bias != nullptr, hc = d.cond != nullptr;
    const float* bp = hb ? d.bias : reinterpret_cast<const float*>(d.wp);      // absent operands read a valid dummy address
    const float* cb = hc ? d.cond + (size_t)b * d.cond_bstride : reinterpret_cast<const float*>(d.wp);
    const int tcol = t0 + 64 * wv + l31;
    const int tz0 = (t0 + 64 * wv) / a.ratio, tz1 = (t0 + 64 * wv + 32) / a.ratio;   // 32 | ratio: one frame per tile row
    const bool s0 = d.save0 != nullptr, s1 = d.save1 != nullptr;
    const float winv = inv_scales(d.w_scale_inv, d.x_scale, d.w_scale);
#pragma unroll
    for (int i = 0; i < MR / 2; ++i)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            const int c0 = HH * mt + 32 * i + 8 * v4 + 4 * lhi;          // first of this lane's four channels
            float addf[4][2], addg[4][2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bfv = bp[hb ? c0 + e : 0], bgv = bp[hb ? R + c0 + e : 0];
                const int cf = (c0 + e) * d.cond_T, cg = (R + c0 + e) * d.cond_T;
                const float f0 = cb[hc ? cf + tz0 : 0], f1 = cb[hc ? cf + tz1 : 0];
                const float g0 = cb[hc ? cg + tz0 : 0], g1 = cb[hc ? cg + tz1 : 0];
                addf[e][0] = (hb ? bfv : 0.0f) + (hc ? f0 : 0.0f); addf[e][1] = (hb ? bfv : 0.0f) + (hc ? f1 : 0.0f);
                addg[e][0] = (hb ? bgv : 0.0f) + (hc ? g0 : 0.0f); addg[e][1] = (hb ? bgv : 0.0f) + (hc ? g1 : 0.0f);
            }
            const size_t off = ((size_t)b * R + c0) * T + tcol;
            float* po = (d.out0 ? d.out0 : d.save1) + off;
            float* p0 = s0 ? d.save0 + off : po;
            float* p1 = s1 ? d.save1 + off : po;
            float gq[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float xf = acc[i][j][v4 * 4 + e] * winv + addf[e][j];
                    const float xg = acc[i + MR / 2][j][v4 * 4 + e] * winv + addg[e][j];
                    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * xf) + 1.0f);
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-xg));
                    const int o = e * T + 32 * j;
                    gq[j][e] = th * sg;
                    if (s0) p0[o] = th;
                    if (s1) p1[o] = sg;
                    if (d.out0) po[o] = gq[j][e];        // (optional where every reader takes the planes: 54 MB less per layer)
                }
            if (d.out_planes) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    store_plane_quad<BF>(d.out_planes, d.out_planes_KC > 0 ? d.out_planes_KC : R / 8, a.NB, d.out_planes_kc0 + (HH / 8) * mt + 4 * i + v4,
                                     n0 + 64 * wv + 32 * j + l31, lhi, gq[j]);
            }
        }
}

// The layer's 1x1 skip + residual conv (wavenet_ops.py:132-136, wavenet.py:72-73) on the gated planes:
// rows 0..S-1: skip += W_s g + b_s; rows S..S+R-1: net' = net + W_r g + b_r (and net' as planes for the next gate conv).
template <bool BF, int MR>
__global__ __launch_bounds__(256, MR == 8 ? 1 : 2) void out_f16x3_kernel(const OutArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const vqw_f16x3_out_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int R = d.R, S = d.S, T = d.T, M = S + R;      // S skip rows, then R residual rows (either may be 0)
    const int Cin = d.Cin > 0 ? d.Cin : R;
    constexpr int HB = 32 * MR;
    const int n_mt = M / HB;
    const int bid = vqw_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid % n_mt, n0 = (bid / n_mt) * 256;
    const int b = n0 / T, t0 = n0 - b * T;
    f32x16 acc[MR][2];
    {
        LoopGeom g;
        g.wp = d.wp; g.xp = d.xp; g.M = M; g.Cin = Cin; g.ks = d.ks > 0 ? d.ks : 1; g.dilation = d.dilation > 0 ? d.dilation : 1; g.NB = a.NB;
        g.xKC = d.xp_KC > 0 ? d.xp_KC : Cin / 8; g.xkc0 = d.xp_kc0; g.dir = d.dir < 0 ? -1 : 1; g.T = T;
        g.m_row0 = mt * HB; g.n0 = n0; g.t0 = t0;
        f16x3_mainloop<BF, MR>(acc, smem, g, wv, lane);
    }
    const bool is_skip = mt * HB < S;     // 256 | S: a block is all skip rows or all residual rows
    const bool hb = d.bias != nullptr;
    const float* bp = hb ? d.bias : reinterpret_cast<const float*>(d.wp);
    const int tcol = t0 + 64 * wv + l31;
    const float winv = inv_scales(d.w_scale_inv, d.x_scale, d.w_scale);
    const float ps = (d.plane_scale > 0.0f ? d.plane_scale : 1.0f) * dev_scale(d.out_scale);
    float gmax = 0.0f;
    bool gbad = false;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            const int m0 = mt * HB + 32 * i + 8 * v4 + 4 * lhi;      // first of this lane's four rows
            const size_t off = is_skip ? ((size_t)b * S + m0) * T + tcol : ((size_t)b * R + (m0 - S)) * T + tcol;
            const bool hin = is_skip || d.net_in != nullptr;
            const float* pin = is_skip ? d.skip + off : (d.net_in ? d.net_in + off : d.net_out + off);
            float* pout = is_skip ? d.skip + off : d.net_out + off;
            float bq[4], old[2][4], nq[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bv = bp[hb ? m0 + e : 0];
                bq[e] = hb ? bv : 0.0f;
#pragma unroll
                for (int j = 0; j < 2; ++j) { const float ov = pin[e * T + 32 * j]; old[j][e] = hin ? ov : 0.0f; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    nq[j][e] = old[j][e] + (acc[i][j][v4 * 4 + e] * winv + bq[e]);
                    pout[e * T + 32 * j] = nq[j][e];
                }
            if (!is_skip && d.net_out_planes) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gmax = fmaxf(gmax, fabsf(nq[j][e])); gbad |= !(fabsf(nq[j][e]) <= 3.0e38f); }
                    store_plane_quad<BF>(d.net_out_planes, d.planes_KC > 0 ? d.planes_KC : R / 8, a.NB, d.planes_kc0 + (m0 - S) / 8,
                                     n0 + 64 * wv + 32 * j + l31, lhi, nq[j], ps);
                }
            }
        }
    if (!is_skip && d.net_out_planes) guard_report(gmax, gbad, ps, d.out_amax, d.flag);
}

// The 1x1 convs around the residual stack (wavenet.py:53-54, 72, 80-96) and their input gradients, epi = 2:
//   out[b][m][t] = mask * (net_in[b][m][t] + (W x)[m] + bias[m] + cond[b][m][t / ratio]),   mask = (aux0[b][m][t] > 0) or 1
// (net_in, bias, cond, aux0 optional; out may be net_in and / or aux0: every element is read and written by one lane), and
// out -- or relu(out), flags bit 0 -- once more as planes for the next contraction, range-checked like the others.
template <int MR, bool BF = false>
__global__ __launch_bounds__(256, MR == 8 ? 1 : 2) void head_f16x3_kernel(const OutArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const vqw_f16x3_out_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int M = d.R, T = d.T;
    constexpr int HB = 32 * MR;
    const int n_mt = M / HB;
    const int bid = vqw_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid % n_mt, n0 = (bid / n_mt) * 256;
    const int b = n0 / T, t0 = n0 - b * T;
    f32x16 acc[MR][2];
    {
        LoopGeom g;
        g.wp = d.wp; g.xp = d.xp; g.M = M; g.Cin = d.Cin; g.ks = 1; g.dilation = 1; g.NB = a.NB;
        g.xKC = d.xp_KC > 0 ? d.xp_KC : d.Cin / 8; g.xkc0 = d.xp_kc0; g.dir = 1; g.T = T;
        g.m_row0 = mt * HB; g.n0 = n0; g.t0 = t0;
        f16x3_mainloop<BF, MR>(acc, smem, g, wv, lane);
    }
    const bool hb = d.bias != nullptr, hc = d.cond != nullptr, hin = d.net_in != nullptr, hm = d.aux0 != nullptr;
    const bool relu_planes = (d.flags & 1) != 0;
    const float* bp = hb ? d.bias : reinterpret_cast<const float*>(d.wp);      // absent operands read a valid dummy address
    const float* cb = hc ? d.cond + (size_t)b * d.cond_bstride : reinterpret_cast<const float*>(d.wp);
    const int ratio = hc ? T / d.cond_T : 1;
    const int tz[2] = {(t0 + 64 * wv) / ratio, (t0 + 64 * wv + 32) / ratio};   // 32 | ratio: one frame per tile row
    const int tcol = t0 + 64 * wv + l31;
    const float winv = inv_scales(d.w_scale_inv, d.x_scale, d.w_scale);
    const float ps = (d.plane_scale > 0.0f ? d.plane_scale : 1.0f) * dev_scale(d.out_scale);
    const int PKC = d.planes_KC > 0 ? d.planes_KC : M / 8;
    float gmax = 0.0f;
    bool gbad = false;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            const int m0 = mt * HB + 32 * i + 8 * v4 + 4 * lhi;      // first of this lane's four rows
            const size_t off = ((size_t)b * M + m0) * T + tcol;
            const float* pin = hin ? d.net_in + off : d.net_out + off;
            const float* pm = hm ? d.aux0 + off : d.net_out + off;
            float* pout = d.net_out + off;
            float add[4][2], old[2][4], mk[2][4], nq[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bv = bp[hb ? m0 + e : 0];
                const int cr = (m0 + e) * d.cond_T;
                const float c0 = cb[hc ? cr + tz[0] : 0], c1 = cb[hc ? cr + tz[1] : 0];
                add[e][0] = (hb ? bv : 0.0f) + (hc ? c0 : 0.0f);
                add[e][1] = (hb ? bv : 0.0f) + (hc ? c1 : 0.0f);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float ov = pin[e * T + 32 * j], mv = pm[e * T + 32 * j];
                    old[j][e] = hin ? ov : 0.0f;
                    mk[j][e] = (!hm || mv > 0.0f) ? 1.0f : 0.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    nq[j][e] = mk[j][e] * (old[j][e] + (acc[i][j][v4 * 4 + e] * winv + add[e][j]));
                    pout[e * T + 32 * j] = nq[j][e];
                }
            if (d.net_out_planes) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gbad |= !(fabsf(nq[j][e]) <= 3.0e38f);
                        if (relu_planes) nq[j][e] = fmaxf(nq[j][e], 0.0f);
                        gmax = fmaxf(gmax, fabsf(nq[j][e]));
                    }
                    store_plane_quad<BF>(d.net_out_planes, PKC, a.NB, d.planes_kc0 + m0 / 8, n0 + 64 * wv + 32 * j + l31, lhi, nq[j], ps);
                }
            }
        }
    if (d.net_out_planes) guard_report(gmax, gbad, ps, d.out_amax, d.flag);
}

// Gate backward (the transpose of gated_cnn's tanh * sigmoid, wavenet_ops.py:112-113, behind the transposed 1x1 convs):
// dg = W_out^T [dskip; dnet] over the gradient planes (dskip in chunks 0..S/8-1, dnet behind it), then
// dpre[filter c] = dg * sg * (1 - th^2), dpre[gate c] = dg * th * sg * (1 - sg); dpre also as planes for the input gradient.
template <bool BF, int MR, int FG = 0>      // FG 1: aux0 holds tanh * sigmoid instead of tanh; 2: ... as the gated PLANES of the forward pass
__global__ __launch_bounds__(256, MR == 8 ? 1 : 2) void gate_bwd_f16x3_kernel(const OutArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const vqw_f16x3_out_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int R = d.R, T = d.T;
    constexpr int HB = 32 * MR;
    const int n_mt = R / HB;
    const int bid = vqw_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid % n_mt, n0 = (bid / n_mt) * 256;
    const int b = n0 / T, t0 = n0 - b * T;
    f32x16 acc[MR][2];
    {
        LoopGeom g;
        g.wp = d.wp; g.xp = d.xp; g.M = R; g.Cin = d.Cin; g.ks = 1; g.dilation = 1; g.NB = a.NB;
        g.m_row0 = mt * HB; g.n0 = n0; g.t0 = t0;
        g.xKC = d.xp_KC > 0 ? d.xp_KC : d.Cin / 8; g.xkc0 = d.xp_kc0; g.dir = 1; g.T = T;
        f16x3_mainloop<BF, MR>(acc, smem, g, wv, lane);
    }
    const int tcol = t0 + 64 * wv + l31;
    const float ps = (d.plane_scale > 0.0f ? d.plane_scale : 1.0f) * dev_scale(d.out_scale);
    const float winv = inv_scales(d.w_scale_inv, d.x_scale, d.w_scale);
    float gmax = 0.0f;
    bool gbad = false;
    const int PKC = d.planes_KC > 0 ? d.planes_KC : 2 * R / 8;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) {
            const int c0 = mt * HB + 32 * i + 8 * v4 + 4 * lhi;      // first of this lane's four gated channels
            const size_t offa = ((size_t)b * R + c0) * T + tcol;      // saved tanh / sigmoid [B][R][T]
            const size_t offo = ((size_t)b * 2 * R + c0) * T + tcol;  // dpre [B][2R][T]: filter rows, then gate rows
            const float* pt = d.aux0 + offa;
            const float* pg = d.aux1 + offa;
            float* pf = d.net_out + offo;
            float* pq = pf + (size_t)R * T;
            float th[2][4], sg[2][4], qf[2][4], qg[2][4];
            if constexpr (FG == 2) {
                // this lane's four channels of one (batch, time) row = 8 bytes of the row's plane entry (store_plane_quad's layout)
                const int AKC = d.aux0_KC > 0 ? d.aux0_KC : R / 8;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const char* pe = reinterpret_cast<const char*>(d.aux0) + ((size_t)(d.aux0_kc0 + c0 / 8) * a.NB + n0 + 64 * wv + 32 * j + l31) * 16 + lhi * 8;
                    const uint2 h1 = *reinterpret_cast<const uint2*>(pe);
                    // (no local arrays here: indexed ones kept the 256-row instantiation from unrolling and put the accumulators in scratch)
                    if constexpr (BF) {
                        th[j][0] = __uint_as_float(h1.x << 16); th[j][1] = __uint_as_float(h1.x & 0xffff0000u);
                        th[j][2] = __uint_as_float(h1.y << 16); th[j][3] = __uint_as_float(h1.y & 0xffff0000u);
                    } else {
                        const uint2 h2 = *reinterpret_cast<const uint2*>(pe + (size_t)AKC * a.NB * 16);
                        auto lo16 = [](unsigned w) { return (float)__builtin_bit_cast(_Float16, (u16)(w & 0xffffu)); };
                        auto hi16 = [](unsigned w) { return (float)__builtin_bit_cast(_Float16, (u16)(w >> 16)); };
                        th[j][0] = lo16(h1.x) + lo16(h2.x); th[j][1] = hi16(h1.x) + hi16(h2.x);
                        th[j][2] = lo16(h1.y) + lo16(h2.y); th[j][3] = hi16(h1.y) + hi16(h2.y);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 2; ++j) sg[j][e] = pg[e * T + 32 * j];
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) { th[j][e] = pt[e * T + 32 * j]; sg[j][e] = pg[e * T + 32 * j]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float dg = acc[i][j][v4 * 4 + e] * winv;
                    if constexpr (FG != 0) {
                        // th[][] holds g = tanh * sigmoid (the forward pass did not store tanh):
                        //   sg (1 - tanh^2) = sg - g^2 / sg  (0 where the sigmoid underflowed),   tanh sg (1 - sg) = g (1 - sg)
                        const float g_ = th[j][e], sv = sg[j][e];
                        // (a DENORMAL sigmoid -- gate pre-activations near -88 -- has an infinite v_rcp_f32 while g^2 underflows to
                        // zero: 0 * inf = NaN; below the smallest normal number the factor is taken as 0, like an underflowed sigmoid)
                        const float ff = sv >= 1.17549435e-38f ? sv - g_ * g_ * __builtin_amdgcn_rcpf(sv) : 0.0f;
                        qf[j][e] = dg * ff;
                        qg[j][e] = dg * g_ * (1.0f - sv);
                    } else {
                        qf[j][e] = dg * sg[j][e] * (1.0f - th[j][e] * th[j][e]);
                        qg[j][e] = dg * th[j][e] * sg[j][e] * (1.0f - sg[j][e]);
                    }
                    // (a NaN / inf in dg or in the saved activations shows in qf / qg: the check below sees it)
                    if (d.net_out) {       // fp32 dpre only where somebody reads it (the batched weight gradients read the planes)
                        pf[e * T + 32 * j] = qf[j][e];
                        pq[e * T + 32 * j] = qg[j][e];
                    }
                }
            if (d.net_out_planes) {
                // The max-abs / finiteness folds sit HERE, next to the plane stores and under the same condition, and are pinned by
                // an empty asm: written in the arithmetic loop above, the optimiser sank them into the conditional guard_report at the
                // end of the kernel and kept every qf / qg of the whole epilogue alive until then -- 150 (128-row blocks) to 290
                // (256-row) spilled registers, 600-1200 bytes of scratch per lane (round 2's "spill-ridden epilogue").
                float gm = 0.0f;
                bool bad = false;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        gm = fmaxf(gm, fmaxf(fabsf(qf[j][e]), fabsf(qg[j][e])));
                        bad |= !(fabsf(qf[j][e]) <= 3.0e38f) || !(fabsf(qg[j][e]) <= 3.0e38f);
                    }
                    store_plane_quad<BF>(d.net_out_planes, PKC, a.NB, d.planes_kc0 + c0 / 8, n0 + 64 * wv + 32 * j + l31, lhi, qf[j], ps);
                    store_plane_quad<BF>(d.net_out_planes, PKC, a.NB, d.planes_kc0 + (R + c0) / 8, n0 + 64 * wv + 32 * j + l31, lhi, qg[j], ps);
                }
                gmax = fmaxf(gmax, gm);
                int gb_i = (int)gbad | (int)bad;
                asm volatile("" : "+v"(gmax), "+v"(gb_i));
                gbad = gb_i != 0;
            }
        }
    if (d.net_out_planes) guard_report(gmax, gbad, ps, d.out_amax, d.flag);
}


// The encoder's stride-2 convs (encoder.py:17-18: conv k=5 stride 2 SAME -> relu -> BatchNorm affine) and their input
// gradients on the same main loop (LoopGeom TAB).  128-row blocks (two per CU), 256 flat (batch, time) columns per block;
// a column tile may straddle two batch rows (T = 1664 is not a multiple of 256): every lane carries its own (batch, time).
//   forward:  out[b][m][t] = bn_scale[m] * relu(sum_j sum_c W[j][c][m] x[b][c][2t + j - pad] + bias[m]) + bn_shift[m],
//             x as space-to-depth planes (split_act_kernel, s2d); relu output optionally saved for the backward pass
//   dgrad:    dx[b][m][2u + r] = sum_{j = r + pad (mod 2)} sum_o Wt[j][o][m] dy[b][o][u + (r + pad - j) / 2], both parities
//             r by the same block one after the other (their stores interleave in the same lines)
struct SconvArgs {
    vqw_f16x3_sconv_desc d;
    int NB;
    int S;           // SPLIT: K splits per tile (1: only the parities of an input gradient are spread over blocks)
};

// Split-K of a conv tile (SPLIT): the S blocks of a tile each run 1/S of the K steps and put their partial accumulators into the slab
// (in register order: 16 bytes per lane, coalesced); the block that arrives LAST at the tile's ticket counter adds the S partial tiles
// up in split order -- its own included, so the sum does not depend on who came last: results stay bitwise reproducible -- and runs
// the ordinary epilogue.  The counter is back at zero when the launch ends.  Returns false for the blocks that are done.
#ifndef VQW_SPLIT_SC1
#define VQW_SPLIT_SC1 1
#endif
#ifndef VQW_SCONV_SPLIT_MAX
#define VQW_SCONV_SPLIT_MAX 8
#endif
template <int MR>
__device__ __forceinline__ bool sconv_split_fixup(f32x16 (&acc)[MR][2], float* slab, int* counter, int S, int ksp, int tid) {
    __shared__ int s_last;
    constexpr int TILE = MR * 32 * 256;
    // The partial tiles cross XCDs (one L2 each): they are written through and read past the L2 (sc1, as agent-scope atomics are)
    // instead of fencing -- a release / acquire fence pair writes back and invalidates the whole L2 of the XCD under the other
    // blocks' main loops (measured: 12 splits of 20 K steps took longer than one block of 240).
    const __amdgpu_buffer_rsrc_t rs = vqw_make_rsrc(slab, (unsigned)((size_t)S * TILE * 4));
    constexpr int SC1 = VQW_SPLIT_SC1 ? 16 : 0;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const f32x4 v = f32x4{acc[i][j][v4 * 4], acc[i][j][v4 * 4 + 1], acc[i][j][v4 * 4 + 2], acc[i][j][v4 * 4 + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (ksp * TILE + ((i * 2 + j) * 4 + v4) * 1024 + tid * 4) * 4, 0, SC1);
            }
    if (VQW_SPLIT_SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this wave has reached memory
    else __threadfence();
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == S - 1;
        if (old == S - 1) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return false;
    if (!VQW_SPLIT_SC1) __threadfence();
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // split order 0 .. S-1 whoever came last; the loads of the next split are in flight while one is added
    constexpr int CH = MR * 8 < 32 ? MR * 8 : 32, NH = MR * 8 / CH;      // 16-byte entries per fetch, fetches per partial tile
    f32x4 nx[CH];
    auto fetch = [&](int k, int h) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
            nx[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (k * TILE + (h * CH + q) * 1024 + tid * 4) * 4, 0, SC1));
    };
    fetch(0, 0);
    for (int k = 0; k < S; ++k) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f32x4 cur[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) cur[q] = nx[q];
            if (h + 1 < NH) fetch(k, h + 1);
            else if (k + 1 < S) fetch(k + 1, 0);
#pragma unroll
            for (int q = 0; q < CH; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qq = h * CH + q;
                    acc[qq >> 3][(qq >> 2) & 1][(qq & 3) * 4 + e] += cur[q][e];
                }
        }
    }
    return true;
}

template <int MR, int DEPTH, bool DGRAD, bool SPLIT = false>
__global__ __launch_bounds__(256, (MR == 4 && DEPTH == 1) ? 2 : 1) void sconv_f16x3_kernel(const SconvArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int HB = 32 * MR;
    const vqw_f16x3_sconv_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int T = d.T, M = d.M;
    const int n_mt = M / HB;
    int bid = vqw_xcd_remap(blockIdx.x, gridDim.x);
    // SPLIT: block -> (K split, [parity of an input gradient,] tile), splits fastest
    int ksp = 0, par = 0, tile_id = 0;
    if constexpr (SPLIT) {
        ksp = bid % a.S; bid /= a.S;
        tile_id = bid;
        if (DGRAD) { par = bid & 1; bid >>= 1; }
    }
    const int mt = bid % n_mt, n0 = (bid / n_mt) * 256;
    const float winv = inv_scales(d.w_scale_inv, d.x_scale, d.w_scale);
    int bcol[2], tcol[2];
    bool cv[2];                      // the last column tile may reach past B * T (its operand rows read garbage or zero, nothing is stored)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 64 * wv + 32 * j + l31;
        cv[j] = n < a.NB;
        bcol[j] = n / T;
        tcol[j] = n - bcol[j] * T;
    }
    f32x16 acc[MR][2];
    LoopGeom g;
    g.wp = d.wp; g.xp = d.xp; g.M = M; g.Cin = d.Cin; g.dilation = 1; g.NB = a.NB; g.xkc0 = 0; g.dir = 1; g.T = T;
    g.m_row0 = mt * HB; g.n0 = n0; g.t0 = 0; g.wks = d.ks;
    if constexpr (!DGRAD) {
        g.ks = d.ks; g.xKC = 2 * d.Cin / 8; g.tj0 = 0; g.tjstep = 1; g.toff = d.pad_left; g.tsgn = -1; g.ts2d = 1;
        g.sn = g.ks * (d.Cin / 16) / (SPLIT ? a.S : 1); g.sb = ksp * g.sn;
        f16x3_mainloop<false, MR, true, DEPTH>(acc, smem, g, wv, lane);
        if constexpr (SPLIT)
            if (a.S > 1 && !sconv_split_fixup<MR>(acc, d.split_slab + (size_t)tile_id * a.S * (HB * 256), d.split_counters + tile_id, a.S, ksp, tid)) return;
        const bool hb = d.bias != nullptr, hs = d.bn_scale != nullptr, sv = d.save_r != nullptr;
        const float* bp = hb ? d.bias : reinterpret_cast<const float*>(d.wp);
        const float* sp = hs ? d.bn_scale : reinterpret_cast<const float*>(d.wp);
        const float* hp = hs ? d.bn_shift : reinterpret_cast<const float*>(d.wp);
        // one 64-bit base per column, 32-bit row offsets inside (out and save_r are < 2 GiB each); the per-row constants are
        // fetched group by group (64-bit lane addresses per element and constants fetched up front spilled 570 registers)
        float* po[2];
        float* pr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t cb = (size_t)bcol[j] * M * T + tcol[j];
            po[j] = d.out + cb;
            pr[j] = sv ? d.save_r + cb : d.out + cb;
        }
        const float rlo = d.relu ? 0.0f : -INFINITY;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                const int m0 = mt * HB + 32 * i + 8 * v4 + 4 * lhi;      // first of this lane's four rows
                const f32x4 bq = hb ? *reinterpret_cast<const f32x4*>(bp + m0) : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 sq = hs ? *reinterpret_cast<const f32x4*>(sp + m0) : f32x4{1.f, 1.f, 1.f, 1.f};
                const f32x4 hq = hs ? *reinterpret_cast<const f32x4*>(hp + m0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ro = (m0 + e) * T;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float y = fmaxf(acc[i][j][v4 * 4 + e] * winv + bq[e], rlo);
                        if (sv && cv[j]) pr[j][ro] = y;
                        if (cv[j]) po[j][ro] = sq[e] * y + hq[e];
                    }
                }
            }
    } else {
        g.xKC = d.Cin / 8; g.tjstep = 2; g.tsgn = 1; g.ts2d = 0;
#pragma unroll 1
        for (int r = SPLIT ? par : 0; r < (SPLIT ? par + 1 : 2); ++r) {
            g.tj0 = (r + d.pad_left) & 1; g.toff = r + d.pad_left;
            g.ks = (d.ks - g.tj0 + 1) / 2;                       // taps of this parity (>= 1 for ks >= 2)
            g.sn = g.ks * (d.Cin / 16) / (SPLIT ? a.S : 1); g.sb = ksp * g.sn;
            if (!SPLIT && r) __syncthreads();                    // the first run's last stage is still being read
            f16x3_mainloop<false, MR, true, DEPTH>(acc, smem, g, wv, lane);
            if constexpr (SPLIT)
                if (a.S > 1 && !sconv_split_fixup<MR>(acc, d.split_slab + (size_t)tile_id * a.S * (HB * 256), d.split_counters + tile_id, a.S, ksp, tid)) return;
            float* pd[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) pd[j] = d.out + (size_t)bcol[j] * M * (2 * T) + 2 * tcol[j] + r;
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    const int m0 = mt * HB + 32 * i + 8 * v4 + 4 * lhi;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if (cv[j]) pd[j][(m0 + e) * (2 * T)] = acc[i][j][v4 * 4 + e] * winv;
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Weight gradients on the fp16 matrix pipe:  dW[j][c][o] += sum_{b,t} p[b][c][t + shift_j] * q[b][o][t]
// (TF Conv2DBackpropFilter of conv1d_v2, wavenet_ops.py:83-86; q = [q0; q1] along o).
//
// The contraction index is TIME, which is the contiguous index of both fp32 operands [B][C][T]: an MFMA fragment entry
// (row, 8 consecutive k) is 32 contiguous bytes of global memory.  The operands are therefore read as fp32 (whole
// 128-byte lines per row and 32-step stage pair), scaled by their power-of-two guard scale, split into the two fp16
// planes IN REGISTERS (same bytes as reading ready-made planes, no conversion pass, no plane copy in HBM) and written
// to the same LDS stage image as the kernels above, so the fragment reads and the 3-term MFMA loop are shared.
// Output tile 256 (c) x 256 (o) per block; the K range B*T is cut into `nsplit` contiguous chunks per tile so that
// tiles * nsplit fills the chip once; partial tiles go to a slab [tile][split][256][256] and a second kernel adds them
// up in a fixed order (no atomics: dW is bitwise reproducible).  Needs T % 32 == 0, C % 256 == 0, shifts <= 0.
// One launch carries up to WG_MAX_BATCH independent problems of the same shape (the weight gradients of several layers):
// every problem has its own operands, scales, sums and output, the tiles of all problems share the K split -- the slab
// holds tiles x splits <= CUs partial tiles whatever the batch size, so a batch of n layers writes and re-reads 1/n of
// the slab bytes per layer that n single launches would.
constexpr int WG_MAX_BATCH = 32;
struct WgProblem {
    const float* p;
    const float* q0;
    const float* q1;
    float* dw;
    const float* sp;      // device scalars (or NULL = 1): power-of-two scales of p, q0, q1
    const float* sq0;
    const float* sq1;
    // sums of q over time, formed from the registers that hold q anyway (blocks of the problem's first row tile and tap only):
    float* q_total;       // [Q0 + Q1] += sum_{b,t} q[b][o][t]  (bias gradients) or NULL
    float* q_seg;         // [B][seg_bstride/..]: q_seg[b * seg_bstride + o * seg_T + t / seg_ratio] += q[b][o][t] (the transpose of
                          //   add_condition's upsampling, wavenet_ops.py:98-100) or NULL; seg_ratio % 32 == 0
    const void* qp;       // QP kernels: q as operand planes [planes][q_KC chunks][B*T rows][8] (scaled by *sq0 * q_hscale), chunks from q_kc0
    int q_KC, q_kc0;
    float q_hscale;       // host-side part of the planes' scale (a power of two; 1 where the scale lives on the device)
    const void* pp;       // PP kernels: p as operand planes, likewise
    int p_KC, p_kc0;
    float p_hscale;
    int pkoff[VQW_MAX_TAPS];   // PP: chunk offset of tap j's rows (space-to-depth planes of a stride-2 conv's input: the parity block)
    int shift[VQW_MAX_TAPS];
};
struct WgArgs {
    WgProblem pr[WG_MAX_BATCH];
    float* slab;
    int nprob;
    int B, T, Cp, Q0, Q1, ntaps;
    int Tp;               // row length of p (= T, or the input length of a stride-2 conv: p index 2 t + shift)
    int p_relu;           // p := max(p, 0) on the way in (the convs behind a relu, wavenet.py:79, 93)
    int nsplit, pairs_row, pairs_total, n_nt;
    long seg_bstride;
    int seg_T, seg_ratio;
    int total_o0, total_o1;   // q_total covers columns [total_o0, total_o1) only (e.g. the residual rows S..S+R)
    long lddw, dw_tap_stride; // (the reduction)
};
// Work item w (after the XCD remap: consecutive w on one XCD) -> (K split, problem, row tile, column tile, tap), taps fastest:
// the taps of one tile and K range read the same q panel and p panels that are the same cache lines a few elements apart, the
// column tiles of a row tile share its p panel, and everything of one K range sits on as few XCDs (L2s) as possible.
struct WgItem {
    int prob, tap, rtl, nt, split, gtile;      // rtl: 256-row tile within the problem; gtile: slab / reduction index
};
__device__ __forceinline__ WgItem wg_decode(int w, const WgArgs& a) {
    const int cpt = a.Cp / 256;
    WgItem it;
    it.tap = w % a.ntaps; w /= a.ntaps;
    it.nt = w % a.n_nt; w /= a.n_nt;
    const int rows_all = a.nprob * cpt, rt = w % rows_all;
    it.split = w / rows_all;
    it.prob = rt / cpt;
    it.rtl = rt - it.prob * cpt;
    it.gtile = ((it.prob * a.ntaps + it.tap) * cpt + it.rtl) * a.n_nt + it.nt;
    // The divisions run on the vector ALU, so the compiler takes everything derived from them for lane-dependent: the problem's
    // pointers were fetched per lane, the buffer resources built from them lived in VGPRs and every operand request of the main
    // loop became a readfirstlane "waterfall" loop of its own (26 of them in the planes variant, each a scheduling barrier
    // between the MFMAs).  One readfirstlane per field states what is true anyway.
    it.prob = __builtin_amdgcn_readfirstlane(it.prob); it.tap = __builtin_amdgcn_readfirstlane(it.tap);
    it.rtl = __builtin_amdgcn_readfirstlane(it.rtl); it.nt = __builtin_amdgcn_readfirstlane(it.nt);
    it.split = __builtin_amdgcn_readfirstlane(it.split); it.gtile = __builtin_amdgcn_readfirstlane(it.gtile);
    return it;
}

template <bool BF>
__device__ __forceinline__ uint2 split4(const f32x4 v, float sc, uint2& lo) {
    u16 a[4], b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xs = v[e] * sc;
        if (BF) { a[e] = bf16_bits(xs); b[e] = 0; continue; }
        const _Float16 h1 = (_Float16)xs;
        a[e] = f16_bits(h1);
        b[e] = f16_bits((_Float16)(xs - (float)h1));
    }
    lo = make_uint2(b[0] | ((unsigned)b[1] << 16), b[2] | ((unsigned)b[3] << 16));
    return make_uint2(a[0] | ((unsigned)a[1] << 16), a[2] | ((unsigned)a[3] << 16));
}

// ODD: some tap shift is not a multiple of 4 (compile-time: a branch would cut the loop body into scheduling regions)
// S2: p is the input of a stride-2 convolution (encoder.py:17-18): dW[j][c][o] += sum p[b][c][2 t + shift_j] q[b][o][t], shifts of
// either sign, indices outside [0, Tp) are the conv's zero padding.  p is then fetched one float per request (stride 8 bytes
// between the four of a chunk), each one range-checked on its own through the buffer descriptor.
// QP: q is not read as fp32 but as the OPERAND PLANES its producer wrote anyway (gate backward writes dpre as planes for the input
// gradient; with this variant it no longer writes the 109 MB of fp32 dpre per layer at all).  Planes are channel-chunk-major --
// 16 bytes = 8 channels of one time step -- while this contraction runs over TIME: the 16-byte entries go to LDS as they are, as a
// [time step][channel] image per plane (rows of 512 + 64 bytes: the four rows of a transposed read fall into four different
// 16-bank groups), and the B fragments are fetched with ds_read_b64_tr_b16, which hands lane (channel n, k half) its 4 + 4
// consecutive time steps.  No conversion arithmetic for q (two thirds of the conversions of the gate kernels' gradient).
constexpr int WG_QSTR = 576, WG_QPL = 16 * WG_QSTR;                       // bytes per time step / per plane of a stage's q image
// PP: the same for p (the layer input planes the forward pass wrote, the gated planes): a tap's shift is then a row offset of
// the planes -- no unaligned 16-byte windows, no ODD variant -- and the loop converts nothing at all.
template <bool QP, bool PP, bool BF> struct WgStage {
    static constexpr int BOFF = PP ? (BF ? 1 : 2) * WG_QPL : 16 * 1024;            // q image behind the p image
    static constexpr int BYTES = BOFF + (QP ? (BF ? 1 : 2) * WG_QPL : 16 * 1024);
};
__device__ __forceinline__ uint2 wg_tr_read(const char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    const h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(p));
    return __builtin_bit_cast(uint2, v);
#else
    return make_uint2(0, 0);
#endif
}

template <bool ODD, bool BF, bool S2 = false, bool QP = false, bool PP = false>
__global__ __launch_bounds__(256, 1) void wgrad_f16x3_kernel(const WgArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int STGB = WgStage<QP, PP, BF>::BYTES, BOFF = WgStage<QP, PP, BF>::BOFF, NPL = BF ? 1 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const WgItem it = wg_decode(vqw_xcd_remap(blockIdx.x, gridDim.x), a);
    const WgProblem& pr = a.pr[it.prob];
    const int split = it.split, tap = it.tap, c0 = it.rtl * 256;
    const int shift = pr.shift[tap];
    const int o0 = it.nt * 256;
    const bool from_q1 = o0 >= a.Q0;
    const float* qsrc = from_q1 ? pr.q1 : pr.q0;
    const int Qs = from_q1 ? a.Q1 : a.Q0, oq = from_q1 ? o0 - a.Q0 : o0;
    const float scp = dev_scale(pr.sp) * (PP ? pr.p_hscale : 1.0f), scq = dev_scale(from_q1 ? pr.sq1 : pr.sq0) * (QP ? pr.q_hscale : 1.0f);
    const float plo = a.p_relu ? 0.0f : -INFINITY;
    const int T = a.T;
    const int s_begin = (int)((long)split * a.pairs_total / a.nsplit), s_end = (int)((long)(split + 1) * a.pairs_total / a.nsplit);
    const __amdgpu_buffer_rsrc_t rp = vqw_make_rsrc(PP ? reinterpret_cast<const float*>(pr.pp) : pr.p, (unsigned)((size_t)a.B * a.Cp * (S2 ? a.Tp : T) * 4));
    const __amdgpu_buffer_rsrc_t rq = vqw_make_rsrc(QP ? reinterpret_cast<const float*>(pr.qp) : qsrc, (unsigned)((size_t)a.B * Qs * T * 4));
    // QP: one resource per plane, based at this block's first chunk (32 chunks = its 256 q rows)
    const size_t NBq = (size_t)a.B * T;
    const char* qpb = reinterpret_cast<const char*>(pr.qp) + (QP ? ((size_t)pr.q_kc0 + o0 / 8) * NBq * 16 : 0);
    const unsigned qspan = (unsigned)(32 * NBq * 16 < 0x7fffffffull ? 32 * NBq * 16 : 0x7fffffffull);
    const __amdgpu_buffer_rsrc_t rqp0 = vqw_make_rsrc(qpb, qspan);
    const __amdgpu_buffer_rsrc_t rqp1 = vqw_make_rsrc(qpb + (QP && !BF ? (size_t)pr.q_KC * NBq * 16 : 0), qspan);
    const char* ppb = reinterpret_cast<const char*>(pr.pp) + (PP ? ((size_t)pr.p_kc0 + pr.pkoff[tap] + c0 / 8) * NBq * 16 : 0);
    const __amdgpu_buffer_rsrc_t rpp0 = vqw_make_rsrc(ppb, qspan);
    const __amdgpu_buffer_rsrc_t rpp1 = vqw_make_rsrc(ppb + (PP && !BF ? (size_t)pr.p_KC * NBq * 16 : 0), qspan);
    float* const q_total = pr.q_total;
    float* const q_seg = pr.q_seg;

    // One load instruction of a wave = 32 rows x 32 bytes (lane -> row lane/2, 16-byte half lane%2); chunk n of a
    // thread: 32-row group g = wv*2 + n/4, quarter nn = n%4 of the 128-byte line = stage nn/2 of the pair, k half nn%2.
    // The four quarters of a line are requested back to back, so the line is fetched once.
    f32x4 rgp[8], rgq[8];
    const int rsub = lane >> 1, hsel = lane & 1;
    int pb = 0, pt0 = 0;                                // batch row / first time step of the pair being requested
    auto issue_one = [&](int n) {
        const int g = wv * 2 + (n >> 2), nn = n & 3;
        const int row = g * 32 + rsub;
        const int tq = pt0 + 8 * nn + 4 * hsel;
        // (S2: T need not be a multiple of the 32-step stage pairs -- steps behind the row's end read as zero)
        if constexpr (QP) {
            // item n of this thread: plane n / 4, chunks (n % 4) * 8 + lane % 8, time step 8 wv + lane / 8 of the pair's 32 --
            // eight lanes = eight chunks of one step (128 contiguous bytes of LDS), eight lane groups = the eight steps of one
            // 128-byte line per chunk
            if (n < 4 * NPL) {
                const int chunk = (n & 3) * 8 + (lane & 7), t = 8 * wv + (lane >> 3);
                // (T need not be a multiple of the 32-step pairs when both operands are planes: steps behind the row's end read as zero)
                const int off = (pt0 + t < T) ? (int)(((size_t)chunk * NBq + (size_t)pb * T + pt0 + t) * 16) : (int)0x80000000;
                rgq[n] = vqw_buf_load4((n >> 2) ? rqp1 : rqp0, off, 0);
            }
        } else {
            rgq[n] = vqw_buf_load4(rq, (!S2 || tq < T) ? (int)((((size_t)pb * Qs + oq + row) * T + tq) * 4) : (int)0x80000000, 0);
        }
        if constexpr (PP) {      // the same item geometry as QP; the tap's shift is a row offset, rows before the batch row read as zero
            if (n < 4 * NPL) {
                const int chunk = (n & 3) * 8 + (lane & 7), tt = pt0 + 8 * wv + (lane >> 3) + shift;
                // (tt >= T: only with the positive row shifts of a stride-2 conv's taps -- its right zero padding)
                const int off = (tt >= 0 && tt < T) ? (int)(((size_t)chunk * NBq + (size_t)pb * T + tt) * 16) : (int)0x80000000;
                rgp[n] = vqw_buf_load4((n >> 2) ? rpp1 : rpp0, off, 0);
            }
            return;
        }
        if (S2) {
            const size_t prow = ((size_t)pb * a.Cp + c0 + row) * a.Tp;
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int idx = 2 * (tq + e) + shift;
                w[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (idx >= 0 && idx < a.Tp && tq < T) ? (int)((prow + idx) * 4) : (int)0x80000000, 0, 0));
            }
            rgp[n] = w;
            return;
        }
        const int tp = tq + shift;                      // shift <= 0: never past the row's end
        const size_t prow = ((size_t)pb * a.Cp + c0 + row) * T;
        // a 16-byte window that straddles t = 0 (only with shifts that are not multiples of 4) is read from t = 0 and moved
        // up inside the register below; windows entirely before t = 0 read as zero through the buffer range check
        const int tl = tp >= 0 ? tp : (tp > -4 ? 0 : -1);
        // (ODD: the straddling window is moved up when the registers are CONVERTED, commit_one -- doing it here, on the value just
        // requested, made every request of the unaligned variant wait for its own data: 205 instead of 150 us per gate kernel)
        rgp[n] = vqw_buf_load4(rp, tl >= 0 ? (int)((prow + tl) * 4) : (int)0x80000000, 0);
    };
    auto set_pair = [&](int s) { pb = s / a.pairs_row; pt0 = (s - pb * a.pairs_row) * 32; };
    int pt0_c = 0;                                      // first time step of the pair being converted (ODD only)
    auto set_commit_pair = [&](int s) { if (ODD) pt0_c = (s % a.pairs_row) * 32; };
    auto commit_one = [&](int n, int pair) {     // raw registers -> two fp16 planes -> LDS stage (2 pair + nn/2) % 4
        const int g = wv * 2 + (n >> 2), nn = n & 3;
        char* st = smem + ((2 * pair + (nn >> 1)) % NSTG) * STGB + ((nn & 1) * 32 + rsub) * 16 + hsel * 8;
        uint2 lo;
        uint2 hi;
        f32x4 pv = rgp[n];
        if constexpr (PP) {
            if (n < 4 * NPL) {
                const int chunk = (n & 3) * 8 + (lane & 7), t = 8 * wv + (lane >> 3);
                char* dp_ = smem + ((2 * pair + (t >> 4)) % NSTG) * STGB + (n >> 2) * WG_QPL + (t & 15) * WG_QSTR + chunk * 16;
                *reinterpret_cast<f32x4*>(dp_) = pv;
            }
        } else {
        if (ODD && !S2) {
            const int tp = pt0_c + 8 * nn + 4 * hsel + shift;
            const int k = (tp < 0 && tp > -4) ? -tp : 0;      // the window was read from t = 0: its first k elements lie before the row
            const f32x4 w = pv;
            pv[0] = k == 0 ? w[0] : 0.0f;
            pv[1] = k == 0 ? w[1] : (k == 1 ? w[0] : 0.0f);
            pv[2] = k == 0 ? w[2] : (k == 1 ? w[1] : (k == 2 ? w[0] : 0.0f));
            pv[3] = k == 0 ? w[3] : (k == 1 ? w[2] : (k == 2 ? w[1] : w[0]));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = fmaxf(pv[e], plo);
        hi = split4<BF>(pv, scp, lo);
        *reinterpret_cast<uint2*>(st + (g * 2 + 0) * 1024) = hi;
        if (!BF) *reinterpret_cast<uint2*>(st + (g * 2 + 1) * 1024) = lo;
        }
        if constexpr (QP) {
            if (n < 4 * NPL) {     // the plane entry as it is: [time step][channel] image of its plane and stage
                const int chunk = (n & 3) * 8 + (lane & 7), t = 8 * wv + (lane >> 3);
                char* dq = smem + ((2 * pair + (t >> 4)) % NSTG) * STGB + BOFF + (n >> 2) * WG_QPL + (t & 15) * WG_QSTR + chunk * 16;
                *reinterpret_cast<f32x4*>(dq) = rgq[n];
            }
        } else {
            hi = split4<BF>(rgq[n], scq, lo);
            *reinterpret_cast<uint2*>(st + BOFF + (g * 2 + 0) * 1024) = hi;
            if (!BF) *reinterpret_cast<uint2*>(st + BOFF + (g * 2 + 1) * 1024) = lo;
        }
    };
    // q sums (bias / condition gradients): the thread's two q rows (groups wv*2 and wv*2+1), its half of the 32 steps
    const bool do_sum = it.rtl == 0 && tap == 0 && (q_seg != nullptr || (q_total != nullptr && o0 < a.total_o1 && o0 + 256 > a.total_o0));
    float qs_pair[2] = {0.f, 0.f}, qs_tot[2] = {0.f, 0.f};
    auto sum_one = [&](int n) {            // called with commit_one(n): rgq[n] still holds the raw fp32 values
#ifdef VQW_ABL_WG_NOSUM
        return;
#endif
        if constexpr (QP) return;          // (QP: the sums are formed from the B fragments, sum_frag)
        const f32x4 v = rgq[n];
        qs_pair[n >> 2] += (v[0] + v[1]) + (v[2] + v[3]);
    };
    // QP: lane (channel n of column tile j, k half) holds 8 consecutive time steps of its channel per plane: their sum, over both
    // planes, descaled -- the fp32 value to 2^-22 (the planes carry 22-23 significand bits)
    const float inv_scq = __builtin_amdgcn_rcpf(scq);
    auto sum_frag = [&](const uint4 (&b)[2][2]) {
#ifdef VQW_ABL_WG_NOSUM
        return;
#endif
        if constexpr (QP) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float acc_ = 0.0f;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const unsigned wds[4] = {b[j][pl].x, b[j][pl].y, b[j][pl].z, b[j][pl].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (BF) {
                            acc_ += __uint_as_float(wds[e] << 16) + __uint_as_float(wds[e] & 0xffff0000u);
                        } else {
                            acc_ += (float)__builtin_bit_cast(_Float16, (u16)(wds[e] & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (u16)(wds[e] >> 16));
                        }
                    }
                }
                qs_pair[j] += acc_ * inv_scq;
            }
        }
    };
    auto flush_pair = [&](int s) {         // pair s is complete in qs_pair: add it to its condition frame, fold it into the totals
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the partner lane holds the other 16 steps (QP: the other k half of the fragment)
            const float both = qs_pair[h] + __shfl_xor(qs_pair[h], QP ? 32 : 1, 64);
            qs_tot[h] += qs_pair[h];
            qs_pair[h] = 0.f;
            if (q_seg && (QP ? lhi == 0 : hsel == 0)) {
                const int b = s / a.pairs_row, t0 = (s - b * a.pairs_row) * 32;
                const int row = o0 + (wv * 2 + h) * 32 + (QP ? l31 : rsub);
                unsafeAtomicAdd(q_seg + (size_t)b * a.seg_bstride + (size_t)row * a.seg_T + t0 / a.seg_ratio, both);
            }
        }
    };
    uint4 fa[8][2], fb[2][2];                        // A fragments (both planes) of 8 row tiles, B fragments of this wave's 2 column tiles
    // transposed reads (QP / PP): 16-lane group G = lane / 16 reads rows (time steps) 8 (G / 2) + 4 r + 0..3, 16 columns (channels)
    // from 16 (G % 2) of the 32-channel tile; its lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3 and receives
    // column lane % 16
    const int tr_lane = (8 * (lane >> 5) + ((lane & 15) >> 2)) * WG_QSTR + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto read_a = [&](int i, int stage) {
        if constexpr (PP) {
            const char* st = smem + (stage % NSTG) * STGB + tr_lane + i * 64;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const uint2 k0 = wg_tr_read(st + pl * WG_QPL), k1 = wg_tr_read(st + pl * WG_QPL + 4 * WG_QSTR);
                fa[i][pl] = make_uint4(k0.x, k0.y, k1.x, k1.y);
            }
        } else {
            const char* st = smem + (stage % NSTG) * STGB + lane * 16;
            fa[i][0] = *reinterpret_cast<const uint4*>(st + (i * 2 + 0) * 1024);
            if (!BF) fa[i][1] = *reinterpret_cast<const uint4*>(st + (i * 2 + 1) * 1024);
        }
    };
    auto read_b = [&](uint4 (&b)[2][2], int stage) {
        if constexpr (QP) {
            const char* st = smem + (stage % NSTG) * STGB + BOFF + tr_lane + wv * 128;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const uint2 k0 = wg_tr_read(st + j * 64 + pl * WG_QPL), k1 = wg_tr_read(st + j * 64 + pl * WG_QPL + 4 * WG_QSTR);
                    b[j][pl] = make_uint4(k0.x, k0.y, k1.x, k1.y);
                }
        } else {
            const char* st = smem + (stage % NSTG) * STGB + lane * 16 + BOFF;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) b[j][pl] = *reinterpret_cast<const uint4*>(st + ((wv * 2 + j) * 2 + pl) * 1024);
        }
    };
    f32x16 acc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // 12 MFMAs: row tiles i, i+1 x this wave's 2 column tiles, the three terms outermost (small first) so that two MFMAs
    // into the same accumulator are three independent ones apart
    auto mfma_rows = [&](int i, const uint4 (&b)[2][2]) {
        if constexpr (BF) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[i + (q >> 1)][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i + (q >> 1)][0]), __builtin_bit_cast(bf16x8, b[q & 1][0]), acc[i + (q >> 1)][q & 1], 0, 0, 0);
        } else {
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ii = i + (q >> 1), j = q & 1;
                    const int pa = term == 1 ? 1 : 0, pb = term == 0 ? 1 : 0;
                    acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ii][pa]), __builtin_bit_cast(f16x8, b[j][pb]), acc[ii][j], 0, 0, 0);
                }
        }
    };

    const int npairs = s_end - s_begin;
    if (npairs > 0) {
        set_pair(s_begin);
        set_commit_pair(s_begin);
#pragma unroll
        for (int n = 0; n < 8; ++n) issue_one(n);
#pragma unroll
        for (int n = 0; n < 8; ++n) { commit_one(n, 0); sum_one(n); }
        if (!QP && do_sum) flush_pair(s_begin);
        set_pair(s_begin + 1);
#pragma unroll
        for (int n = 0; n < 8; ++n) issue_one(n);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) read_a(i, 0);
        read_b(fb, 0);
        sum_frag(fb);
        for (int it = 0; it < npairs; ++it) {
            // Stages 2it, 2it+1 (pair `it`) are in LDS, the fragments of stage 2it in registers; the raw operands of pair
            // it+1 are in rgp / rgq (requested one iteration ago).  Row tile by row tile: 6 MFMAs of the first stage, behind
            // them the tile's fragments of the second stage are fetched and one chunk of pair it+1 is converted and written
            // to the two LDS stages pair it-1 was read from; then the second stage's MFMAs with the requests of pair it+2
            // behind them.  VALU / LDS / VMEM work is issued in the shadow of the MFMAs.
            // (No conditionals inside: past the block's last pair the conversion rewrites stale registers into LDS stages
            // nobody reads again and the requests fall behind the end of the buffers, where raw buffer loads return zero --
            // branches would cut the body into scheduling regions and the MFMAs could no longer be interleaved.)
            set_commit_pair(s_begin + it + 1);
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                mfma_rows(i, fb);
                read_a(i, 2 * it + 1);
                read_a(i + 1, 2 * it + 1);
                commit_one(i, it + 1);
                commit_one(i + 1, it + 1);
                sum_one(i);
                sum_one(i + 1);
#pragma unroll
                for (int k_ = 0; k_ < 12; ++k_) {         // 1 MFMA, then ~1/12 of the group's VALU / LDS work in its shadow
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x100 | 0x200, VQW_WG_PAT_A, 0);
                }
            }
            if (!QP && do_sum && it + 1 < npairs) flush_pair(s_begin + it + 1);
            set_pair(s_begin + it + 2);
            read_b(fb, 2 * it + 1);        // (after the first stage's last use of fb: 16 registers instead of 32)
            sum_frag(fb);
            if (QP && do_sum) flush_pair(s_begin + it);     // both stages of pair `it` have been read
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                mfma_rows(i, fb);
                issue_one(i);
                issue_one(i + 1);
#pragma unroll
                for (int k_ = 0; k_ < 12; ++k_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x020, VQW_WG_PAT_B, 0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) read_a(i, 2 * it + 2);
            read_b(fb, 2 * it + 2);
            sum_frag(fb);                  // (behind the last pair: a stale stage, never flushed)
        }
    }
    if (do_sum && q_total) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float both = qs_tot[h] + __shfl_xor(qs_tot[h], QP ? 32 : 1, 64);
            const int row = o0 + (wv * 2 + h) * 32 + (QP ? l31 : rsub);
            if ((QP ? lhi == 0 : hsel == 0) && row >= a.total_o0 && row < a.total_o1) unsafeAtomicAdd(q_total + row, both);
        }
    }
    // ---- partial tile -> slab [tile][split][256][256], rows c, columns o (32 lanes = 128 contiguous bytes)
    const float inv = __builtin_amdgcn_rcpf(scp * scq);
    float* out = a.slab + ((size_t)it.gtile * a.nsplit + split) * 65536;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    out[(32 * i + 8 * v4 + 4 * lhi + e) * 256 + 64 * wv + 32 * j + l31] = acc[i][j][v4 * 4 + e] * inv;
}

// dW[tap][c][o] += sum over splits (fixed order) of the slab tiles; grid (64, tiles of all problems)
__global__ void wgrad_reduce_kernel(const WgArgs a) {
    const int gtile = blockIdx.y, cpt = a.Cp / 256, nsplit = a.nsplit;
    int w = gtile;
    const int nt = w % a.n_nt; w /= a.n_nt;
    const int rtl = w % cpt; w /= cpt;
    const int tap = w % a.ntaps, prob = w / a.ntaps;
    const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;     // 4 consecutive columns
    if (idx >= 65536) return;
    const int r = idx >> 8, col = idx & 255;
    const float* sp = a.slab + (size_t)gtile * nsplit * 65536 + idx;
    // four running sums (splits k = 0, 1, 2, 3 mod 4), combined at the end: a FIXED order, eight loads in flight per thread
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + (size_t)(k + u) * 65536));
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
    }
    for (; k < nsplit; ++k) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + (size_t)k * 65536));
        switch (k & 3) { case 0: s0 += v; break; case 1: s1 += v; break; case 2: s2 += v; break; default: s3 += v; }
    }
    const f32x4 sum = (s0 + s1) + (s2 + s3);
    float* d = a.pr[prob].dw + (size_t)tap * a.dw_tap_stride + (size_t)(rtl * 256 + r) * a.lddw + nt * 256 + col;
    f32x4 old = *reinterpret_cast<const f32x4*>(d);
    *reinterpret_cast<f32x4*>(d) = old + sum;
}

}  // namespace

extern "C" {

int vqw_f16x3_amax(const float* x, int64_t rows, int cols, int64_t ld, int64_t mstride, int count, uint32_t* amax, int32_t* flag, vqw_stream_t s_) {
    VQW_CHECK(x && amax, "vqw_f16x3_amax: null pointer");
    VQW_CHECK(rows > 0 && cols > 0 && ld >= cols && count >= 1 && count <= 65535, "vqw_f16x3_amax: bad shape (rows=%lld cols=%d ld=%lld count=%d)", (long long)rows, cols, (long long)ld, count);
    const size_t n = (size_t)rows * cols;
    unsigned g = (unsigned)((n + 256 * 32 - 1) / (256 * 32));
    if (g > 512) g = 512;
    hipLaunchKernelGGL(amax_kernel, dim3(g, count), dim3(256), 0, (hipStream_t)s_, x, (long)rows, cols, (long)ld, (long)mstride, amax, flag);
    VQW_LAUNCH_CHECK("vqw_f16x3_amax");
    return 0;
}

int vqw_f16x3_update_scales_guarded(uint32_t* amax, float* scale, int n, int target_exp, int reset, int32_t* flag, const int32_t* skip,
                                    vqw_stream_t s_) {
    VQW_CHECK(amax && scale && n > 0, "vqw_f16x3_update_scales: null pointer");
    VQW_CHECK(target_exp >= 1 && target_exp <= 15, "vqw_f16x3_update_scales: target_exp must be in 1..15 (fp16 holds |x| < 2^16)");
    hipLaunchKernelGGL(update_scales_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)s_, amax, scale, n, target_exp, reset, flag, skip);
    VQW_LAUNCH_CHECK("vqw_f16x3_update_scales");
    return 0;
}

int vqw_f16x3_update_scales(uint32_t* amax, float* scale, int n, int target_exp, int reset, int32_t* flag, vqw_stream_t s_) {
    return vqw_f16x3_update_scales_guarded(amax, scale, n, target_exp, reset, flag, nullptr, s_);
}

int vqw_f16x3_split_activations(const float* x, void* planes, int B, int C, int T, float scale, int kc0, int KC,
                                const float* scale_dev, uint32_t* amax, int32_t* flag, int mode, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(x && planes, "vqw_f16x3_split_activations: null pointer");
    VQW_CHECK(B > 0 && T > 0 && C > 0 && C % 8 == 0, "vqw_f16x3_split_activations: C must be a positive multiple of 8 (got %d)", C);
    if (KC <= 0) { KC = C / 8; kc0 = 0; }
    VQW_CHECK(kc0 >= 0 && kc0 + C / 8 <= KC, "vqw_f16x3_split_activations: bad chunk range (kc0=%d KC=%d)", kc0, KC);
    const size_t n = (size_t)B * T * (C / 8);
    VQW_CHECK(mode >= 0 && mode <= 7, "vqw_f16x3_split_activations: mode is a bit set of VQW_X3_BF16 | VQW_X3_HALF_BLOCKS | VQW_X3_S2D");
    const int s2d = (mode & VQW_X3_S2D) ? 1 : 0;
    VQW_CHECK(!s2d || (T % 2 == 0 && kc0 == 0 && KC == C / 8), "vqw_f16x3_split_activations: space-to-depth planes need an even T and the whole plane (T=%d kc0=%d)", T, kc0);
    if (mode & 1) hipLaunchKernelGGL(split_act_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, (uint4*)planes, B, C, T, scale, kc0, KC,
                                 scale_dev, amax, flag, s2d);
    else hipLaunchKernelGGL(split_act_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, (uint4*)planes, B, C, T, scale, kc0, KC,
                            scale_dev, amax, flag, s2d);
    VQW_LAUNCH_CHECK("vqw_f16x3_split_activations");
    return 0;
}

int vqw_f16x3_pack_gate_weights(const float* w, void* planes, int ks, int R, int ldw, float scale, int count, const float* scale_dev, int mode, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(w && planes, "vqw_f16x3_pack_gate_weights: null pointer");
    VQW_CHECK(ks >= 1 && R > 0 && R % 128 == 0 && ldw >= 2 * R && count >= 1 && count <= 65535, "vqw_f16x3_pack_gate_weights: needs R %% 128 == 0, ldw >= 2R, 1 <= count <= 65535 (R=%d ldw=%d count=%d)", R, ldw, count);
    const int n = (ks * R / 8) * 2 * R;
    const int hb = (mode & VQW_X3_HALF_BLOCKS) ? 128 : 256;      // must match the block height of the gate conv that reads the planes
    if (mode & 1) hipLaunchKernelGGL(pack_gate_w_kernel<true>, dim3((n + 255) / 256, count), dim3(256), 0, st, w, (uint4*)planes, ks, R, ldw, scale, scale_dev, hb);
    else hipLaunchKernelGGL(pack_gate_w_kernel<false>, dim3((n + 255) / 256, count), dim3(256), 0, st, w, (uint4*)planes, ks, R, ldw, scale, scale_dev, hb);
    VQW_LAUNCH_CHECK("vqw_f16x3_pack_gate_weights");
    return 0;
}

int vqw_f16x3_pack_weights(const float* w, void* planes, int K, int M, int ldw, float scale, int count, const float* scale_dev, int mode, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(w && planes, "vqw_f16x3_pack_weights: null pointer");
    VQW_CHECK(K > 0 && K % 8 == 0 && M > 0 && ldw >= M && count >= 1 && count <= 65535, "vqw_f16x3_pack_weights: needs K %% 8 == 0, ldw >= M, 1 <= count <= 65535 (K=%d M=%d ldw=%d count=%d)", K, M, ldw, count);
    const int n = (K / 8) * M;
    if (mode & 1) hipLaunchKernelGGL(pack_w_kernel<true>, dim3((n + 255) / 256, count), dim3(256), 0, st, w, (uint4*)planes, K, M, ldw, scale, scale_dev);
    else hipLaunchKernelGGL(pack_w_kernel<false>, dim3((n + 255) / 256, count), dim3(256), 0, st, w, (uint4*)planes, K, M, ldw, scale, scale_dev);
    VQW_LAUNCH_CHECK("vqw_f16x3_pack_weights");
    return 0;
}

int vqw_f16x3_pack_weights_t(const float* w, void* planes, int K, int M, int k_inner, int ld_src, int64_t blk_stride, float scale, int count,
                             const float* scale_dev, int mode, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(w && planes, "vqw_f16x3_pack_weights_t: null pointer");
    VQW_CHECK(K > 0 && M > 0 && k_inner > 0 && k_inner % 8 == 0 && K % k_inner == 0 && ld_src >= k_inner && ld_src % 4 == 0 && blk_stride % 4 == 0 &&
              blk_stride >= 0 && count >= 1 && count <= 65535 && (reinterpret_cast<uintptr_t>(w) & 15) == 0,
              "vqw_f16x3_pack_weights_t: needs k_inner %% 8 == 0, K %% k_inner == 0, ld_src >= k_inner, 16-byte aligned rows (K=%d M=%d k_inner=%d ld_src=%d)",
              K, M, k_inner, ld_src);
    const int n = (K / 8) * M;
    if (k_inner % 64 == 0) {        // whole 64-row tiles inside every block of k: the LDS-transposing kernel
        const dim3 grid(((M + 63) / 64) * (K / 64), count);
        if (mode & 1) hipLaunchKernelGGL(pack_w_t_tile_kernel<true>, grid, dim3(256), 0, st, w, (uint4*)planes, K, M, k_inner, ld_src, (long)blk_stride, scale, scale_dev);
        else hipLaunchKernelGGL(pack_w_t_tile_kernel<false>, grid, dim3(256), 0, st, w, (uint4*)planes, K, M, k_inner, ld_src, (long)blk_stride, scale, scale_dev);
    } else if (mode & 1) hipLaunchKernelGGL(pack_w_t_kernel<true>, dim3((n + 255) / 256, count), dim3(256), 0, st, w, (uint4*)planes, K, M, k_inner, ld_src, (long)blk_stride, scale, scale_dev);
    else hipLaunchKernelGGL(pack_w_t_kernel<false>, dim3((n + 255) / 256, count), dim3(256), 0, st, w, (uint4*)planes, K, M, k_inner, ld_src, (long)blk_stride, scale, scale_dev);
    VQW_LAUNCH_CHECK("vqw_f16x3_pack_weights_t");
    return 0;
}

int vqw_f16x3_out_conv(const vqw_f16x3_out_desc* dp, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(dp, "vqw_f16x3_out_conv: null descriptor");
    const vqw_f16x3_out_desc& d = *dp;
    VQW_CHECK(d.xp && d.wp, "vqw_f16x3_out_conv: null operand");
    VQW_CHECK((d.S == 0 || d.skip) && (d.R == 0 || d.net_out || (d.epi == 1 && d.net_out_planes)), "vqw_f16x3_out_conv: null output");
    VQW_CHECK(d.ks >= 0 && d.ks <= 8 && d.dilation >= 0, "vqw_f16x3_out_conv: bad kernel size %d / dilation %d", d.ks, d.dilation);
    VQW_CHECK(d.B > 0 && d.T > 0 && d.T % 256 == 0, "vqw_f16x3_out_conv: T must be a positive multiple of 256 (got %d)", d.T);
    VQW_CHECK(d.R >= 0 && d.R % 256 == 0 && d.S >= 0 && d.S % 256 == 0 && d.S + d.R > 0, "vqw_f16x3_out_conv: R and S must be multiples of 256 (R=%d S=%d)", d.R, d.S);
    const int cin = d.Cin > 0 ? d.Cin : d.R, xkc = d.xp_KC > 0 ? d.xp_KC : cin / 8;
    VQW_CHECK(cin >= 64 && cin % 32 == 0 && d.xp_kc0 >= 0 && d.xp_kc0 + cin / 8 <= xkc && (d.ks <= 1 || d.xp_kc0 == 0), "vqw_f16x3_out_conv: bad contraction range (Cin=%d kc0=%d KC=%d)", cin, d.xp_kc0, xkc);
    VQW_CHECK((size_t)(cin / 8) * d.B * d.T * 16 < (size_t)1 << 31, "vqw_f16x3_out_conv: the chunks of one contraction (Cin = %d) exceed 2 GiB per plane: cut it into channel groups", cin);
    VQW_CHECK(d.w_scale_inv > 0.0f, "vqw_f16x3_out_conv: w_scale_inv must be positive");
    OutArgs a;
    a.d = d;
    a.NB = d.B * d.T;
    VQW_CHECK(d.mode >= 0 && d.mode <= 3, "vqw_f16x3_out_conv: mode is a bit set of VQW_X3_BF16 | VQW_X3_HALF_BLOCKS");
    const bool bf = (d.mode & 1) != 0, half = (d.mode & VQW_X3_HALF_BLOCKS) != 0;
    typedef void (*kfn_t)(OutArgs);
    const kfn_t kouts[4] = {out_f16x3_kernel<false, 8>, out_f16x3_kernel<true, 8>, out_f16x3_kernel<false, 4>, out_f16x3_kernel<true, 4>};
    const kfn_t kbwds[4] = {gate_bwd_f16x3_kernel<false, 8>, gate_bwd_f16x3_kernel<true, 8>, gate_bwd_f16x3_kernel<false, 4>, gate_bwd_f16x3_kernel<true, 4>};
    const int lds = half ? X3Shape<4>::LDS_BYTES : X3Shape<8>::LDS_BYTES, hb = half ? 128 : 256;
    const bool bwd = d.epi == 1;
    if (d.epi == 2) {
        VQW_CHECK(d.S == 0 && d.R > 0 && d.net_out && d.Cin > 0 && d.ks <= 1, "vqw_f16x3_out_conv: epi 2 needs S = 0, net_out, Cin and a 1x1 kernel");
        VQW_CHECK(!d.cond || (d.cond_T > 0 && d.T % d.cond_T == 0 && (d.T / d.cond_T) % 32 == 0 && d.cond_bstride >= (int64_t)d.R * d.cond_T),
                  "vqw_f16x3_out_conv: T / cond_T must be a multiple of 32 (T=%d cond_T=%d)", d.T, d.cond_T);
        const kfn_t kfn = bf ? (half ? head_f16x3_kernel<4, true> : head_f16x3_kernel<8, true>) : (half ? head_f16x3_kernel<4> : head_f16x3_kernel<8>);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return vqw_set_error("vqw_f16x3_out_conv: cannot reserve %d bytes of LDS", lds);
        hipLaunchKernelGGL(kfn, dim3((d.R / hb) * (a.NB / 256)), dim3(256), lds, st, a);
        VQW_LAUNCH_CHECK("vqw_f16x3_out_conv");
        return 0;
    }
    if (bwd) VQW_CHECK(d.S == 0 && d.R > 0 && d.aux0 && d.aux1 && d.Cin > 0, "vqw_f16x3_out_conv: gate backward needs S = 0, saved tanh (aux0) and sigmoid (aux1), Cin");
    kfn_t kfn = (bwd ? kbwds : kouts)[(bf ? 1 : 0) + (half ? 2 : 0)];
    if (bwd && (d.flags & 6)) {
        const bool pl = (d.flags & 4) != 0;      // bit 2: aux0 = the gated planes [planes][aux0_KC][B*T][8] from chunk aux0_kc0

        VQW_CHECK(!pl || (d.aux0_kc0 >= 0 && d.aux0_kc0 + d.R / 8 <= (d.aux0_KC > 0 ? d.aux0_KC : d.R / 8)), "vqw_f16x3_out_conv: bad chunk range of the gated planes");
        const kfn_t kg[8] = {gate_bwd_f16x3_kernel<false, 8, 1>, gate_bwd_f16x3_kernel<false, 4, 1>, gate_bwd_f16x3_kernel<false, 8, 2>,
                             gate_bwd_f16x3_kernel<false, 4, 2>, gate_bwd_f16x3_kernel<true, 8, 2>, gate_bwd_f16x3_kernel<true, 4, 2>,
                             gate_bwd_f16x3_kernel<true, 8, 1>, gate_bwd_f16x3_kernel<true, 4, 1>};
        kfn = kg[(pl ? (bf ? 4 : 2) : (bf ? 6 : 0)) + (half ? 1 : 0)];
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return vqw_set_error("vqw_f16x3_out_conv: cannot reserve %d bytes of LDS", lds);
    const int rows = bwd ? d.R : d.S + d.R;
    hipLaunchKernelGGL(kfn, dim3((rows / hb) * (a.NB / 256)), dim3(256), lds, st, a);
    VQW_LAUNCH_CHECK("vqw_f16x3_out_conv");
    return 0;
}

int vqw_f16x3_gate_conv(const vqw_f16x3_gate_desc* dp, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(dp, "vqw_f16x3_gate_conv: null descriptor");
    const vqw_f16x3_gate_desc& d = *dp;
    VQW_CHECK(d.xp && d.wp && (d.out0 || (d.out_planes && d.save1)), "vqw_f16x3_gate_conv: null operand (out0 may be NULL only with out_planes and save1)");
    VQW_CHECK(d.B > 0 && d.T > 0 && d.T % 256 == 0, "vqw_f16x3_gate_conv: T must be a positive multiple of 256 (got %d)", d.T);
    VQW_CHECK(d.R > 0 && d.R % 128 == 0, "vqw_f16x3_gate_conv: R must be a multiple of 128 (got %d)", d.R);
    VQW_CHECK(d.ks >= 1 && d.ks <= 8 && d.dilation >= 1, "vqw_f16x3_gate_conv: bad kernel size %d / dilation %d", d.ks, d.dilation);
    VQW_CHECK((size_t)(d.R / 8) * d.B * d.T * 16 < (size_t)1 << 31, "vqw_f16x3_gate_conv: one activation plane exceeds 2 GiB");
    VQW_CHECK(d.out_planes_KC == 0 || (d.out_planes_kc0 >= 0 && d.out_planes_kc0 + d.R / 8 <= d.out_planes_KC),
              "vqw_f16x3_gate_conv: bad output plane range (kc0=%d KC=%d)", d.out_planes_kc0, d.out_planes_KC);
    VQW_CHECK(d.w_scale_inv > 0.0f, "vqw_f16x3_gate_conv: w_scale_inv must be positive");
    GateArgs a;
    a.d = d;
    a.NB = d.B * d.T;
    a.ratio = 1;
    if (d.cond) {
        VQW_CHECK(d.cond_T > 0 && d.T % d.cond_T == 0 && (d.T / d.cond_T) % 32 == 0,
                  "vqw_f16x3_gate_conv: T / cond_T must be a multiple of 32 (T=%d cond_T=%d)", d.T, d.cond_T);
        a.ratio = d.T / d.cond_T;
    }
    VQW_CHECK(d.mode >= 0 && d.mode <= 3, "vqw_f16x3_gate_conv: mode is a bit set of VQW_X3_BF16 | VQW_X3_HALF_BLOCKS");
    const bool bf = (d.mode & 1) != 0, half = (d.mode & VQW_X3_HALF_BLOCKS) != 0;
    typedef void (*kfn_t)(GateArgs);
    const kfn_t ks4[4] = {gate_f16x3_kernel<false, 8>, gate_f16x3_kernel<true, 8>, gate_f16x3_kernel<false, 4>, gate_f16x3_kernel<true, 4>};
    const kfn_t kfn = ks4[(bf ? 1 : 0) + (half ? 2 : 0)];
    const int lds = half ? X3Shape<4>::LDS_BYTES : X3Shape<8>::LDS_BYTES;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return vqw_set_error("vqw_f16x3_gate_conv: cannot reserve %d bytes of LDS", lds);
    const int blocks = (d.R / (half ? 64 : 128)) * (a.NB / 256);
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, st, a);
    VQW_LAUNCH_CHECK("vqw_f16x3_gate_conv");
    return 0;
}

int vqw_f16x3_strided_conv(const vqw_f16x3_sconv_desc* dp, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(dp, "vqw_f16x3_strided_conv: null descriptor");
    const vqw_f16x3_sconv_desc& d = *dp;
    VQW_CHECK(d.xp && d.wp && d.out, "vqw_f16x3_strided_conv: null operand");
    VQW_CHECK(d.B > 0 && d.T > 0, "vqw_f16x3_strided_conv: bad shape (B=%d T=%d)", d.B, d.T);
    VQW_CHECK(d.Cin >= 64 && d.Cin % 32 == 0 && d.M > 0 && d.M % 128 == 0, "vqw_f16x3_strided_conv: Cin %% 32, M %% 128 (Cin=%d M=%d)", d.Cin, d.M);
    VQW_CHECK(d.ks >= 2 && d.ks <= 8 && d.pad_left >= 0 && d.pad_left < d.ks, "vqw_f16x3_strided_conv: 2..8 taps, 0 <= pad_left < ks (ks=%d pad_left=%d)", d.ks, d.pad_left);
    VQW_CHECK(!d.bn_scale || d.bn_shift, "vqw_f16x3_strided_conv: bn_scale needs bn_shift");
    VQW_CHECK(d.w_scale_inv > 0.0f, "vqw_f16x3_strided_conv: w_scale_inv must be positive");
    const size_t xbytes = (size_t)2 * (d.dgrad ? 1 : 2) * (d.Cin / 8) * d.B * d.T * 16, obytes = (size_t)d.B * d.M * d.T * (d.dgrad ? 2 : 1) * 4;
    VQW_CHECK(xbytes < ((size_t)1 << 31) && obytes < ((size_t)1 << 31) && (size_t)2 * d.ks * (d.Cin / 8) * d.M * 16 < ((size_t)1 << 31),
              "vqw_f16x3_strided_conv: operands exceed 2 GiB");
    SconvArgs a;
    a.d = d;
    a.NB = d.B * d.T;
    // block shape (d.shape forces one).  Measured on the benchmark's layers (tools/sconv_bench.py, us, shapes 1 / 2 / 3 / 4):
    //   forward  T=1664: 299 / 290 / 228 / 196   T=832: 154 / 129 / 196 / 155   T=416: 150 / 125 / 196 / 152
    //   dgrad    T=1664: 314 / 311 / 303 / 236   T=832: 198 / 142 / 275 / 184   T=416: 196 / 134 / 273 / 182
    // -> 256-row blocks for a launch that fills more than half of the chip with them, otherwise the deep 128-row shape
    const int cus = vqw_device_cus(), nt = (a.NB + 255) / 256;      // (a partial last column tile is masked in the epilogue)
    int shape = d.shape;
    if (shape == 0) shape = (d.M % 256 == 0 && (d.M / 256) * nt * 2 > cus) ? 3 : 2;
    // 192-row blocks where they fill more CUs than 256-row blocks without a second round (768 rows x 52 column tiles: 208 blocks
    // instead of 156 on 256 CUs)
    if (d.shape == 0 && shape == 3 && d.M % 192 == 0 && (d.M / 192) * nt <= cus && VQW_SCONV_192) shape = 4;
    // 64-row blocks for the shortest layers (encoder layers 4 and 5: 42 / 24 blocks of 128 rows on 256 CUs): twice the blocks at half
    // the MFMAs per step -- 105 against 125 us per launch, tools/sconv_bench.py (a step then costs ~880 cycles for 384 of MFMA: the
    // issue pattern has 12 MFMA shadows for ~20 memory instructions; four stages of requests in flight measured no better, 108 us)
    if (d.shape == 0 && shape == 2 && (d.M / 128) * nt * 4 <= cus && VQW_SCONV_64) shape = 5;
    VQW_CHECK(shape >= 1 && shape <= 5 && (shape != 3 || d.M % 256 == 0) && (shape != 4 || d.M % 192 == 0) && (shape != 5 || d.M % 64 == 0),
              "vqw_f16x3_strided_conv: shape is 0 (auto), 1, 2, 3 (256-row blocks: M %% 256 == 0), 4 (192-row blocks: M %% 192 == 0) or 5 (64-row blocks)");
    typedef void (*kfn_t)(SconvArgs);
    // Split-K (scratch given): a launch of few 128-row tiles cuts each tile's K steps over S blocks -- as many as fill the chip once, the
    // steps of every parity in even parts of at least 8 steps, at most 8 partial tiles per fix-up (256-row tiles measured slower at every
    // split count, tools/sconv_bench.py, and their fix-up does not fit the register file).
    if (d.split_slab && d.ksplit != 1 && (d.shape == 0 || d.shape == 2)) {
        VQW_CHECK(d.split_counters, "vqw_f16x3_strided_conv: split_slab needs split_counters");
        const int spt = d.Cin / 16, par = d.dgrad ? 2 : 1;
        const int st0 = d.dgrad ? ((d.ks - (d.pad_left & 1) + 1) / 2) * spt : d.ks * spt;          // K steps of parity 0 / the forward conv
        const int st1 = d.dgrad ? ((d.ks - ((1 + d.pad_left) & 1) + 1) / 2) * spt : st0;
        auto fits = [&](int S) { return S >= 1 && st0 % (2 * S) == 0 && st1 % (2 * S) == 0 && st0 / S >= 8 && st1 / S >= 8; };
        const int tiles = (d.M / 128) * nt * par;
        int S = d.ksplit;
        if (S == 0) {
            // one round of blocks; tools/sconv_bench.py (us per launch, 24 tiles: S = 2 / 4 / 8 / 10 / 12 -> 82 / 54 / 47.5 / 49 / 65): a K step
            // costs ~0.55 us, a partial tile in the fix-up ~2.5 us.  Where even two blocks per tile do not fit in one round but the unsplit
            // launch would leave two thirds of the chip idle (the input gradient of layer 3: 78 blocks of 120 steps), up to two rounds
            // (130 -> 90 us with three splits).
            for (int c = 2; c <= VQW_SCONV_SPLIT_MAX; ++c) if (fits(c) && tiles * c <= cus) S = c;
            if (S == 0 && (tiles / par) * 3 <= cus)
                for (int c = 2; c <= VQW_SCONV_SPLIT_MAX; ++c) if (fits(c) && tiles * c <= 2 * cus) S = c;
        }
        if (S > 1) {
            VQW_CHECK(fits(S), "vqw_f16x3_strided_conv: the K steps (%d / %d) do not divide into %d even parts of >= 8", st0, st1, S);
            VQW_CHECK(tiles <= d.split_counters_n && (int64_t)tiles * S * 128 * 256 <= d.split_slab_floats,
                      "vqw_f16x3_strided_conv: split scratch too small (%d tiles x %d splits)", tiles, S);
            a.S = S;
            const kfn_t kfn = d.dgrad ? sconv_f16x3_kernel<4, 2, true, true> : sconv_f16x3_kernel<4, 2, false, true>;
            const int lds = 4 * (4 * 2 + 16) * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return vqw_set_error("vqw_f16x3_strided_conv: cannot reserve %d bytes of LDS", lds);
            hipLaunchKernelGGL(kfn, dim3(tiles * S), dim3(256), lds, st, a);
            VQW_LAUNCH_CHECK("vqw_f16x3_strided_conv");
            return 0;
        }
    }
    a.S = 1;
    const kfn_t kfn = d.dgrad ? (shape == 5 ? sconv_f16x3_kernel<2, 2, true> : (shape == 4 ? sconv_f16x3_kernel<6, 2, true> : (shape == 3 ? sconv_f16x3_kernel<8, 2, true> : (shape == 2 ? sconv_f16x3_kernel<4, 2, true> : sconv_f16x3_kernel<4, 1, true>))))
                              : (shape == 5 ? sconv_f16x3_kernel<2, 2, false> : (shape == 4 ? sconv_f16x3_kernel<6, 2, false> : (shape == 3 ? sconv_f16x3_kernel<8, 2, false> : (shape == 2 ? sconv_f16x3_kernel<4, 2, false> : sconv_f16x3_kernel<4, 1, false>))));
    const int mr = shape == 5 ? 2 : (shape == 4 ? 6 : (shape == 3 ? 8 : 4)), lds = (shape == 1 ? 3 : 4) * (mr * 2 + 16) * 1024;      // DEPTH + 2 stages
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return vqw_set_error("vqw_f16x3_strided_conv: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL(kfn, dim3((d.M / (32 * mr)) * nt), dim3(256), lds, st, a);
    VQW_LAUNCH_CHECK("vqw_f16x3_strided_conv");
    return 0;
}

int vqw_f16x3_wgrad_batch(const vqw_f16x3_wgrad_desc* dp, int nprob, vqw_stream_t s_) {
    hipStream_t st = (hipStream_t)s_;
    VQW_CHECK(dp, "vqw_f16x3_wgrad: null descriptor");
    VQW_CHECK(nprob >= 1 && nprob <= WG_MAX_BATCH, "vqw_f16x3_wgrad_batch: 1..%d problems per launch (got %d)", WG_MAX_BATCH, nprob);
    const vqw_f16x3_wgrad_desc& d = dp[0];
    VQW_CHECK(d.slab, "vqw_f16x3_wgrad: null operand");
    VQW_CHECK(d.B > 0 && d.T > 0 && (d.T % 32 == 0 || (d.p_stride == 2 && d.T % 4 == 0) || (d.p_planes && d.q_planes && !d.q_seg)),
              "vqw_f16x3_wgrad: T must be a positive multiple of 32 (of 4 with p_stride 2; any with both operands as planes) (got %d)", d.T);
    VQW_CHECK(d.Cp > 0 && d.Cp % 256 == 0 && d.Q0 > 0 && d.Q0 % 256 == 0 && d.Q1 >= 0 && d.Q1 % 256 == 0,
              "vqw_f16x3_wgrad: Cp, Q0, Q1 must be multiples of 256 (Cp=%d Q0=%d Q1=%d)", d.Cp, d.Q0, d.Q1);
    VQW_CHECK(d.ntaps >= 1 && d.ntaps <= VQW_MAX_TAPS, "vqw_f16x3_wgrad: 1..%d taps", VQW_MAX_TAPS);
    VQW_CHECK(d.p_stride == 0 || d.p_stride == 1 || d.p_stride == 2, "vqw_f16x3_wgrad: p_stride is 1 or 2 (got %d)", d.p_stride);
    const bool s2 = d.p_stride == 2;
    const int Tp = s2 ? d.Tp : d.T;
    VQW_CHECK(!s2 || (Tp > 0 && !(d.mode & 1)), "vqw_f16x3_wgrad: p_stride 2 needs Tp > 0 and the fp16x3 mode (Tp=%d mode=%d)", Tp, d.mode);
    const size_t pbytes = (size_t)d.B * d.Cp * Tp * 4, qbytes = (size_t)d.B * (d.Q0 > d.Q1 ? d.Q0 : d.Q1) * d.T * 4;
    VQW_CHECK(pbytes < ((size_t)1 << 31) && qbytes < ((size_t)1 << 31), "vqw_f16x3_wgrad: operands exceed 2 GiB");
    const int lddw = d.lddw > 0 ? d.lddw : d.Q0 + d.Q1;
    VQW_CHECK(lddw >= d.Q0 + d.Q1 && lddw % 4 == 0, "vqw_f16x3_wgrad: lddw must be a multiple of 4 and >= Q0 + Q1");
    VQW_CHECK(d.mode >= 0 && d.mode <= 3, "vqw_f16x3_wgrad: mode is a bit set of VQW_X3_BF16 | VQW_X3_HALF_BLOCKS (the latter has no effect here)");
    WgArgs a;
    memset(&a, 0, sizeof(a));
    bool odd = false;
    for (int i = 0; i < nprob; ++i) {
        const vqw_f16x3_wgrad_desc& e = dp[i];
        // one shape per launch: only the operands, scales, sums, tap shifts and the output differ between the problems
        VQW_CHECK(e.B == d.B && e.T == d.T && e.Cp == d.Cp && e.Q0 == d.Q0 && e.Q1 == d.Q1 && e.ntaps == d.ntaps && e.lddw == d.lddw &&
                  e.dw_tap_stride == d.dw_tap_stride && e.seg_bstride == d.seg_bstride && e.seg_T == d.seg_T && e.total_o0 == d.total_o0 &&
                  e.total_o1 == d.total_o1 && e.mode == d.mode && e.p_relu == d.p_relu && e.p_stride == d.p_stride && e.Tp == d.Tp &&
                  e.slab == d.slab && (e.q_seg != nullptr) == (d.q_seg != nullptr),
                  "vqw_f16x3_wgrad_batch: problem %d differs from problem 0 in shape or layout", i);
        VQW_CHECK((e.p || e.p_planes) && (e.q0 || e.q_planes) && e.dw && (e.Q1 == 0 || e.q1), "vqw_f16x3_wgrad: null operand (problem %d)", i);
        VQW_CHECK((e.q_planes != nullptr) == (d.q_planes != nullptr) && (e.p_planes != nullptr) == (d.p_planes != nullptr),
                  "vqw_f16x3_wgrad_batch: an operand as planes in every problem or in none");
        if (e.p_planes) {
            const int kc = e.p_planes_KC > 0 ? e.p_planes_KC : d.Cp / 8;
            for (int j = 0; j < d.ntaps; ++j)
                VQW_CHECK(e.p_planes_kc0 >= 0 && e.p_tap_chunk[j] >= 0 && e.p_planes_kc0 + e.p_tap_chunk[j] + d.Cp / 8 <= kc,
                          "vqw_f16x3_wgrad: bad chunk range of the p planes (kc0=%d tap %d chunk offset %d KC=%d)", e.p_planes_kc0, j, e.p_tap_chunk[j], kc);
        }
        if (e.q_planes) {
            const int kc = e.q_planes_KC > 0 ? e.q_planes_KC : d.Q0 / 8;
            VQW_CHECK(e.q_planes_kc0 >= 0 && e.q_planes_kc0 + d.Q0 / 8 <= kc, "vqw_f16x3_wgrad: bad chunk range of the q planes (kc0=%d KC=%d)", e.q_planes_kc0, kc);
        }
        VQW_CHECK((reinterpret_cast<uintptr_t>(e.dw) & 15u) == 0, "vqw_f16x3_wgrad: dw must be 16-byte aligned (problem %d)", i);
        WgProblem& q = a.pr[i];
        q.p = e.p; q.q0 = e.q0; q.q1 = e.q1; q.dw = e.dw; q.sp = e.p_scale; q.sq0 = e.q0_scale; q.sq1 = e.q1_scale;
        q.q_total = e.q_total; q.q_seg = e.q_seg;
        q.qp = e.q_planes; q.q_KC = e.q_planes_KC > 0 ? e.q_planes_KC : d.Q0 / 8; q.q_kc0 = e.q_planes_kc0;
        q.q_hscale = e.q_planes_scale > 0.0f ? e.q_planes_scale : 1.0f;
        q.pp = e.p_planes; q.p_KC = e.p_planes_KC > 0 ? e.p_planes_KC : d.Cp / 8; q.p_kc0 = e.p_planes_kc0;
        q.p_hscale = e.p_planes_scale > 0.0f ? e.p_planes_scale : 1.0f;
        for (int j = 0; j < d.ntaps; ++j) {
            VQW_CHECK(((s2 || e.p_planes) ? e.tap_shift[j] < (1 << 24) : e.tap_shift[j] <= 0) && e.tap_shift[j] > -(1 << 24),
                      "vqw_f16x3_wgrad: tap shifts must be <= 0 unless p_stride is 2 or p comes as planes (problem %d, tap %d: %d)", i, j, e.tap_shift[j]);
            q.shift[j] = e.tap_shift[j];
            q.pkoff[j] = e.p_planes ? e.p_tap_chunk[j] : 0;
            odd |= (e.tap_shift[j] & 3) != 0;
        }
    }
    a.slab = d.slab; a.nprob = nprob;
    a.B = d.B; a.T = d.T; a.Cp = d.Cp; a.Q0 = d.Q0; a.Q1 = d.Q1; a.ntaps = d.ntaps; a.Tp = Tp; a.p_relu = d.p_relu;
    a.pairs_row = (d.T + 31) / 32; a.pairs_total = d.B * a.pairs_row;
    a.n_nt = (d.Q0 + d.Q1) / 256;
    const int tiles = nprob * d.ntaps * (d.Cp / 256) * a.n_nt;
    int nsplit = d.nsplit > 0 ? d.nsplit : vqw_device_cus() / tiles;     // one round of blocks
    if (nsplit < 1) nsplit = 1;
    if (nsplit > a.pairs_total) nsplit = a.pairs_total;
    a.nsplit = nsplit;
    a.seg_bstride = (long)d.seg_bstride; a.seg_T = d.seg_T;
    a.total_o0 = d.total_o0; a.total_o1 = d.total_o1 > 0 ? d.total_o1 : d.Q0 + d.Q1;
    a.seg_ratio = 1;
    if (d.q_seg) {
        VQW_CHECK(d.seg_T > 0 && d.T % d.seg_T == 0 && (d.T / d.seg_T) % 32 == 0 && d.seg_bstride >= (int64_t)(d.Q0 + d.Q1) * d.seg_T,
                  "vqw_f16x3_wgrad: q_seg needs T / seg_T to be a multiple of 32 (T=%d seg_T=%d)", d.T, d.seg_T);
        a.seg_ratio = d.T / d.seg_T;
    }
    a.lddw = lddw;
    a.dw_tap_stride = d.dw_tap_stride > 0 ? (long)d.dw_tap_stride : (long)d.Cp * lddw;
    VQW_CHECK((size_t)tiles * nsplit * 65536 <= (size_t)d.slab_floats, "vqw_f16x3_wgrad: slab too small (%d tiles x %d splits x 65536 floats)", tiles, nsplit);
    typedef void (*kfn_t)(WgArgs);
    const kfn_t ktab[4] = {wgrad_f16x3_kernel<false, false>, wgrad_f16x3_kernel<true, false>, wgrad_f16x3_kernel<false, true>,
                           wgrad_f16x3_kernel<true, true>};
    const kfn_t kqp[4] = {wgrad_f16x3_kernel<false, false, false, true>, wgrad_f16x3_kernel<true, false, false, true>,
                          wgrad_f16x3_kernel<false, true, false, true>, wgrad_f16x3_kernel<true, true, false, true>};
    const bool qp = d.q_planes != nullptr, pp = d.p_planes != nullptr, bf = (d.mode & 1) != 0;
    VQW_CHECK(!qp || (!s2 && d.Q1 == 0 && (size_t)32 * d.B * d.T * 16 < ((size_t)1 << 31)),
              "vqw_f16x3_wgrad: q as planes needs p_stride 1, Q1 = 0 and B * T < 4 M");
    VQW_CHECK(!pp || (!s2 && !d.p_relu && (size_t)32 * d.B * d.T * 16 < ((size_t)1 << 31)),
              "vqw_f16x3_wgrad: p as planes needs p_stride 1, no p_relu and B * T < 4 M");
    // (p as planes: a tap's shift is a row offset -- the unaligned-window variant is not needed)
    const kfn_t kpp[4] = {wgrad_f16x3_kernel<false, false, false, false, true>, wgrad_f16x3_kernel<false, true, false, false, true>,
                          wgrad_f16x3_kernel<false, false, false, true, true>, wgrad_f16x3_kernel<false, true, false, true, true>};
    const kfn_t kfn = s2 ? wgrad_f16x3_kernel<false, false, true>
                         : (pp ? kpp[(bf ? 1 : 0) + (qp ? 2 : 0)] : (qp ? kqp : ktab)[(odd ? 1 : 0) + 2 * (bf ? 1 : 0)]);
    const int npl = bf ? 1 : 2;
    const int lds = NSTG * ((pp ? npl * WG_QPL : 16 * 1024) + (qp ? npl * WG_QPL : 16 * 1024));
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return vqw_set_error("vqw_f16x3_wgrad: cannot reserve %d bytes of LDS", lds);
    hipLaunchKernelGGL(kfn, dim3(tiles * nsplit), dim3(256), lds, st, a);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(65536 / 4 / 256, tiles), dim3(256), 0, st, a);
    VQW_LAUNCH_CHECK("vqw_f16x3_wgrad");
    return 0;
}

int vqw_f16x3_wgrad(const vqw_f16x3_wgrad_desc* dp, vqw_stream_t s_) { return vqw_f16x3_wgrad_batch(dp, 1, s_); }

}  // extern "C"

// Experiment, not part of libvqwave: an fp32-accurate GEMM on the 16-bit matrix pipe of gfx950.
//
//   a = a1 + a2 + a3 exactly, a_i bf16 (3 x 8 significand bits = fp32's 24), likewise b.
//   a*b = sum_{i,j} a_i b_j; the six terms with i + j <= 4 carry the product to 2^-25 relative (below fp32's
//   own unit roundoff), every a_i b_j is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.
//   NT = 6 -> "fp32 on the bf16 pipe" at 1/6 of the bf16 rate (paper: 2.5 PF / 6 = 417 TF vs 157 TF native fp32);
//   NT = 3 (i + j <= 3) is shown for contrast (2^-16 relative: NOT fp32).
//   fp16 variant: a = h1 + h2 (2 x 11 bits), h1 h1 + h1 h2 + h2 h1 on v_mfma_f32_32x32x16_f16: 3 terms, 4 bytes per
//   element, measured at least as accurate as sequential fp32 fma (the dropped h2 h2 averages out).
//
// Shape: the decoder's gate conv as a GEMM, C[M=512][N] = A[M][K=768] B[K][N], N = B*T.
// Operands live in HBM as three bf16 planes in "chunk-major" order P[plane][K/8][rows][8]: the 16-byte entries of 32
// consecutive rows are contiguous, so one LDS-DMA instruction (64 lanes x 16 B) fetches a 32-row x 16-k MFMA
// operand fragment as 2 x 512 contiguous bytes and lands it lane-linear in LDS = exactly the order ds_read_b128
// hands it to the MFMA (no padding, no VGPR staging, a dilation shift of the conv is a row offset).
// Block 256 x 256, 4 waves (2 x 2) of 128 x 128, BK = 16, three LDS stages of 48 KB (bf16) / four of 32 KB (fp16).
//
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/micro/split_bf16_gemm.hip -o tools/micro/split_bf16_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../vq-vae-wavenet_amd/csrc/vqw_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u16 bf16_rn(float x) {
    const unsigned u = __float_as_uint(x);
    return (u16)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ u16 f16_bits(_Float16 h) { return __builtin_bit_cast(u16, h); }

// src[rows][K] fp32 row-major -> planes[NP][K/8][rows][8]; NP = 3: bf16 pieces; NP = 2: fp16 pieces of scale * src
template <int NP>
__global__ void split_kernel(const float* __restrict__ src, u16* __restrict__ planes, int rows, int ld, int K, float scale) {
    const int KC = K / 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * KC) return;
    const int row = (int)(i % rows), kc = (int)(i / rows);
    u16 o[3][8];
    for (int e = 0; e < 8; ++e) {
        const float a = src[(size_t)row * K + kc * 8 + e] * scale;
        if (NP == 3) {
            const u16 h1 = bf16_rn(a);
            const float r1 = a - bf16_f(h1);
            const u16 h2 = bf16_rn(r1);
            const float r2 = r1 - bf16_f(h2);
            o[0][e] = h1; o[1][e] = h2; o[2][e] = bf16_rn(r2);
        } else {
            const _Float16 h1 = (_Float16)a;             // round to nearest even
            const float r1 = a - (float)h1;              // exact
            o[0][e] = f16_bits(h1); o[1][e] = f16_bits((_Float16)r1);
        }
    }
    for (int p = 0; p < NP; ++p) {
        uint4 v;
        v.x = o[p][0] | ((unsigned)o[p][1] << 16); v.y = o[p][2] | ((unsigned)o[p][3] << 16);
        v.z = o[p][4] | ((unsigned)o[p][5] << 16); v.w = o[p][6] | ((unsigned)o[p][7] << 16);
        *reinterpret_cast<uint4*>(planes + (((size_t)p * KC + kc) * ld + row) * 8) = v;
    }
}

constexpr int BM = 256, BN = 256, BK = 16;
template <int NP> struct Cfg {
    static constexpr int PA = (BM / 32) * NP, PB = (BN / 32) * NP;   // 1-KiB pieces per stage
    static constexpr int STAGE = (PA + PB) * 1024;                    // 48 KB (bf16 x 3 planes) / 32 KB (fp16 x 2)
    static constexpr int PER_WAVE = (PA + PB) / 4;                    // DMA instructions per wave and stage
    static constexpr int NST = NP == 3 ? 3 : 4;                       // LDS stages
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NP> __device__ __forceinline__ f32x16 mma(uint4 a, uint4 b, f32x16 c) {
    if (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// NT terms, least significant first.  bf16 (NP = 3): 6 = fp32-accurate, 3, 1.  fp16 (NP = 2): 3 = fp32-accurate
// (h1 h1 + h1 h2 + h2 h1; the dropped h2 h2 is 2^-22 relative per product and averages out below the fp32
// accumulation error), 1.
template <int NP, int NT, int DMA>   // DMA: 0 no operand traffic (timing only), 1 LDS-DMA, 2 through VGPRs
__global__ __launch_bounds__(256, 1) void gemm_kernel(const u16* __restrict__ Ap, const u16* __restrict__ Bp, float* __restrict__ C,
                                                       int M, int N, int K, int ldA, int ldB, float out_scale) {
    using G = Cfg<NP>;
    constexpr int PA = G::PA, PER_WAVE = G::PER_WAVE, STAGE = G::STAGE, NST = G::NST, AHEAD = NST - 1;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int n_mt = M / BM;
    const int bid = vqw_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid % n_mt) * BM, n0 = (bid / n_mt) * BN;
    const int KC = K / 8, nsteps = K / BK;
    const __amdgpu_buffer_rsrc_t ra = vqw_make_rsrc(Ap, (unsigned)((size_t)NP * KC * ldA * 16));
    const __amdgpu_buffer_rsrc_t rb = vqw_make_rsrc(Bp, (unsigned)((size_t)NP * KC * ldB * 16));

    // this wave's DMA pieces: q = wv * PER_WAVE + i; q < PA: A piece (tile q / NP, plane q % NP), else B
    int voff[PER_WAVE];
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int q = wv * PER_WAVE + i;
        const bool isA = q < PA;
        const int qq = isA ? q : q - PA;
        const int tile = qq / NP, p = qq % NP;
        const int rows = isA ? ldA : ldB, r0 = isA ? m0 : n0;
        voff[i] = ((p * KC + (lane >> 5)) * rows + r0 + tile * 32 + (lane & 31)) * 16;
    }
    auto issue = [&](int s) {
        char* dst = smem + (s % NST) * STAGE + wv * PER_WAVE * 1024;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wv * PER_WAVE + i;      // wave-uniform
            const bool isA = q < PA;
            const int soff = s * 2 * (isA ? ldA : ldB) * 16;
            vqw_buf_load_lds16(isA ? ra : rb, reinterpret_cast<float*>(dst + i * 1024), voff[i], soff);
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // register path: stage s + 2 is requested in step s, written to LDS in step s + 1, read in step s + 2
    f32x4 rg[PER_WAVE];
    auto rissue = [&](int s) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wv * PER_WAVE + i;
            const bool isA = q < PA;
            rg[i] = vqw_buf_load4(isA ? ra : rb, voff[i], s * 2 * (isA ? ldA : ldB) * 16);
        }
    };
    auto rcommit = [&](int s) {
        char* dst = smem + (s % NST) * STAGE + wv * PER_WAVE * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) *reinterpret_cast<f32x4*>(dst + i * 1024) = rg[i];
    };
    if (DMA == 2) {
        rissue(0); rcommit(0);
        if (nsteps > 1) { rissue(1); rcommit(1); }
        if (nsteps > 2) rissue(2);
    }
    if (DMA == 1)
        for (int s = 0; s < AHEAD && s < nsteps; ++s) issue(s);
    for (int s = 0; s < nsteps; ++s) {
        if (DMA == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage s + 2's buffer (= stage s - 1's for NST = 3) is free
            if (s + 2 < nsteps) rcommit(s + 2);
            if (s + 3 < nsteps) rissue(s + 3);
        }
        if (DMA == 1) {                                  // stage s has landed when at most the later stages are outstanding
            const int later = nsteps - 1 - s;
            if (later >= AHEAD - 1) wait_vm<PER_WAVE * (AHEAD - 1)>();
            else if (AHEAD >= 3 && later == 1) wait_vm<PER_WAVE>();
            else wait_vm<0>();
        }
        if (DMA != 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (DMA == 1 && s + AHEAD < nsteps) issue(s + AHEAD);
        const char* st = smem + (s % NST) * STAGE;
        uint4 a[4][NP], b[4][NP];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                a[i][p] = *reinterpret_cast<const uint4*>(st + ((wm * 4 + i) * NP + p) * 1024 + lane * 16);
                b[i][p] = *reinterpret_cast<const uint4*>(st + PA * 1024 + ((wn * 4 + i) * NP + p) * 1024 + lane * 16);
            }
        constexpr int TA3[6] = {1, 0, 2, 0, 1, 0}, TB3[6] = {1, 2, 0, 1, 0, 0};   // bf16 terms, small first
        constexpr int TA2[3] = {0, 1, 0}, TB2[3] = {1, 0, 0};                     // fp16 terms
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (NP == 3) {
#pragma unroll
                    for (int t = 6 - NT; t < 6; ++t) acc[i][j] = mma<NP>(a[i][TA3[t]], b[j][TB3[t]], acc[i][j]);
                } else {
#pragma unroll
                    for (int t = 3 - NT; t < 3; ++t) acc[i][j] = mma<NP>(a[i][TA2[t]], b[j][TB2[t]], acc[i][j]);
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = m0 + (wm * 4 + i) * 32 + 8 * (v / 4) + 4 * (lane >> 5) + (v & 3);
                const int n = n0 + (wn * 4 + j) * 32 + (lane & 31);
                C[(size_t)m * N + n] = acc[i][j][v] * out_scale;
            }
}

// what the bf16 pipe sustains on random operands with no memory traffic at all (16 independent accumulators)
__global__ __launch_bounds__(256, 1) void bare_kernel(const u16* __restrict__ Ap, float* __restrict__ out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(Ap + ((size_t)threadIdx.x * 8 + i) * 8);
        b[i] = *reinterpret_cast<const bf16x8*>(Ap + ((size_t)threadIdx.x * 8 + 4 + i) * 8);
    }
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

static double urand(unsigned long long& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return ((s >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53);
}

template <int NP, int NT, int DMA = 1>
static void run(const char* name, const u16* Ap, const u16* Bp, float* C, int M, int N, int K, int ldA, int ldB, float out_scale,
                const std::vector<float>& hA, const std::vector<float>& hB) {
    const int blocks = (M / BM) * (N / BN);
    const size_t lds = (size_t)Cfg<NP>::NST * Cfg<NP>::STAGE;
    auto kfn = gemm_kernel<NP, NT, DMA>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, 0, Ap, Bp, C, M, N, K, ldA, ldB, out_scale);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kfn, dim3(blocks), dim3(256), lds, 0, Ap, Bp, C, M, N, K, ldA, ldB, out_scale);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    // accuracy on sampled outputs against fp64; error scaled by sum |a||b| (the quantity fp32 rounding is relative to)
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long s = 99;
    double emax = 0, e2 = 0, f32max = 0, f322 = 0;
    const int ns = 4096;
    for (int t = 0; t < ns; ++t) {
        const int m = (int)(urand(s) * M), n = (int)(urand(s) * N);
        double ref = 0, mag = 0;
        float f = 0.f;
        for (int k = 0; k < K; ++k) {
            const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
            ref += p; mag += fabs(p);
            f = fmaf(hA[(size_t)m * K + k], hB[(size_t)n * K + k], f);
        }
        const double e = fabs(hC[(size_t)m * N + n] - ref) / mag, ef = fabs((double)f - ref) / mag;
        emax = fmax(emax, e); e2 += e * e; f32max = fmax(f32max, ef); f322 += ef * ef;
    }
    const double flop = 2.0 * M * N * K;
    printf("%-8s N=%6d blocks=%4d: %.1f us, %.1f TFLOP/s fp32-equivalent (%.0f TFLOP/s executed); error / sum|a||b|: max %.2e rms %.2e"
           " (sequential fp32 fma on the host: max %.2e rms %.2e)\n",
           name, N, blocks, ms * 1e3, flop / ms * 1e-9, flop * NT / ms * 1e-9, emax, sqrt(e2 / ns), f32max, sqrt(f322 / ns));
}

int main(int argc, char** argv) {
    const int M = 512, K = 768;
    const int Ns[2] = {53248, 65536};
    for (int c = 0; c < 2; ++c) {
        const int N = Ns[c];
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned long long s = 1234;
        for (auto& x : hA) x = (float)((urand(s) * 2 - 1) * 0.0625);
        for (auto& x : hB) x = (float)((urand(s) * 2 - 1) * (urand(s) < 0.5 ? 1.0 : 0.01));
        float *dA, *dB, *dC;
        u16 *pA, *pB;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
        const int pad = argc > 1 ? atoi(argv[1]) : 0;   // rows of padding between the 16-byte-entry columns of a plane
        const int ldA = M + pad, ldB = N + pad;
        printf("---- N = %d, plane leading dimension = rows + %d\n", N, pad);
        CK(hipMalloc(&pA, (size_t)ldA * K * 6)); CK(hipMalloc(&pB, (size_t)ldB * K * 6));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0;
        const unsigned gA = (unsigned)(((size_t)M * K / 8 + 255) / 256), gB = (unsigned)(((size_t)N * K / 8 + 255) / 256);
        // ---- bf16, three planes
        hipLaunchKernelGGL(split_kernel<3>, dim3(gA), dim3(256), 0, 0, dA, pA, M, ldA, K, 1.0f);
        hipLaunchKernelGGL(split_kernel<3>, dim3(gB), dim3(256), 0, 0, dB, pB, N, ldB, K, 1.0f);
        CK(hipDeviceSynchronize());
        if (c == 0) {
            for (int blocks = 256; blocks <= 512; blocks *= 2) {
                const int iters = 4000;
                hipLaunchKernelGGL(bare_kernel, dim3(blocks), dim3(256), 0, 0, pB, dC, iters / 10);
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(bare_kernel, dim3(blocks), dim3(256), 0, 0, pB, dC, iters);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("bare v_mfma_f32_32x32x16_bf16 loop, %d blocks of 4 waves: %.1f us, %.0f TFLOP/s\n", blocks, ms * 1e3,
                       (double)blocks * 4 * iters * 16 * 32768.0 / ms * 1e-9);
            }
        }
        run<3, 6>("bf16x6", pA, pB, dC, M, N, K, ldA, ldB, 1.0f, hA, hB);
        run<3, 6, 0>("x6 nodma", pA, pB, dC, M, N, K, ldA, ldB, 1.0f, hA, hB);
        run<3, 3>("bf16x3", pA, pB, dC, M, N, K, ldA, ldB, 1.0f, hA, hB);
        run<3, 1>("bf16x1", pA, pB, dC, M, N, K, ldA, ldB, 1.0f, hA, hB);
        run<3, 1, 0>("x1 nodma", pA, pB, dC, M, N, K, ldA, ldB, 1.0f, hA, hB);
        // ---- fp16, two planes; the weights (|w| <= 2^-4) are scaled by 2^8 so that their residual plane stays in
        // fp16's normal range; the activations (|x| <= 1) are taken as they are
        const float sa = 256.0f;
        hipLaunchKernelGGL(split_kernel<2>, dim3(gA), dim3(256), 0, 0, dA, pA, M, ldA, K, sa);
        hipLaunchKernelGGL(split_kernel<2>, dim3(gB), dim3(256), 0, 0, dB, pB, N, ldB, K, 1.0f);
        CK(hipDeviceSynchronize());
        run<2, 3>("fp16x3", pA, pB, dC, M, N, K, ldA, ldB, 1.0f / sa, hA, hB);
        run<2, 3, 0>("x3 nodma", pA, pB, dC, M, N, K, ldA, ldB, 1.0f / sa, hA, hB);
        run<2, 3, 2>("x3 regs", pA, pB, dC, M, N, K, ldA, ldB, 1.0f / sa, hA, hB);
        run<2, 1, 2>("x1 regs", pA, pB, dC, M, N, K, ldA, ldB, 1.0f / sa, hA, hB);
        run<2, 1>("fp16x1", pA, pB, dC, M, N, K, ldA, ldB, 1.0f / sa, hA, hB);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(pA)); CK(hipFree(pB));
    }
    return 0;
}

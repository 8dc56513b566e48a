// Implicit-GEMM convolution engine for gfx950 (MI355X), true-fp32 MFMA.
//
// Replaces the TensorFlow runtime work behind the reference's conv call sites:
// wavenet_ops.py:81-89 (tf.pad + tf.nn.conv2d + bias), 93-101 (add_condition),
// 112-113 (gate), 132-136 + wavenet.py:72-73 (1x1 skip/residual + accumulation),
// encoder.py:15-25 (Conv1D + relu + BatchNorm) and their input gradients.
//
// GEMM view (per batch element):  D[m][t] = sum_k Wt[m][k] * X[k][t],  k = (tap, channel)
//   MFMA A operand = weights  (rows m = output channels), staged in LDS as As[k][m]
//   MFMA B operand = activations (cols t = time),          staged in LDS as Bs[k][t]
// Both LDS images are k-major with the MFMA row/col index contiguous, which is exactly the
// global layout of kernel[k][Cin][Cout] and of (batch, channel, time) activations: global
// reads are full-line coalesced and the dilation of a tap is only a shifted start address
// of the staged window ("LDS-staged dilation windows").
//
// v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31].  A wave
// owns MT x NT tiles of 32x32; tile e of the M direction holds rows m = MT*r + e and tile f
// of the N direction holds columns t = NT*c + f, so one ds_read_b64/b128 per lane fetches
// the operands of all MT (NT) tiles and every accumulator row stores NT contiguous floats.
#include <type_traits>

#include "vqw_common.h"

namespace {

#ifndef VQW_BK
#define VQW_BK 16
#endif
#ifndef VQW_SETPRIO
#define VQW_SETPRIO 0
#endif
#ifndef VQW_SCHED
#define VQW_SCHED 1
#endif
constexpr int BK = VQW_BK;  // channels per K-step
#ifndef VQW_NST
#define VQW_NST 3
#endif
constexpr int NST_DMA = VQW_NST;   // LDS stages of the LDS-DMA pipeline (experiment: 2 = one more block per CU)
#if VQW_SCHED
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define SCHED_FENCE() ((void)0)
#endif
#if VQW_SETPRIO
#define MMA_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define MMA_PRIO(p) ((void)0)
#endif

struct BlockGrid {   // one rectangle of output tiles: all M tiles x n_nt column tiles from t_begin x all batch rows
    int n_mt, n_nt, nwg, t_begin;
};

struct ConvArgs {
    vqw_conv_desc d;
    BlockGrid main, tail;   // tail.nwg == 0: no tail tiles
    int H;       // GATE / GATE_BWD: half width
    int ratio;   // cond: T_out / cond_T
    int vec_ok;  // output rows allow aligned vector stores
    int use_dma; // LDS-DMA pipeline for interior blocks (VQW_CONV_DMA=0 disables it)
    int ksplit;  // > 1: the K range of every tile is cut over ksplit blocks, partial tiles meet by fp32 atomics
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

template <int NT>
__device__ __forceinline__ void store_row(float* __restrict__ rowp, int tb, int T_out,
                                          int tstride, int toff, int vec_ok,
                                          const float (&v)[NT]) {
    if (NT > 1 && vec_ok && tb + NT <= T_out) {
        if constexpr (NT == 4) {
            *reinterpret_cast<f32x4*>(rowp + tb) = f32x4{v[0], v[1], v[2], v[3]};
        } else if constexpr (NT == 2) {
            *reinterpret_cast<f32x2*>(rowp + tb) = f32x2{v[0], v[1]};
        }
    } else {
#pragma unroll
        for (int f = 0; f < NT; ++f)
            if (tb + f < T_out) rowp[tstride * (tb + f) + toff] = v[f];
    }
}

template <int NT>
__device__ __forceinline__ void load_row(const float* __restrict__ rowp, int tb, int T_out,
                                         int tstride, int toff, int vec_ok, float (&v)[NT]) {
    if (NT > 1 && vec_ok && tb + NT <= T_out) {
        if constexpr (NT == 4) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(rowp + tb);
            v[0] = u[0]; v[1] = u[1]; v[2] = u[2]; v[3] = u[3];
        } else if constexpr (NT == 2) {
            const f32x2 u = *reinterpret_cast<const f32x2*>(rowp + tb);
            v[0] = u[0]; v[1] = u[1];
        }
    } else {
#pragma unroll
        for (int f = 0; f < NT; ++f)
            v[f] = (tb + f < T_out) ? rowp[tstride * (tb + f) + toff] : 0.0f;
    }
}

// One output tile.  `smem` holds NST stages of BK*(BM+BN) floats (the caller's ONE LDS array).
template <int MT, int NT, int EPI>
__device__ __forceinline__ void conv_block(const ConvArgs& a, float* const smem, const int bid, const BlockGrid g, const int ks) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int A_F4 = (BK * BM / 4) / 256;  // float4 per thread, weight tile
    constexpr int B_F4 = (BK * BN / 4) / 256;  // float4 per thread, activation tile
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small");
    static_assert(EPI != VQW_EPI_GATE || MT == 2, "gate epilogue pairs two M tiles");
    constexpr bool GATE = (EPI == VQW_EPI_GATE);

    // stage-major LDS: stage s = weight image [BK][BM], then activation image [BK][BN].  The register pipeline
    // uses stages 0-1, the LDS-DMA pipeline all three.
    constexpr int STG = BK * (BM + BN);

    const vqw_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;

    // block -> (m tile, n tile, batch); m fastest so co-scheduled blocks share the
    // activation panel in one XCD's L2.
    int L = vqw_xcd_remap(bid, g.nwg);
    const int mt = L % g.n_mt;
    L /= g.n_mt;
    const int nt = L % g.n_nt;
    const int b = L / g.n_nt;
    const int t0 = g.t_begin + nt * BN;
    const int o0 = GATE ? mt * (BM / 2) : mt * BM;  // first output row / gated channel
    const int Ctot = d.C0 + d.C1;
    const int kchunks = Ctot / BK;

    // taps whose staged window lies entirely outside [0,T_in) contribute nothing: skip them
    unsigned act = 0;
    int nact = 0;
    {
        const int tl = min(t0 + BN, d.T_out) - 1;
        for (int j = 0; j < d.ntaps; ++j) {
            const int lo = d.in_stride * t0 + d.tap_shift[j];
            const int hi = d.in_stride * tl + d.tap_shift[j];
            if (hi >= 0 && lo < d.T_in) { act |= 1u << j; ++nact; }
        }
    }
    // split-K (linear STORE only): this block multiplies K-steps [s_begin, s_begin + nsteps) of the tile
    const int nsteps_all = nact * kchunks;
    const int s_begin = (a.ksplit > 1) ? (int)((long)ks * nsteps_all / a.ksplit) : 0;
    const int nsteps = (a.ksplit > 1) ? (int)((long)(ks + 1) * nsteps_all / a.ksplit) - s_begin : nsteps_all;
    auto first_pos = [&](int& tap, int& kc) {   // (tap, channel block) of K-step s_begin
        tap = -1;
        for (int i = 0; i <= s_begin / kchunks; ++i) {
            ++tap;
            while (tap < d.ntaps && !((act >> tap) & 1u)) ++tap;
        }
        kc = (s_begin % kchunks) * BK;
    };

    // Staging registers.  A "piece" is what one thread moves per K-step in one go: one float4
    // of activations, or one float4 of weights (GATE: the matching filter+gate float4 pair).
    constexpr int NA = GATE ? A_F4 / 2 : A_F4;   // weight pieces
    constexpr int NPIECE = NA + B_F4;
    static_assert(NPIECE <= BK / 2, "one staging piece per MFMA group");
    f32x4 ra[A_F4], rb[B_F4];
    const __amdgpu_buffer_rsrc_t rs0 = vqw_make_rsrc(d.x0 + (size_t)b * d.C0 * d.T_in, (unsigned)d.C0 * d.T_in * 4u);
    const __amdgpu_buffer_rsrc_t rs1 = vqw_make_rsrc(d.C1 ? d.x1 + (size_t)b * d.C1 * d.T_in : d.x0, (unsigned)(d.C1 ? d.C1 : 1) * d.T_in * 4u);

    auto next_tap = [&](int j) {
        ++j;
        while (j < d.ntaps && !((act >> j) & 1u)) ++j;
        return j;
    };

    // global -> registers, piece p of the tile (tap, kc).  Every decision is block-uniform.
    auto piece_load = [&](auto fast_tag, int p, int tap, int kc) {
        constexpr bool FAST = decltype(fast_tag)::value;
        if (p < NA) {  // ---- weights (columns beyond M are clamped, then zeroed: no divergent branch)
            const float* wt = d.w + (size_t)tap * d.w_tap_stride + (size_t)kc * d.ldw;
            const int idx = tid + p * 256;
            if constexpr (GATE) {
                const int kk = idx / (BM / 8), u2 = idx % (BM / 8);
                const int g = o0 + 4 * u2;
                const int gc = min(g, a.H - 4);
                const float* wr = wt + (size_t)kk * d.ldw;
                ra[2 * p] = *reinterpret_cast<const f32x4*>(wr + gc);
                ra[2 * p + 1] = *reinterpret_cast<const f32x4*>(wr + a.H + gc);
                if (g >= a.H) { ra[2 * p] = f32x4{0, 0, 0, 0}; ra[2 * p + 1] = f32x4{0, 0, 0, 0}; }
            } else {
                const int kk = idx / (BM / 4), u = idx % (BM / 4);
                const int o = o0 + 4 * u;
                const int oc = min(o, d.M - 4);
                ra[p] = *reinterpret_cast<const f32x4*>(wt + (size_t)kk * d.ldw + oc);
                if (o >= d.M) ra[p] = f32x4{0, 0, 0, 0};
            }
        } else {       // ---- activations (the dilated / strided window of this tap)
            const int i = p - NA;
            const int shift = d.tap_shift[tap];
            const float* xb = (kc < d.C0) ? d.x0 + ((size_t)b * d.C0 + kc) * d.T_in
                                          : d.x1 + ((size_t)b * d.C1 + (kc - d.C0)) * d.T_in;
            const int lo = d.in_stride * t0 + shift;
            const int idx = tid + i * 256;
            const int kk = idx / (BN / 4), u = idx % (BN / 4);
            if constexpr (FAST) {   // every active tap of this block is fully inside [0,T_in)
                // one batch element per descriptor (< 4 GB); voffset = per-thread part, soffset = uniform part
                const bool first = kc < d.C0;
                const __amdgpu_buffer_rsrc_t rs = first ? rs0 : rs1;
                const int soff = ((first ? kc : kc - d.C0) * d.T_in + lo) * 4;
                if (d.in_stride == 1) {
                    rb[i] = vqw_buf_load4(rs, (kk * d.T_in + 4 * u) * 4, soff);
                } else {
                    const int vo = (kk * d.T_in + 8 * u) * 4;
                    const f32x4 v0 = vqw_buf_load4(rs, vo, soff);
                    const f32x4 v1 = vqw_buf_load4(rs, vo + 16, soff);
                    rb[i] = f32x4{v0[0], v0[2], v1[0], v1[2]};
                }
            } else {
                const float* row = xb + (size_t)kk * d.T_in;
                const int ti = lo + d.in_stride * 4 * u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tt = ti + d.in_stride * e;
                    rb[i][e] = (tt >= 0 && tt < d.T_in) ? row[tt] : 0.0f;
                }
            }
        }
    };

    // registers -> LDS, piece p
    auto piece_store = [&](int p, int buf) {
        if (p < NA) {
            float* Ab = smem + buf * STG;
            const int idx = tid + p * 256;
            if constexpr (GATE) {   // row image = [filter BM/2 | gate BM/2]
                const int kk = idx / (BM / 8), u2 = idx % (BM / 8);
                float* q = Ab + kk * BM + 4 * u2;
                *reinterpret_cast<f32x4*>(q) = ra[2 * p];
                *reinterpret_cast<f32x4*>(q + BM / 2) = ra[2 * p + 1];
            } else {
                const int kk = idx / (BM / 4), u = idx % (BM / 4);
                *reinterpret_cast<f32x4*>(Ab + kk * BM + 4 * u) = ra[p];
            }
        } else {
            const int i = p - NA;
            float* Bb = smem + buf * STG + BK * BM;
            const int idx = tid + i * 256;
            const int kk = idx / (BN / 4), u = idx % (BN / 4);
            f32x4 v = rb[i];
            if (d.in_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
            }
            *reinterpret_cast<f32x4*>(Bb + kk * BN + 4 * u) = v;
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int e = 0; e < MT; ++e)
#pragma unroll
        for (int f = 0; f < NT; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.0f;

    // one k-step (2 channels) of operand fragments: lanes 0-31 take channel 2ks, lanes 32-63 2ks+1
    auto read_frags = [&](const float* Ab, const float* Bb, int ks, float (&av)[MT], float (&bv)[NT]) {
        const int kk = 2 * ks + lhi;
        if constexpr (GATE) {           // filter row and gate row of the same channel: one ds_read2_b32
            av[0] = Ab[kk * BM];
            av[1] = Ab[kk * BM + BM / 2];
        } else if constexpr (MT == 2) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(Ab + kk * BM);
            av[0] = t[0]; av[1] = t[1];
        } else {
            av[0] = Ab[kk * BM];
        }
        if constexpr (NT == 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(Bb + kk * BN);
            bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
        } else if constexpr (NT == 2) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(Bb + kk * BN);
            bv[0] = t[0]; bv[1] = t[1];
        } else {
            bv[0] = Bb[kk * BN];
        }
    };
    auto mma = [&](const float (&av)[MT], const float (&bv)[NT]) {
#pragma unroll
        for (int e = 0; e < MT; ++e)
#pragma unroll
            for (int f = 0; f < NT; ++f)
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[f], acc[e][f], 0, 0, 0);
    };

    // Software pipeline.  K-step s computes from LDS buffer s&1.  Its BK/2 MFMA groups are each
    // followed by ONE staging piece: tile s+1 goes registers -> LDS (other buffer) and the
    // same registers are immediately re-loaded with tile s+2, so a wave never has a long
    // MFMA-free stretch, every global load has a whole K-step to land and the LDS writes
    // retire under the following MFMAs.  (Measured: staging in one lump cost ~20 % of the MFMA
    // rate because co-resident workgroups run in lockstep and all hit the lump together.)
    // The barrier is a raw s_barrier behind lgkmcnt(0) only: __syncthreads() would also drain
    // vmcnt, i.e. wait for the loads that were just issued.  The K loop exists twice: a
    // straight-line version for blocks whose active taps are all interior and a generic one
    // for the few blocks that touch the left/right edge of the signal.
    auto k_loop = [&](auto fast_tag) {
        int ld_tap, ld_kc;
        first_pos(ld_tap, ld_kc);
        auto advance = [&]() {
            ld_kc += BK;
            if (ld_kc == Ctot) { ld_kc = 0; ld_tap = next_tap(ld_tap); }
        };
        if (nsteps > 0) {
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) piece_load(fast_tag, p, ld_tap, ld_kc);
            advance();
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) piece_store(p, 0);
            if (nsteps > 1) {
#pragma unroll
                for (int p = 0; p < NPIECE; ++p) piece_load(fast_tag, p, ld_tap, ld_kc);
                advance();
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        // MODE 2: store tile s+1 and load tile s+2 (steady state, no conditionals around the
        // loads so the compiler can count vmcnt exactly); 1: store only; 0: last K-step.
        auto kstep = [&](auto mode_tag, int s) {
            constexpr int MODE = decltype(mode_tag)::value;
            const int buf = s & 1;
            const float* Ab = smem + buf * STG + (GATE ? wm * 32 + l31 : wm * (MT * 32) + MT * l31);
            const float* Bb = smem + buf * STG + BK * BM + wn * (NT * 32) + NT * l31;
            // fragments of k-step ks+1 are fetched from LDS while the MFMAs of k-step ks run;
            // sched_barrier pins that order (hipcc otherwise sinks the reads next to their use)
            float a0[MT], b0[NT], a1[MT], b1[NT];
            read_frags(Ab, Bb, 0, a0, b0);
#pragma unroll
            for (int ks = 0; ks < BK / 2; ks += 2) {
                read_frags(Ab, Bb, ks + 1, a1, b1);
                SCHED_FENCE();
                MMA_PRIO(1);
                mma(a0, b0);
                MMA_PRIO(0);
                SCHED_FENCE();
                if constexpr (MODE >= 1) {
                    if (ks < NPIECE) {
#ifndef VQW_ABL_NOSTORE
                        piece_store(ks, buf ^ 1);
#endif
                        SCHED_FENCE();  // keep the re-load behind the store: same registers, no copy
#ifndef VQW_ABL_NOLOAD
                        if constexpr (MODE == 2) piece_load(fast_tag, ks, ld_tap, ld_kc);
#endif
                    }
                }
                if (ks + 2 < BK / 2) read_frags(Ab, Bb, ks + 2, a0, b0);
                SCHED_FENCE();
                MMA_PRIO(1);
                mma(a1, b1);
                MMA_PRIO(0);
                SCHED_FENCE();
                if constexpr (MODE >= 1) {
                    if (ks + 1 < NPIECE) {
#ifndef VQW_ABL_NOSTORE
                        piece_store(ks + 1, buf ^ 1);
#endif
                        SCHED_FENCE();
#ifndef VQW_ABL_NOLOAD
                        if constexpr (MODE == 2) piece_load(fast_tag, ks + 1, ld_tap, ld_kc);
#endif
                    }
                }
                SCHED_FENCE();
            }
            if constexpr (MODE == 2) advance();
            __builtin_amdgcn_s_waitcnt(0xC07F);  // own LDS writes done (lgkmcnt(0)); loads stay in flight
            __builtin_amdgcn_s_barrier();
        };
        int s = 0;
        for (; s + 2 < nsteps; ++s) kstep(std::integral_constant<int, 2>{}, s);
        if (s + 1 < nsteps) { kstep(std::integral_constant<int, 1>{}, s); ++s; }
        if (s < nsteps) kstep(std::integral_constant<int, 0>{}, s);
    };
    // ---- LDS-DMA pipeline.  PMC on the register pipeline above (profiles/round1_gate_conv_pmc_sq.txt): MFMA busy
    // 70 % with the waves parked at the vmcnt wait in front of their ds_writes -- one K-step (2048 MFMA cycles per
    // wave, shared three ways) is not enough for a global load to land.  Here tile s+2 travels global -> LDS by
    // buffer_load_dwordx4 ... lds (no VGPRs, no ds_write) into a third stage while tile s is multiplied and tile
    // s+1 is already resident: two K-steps of latency budget.  Every wave issues its own 1-KiB pieces, one behind
    // each MFMA group; a counted vmcnt (the next tile's pieces stay in flight) and a raw s_barrier retire tile s+1
    // one phase before it is read.  Used by interior blocks of unit-stride convolutions with a full M tile.
    auto k_loop_dma = [&]() {
        constexpr int NDMA = A_F4 + B_F4;       // 1-KiB pieces per wave and K-step
        static_assert(NDMA <= BK / 2, "one DMA piece per MFMA group");
        constexpr int WAIT_NEXT = (NDMA & 0xF) | (0x7 << 4) | (0xF << 8) | ((NDMA >> 4) << 14);   // vmcnt(NDMA) only
        constexpr int WAIT_ALL = 0 | (0x7 << 4) | (0xF << 8);                                     // vmcnt(0) only
        const long wfloats = (long)(d.ntaps - 1) * d.w_tap_stride + (long)Ctot * d.ldw;
        const __amdgpu_buffer_rsrc_t rsw = vqw_make_rsrc(d.w, (unsigned)(wfloats * 4));
        int va[A_F4], vb[B_F4];
#pragma unroll
        for (int q = 0; q < A_F4; ++q) {
            const int f = (wid * A_F4 + q) * 256 + 4 * lane;
            const int kk = f / BM, m = f % BM;
            const int col = GATE ? ((m < BM / 2) ? o0 + m : a.H + o0 + (m - BM / 2)) : o0 + m;
            va[q] = (kk * d.ldw + col) * 4;
        }
#pragma unroll
        for (int q = 0; q < B_F4; ++q) {
            const int f = (wid * B_F4 + q) * 256 + 4 * lane;
            vb[q] = ((f / BN) * d.T_in + (f % BN)) * 4;
        }
        int ld_tap, ld_kc;
        first_pos(ld_tap, ld_kc);
        auto advance = [&]() {
            ld_kc += BK;
            if (ld_kc == Ctot) { ld_kc = 0; ld_tap = next_tap(ld_tap); }
        };
        auto dma = [&](int q, int st) {   // piece q of tile (ld_tap, ld_kc) -> stage st
#ifdef VQW_ABL_NODMA
            return;   // timing ablation only: operands never staged, wrong results
#endif
            if (q < A_F4) {
                const int soff = (int)(((long)ld_tap * d.w_tap_stride + (long)ld_kc * d.ldw) * 4);
                vqw_buf_load_lds16(rsw, smem + st * STG + (wid * A_F4 + q) * 256, va[q], soff);
            } else {
                const int i = q - A_F4;
                const bool first = ld_kc < d.C0;
                const int soff = ((first ? ld_kc : ld_kc - d.C0) * d.T_in + t0 + d.tap_shift[ld_tap]) * 4;
                vqw_buf_load_lds16(first ? rs0 : rs1, smem + st * STG + BK * BM + (wid * B_F4 + i) * 256, vb[i], soff);
            }
        };
#pragma unroll
        for (int q = 0; q < NDMA; ++q) dma(q, 0);
        advance();
        if (NST_DMA == 3 && nsteps > 1) {
#pragma unroll
            for (int q = 0; q < NDMA; ++q) dma(q, 1);
            advance();
            __builtin_amdgcn_s_waitcnt(WAIT_NEXT);
        } else {
            __builtin_amdgcn_s_waitcnt(WAIT_ALL);
        }
        __builtin_amdgcn_s_barrier();
        // MODE 1: tile s+2 is issued into stage nxt and stays in flight across the barrier; 0: nothing left to issue
        auto kstep = [&](auto mode_tag, int cur, int nxt) {
            constexpr int MODE = decltype(mode_tag)::value;
            const float* Ab = smem + cur * STG + (GATE ? wm * 32 + l31 : wm * (MT * 32) + MT * l31);
            const float* Bb = smem + cur * STG + BK * BM + wn * (NT * 32) + NT * l31;
            float a0[MT], b0[NT], a1[MT], b1[NT];
            read_frags(Ab, Bb, 0, a0, b0);
#pragma unroll
            for (int ks = 0; ks < BK / 2; ks += 2) {
                read_frags(Ab, Bb, ks + 1, a1, b1);
                SCHED_FENCE();
                MMA_PRIO(1);
                mma(a0, b0);
                MMA_PRIO(0);
                SCHED_FENCE();
                if constexpr (MODE == 1) {
                    if (ks < NDMA) dma(ks, nxt);
                }
                if (ks + 2 < BK / 2) read_frags(Ab, Bb, ks + 2, a0, b0);
                SCHED_FENCE();
                MMA_PRIO(1);
                mma(a1, b1);
                MMA_PRIO(0);
                SCHED_FENCE();
                if constexpr (MODE == 1) {
                    if (ks + 1 < NDMA) dma(ks + 1, nxt);
                }
                SCHED_FENCE();
            }
            if constexpr (MODE == 1) {
                advance();
                if constexpr (NST_DMA == 3) __builtin_amdgcn_s_waitcnt(WAIT_NEXT);   // my pieces of tile s+1 have landed; tile s+2 stays in flight
                else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            } else {
                __builtin_amdgcn_s_waitcnt(WAIT_ALL);
            }
            __builtin_amdgcn_s_barrier();                // everybody's pieces of tile s+1 have landed; stage cur is free
        };
        int s = 0, cur = 0;
        if constexpr (NST_DMA == 3) {
            for (; s + 2 < nsteps; ++s) {
                const int nxt = (cur == 0) ? 2 : cur - 1;    // (cur + 2) % 3
                kstep(std::integral_constant<int, 1>{}, cur, nxt);
                cur = (cur == 2) ? 0 : cur + 1;
            }
            for (; s < nsteps; ++s) {
                kstep(std::integral_constant<int, 0>{}, cur, 0);
                cur = (cur == 2) ? 0 : cur + 1;
            }
        } else {   // two stages: tile s+1 travels while tile s is multiplied, nothing in flight across the barrier
            for (; s + 1 < nsteps; ++s) {
                kstep(std::integral_constant<int, 1>{}, cur, cur ^ 1);
                cur ^= 1;
            }
            if (s < nsteps) kstep(std::integral_constant<int, 0>{}, cur, 0);
        }
    };
    bool all_interior = true;
    for (int j = 0; j < d.ntaps; ++j) {
        if (!((act >> j) & 1u)) continue;
        const int lo = d.in_stride * t0 + d.tap_shift[j];
        const int hi = d.in_stride * (t0 + BN - 1) + d.tap_shift[j] + (d.in_stride - 1);  // last element touched
        all_interior = all_interior && (lo >= 0) && (hi < d.T_in);
    }
    const bool full_m = GATE ? (o0 + BM / 2 <= a.H) : (o0 + BM <= d.M);
    if (all_interior && d.in_stride == 1 && !d.in_relu && full_m && nsteps > 0 && a.use_dma) k_loop_dma();
    else if (all_interior) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    // ------------------------------------------------------------------ epilogue
    const int tb = t0 + wn * (NT * 32) + NT * l31;  // first of this lane's NT output times
    if (tb >= d.T_out) return;
#ifdef VQW_ABL_NOEPI
    if (acc[0][0][0] != 12345.678f) return;   // timing ablation only: no epilogue, wrong results
#endif
    const int Ts = d.T_store;
    const int tstr = d.out_tstride, toff = d.out_toffset, vok = a.vec_ok;

    int tz[NT];
#pragma unroll
    for (int f = 0; f < NT; ++f) tz[f] = (d.cond_T > 0) ? min(tb + f, d.T_out - 1) / a.ratio : 0;

    float addf[4][NT], addg[4][NT];   // GATE: bias + condition of the current group of four rows
    float oldv[4][MT][NT], oldb[4][MT];   // ACCUM_SPLIT: old output values + bias of the group (GATE_BWD: saved tanh)
    float oldw[4][MT][NT];                // GATE_BWD: saved sigmoid
    (void)addf; (void)addg; (void)oldv; (void)oldb; (void)oldw;
#pragma unroll
    for (int rho = 0; rho < 16; ++rho) {
        const int r = (rho & 3) + 8 * (rho >> 2) + 4 * lhi;  // row inside the 32x32 tile
        if constexpr (EPI == VQW_EPI_GATE) {
            // bias and condition of FOUR rows are requested together, without a branch in between (absent operands
            // read a valid dummy address and are masked by a select): with per-row conditionals hipcc waited
            // vmcnt(0) three times per row, 48 dependent round trips per tile
            if ((rho & 3) == 0) {
                const bool hb = d.bias != nullptr, hc = d.cond_T > 0;
                const float* bp = hb ? d.bias : d.w;
                const float* cb = hc ? d.cond + (size_t)b * d.cond_bstride : d.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int gi = min(o0 + wm * 32 + r + i, a.H - 1);
                    const float bf = bp[hb ? gi : 0], bg = bp[hb ? a.H + gi : 0];
#pragma unroll
                    for (int f = 0; f < NT; ++f) {
                        const float cf = cb[hc ? (size_t)gi * d.cond_T + tz[f] : 0];
                        const float cg = cb[hc ? (size_t)(a.H + gi) * d.cond_T + tz[f] : 0];
                        addf[i][f] = (hb ? bf : 0.0f) + (hc ? cf : 0.0f);
                        addg[i][f] = (hb ? bg : 0.0f) + (hc ? cg : 0.0f);
                    }
                }
            }
            const int g = o0 + wm * 32 + r;  // gated channel
            if (g < a.H) {
                float og[NT], ot[NT], os[NT];
#pragma unroll
                for (int f = 0; f < NT; ++f) {
                    ot[f] = tanh_f(acc[0][f][rho] + addf[rho & 3][f]);
                    os[f] = sigmoid_f(acc[1][f][rho] + addg[rho & 3][f]);
                    og[f] = ot[f] * os[f];
                }
                const size_t ro = ((size_t)b * a.H + g) * Ts;
                store_row<NT>(d.out0 + ro, tb, d.T_out, tstr, toff, vok, og);
                if (d.save0) store_row<NT>(d.save0 + ro, tb, d.T_out, tstr, toff, vok, ot);
                if (d.save1) store_row<NT>(d.save1 + ro, tb, d.T_out, tstr, toff, vok, os);
            }
        } else {
            if constexpr (EPI == VQW_EPI_STORE) {
                if ((rho & 3) == 0 && a.ksplit <= 1) {   // bias, condition, BN affine of four rows at once, branch-free
                    const bool hb = d.bias != nullptr, hc = d.cond_T > 0, hs = d.scale != nullptr;
                    const float* bp = hb ? d.bias : d.w;
                    const float* cb = hc ? d.cond + (size_t)b * d.cond_bstride : d.w;
                    const float* sp = hs ? d.scale : d.w;
                    const float* hp = hs ? d.shift : d.w;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < MT; ++e) {
                            const int rw = min(o0 + wm * (MT * 32) + MT * (r + i) + e, d.M - 1);
                            const float bv = bp[hb ? rw : 0];
                            oldb[i][e] = hb ? bv : 0.0f;
#pragma unroll
                            for (int f = 0; f < NT; ++f) {
                                const float cv = cb[hc ? (size_t)rw * d.cond_T + tz[f] : 0];
                                oldv[i][e][f] = hc ? cv : 0.0f;
                            }
                            oldw[i][e][0] = sp[hs ? rw : 0];
                            if (NT > 1) oldw[i][e][1] = hp[hs ? rw : 0];
                        }
                }
            }
            if constexpr (EPI == VQW_EPI_GATE_BWD) {
                if ((rho & 3) == 0) {   // saved tanh / sigmoid of four rows at once
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < MT; ++e) {
                            const int rw = min(o0 + wm * (MT * 32) + MT * (r + i) + e, d.M - 1);
                            const size_t ri = ((size_t)b * a.H + rw) * Ts;
                            load_row<NT>(d.aux0 + ri, tb, d.T_out, tstr, toff, vok, oldv[i][e]);
                            load_row<NT>(d.aux1 + ri, tb, d.T_out, tstr, toff, vok, oldw[i][e]);
                        }
                }
            }
            if constexpr (EPI == VQW_EPI_ACCUM_SPLIT) {
                // the old values and biases of FOUR rows are requested together (the loads of a row otherwise wait
                // behind the previous row's store to the same tensor)
                if ((rho & 3) == 0) {
                    const bool hb = d.bias != nullptr;
                    const float* bp = hb ? d.bias : d.w;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < MT; ++e) {
                            const int rw = min(o0 + wm * (MT * 32) + MT * (r + i) + e, d.M - 1);
                            const float* src = (rw < d.M0) ? d.out0 + ((size_t)b * d.M0 + rw) * Ts
                                                           : d.aux1 + ((size_t)b * (d.M - d.M0) + (rw - d.M0)) * Ts;
                            load_row<NT>(src, tb, d.T_out, tstr, toff, vok, oldv[i][e]);
                            const float bv = bp[hb ? rw : 0];
                            oldb[i][e] = hb ? bv : 0.0f;
                        }
                }
            }
#pragma unroll
            for (int e = 0; e < MT; ++e) {
                const int row = o0 + wm * (MT * 32) + MT * r + e;
                if (row >= d.M) continue;
                float v[NT];
#pragma unroll
                for (int f = 0; f < NT; ++f) v[f] = acc[e][f][rho];

                if constexpr (EPI == VQW_EPI_STORE) {
                    if (a.ksplit > 1) {   // partial tile: the launcher zeroed out0 and vetted the epilogue as linear
                        const float bv0 = (ks == 0 && d.bias) ? d.bias[row] : 0.0f;
                        float* dst = d.out0 + ((size_t)b * d.M + row) * Ts;
#pragma unroll
                        for (int f = 0; f < NT; ++f)
                            if (tb + f < d.T_out) unsafeAtomicAdd(dst + tstr * (tb + f) + toff, v[f] + bv0);
                        continue;
                    }
#pragma unroll
                    for (int f = 0; f < NT; ++f) {
                        v[f] = (v[f] + oldb[rho & 3][e]) + oldv[rho & 3][e][f];     // + bias, + condition
                        if (d.out_relu) v[f] = fmaxf(v[f], 0.0f);
                    }
                    if (d.save0)
                        store_row<NT>(d.save0 + ((size_t)b * d.M + row) * Ts, tb, d.T_out, tstr,
                                      toff, vok, v);
                    if (d.scale) {
                        const float sc = oldw[rho & 3][e][0], sh = oldw[rho & 3][e][NT > 1 ? 1 : 0];
#pragma unroll
                        for (int f = 0; f < NT; ++f) v[f] = sc * v[f] + (NT > 1 ? sh : d.shift[row]);
                    }
                    if (d.aux1) {   // residual added after the activation: net = act(conv) + net
                        float res[NT];
                        load_row<NT>(d.aux1 + ((size_t)b * d.M + row) * Ts, tb, d.T_out, tstr, toff, vok, res);
#pragma unroll
                        for (int f = 0; f < NT; ++f) v[f] += res[f];
                    }
                    float* dst = (row < d.M0)
                                     ? d.out0 + ((size_t)b * d.M0 + row) * Ts
                                     : d.out1 + ((size_t)b * (d.M - d.M0) + (row - d.M0)) * Ts;
                    store_row<NT>(dst, tb, d.T_out, tstr, toff, vok, v);
                } else if constexpr (EPI == VQW_EPI_ACCUM_SPLIT) {
                    float* dst = (row < d.M0) ? d.out0 + ((size_t)b * d.M0 + row) * Ts
                                              : d.out1 + ((size_t)b * (d.M - d.M0) + (row - d.M0)) * Ts;
#pragma unroll
                    for (int f = 0; f < NT; ++f) v[f] = oldv[rho & 3][e][f] + (v[f] + oldb[rho & 3][e]);
                    store_row<NT>(dst, tb, d.T_out, tstr, toff, vok, v);
                } else if constexpr (EPI == VQW_EPI_GATE_BWD) {
                    float o1[NT], o2[NT];
#pragma unroll
                    for (int f = 0; f < NT; ++f) {
                        const float th = oldv[rho & 3][e][f], sg = oldw[rho & 3][e][f];
                        o1[f] = v[f] * sg * (1.0f - th * th);
                        o2[f] = v[f] * th * sg * (1.0f - sg);
                    }
                    store_row<NT>(d.out0 + ((size_t)b * 2 * a.H + row) * Ts, tb, d.T_out, tstr,
                                  toff, vok, o1);
                    store_row<NT>(d.out0 + ((size_t)b * 2 * a.H + a.H + row) * Ts, tb, d.T_out,
                                  tstr, toff, vok, o2);
                } else {  // VQW_EPI_MASK
                    const size_t ro = ((size_t)b * d.M + row) * Ts;
                    float m[NT];
                    load_row<NT>(d.aux0 + ro, tb, d.T_out, tstr, toff, vok, m);
                    const float sc = d.scale ? d.scale[row] : 1.0f;
#pragma unroll
                    for (int f = 0; f < NT; ++f) v[f] = (m[f] > 0.0f) ? v[f] * sc : 0.0f;
                    store_row<NT>(d.out0 + ro, tb, d.T_out, tstr, toff, vok, v);
                }
            }
        }
    }
}

// The launch: `main` tiles first, then (optionally) the last columns of every row as tiles half as wide, so that
// the chip's last round is made of small blocks that fill in behind the big ones instead of a thin round of big
// blocks (B*T = 13*2^12 makes every tiling a multiple of 13 blocks).  ONE LDS array, sized for the main tile
// (a second __shared__ object beside LDS-DMA traffic makes hipcc drain vmcnt before every ds_read).
template <int MT, int NT, int EPI>
__global__ __launch_bounds__(256, (MT * NT >= 8) ? 2 : ((MT * NT >= 4) ? (NST_DMA == 2 ? 4 : 3) : 4)) void conv_gemm_kernel(const ConvArgs a) {
    constexpr int NST = NST_DMA;
    __shared__ __attribute__((aligned(16))) float smem[NST * BK * 64 * (MT + NT)];
    int bid = blockIdx.x, ks = 0;
    if (a.ksplit > 1) {
        const int ntiles = a.main.nwg + a.tail.nwg;
        ks = bid / ntiles;
        bid -= ks * ntiles;
    }
    if (bid < a.main.nwg) {
        conv_block<MT, NT, EPI>(a, smem, bid, a.main, ks);
    } else {
        if constexpr (NT > 1) conv_block<MT, NT / 2, EPI>(a, smem, bid - a.main.nwg, a.tail, ks);
    }
}

template <int MT, int NT>
int launch_cfg(ConvArgs& a, hipStream_t st, int tail_nt) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    const vqw_conv_desc& d = a.d;
    const bool gate = d.epilogue == VQW_EPI_GATE;
    const int n_mt = gate ? vqw_cdiv(a.H, BM / 2) : vqw_cdiv(d.M, BM);
    const int n_nt = vqw_cdiv(d.T_out, BN);
    if (NT == 1 || tail_nt <= 0 || tail_nt >= n_nt) tail_nt = 0;
    a.main = BlockGrid{n_mt, n_nt - tail_nt, n_mt * (n_nt - tail_nt) * d.B, 0};
    a.tail = BlockGrid{n_mt, 0, 0, 0};
    if (tail_nt > 0) {
        const int t_main = (n_nt - tail_nt) * BN;
        a.tail.n_nt = vqw_cdiv(d.T_out - t_main, BN / 2);
        a.tail.nwg = n_mt * a.tail.n_nt * d.B;
        a.tail.t_begin = t_main;
    }
    dim3 grid((a.main.nwg + a.tail.nwg) * (a.ksplit > 1 ? a.ksplit : 1)), block(256);
    switch (d.epilogue) {
        case VQW_EPI_STORE:
            hipLaunchKernelGGL((conv_gemm_kernel<MT, NT, VQW_EPI_STORE>), grid, block, 0, st, a);
            break;
        case VQW_EPI_ACCUM_SPLIT:
            hipLaunchKernelGGL((conv_gemm_kernel<MT, NT, VQW_EPI_ACCUM_SPLIT>), grid, block, 0, st, a);
            break;
        case VQW_EPI_GATE:
            if constexpr (MT == 2) {
                hipLaunchKernelGGL((conv_gemm_kernel<2, NT, VQW_EPI_GATE>), grid, block, 0, st, a);
            } else {
                return vqw_set_error("vqw_conv_gemm: GATE needs an MT=2 tile");
            }
            break;
        case VQW_EPI_GATE_BWD:
            hipLaunchKernelGGL((conv_gemm_kernel<MT, NT, VQW_EPI_GATE_BWD>), grid, block, 0, st, a);
            break;
        case VQW_EPI_MASK:
            hipLaunchKernelGGL((conv_gemm_kernel<MT, NT, VQW_EPI_MASK>), grid, block, 0, st, a);
            break;
        default:
            return vqw_set_error("vqw_conv_gemm: unknown epilogue %d", d.epilogue);
    }
    VQW_LAUNCH_CHECK("vqw_conv_gemm");
    return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int vqw_conv_gemm(const vqw_conv_desc* dp, vqw_stream_t s) {
    VQW_CHECK(dp != nullptr, "vqw_conv_gemm: null descriptor");
    ConvArgs a;
    a.d = *dp;
    vqw_conv_desc& d = a.d;
    VQW_CHECK(d.B > 0 && d.T_out > 0 && d.T_in > 0 && d.M > 0, "vqw_conv_gemm: bad shape B=%d T_out=%d T_in=%d M=%d", d.B, d.T_out, d.T_in, d.M);
    VQW_CHECK(d.C0 > 0 && d.C0 % BK == 0 && d.C1 >= 0 && d.C1 % BK == 0, "vqw_conv_gemm: C0=%d / C1=%d must be multiples of %d", d.C0, d.C1, BK);
    VQW_CHECK(d.ntaps >= 1 && d.ntaps <= VQW_MAX_TAPS, "vqw_conv_gemm: ntaps=%d out of range", d.ntaps);
    VQW_CHECK(d.in_stride == 1 || d.in_stride == 2, "vqw_conv_gemm: in_stride must be 1 or 2");
    VQW_CHECK(d.M % 4 == 0 && d.ldw % 4 == 0 && d.ldw >= d.M, "vqw_conv_gemm: M=%d ldw=%d must be multiples of 4, ldw>=M", d.M, d.ldw);
    VQW_CHECK(d.x0 && d.w && d.out0, "vqw_conv_gemm: x0/w/out0 must not be null");
    VQW_CHECK(d.C1 == 0 || d.x1, "vqw_conv_gemm: C1>0 needs x1");
    VQW_CHECK(aligned16(d.w), "vqw_conv_gemm: w must be 16-byte aligned");
    if (d.out_tstride <= 0) d.out_tstride = 1;
    if (d.w_tap_stride <= 0) d.w_tap_stride = (int64_t)(d.C0 + d.C1) * d.ldw;
    if (d.T_store <= 0) d.T_store = d.out_tstride * d.T_out;
    a.H = 0;
    a.ratio = 1;
    if (d.cond_T > 0) {
        VQW_CHECK(d.cond != nullptr, "vqw_conv_gemm: cond_T>0 needs cond");
        VQW_CHECK(d.T_out % d.cond_T == 0, "vqw_conv_gemm: T_out=%d not a multiple of cond_T=%d", d.T_out, d.cond_T);
        a.ratio = d.T_out / d.cond_T;
    }
    switch (d.epilogue) {
        case VQW_EPI_GATE:
            VQW_CHECK(d.M % 8 == 0, "vqw_conv_gemm: GATE needs M=2H with H%%4==0");
            a.H = d.M / 2;
            break;
        case VQW_EPI_GATE_BWD:
            VQW_CHECK(d.aux0 && d.aux1, "vqw_conv_gemm: GATE_BWD needs aux0 (tanh) and aux1 (sigmoid)");
            a.H = d.M;
            break;
        case VQW_EPI_MASK:
            VQW_CHECK(d.aux0, "vqw_conv_gemm: MASK needs aux0");
            break;
        case VQW_EPI_ACCUM_SPLIT:
            if (d.M0 < 0) d.M0 = 0;
            if (d.M0 > d.M) d.M0 = d.M;
            VQW_CHECK(d.M0 == d.M || (d.out1 && d.aux1), "vqw_conv_gemm: ACCUM_SPLIT rows >= M0 need out1 and aux1");
            break;
        case VQW_EPI_STORE:
            if (d.M0 <= 0 || d.M0 > d.M || !d.out1) d.M0 = d.M;
            VQW_CHECK(!d.scale || d.shift, "vqw_conv_gemm: scale needs shift");
            break;
        default:
            return vqw_set_error("vqw_conv_gemm: unknown epilogue %d", d.epilogue);
    }
    a.vec_ok = (d.out_tstride == 1 && d.out_toffset == 0 && d.T_store % 4 == 0 &&
                aligned16(d.out0) && (!d.out1 || aligned16(d.out1)) &&
                (!d.aux0 || aligned16(d.aux0)) && (!d.aux1 || aligned16(d.aux1)) &&
                (!d.save0 || aligned16(d.save0)) && (!d.save1 || aligned16(d.save1)))
                   ? 1 : 0;

    a.use_dma = vqw_env_enabled("VQW_CONV_DMA");
    hipStream_t st = static_cast<hipStream_t>(s);
    const int cus = vqw_device_cus();
    const bool gate = d.epilogue == VQW_EPI_GATE;
    auto nblocks = [&](int t) {
        const int mt = t / 10, nt = t % 10;
        return (long)(gate ? vqw_cdiv(a.H, 32 * mt) : vqw_cdiv(d.M, 64 * mt)) * vqw_cdiv(d.T_out, 64 * nt) * d.B;
    };
    int tile = d.tile;
    if (tile == 0) {
        // the widest tile that still gives the chip a round and a half of blocks: the short layers of the encoder
        // (T_out down to 104) ran 48..192 blocks of 128x128 on 256 CUs at 13..43 TFLOP/s
        tile = 22;
        if (gate) {
            if (nblocks(22) * 2 < 3L * cus) tile = 21;
        } else {
            if (d.M <= 64) tile = 12;
            if (nblocks(tile) * 2 < 3L * cus) tile = 12;
            if (nblocks(tile) * 2 < 3L * cus) tile = 11;
        }
    }
    // split-K for a linear STORE whose tiles cannot fill the chip but whose K loop is long (the input gradient of
    // the 31 stacked condition projections: 128 x 104 outputs per batch row, K = 15 872: 8 blocks, 1.4 ms)
    a.ksplit = 1;
    bool caller_zeroed = false;
    {
        const bool linear = d.epilogue == VQW_EPI_STORE && !d.out_relu && !d.scale && d.cond_T == 0 && !d.save0 && !d.aux1 &&
                            d.M0 == d.M;
        const long ksteps = (long)d.ntaps * ((d.C0 + d.C1) / BK);
        int want = d.split_k < 0 ? -d.split_k : d.split_k;
        caller_zeroed = d.split_k < 0;
        if (want == 0 && linear && d.out_tstride == 1 && d.out_toffset == 0) {
            const long nb = nblocks(tile % 100);
            if (nb * 2 <= cus && ksteps >= 64) {
                want = (int)((2L * cus + nb - 1) / nb);
                if (want > ksteps / 16) want = (int)(ksteps / 16);
            }
        }
        if (want > 1) {
            VQW_CHECK(linear, "vqw_conv_gemm: split_k needs a plain STORE epilogue (bias only)");
            if (want > ksteps) want = (int)ksteps;
            a.ksplit = want;
        }
    }
    // A GEMM whose block count leaves a thin last round (B*T = 13*2^12 makes every tiling a multiple of 13
    // blocks; measured on the gate conv: two full rounds of 768 blocks, then 128 blocks on a mostly idle chip)
    // gives the last columns of every row to tiles half as wide, appended to the same grid: they start as the
    // big blocks of the last full round retire and spread over all CUs.  Explicit form of `tile`:
    // main + 10000*(main-tile columns given to the half-width tiles); VQW_CONV_TAIL=0 disables the auto choice.
    const int tail_env = vqw_env_enabled("VQW_CONV_TAIL");
    const int main_tile = tile % 100;
    int tail_nt = tile / 10000;
    const int mtm = main_tile / 10, ntm = main_tile % 10;
    if (mtm < 1 || mtm > 2 || (ntm != 1 && ntm != 2 && ntm != 4)) return vqw_set_error("vqw_conv_gemm: unsupported tile %d", tile);
    if (tile < 10000 && tail_env && ntm > 1) {
        const int occ = (main_tile == 22) ? (NST_DMA == 2 ? 4 : 3) : (main_tile == 24 || main_tile == 14) ? 2 : 4;
        const int n_mt = (d.epilogue == VQW_EPI_GATE) ? vqw_cdiv(a.H, 32 * mtm) : vqw_cdiv(d.M, 64 * mtm);
        const long slots = (long)cus * occ, per_col = (long)n_mt * d.B, nblk = per_col * vqw_cdiv(d.T_out, 64 * ntm);
        const long rem = nblk % slots;
        if (nblk > slots && rem > 0 && rem * 10 < slots * 7) tail_nt = (int)((rem + per_col - 1) / per_col);
    }
    if (a.ksplit > 1 && !caller_zeroed) {
        hipError_t e_ = hipMemsetAsync(d.out0, 0, (size_t)d.B * d.M * d.T_store * sizeof(float), st);
        if (e_ != hipSuccess) return vqw_set_error("vqw_conv_gemm: hipMemsetAsync failed: %s", hipGetErrorString(e_));
    }
    switch (main_tile) {
        case 24: return launch_cfg<2, 4>(a, st, tail_nt);
        case 22: return launch_cfg<2, 2>(a, st, tail_nt);
        case 21: return launch_cfg<2, 1>(a, st, 0);
        case 14: return launch_cfg<1, 4>(a, st, tail_nt);
        case 12: return launch_cfg<1, 2>(a, st, tail_nt);
        case 11: return launch_cfg<1, 1>(a, st, 0);
        default: return vqw_set_error("vqw_conv_gemm: unsupported tile %d", tile);
    }
}

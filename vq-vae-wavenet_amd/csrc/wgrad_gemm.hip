// Weight-gradient engine for gfx950 (MI355X), true-fp32 MFMA, split-K over (batch, time).
//
// Replaces TensorFlow autodiff's Conv2DBackpropFilter for every conv on the hot path
// (wavenet_ops.py:59-90 call sites, encoder.py:15-24):
//   dw[j][c][o] += sum_{b,t} p[b][c][p_stride*t + shift_j] * q[b][o][t]
// GEMM view: D[c][o] = sum_t P[c][t] Q[o][t] -- both operands have the reduction index t
// contiguous in memory, so the LDS images are row-major [row][32 t (+4 pad)] and each lane
// reads 4 consecutive t with one ds_read_b128: lanes 0-31 take t = 8s..8s+3, lanes 32-63
// t = 8s+4..8s+7, which feeds four v_mfma_f32_32x32x2_f32 (the k order inside a chunk is
// permuted identically for A and B, so the sum is unchanged).  Row stride 36 floats makes
// those reads bank-conflict-free.  Partial tiles are combined with global fp32 atomics
// (128-byte row segments per wave instruction).
#include <type_traits>

#include "vqw_common.h"

namespace {

constexpr int BT = 32;   // time steps per K-step
constexpr int LDT = 36;  // padded LDS row (floats)

struct WgradArgs {
    vqw_wgrad_desc d;
    int n_ct, n_ot, chunks, chunk_len, nwg;
};

__device__ __forceinline__ f32x4 wg_load4(const float* __restrict__ row, int ti, int T,
                                          int stride, int relu) {
    f32x4 v;
    if (stride == 1) {
        if (ti >= 0 && ti + 3 < T) {
            const F4U u = *reinterpret_cast<const F4U*>(row + ti);
            v = {u.x, u.y, u.z, u.w};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tt = ti + e;
                v[e] = (tt >= 0 && tt < T) ? row[tt] : 0.0f;
            }
        }
    } else {
        if (ti >= 0 && ti + 7 < T) {
            const F4U u0 = *reinterpret_cast<const F4U*>(row + ti);
            const F4U u1 = *reinterpret_cast<const F4U*>(row + ti + 4);
            v = {u0.x, u0.z, u1.x, u1.z};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tt = ti + stride * e;
                v[e] = (tt >= 0 && tt < T) ? row[tt] : 0.0f;
            }
        }
    }
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
    }
    return v;
}

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int BM = 64 * MT, BN = 64 * NT;  // dw tile: BM input channels x BN output channels
    constexpr int P_F4 = (BM * BT / 4) / 256;
    constexpr int Q_F4 = (BN * BT / 4) / 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDT];
    float* const Ps = smem;
    float* const Qs = smem + 2 * BM * LDT;

    const vqw_wgrad_desc& d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;

    int L = vqw_xcd_remap(blockIdx.x, a.nwg);
    const int ot = L % a.n_ot; L /= a.n_ot;
    const int ct = L % a.n_ct; L /= a.n_ct;
    const int tap = L % d.ntaps; L /= d.ntaps;
    const int chunk = L % a.chunks;
    const int b = L / a.chunks;
    const int c0 = ct * BM, o0 = ot * BN;
    const int Qtot = d.Q0 + d.Q1;
    const int tbeg = chunk * a.chunk_len;
    const int tend = min(tbeg + a.chunk_len, d.T_q);
    const int nsteps = (tend > tbeg) ? (tend - tbeg + BT - 1) / BT : 0;
    const int shift = d.tap_shift[tap];

    f32x4 rp[P_F4], rq[Q_F4];
    const __amdgpu_buffer_rsrc_t rsp = vqw_make_rsrc(d.p + (size_t)b * d.Cp * d.T_p, (unsigned)d.Cp * d.T_p * 4u);
    const __amdgpu_buffer_rsrc_t rsq0 = vqw_make_rsrc(d.q0 + (size_t)b * d.Q0 * d.T_q, (unsigned)d.Q0 * d.T_q * 4u);
    const __amdgpu_buffer_rsrc_t rsq1 = vqw_make_rsrc(d.Q1 ? d.q1 + (size_t)b * d.Q1 * d.T_q : d.q0, (unsigned)(d.Q1 ? d.Q1 : 1) * d.T_q * 4u);

    // Block-uniform classification: a block is FAST when its whole dw tile is inside
    // [0,Cp) x [0,Qtot), its time range is whole K-steps and every p window is inside [0,T_p).
    const int p_lo = d.p_stride * tbeg + shift;
    const int p_hi = d.p_stride * (tbeg + nsteps * BT - 1) + shift + (d.p_stride - 1);
    const bool fast = (c0 + BM <= d.Cp) && (o0 + BN <= Qtot) && (tbeg + nsteps * BT <= tend) &&
                      (p_lo >= 0) && (p_hi < d.T_p) && (o0 + BN <= d.Q0 || o0 >= d.Q0);

    // One staging piece = one float4 per thread: pieces [0,P_F4) belong to p, the rest to q.
    constexpr int NPIECE = P_F4 + Q_F4;
    static_assert(NPIECE <= BT / 2, "one staging piece per 4-MFMA group");
    auto piece_load = [&](auto fast_tag, int pc, int t0) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int i = pc < P_F4 ? pc : pc - P_F4;
        const int idx = tid + i * 256;
        const int row = idx / (BT / 4), tq = idx % (BT / 4);
        if (pc < P_F4) {
            if constexpr (FAST) {
                const int soff = (c0 * d.T_p + d.p_stride * t0 + shift) * 4;   // block-uniform part
                if (d.p_stride == 1) {
                    rp[i] = vqw_buf_load4(rsp, (row * d.T_p + 4 * tq) * 4, soff);
                } else {
                    const int vo = (row * d.T_p + 8 * tq) * 4;
                    const f32x4 v0 = vqw_buf_load4(rsp, vo, soff);
                    const f32x4 v1 = vqw_buf_load4(rsp, vo + 16, soff);
                    rp[i] = f32x4{v0[0], v0[2], v1[0], v1[2]};
                }
            } else {
                const int c = c0 + row;
                const int t = t0 + 4 * tq;
                if (c < d.Cp && t < tend) {
                    const float* rowp = d.p + ((size_t)b * d.Cp + c) * d.T_p;
                    rp[i] = wg_load4(rowp, d.p_stride * t + shift, d.T_p, d.p_stride, 0);
                } else {
                    rp[i] = f32x4{0, 0, 0, 0};
                }
            }
        } else {
            if constexpr (FAST) {
                const bool first = o0 < d.Q0;
                const int soff = ((first ? o0 : o0 - d.Q0) * d.T_q + t0) * 4;
                rq[i] = vqw_buf_load4(first ? rsq0 : rsq1, (row * d.T_q + 4 * tq) * 4, soff);
            } else {
                const int o = o0 + row;
                const int t = t0 + 4 * tq;
                f32x4 v = f32x4{0, 0, 0, 0};
                if (o < Qtot && t < tend) {
                    const float* rowp = (o < d.Q0) ? d.q0 + ((size_t)b * d.Q0 + o) * d.T_q
                                                   : d.q1 + ((size_t)b * d.Q1 + (o - d.Q0)) * d.T_q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (t + e < tend) ? rowp[t + e] : 0.0f;
                }
                rq[i] = v;
            }
        }
    };
    auto piece_store = [&](int pc, int buf) {
        const int i = pc < P_F4 ? pc : pc - P_F4;
        const int idx = tid + i * 256;
        const int off = (idx / (BT / 4)) * LDT + 4 * (idx % (BT / 4));
        if (pc < P_F4) {
            f32x4 v = rp[i];
            if (d.p_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
            }
            *reinterpret_cast<f32x4*>(Ps + buf * BM * LDT + off) = v;
        } else {
            *reinterpret_cast<f32x4*>(Qs + buf * BN * LDT + off) = rq[i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int e = 0; e < MT; ++e)
#pragma unroll
        for (int f = 0; f < NT; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.0f;

    auto read_frags = [&](const float* Pb, const float* Qb, int kb, f32x4 (&av)[MT], f32x4 (&bv)[NT]) {
#pragma unroll
        for (int e = 0; e < MT; ++e) av[e] = *reinterpret_cast<const f32x4*>(Pb + e * 32 * LDT + kb * 8);
#pragma unroll
        for (int f = 0; f < NT; ++f) bv[f] = *reinterpret_cast<const f32x4*>(Qb + f * 32 * LDT + kb * 8);
    };
    auto mma_u = [&](const f32x4 (&av)[MT], const f32x4 (&bv)[NT], int u) {
#pragma unroll
        for (int e = 0; e < MT; ++e)
#pragma unroll
            for (int f = 0; f < NT; ++f)
                acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e][u], bv[f][u], acc[e][f], 0, 0, 0);
    };

    // Same software pipeline as the conv engine: after every group of MT*NT MFMAs one staging
    // piece moves tile s+1 registers -> LDS (other buffer) and re-loads the registers with
    // tile s+2; raw s_barrier behind lgkmcnt(0) only.
    auto k_loop = [&](auto fast_tag) {
        if (nsteps > 0) {
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) piece_load(fast_tag, pc, tbeg);
#pragma unroll
            for (int pc = 0; pc < NPIECE; ++pc) piece_store(pc, 0);
            if (nsteps > 1) {
#pragma unroll
                for (int pc = 0; pc < NPIECE; ++pc) piece_load(fast_tag, pc, tbeg + BT);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        auto kstep = [&](auto mode_tag, int s) {
            constexpr int MODE = decltype(mode_tag)::value;   // 2: store+load, 1: store, 0: none
            const int buf = s & 1;
            const int t2 = tbeg + (s + 2) * BT;
            const float* Pb = Ps + buf * BM * LDT + (wm * MT * 32 + l31) * LDT + 4 * lhi;
            const float* Qb = Qs + buf * BN * LDT + (wn * NT * 32 + l31) * LDT + 4 * lhi;
            f32x4 a0[MT], b0[NT], a1[MT], b1[NT];
            read_frags(Pb, Qb, 0, a0, b0);
#pragma unroll
            for (int kb = 0; kb < BT / 8; kb += 2) {
                read_frags(Pb, Qb, kb + 1, a1, b1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
                    mma_u(a0, b0, u);
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = kb * 4 + u;          // 4-MFMA group index inside the K-step
                    if constexpr (MODE >= 1) {
                        if ((g & 1) == 0 && (g >> 1) < NPIECE) {
                            piece_store(g >> 1, buf ^ 1);
                            __builtin_amdgcn_sched_barrier(0);  // re-load behind the store: same registers
                            if constexpr (MODE == 2) piece_load(fast_tag, g >> 1, t2);
                        }
                    }
                }
                if (kb + 2 < BT / 8) read_frags(Pb, Qb, kb + 2, a0, b0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    __builtin_amdgcn_sched_barrier(0);
                    mma_u(a1, b1, u);
                    __builtin_amdgcn_sched_barrier(0);
                    const int g = (kb + 1) * 4 + u;
                    if constexpr (MODE >= 1) {
                        if ((g & 1) == 0 && (g >> 1) < NPIECE) {
                            piece_store(g >> 1, buf ^ 1);
                            __builtin_amdgcn_sched_barrier(0);  // re-load behind the store: same registers
                            if constexpr (MODE == 2) piece_load(fast_tag, g >> 1, t2);
                        }
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        };
        int s = 0;
        for (; s + 2 < nsteps; ++s) kstep(std::integral_constant<int, 2>{}, s);
        if (s + 1 < nsteps) { kstep(std::integral_constant<int, 1>{}, s); ++s; }
        if (s < nsteps) kstep(std::integral_constant<int, 0>{}, s);
    };
    if (fast) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    if (nsteps == 0) return;

    float* dwt = d.dw + (size_t)tap * d.dw_tap_stride;
#pragma unroll
    for (int e = 0; e < MT; ++e)
#pragma unroll
        for (int f = 0; f < NT; ++f) {
            const int o = o0 + wn * NT * 32 + f * 32 + l31;
#pragma unroll
            for (int rho = 0; rho < 16; ++rho) {
                const int c = c0 + wm * MT * 32 + e * 32 + (rho & 3) + 8 * (rho >> 2) + 4 * lhi;
                if (c < d.Cp && o < Qtot) unsafeAtomicAdd(dwt + (size_t)c * d.lddw + o, acc[e][f][rho]);
            }
        }
}

}  // namespace

extern "C" int vqw_wgrad_gemm(const vqw_wgrad_desc* dp, vqw_stream_t s) {
    VQW_CHECK(dp != nullptr, "vqw_wgrad_gemm: null descriptor");
    WgradArgs a;
    a.d = *dp;
    vqw_wgrad_desc& d = a.d;
    VQW_CHECK(d.B > 0 && d.T_q > 0 && d.T_p > 0 && d.Cp > 0 && d.Q0 > 0 && d.Q1 >= 0, "vqw_wgrad_gemm: bad shape");
    VQW_CHECK(d.ntaps >= 1 && d.ntaps <= VQW_MAX_TAPS, "vqw_wgrad_gemm: ntaps=%d out of range", d.ntaps);
    VQW_CHECK(d.p_stride == 1 || d.p_stride == 2, "vqw_wgrad_gemm: p_stride must be 1 or 2");
    VQW_CHECK(d.p && d.q0 && d.dw && (d.Q1 == 0 || d.q1), "vqw_wgrad_gemm: null pointer");
    VQW_CHECK(d.lddw >= d.Q0 + d.Q1, "vqw_wgrad_gemm: lddw too small");
    constexpr int BM = 128, BN = 128;
    a.n_ct = vqw_cdiv(d.Cp, BM);
    a.n_ot = vqw_cdiv(d.Q0 + d.Q1, BN);
    const int tiles = a.n_ct * a.n_ot * d.ntaps;
    int chunks = d.splits;
    if (chunks <= 0) {
        // aim for ~6 blocks per CU over the whole chip (measured best on MI355X: bench_kernels.py)
        chunks = vqw_cdiv(1536, tiles * d.B);
        const int max_chunks = vqw_cdiv(d.T_q, 4 * BT);
        // ... but keep >= 24 K-steps per block while that still gives two blocks per CU: the 1x1 skip|residual
        // kernel gradient (12 tiles) ran 102 TFLOP/s with 16 chunks of 13 K-steps, 111 with 8 chunks of 26
        const int by_len = d.T_q / (24 * BT) > 1 ? d.T_q / (24 * BT) : 1;
        if (chunks > by_len) {
            chunks = by_len;
            if ((long)tiles * d.B * chunks < 512) chunks = vqw_cdiv(512, tiles * d.B);
        }
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
    }
    a.chunk_len = vqw_cdiv(vqw_cdiv(d.T_q, chunks), BT) * BT;
    a.chunks = vqw_cdiv(d.T_q, a.chunk_len);
    a.nwg = tiles * a.chunks * d.B;
    hipLaunchKernelGGL((wgrad_kernel<2, 2>), dim3(a.nwg), dim3(256), 0, static_cast<hipStream_t>(s), a);
    VQW_LAUNCH_CHECK("vqw_wgrad_gemm");
    return 0;
}

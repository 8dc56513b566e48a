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

    // Block-uniform classification: a block is FAST when its whole dw tile is inside
    // [0,Cp) x [0,Qtot), its time range is whole K-steps and every p window is inside [0,T_p).
    const int p_lo = d.p_stride * tbeg + shift;
    const int p_hi = d.p_stride * (tbeg + nsteps * BT - 1) + shift + (d.p_stride - 1);
    const bool fast = (c0 + BM <= d.Cp) && (o0 + BN <= Qtot) && (tbeg + nsteps * BT <= tend) &&
                      (p_lo >= 0) && (p_hi < d.T_p) && (o0 + BN <= d.Q0 || o0 >= d.Q0);

    auto load_tiles = [&](auto fast_tag, int t0) {
        constexpr bool FAST = decltype(fast_tag)::value;
        if constexpr (FAST) {
            const float* pb = d.p + ((size_t)b * d.Cp + c0) * d.T_p + (d.p_stride * t0 + shift);
            const float* qb = (o0 < d.Q0) ? d.q0 + ((size_t)b * d.Q0 + o0) * d.T_q + t0
                                          : d.q1 + ((size_t)b * d.Q1 + (o0 - d.Q0)) * d.T_q + t0;
#pragma unroll
            for (int i = 0; i < P_F4; ++i) {
                const int idx = tid + i * 256;
                const int row = idx / (BT / 4), tq = idx % (BT / 4);
                if (d.p_stride == 1) {
                    const F4U v = *reinterpret_cast<const F4U*>(pb + (size_t)row * d.T_p + 4 * tq);
                    rp[i] = f32x4{v.x, v.y, v.z, v.w};
                } else {
                    const float* pp = pb + (size_t)row * d.T_p + 8 * tq;
                    const F4U v0 = *reinterpret_cast<const F4U*>(pp);
                    const F4U v1 = *reinterpret_cast<const F4U*>(pp + 4);
                    rp[i] = f32x4{v0.x, v0.z, v1.x, v1.z};
                }
            }
#pragma unroll
            for (int i = 0; i < Q_F4; ++i) {
                const int idx = tid + i * 256;
                const int row = idx / (BT / 4), tq = idx % (BT / 4);
                const F4U v = *reinterpret_cast<const F4U*>(qb + (size_t)row * d.T_q + 4 * tq);
                rq[i] = f32x4{v.x, v.y, v.z, v.w};
            }
        } else {
#pragma unroll
            for (int i = 0; i < P_F4; ++i) {
                const int idx = tid + i * 256;
                const int row = idx / (BT / 4), tq = idx % (BT / 4);
                const int c = c0 + row;
                const int t = t0 + 4 * tq;
                if (c < d.Cp && t < tend) {
                    const float* rowp = d.p + ((size_t)b * d.Cp + c) * d.T_p;
                    rp[i] = wg_load4(rowp, d.p_stride * t + shift, d.T_p, d.p_stride, 0);
                } else {
                    rp[i] = f32x4{0, 0, 0, 0};
                }
            }
#pragma unroll
            for (int i = 0; i < Q_F4; ++i) {
                const int idx = tid + i * 256;
                const int row = idx / (BT / 4), tq = idx % (BT / 4);
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
    auto store_tiles = [&](int buf) {
        float* Pb = Ps + buf * BM * LDT;
        float* Qb = Qs + buf * BN * LDT;
#pragma unroll
        for (int i = 0; i < P_F4; ++i) {
            const int idx = tid + i * 256;
            f32x4 v = rp[i];
            if (d.p_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
            }
            *reinterpret_cast<f32x4*>(Pb + (idx / (BT / 4)) * LDT + 4 * (idx % (BT / 4))) = v;
        }
#pragma unroll
        for (int i = 0; i < Q_F4; ++i) {
            const int idx = tid + i * 256;
            *reinterpret_cast<f32x4*>(Qb + (idx / (BT / 4)) * LDT + 4 * (idx % (BT / 4))) = rq[i];
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
    auto mma = [&](const f32x4 (&av)[MT], const f32x4 (&bv)[NT]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < MT; ++e)
#pragma unroll
                for (int f = 0; f < NT; ++f)
                    acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e][u], bv[f][u], acc[e][f], 0, 0, 0);
    };

    // Same software pipeline as the conv engine: tile k+1 goes registers -> LDS in the middle of
    // K-step k, the loads of tile k+2 are issued right behind it; raw s_barrier + lgkmcnt(0).
    auto k_loop = [&](auto fast_tag) {
        if (nsteps > 0) {
            load_tiles(fast_tag, tbeg);
            store_tiles(0);
            if (nsteps > 1) load_tiles(fast_tag, tbeg + BT);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsteps; ++s) {
            const int buf = s & 1;
            const float* Pb = Ps + buf * BM * LDT + (wm * MT * 32 + l31) * LDT + 4 * lhi;
            const float* Qb = Qs + buf * BN * LDT + (wn * NT * 32 + l31) * LDT + 4 * lhi;
            f32x4 a0[MT], b0[NT], a1[MT], b1[NT];
            read_frags(Pb, Qb, 0, a0, b0);
#pragma unroll
            for (int kb = 0; kb < BT / 8; kb += 2) {
                read_frags(Pb, Qb, kb + 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mma(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (kb + 2 < BT / 8) read_frags(Pb, Qb, kb + 2, a0, b0);
                if (kb == 0 && s + 1 < nsteps) {
                    store_tiles(buf ^ 1);
                    if (s + 2 < nsteps) load_tiles(fast_tag, tbeg + (s + 2) * BT);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
        }
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
        // aim for ~3 resident blocks per CU over the whole chip
        chunks = vqw_cdiv(768, tiles * d.B);
        const int max_chunks = vqw_cdiv(d.T_q, 4 * BT);
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

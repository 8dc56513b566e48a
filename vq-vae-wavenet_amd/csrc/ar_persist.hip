// Persistent fast-generation kernel for gfx950: the whole sample loop of generate.py:103-113
// (216 tf.matmul + 91 FIFOQueue ops + numpy sampling per sample in the reference,
// wavenet.py:103-172 / wavenet_ops.py:147-267 / utils.py:13-46) in ONE launch.
//
// Decomposition: R/8 workgroups (one per CU), workgroup bi owns channels 8 bi .. 8 bi + 7 of
// every matrix of the model (plus the same columns of the wider ones), so a sample is a chain
// of 2L+3 phases, each = "gather the full input vector, multiply by my columns, publish my
// outputs".  The exchange follows the R2 recipe of the CDNA guide (Guideline 16): every value
// travels as ONE naturally aligned 8-byte granule {tag, fp32 bits} written with a relaxed
// agent-scope atomic store (sc1, write-through) and polled with relaxed agent-scope atomic
// loads; the tag is a per-phase sequence number, so no flags, fences or resets are needed and
// nothing depends on dispatch order or XCD placement.  The per-layer exchange buffer of the
// layer input doubles as the dilation queue (ring of (k-1)d+1 time slots).
//
// A workgroup has two roles (measured with tools/ar_trace.py: a wave that polls behind its own
// weight loads waits for them, vmcnt returns in order -- 6.5 of 8.1 us per layer were weight-load
// latency in the single-role version):
//   waves 0-3 "compute": keep the NEXT phase's weight columns in registers (16-byte loads from
//             a per-thread blocked copy, issued one phase ahead, served by L2 / Infinity
//             Cache while the exchange is in flight), multiply, reduce over a half wave,
//             publish.  They never poll global memory.
//   waves 4-7 "gather":  poll the granules of the phase (all loads of a phase in flight at
//             once), strip the tags into LDS, meet the compute waves at a raw s_barrier.
// Head weights (skip0, postprocess1/2, preprocess), all biases and the condition projections
// of the current frame live in LDS.  Every spin is bounded by a wall-clock timeout
// (s_memrealtime) that sets an error word and drains the grid.
#include <string.h>

#include <vector>

#include "ar_persist.h"

namespace {

typedef unsigned long long u64;
constexpr int PB = 8;        // max batch rows
constexpr int PJ = 8;        // max Q / R
constexpr int CPB = 8;       // channels per workgroup
constexpr int KSMAX = 4;     // taps of the dilated conv handled by the persistent kernel
constexpr int NCT = 256;     // compute threads (waves 0-3); gather threads are 256..511
constexpr unsigned long long TIMEOUT_TICKS = 300000000ull;  // 3 s of s_memrealtime (100 MHz)

struct PArgs {
    int L, ks, R, S, Q, B, nS, nQ, pre_k;
    int ngq, noq;                 // 16-byte groups per compute thread: gate / out weights of a layer
    const float* gw;              // [L][nwg][ngq][256][4]   element e = (tap*RL + i)*2 + {filter, gate}
    const float* ow;              // [L][nwg][noq][256][4]   element e = i*(nS+1) + col (skip cols, residual col)
    const float* headw;           // [nwg][nhead][256]       post1 | post2 | skip0 | preprocess
    const float* bias;            // [nwg][nbias]
    const int* dil;               // [L]
    const int* ring_off;          // [L] granule offset of each layer's ring
    u64* rings;                   // sum_l depth_l * B * R granules: layer inputs == dilation queues
    const float* const* cond;     // [L+1] this run's condition projections ([B][2R][Tz] ..., [B][S][Tz])
    u64 *ex_g, *ex_s, *ex_h, *ex_l;  // [B][R], [B][S], [B][S], [B][Q]
    float* xhist;                 // [B][pre_k] encoded input history (state across runs)
    float* prev;                  // [B] last decoded sample
    int* state;                   // [0] = step, [1] = error
    int Tz, ratio, mode, n_steps;
    const float* uniforms;
    float* audio;
    int* indices;
    float* probs_last;
#ifdef VQW_AR_TRACE
    u64* trace;                   // [gridDim][8 waves][16] accumulated s_memrealtime ticks (tools/ar_trace.py)
#endif
};

// LDS carve (in floats), shared by the host (size) and the device (offsets)
struct Carve {
    int xg, xo, hw, bias, condc, xh, misc, tab, total;
    int n_p1, n_p2, n_s0, n_pre, nhead, nbias, bias_head;
};
__host__ __device__ inline Carve make_carve(int L, int ks, int R, int S, int Q, int B, int nS, int nQ, int pre_k) {
    Carve c;
    const int RL = R / 32, SL = S / 32;
    c.n_p1 = SL * nS; c.n_p2 = SL * nQ; c.n_s0 = RL * nS; c.n_pre = (pre_k + 31) / 32;
    c.nhead = c.n_p1 + c.n_p2 + c.n_s0 + c.n_pre;
    c.bias_head = L * (nS + 3) * CPB;                    // per layer: gate {filter, gate}, out {skip cols, residual}
    c.nbias = c.bias_head + (2 * nS + nQ + 1) * CPB;     // skip0, post1, post2, preprocess
    int nx = ks * B * R;
    if (B * S > nx) nx = B * S;
    if (B * Q > nx) nx = B * Q;
    c.xg = 0;
    c.xo = c.xg + nx;
    c.hw = c.xo + B * S;
    c.bias = c.hw + c.nhead * NCT;
    c.condc = c.bias + ((c.nbias + 3) & ~3);
    c.xh = c.condc + (L * 2 + nS) * CPB * B;
    c.misc = c.xh + ((B * pre_k + 3) & ~3);
    c.tab = c.misc + 64;
    c.total = c.tab + 2 * L;
    return c;
}

__device__ __forceinline__ float p_mu_enc(float x) {
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    const float s = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    return s * log1pf(255.0f * fabsf(x)) / 5.5451774444795623f;
}
__device__ __forceinline__ float p_mu_dec(float idx) {
    const float y = 2.0f * idx / 255.0f - 1.0f;
    const float s = (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f);
    return s * (powf(256.0f, fabsf(y)) - 1.0f) / 255.0f;
}

__device__ __forceinline__ void publish(u64* p, unsigned tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// This wave's LDS traffic is done, then meet the other role.  Raw s_barrier: __syncthreads() would
// also drain vmcnt, i.e. make the compute waves wait for the weights they have just requested.
__device__ __forceinline__ void role_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Spin on one granule.  Two polls are kept in flight (returns are in order), so an arrival is seen one
// poll spacing after it becomes visible instead of up to a full round trip later.
__device__ __forceinline__ u64 poll_granule(int* fail, const u64* p, unsigned tag, u64 g) {
    if ((unsigned)(g >> 32) == tag) return g;
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    u64 g1 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(2);
        const u64 g2 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(g1 >> 32) == tag) return g1;
        g1 = g2;
        if ((spins & 255u) == 0 &&
            (__builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS || *reinterpret_cast<volatile int*>(fail))) {
            *fail = 1;
            return g1;
        }
    }
}

// Gather lanes: nseg segments of n granules each -> dst[s*n + i].  All first loads of a chunk are
// issued before the first tag is looked at; src[s] == nullptr reads as zeros (zero-filled queues).
template <int NSEG>
__device__ __forceinline__ void gather_segs(int gid, int* fail, const u64* const* src, const unsigned* tag,
                                            int nseg, int n, float* dst, int relu) {
    constexpr int CH = 4;
    for (int base = gid; base < n; base += NCT * CH) {
        u64 v[NSEG][CH];
#pragma unroll
        for (int s = 0; s < NSEG; ++s)
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                const int idx = base + NCT * m;
                v[s][m] = (s < nseg && src[s] && idx < n)
                              ? __hip_atomic_load(src[s] + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
#pragma unroll
        for (int s = 0; s < NSEG; ++s)
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                const int idx = base + NCT * m;
                if (s < nseg && idx < n) {
                    float x = 0.0f;
                    if (src[s]) x = __uint_as_float((unsigned)poll_granule(fail, src[s] + idx, tag[s], v[s][m]));
                    dst[(size_t)s * n + idx] = relu ? fmaxf(x, 0.0f) : x;
                }
            }
    }
}

// Sum over the 32 lanes of a half wave, every lane receives the total: four DPP steps inside each row of
// 16 (full-rate VALU, no LDS round trip) and one cross-row ds_bpermute.  (Five ds_bpermute round trips
// per dot product were a third of a phase's compute time.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_sum(float v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror
    v += __shfl_xor(v, 16);
    return v;
}

// softmax + sampling + mu-law decode of the batch rows this wave owns (row b -> wave b): utils.py:13-46,
// mu_law_ops.py:26-31.  Every workgroup does it redundantly (identical bits everywhere); xh receives
// x_in(t+1) = mu_law_encode(decoded sample)  (wavenet.py:113).
__device__ __forceinline__ void decode_rows(const PArgs& a, int bi, int tid, int t, int it, float* xg, float* xh) {
    const int wv = tid >> 6, lane = tid & 63, Q = a.Q;
    for (int b = wv; b < a.B; b += 8) {
        float* lg = xg + (size_t)b * Q;
        float m = -INFINITY;
        int mi = 0x7fffffff;
        for (int q = lane; q < Q; q += 64) {
            const float v = lg[q];
            if (v > m) { m = v; mi = q; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o);
            const int oi = __shfl_xor(mi, o);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        const bool last = (it == a.n_steps - 1);
        if (a.mode != 0 || (a.probs_last && last)) {
            float s = 0.0f;
            for (int q = lane; q < Q; q += 64) {
                const float e = __expf(lg[q] - m);
                lg[q] = e;
                s += e;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float inv = 1.0f / s;
            for (int q = lane; q < Q; q += 64) {
                const float p = lg[q] * inv;
                lg[q] = p;
                if (bi == 0 && a.probs_last && last) a.probs_last[(size_t)b * Q + q] = p;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // lane 0 reads the other lanes' LDS writes (same wave)
        }
        if (lane == 0) {
            int idx;
            if (a.mode == 0) {
                idx = mi;                                  // greedy: first maximum (np.argmax)
            } else {                                       // utils.py:20-25: searchsorted(cumsum(pdf), u), side='left'
                const float u = a.uniforms[(size_t)b * a.n_steps + it];
                float cdf = 0.0f;
                idx = 0;
                for (int q = 0; q < Q; ++q) {
                    cdf += lg[q];
                    if (cdf < u) idx = q + 1;
                }
            }
            const float dec = p_mu_dec((float)idx);
            xh[b * a.pre_k + ((t + 1) % a.pre_k)] = p_mu_enc(dec);
            if (bi == 0) {
                if (a.audio) a.audio[(size_t)b * a.n_steps + it] = dec;
                if (a.indices) a.indices[(size_t)b * a.n_steps + it] = idx;
                if (last) a.prev[b] = dec;
            }
        }
    }
}

#ifdef VQW_AR_TRACE
#define TR_DECL u64 tr_acc[16] = {0}; u64 tr_t = __builtin_amdgcn_s_memrealtime();
#define TR(i) { const u64 n_ = __builtin_amdgcn_s_memrealtime(); tr_acc[i] += n_ - tr_t; tr_t = n_; }
#define TR_DUMP if (a.trace && (tid & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) a.trace[((size_t)bi * 8 + (tid >> 6)) * 16 + i_] = tr_acc[i_];
#else
#define TR_DECL
#define TR(i)
#define TR_DUMP
#endif

// TB: compile-time bound of the batch rows; RLT = R/32 rows of a weight column per lane; NS = S/R; KS taps.
// (Compile-time load counts let the compiler wait with exact vmcnt values: with a run-time count it fell back
// to vmcnt(0) in front of the out phase, i.e. waited for the gate weights it had requested a moment earlier.)
template <int TB, int RLT, int NS, int KS>
__global__ __launch_bounds__(512, 1) void ar_persist_kernel(const PArgs a) {
    extern __shared__ float lds[];
    constexpr int SL = RLT * NS;                       // S / 32
    constexpr int NGQ = (KS * RLT * 2 + 3) / 4;        // float4 groups of gate weights per thread
    constexpr int NOQ = (RLT * (NS + 1) + 3) / 4;      // float4 groups of out weights per thread
    const int bi = blockIdx.x, tid = threadIdx.x, nwg = gridDim.x;
    const bool compute = tid < NCT;
    const int ct = tid & (NCT - 1);                    // index inside the role
    const int cg = ct >> 5, kl = ct & 31;
    const int c = bi * CPB + cg;                       // compute thread's channel
    constexpr int ks = KS;
    const int B = a.B, R = a.R, S = a.S, Q = a.Q, L = a.L, nQ = a.nQ;
    const Carve cv = make_carve(L, ks, R, S, Q, B, NS, nQ, a.pre_k);
    float* const xg = lds + cv.xg;                     // gate inputs [ks][B][R] / head inputs [B][S] / logits [B][Q]
    float* const xo = lds + cv.xo;                     // out inputs [B][R] / postprocess2 inputs [B][S]
    const float* const hw_p1 = lds + cv.hw;
    const float* const hw_p2 = hw_p1 + cv.n_p1 * NCT;
    const float* const hw_s0 = hw_p2 + cv.n_p2 * NCT;
    const float* const hw_pre = hw_s0 + cv.n_s0 * NCT;
    const float* const bs = lds + cv.bias;
    float* const condc = lds + cv.condc;               // [L][2][8][B], then postprocess1 [NS][8][B]
    float* const xh = lds + cv.xh;                     // [B][pre_k]
    float* const misc = lds + cv.misc;
    int* const fail = reinterpret_cast<int*>(misc + 60);
    int* const tab = reinterpret_cast<int*>(lds + cv.tab);   // [L] dilation, [L] ring offset

    const int t0 = a.state[0];
    // ---------------- one-time LDS fill
    for (int i = tid; i < cv.nhead * NCT; i += 512) lds[cv.hw + i] = a.headw[(size_t)bi * cv.nhead * NCT + i];
    for (int i = tid; i < cv.nbias; i += 512) lds[cv.bias + i] = a.bias[(size_t)bi * cv.nbias + i];
    for (int i = tid; i < L; i += 512) { tab[i] = a.dil[i]; tab[L + i] = a.ring_off[i]; }
    for (int i = tid; i < B * a.pre_k; i += 512) xh[i] = a.xhist[i];
    if (tid == 0) *fail = 0;
    __syncthreads();
    if (tid < B) xh[tid * a.pre_k + (t0 % a.pre_k)] = p_mu_enc(a.prev[tid]);   // x_in(t0) = mu_law_encode(previous sample)
    __syncthreads();

    const int PH = 2 * L + 4;
    const size_t nBR = (size_t)B * R;
    TR_DECL

    if (compute) {
        // ======================================================================== compute waves
        f32x4 gq[NGQ], oq[NOQ];
        auto load_gate = [&](int l) {
            const f32x4* p = reinterpret_cast<const f32x4*>(a.gw) + ((size_t)(l * nwg + bi) * NGQ) * NCT + ct;
#pragma unroll
            for (int q = 0; q < NGQ; ++q) gq[q] = p[(size_t)q * NCT];
        };
        auto load_out = [&](int l) {
            const f32x4* p = reinterpret_cast<const f32x4*>(a.ow) + ((size_t)(l * nwg + bi) * NOQ) * NCT + ct;
#pragma unroll
            for (int q = 0; q < NOQ; ++q) oq[q] = p[(size_t)q * NCT];
        };
        load_gate(0);
        load_out(0);
        float cur[TB], skipacc[NS][TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            cur[b] = 0.0f;
#pragma unroll
            for (int j = 0; j < NS; ++j) skipacc[j][b] = 0.0f;
        }

        for (int it = 0; it < a.n_steps; ++it) {
            const int t = t0 + it;
            const unsigned seq = (unsigned)t * (unsigned)PH + 1u;
            const unsigned ttag = (unsigned)t + 1u;
            // ---------------- preprocess: causal conv over the encoded input history, lane kl = tap kl (+32)
            {
                const int depth0 = (ks - 1) * tab[0] + 1;
                u64* ring0 = a.rings + tab[L] + (size_t)(t % depth0) * nBR;
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    if (b < B) {
                        float acc = 0.0f;
                        for (int i = 0; i < cv.n_pre; ++i) {
                            const int j = kl + 32 * i;
                            if (j < a.pre_k) {
                                const int tau = t - (a.pre_k - 1 - j);
                                const int slot = ((tau % a.pre_k) + a.pre_k) % a.pre_k;
                                acc = fmaf(hw_pre[i * NCT + ct], xh[b * a.pre_k + slot], acc);
                            }
                        }
                        cur[b] = half_sum(acc) + bs[cv.bias_head + (2 * NS + nQ) * CPB + cg];
                        if (kl == 0) publish(ring0 + (size_t)b * R + c, ttag, cur[b]);
                    }
                }
            }
            TR(0)
            for (int l = 0; l < L; ++l) {
                const float* bl = bs + l * (NS + 3) * CPB;
                // the taps t-(ks-1)d .. t-d are old news: their share of the dot products is done while the
                // current layer input is still travelling
                float af[TB], ag[TB];
                role_barrier();                                    // BAR0: xg[0..ks-2] = cur_l(t - (ks-1-j) d)
                TR(1)
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    af[b] = 0.0f; ag[b] = 0.0f;
                    if (b < B) {
#pragma unroll
                        for (int j = 0; j < KS - 1; ++j) {
#pragma unroll
                            for (int i = 0; i < RLT; ++i) {
                                const float xv = xg[((size_t)j * B + b) * R + kl + 32 * i];
                                const int e = (j * RLT + i) * 2;
                                af[b] = fmaf(gq[e >> 2][e & 3], xv, af[b]);
                                ag[b] = fmaf(gq[(e + 1) >> 2][(e + 1) & 3], xv, ag[b]);
                            }
                        }
                        if (TB > 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                TR(12)
                role_barrier();                                    // BAR1: xg[ks-1] = cur_l(t)
                TR(13)
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    if (b < B) {
                        const float* xc = xg + ((size_t)(KS - 1) * B + b) * R;
                        float f = af[b], g = ag[b];
#pragma unroll
                        for (int i = 0; i < RLT; ++i) {
                            const float xv = xc[kl + 32 * i];
                            const int e = ((KS - 1) * RLT + i) * 2;
                            f = fmaf(gq[e >> 2][e & 3], xv, f);
                            g = fmaf(gq[(e + 1) >> 2][(e + 1) & 3], xv, g);
                        }
                        f = half_sum(f);
                        g = half_sum(g);
                        if (l == 0) {   // skip = linear(current)  (wavenet.py:127-128)
#pragma unroll
                            for (int j = 0; j < NS; ++j) {
                                float s0 = 0.0f;
#pragma unroll
                                for (int i = 0; i < RLT; ++i) s0 = fmaf(hw_s0[(i * NS + j) * NCT + ct], xc[kl + 32 * i], s0);
                                skipacc[j][b] = half_sum(s0) + bs[cv.bias_head + j * CPB + cg];
                            }
                        }
                        if (kl == 0) {
                            const float vf = f + bl[cg] + condc[((l * 2 + 0) * CPB + cg) * B + b];
                            const float vg = g + bl[CPB + cg] + condc[((l * 2 + 1) * CPB + cg) * B + b];
                            const float th = 1.0f - 2.0f / (__expf(2.0f * vf) + 1.0f);
                            const float sg = 1.0f / (1.0f + __expf(-vg));
                            publish(a.ex_g + (size_t)b * R + c, seq + 2 * l, th * sg);
                        }
                        if (TB > 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                TR(14)
                load_gate(l + 1 < L ? l + 1 : 0);                  // next gate phase's columns: a whole layer to land
                TR(2)
                role_barrier();                                    // BAR2: xo = gated_l(t)
                TR(3)
                // ---------------- out phase: skip columns (private accumulators) + residual column
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    if (b < B) {
#pragma unroll
                        for (int j = 0; j <= NS; ++j) {
                            float s = 0.0f;
#pragma unroll
                            for (int i = 0; i < RLT; ++i) {
                                const int e = i * (NS + 1) + j;
                                s = fmaf(oq[e >> 2][e & 3], xo[(size_t)b * R + kl + 32 * i], s);
                            }
                            s = half_sum(s) + bl[(2 + j) * CPB + cg];
                            if (j < NS) skipacc[j < NS ? j : 0][b] += s;
                            else cur[b] += s;
                        }
                        if (l + 1 < L && kl == 0) {
                            const int depth = (ks - 1) * tab[l + 1] + 1;
                            publish(a.rings + tab[L + l + 1] + ((size_t)(t % depth) * B + b) * R + c, ttag, cur[b]);
                        }
                        if (TB > 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (l + 1 == L && kl == 0) {   // postprocess1 input
#pragma unroll
                    for (int b = 0; b < TB; ++b)
#pragma unroll
                        for (int j = 0; j < NS; ++j)
                            if (b < B) publish(a.ex_s + (size_t)b * S + c + (size_t)j * R, seq + 2 * L, skipacc[j][b]);
                }
                load_out(l + 1 < L ? l + 1 : 0);
                TR(4)
            }
            // ---------------- postprocess1: relu(skip) -> 1x1 + condition  (wavenet.py:152-162)
            role_barrier();                                        // BAR3: xg = relu(skip) [B][S]
            TR(5)
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        float s = 0.0f;
#pragma unroll
                        for (int i = 0; i < SL; ++i) s = fmaf(hw_p1[(i * NS + j) * NCT + ct], xg[(size_t)b * S + kl + 32 * i], s);
                        s = half_sum(s);
                        if (kl == 0) {
                            const float v = s + bs[cv.bias_head + (NS + j) * CPB + cg] + condc[((L * 2 + j) * CPB + cg) * B + b];
                            publish(a.ex_h + (size_t)b * S + c + (size_t)j * R, seq + 2 * L + 1, v);
                        }
                    }
                }
            }
            TR(6)
            // ---------------- postprocess2: relu(h) -> logits  (wavenet.py:165-167)
            role_barrier();                                        // BAR4: xo = relu(h) [B][S]
            TR(7)
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
                    for (int j = 0; j < nQ; ++j) {
                        float s = 0.0f;
#pragma unroll
                        for (int i = 0; i < SL; ++i) s = fmaf(hw_p2[(i * nQ + j) * NCT + ct], xo[(size_t)b * S + kl + 32 * i], s);
                        s = half_sum(s);
                        if (kl == 0) publish(a.ex_l + (size_t)b * Q + c + (size_t)j * R, seq + 2 * L + 2, s + bs[cv.bias_head + (2 * NS + j) * CPB + cg]);
                    }
                }
            }
            TR(8)
            role_barrier();                                        // BAR5: xg = logits [B][Q]
            TR(9)
            decode_rows(a, bi, tid, t, it, xg, xh);
            TR(10)
            if (*reinterpret_cast<volatile int*>(fail)) break;     // stable between BAR5 and BAR6: nobody polls there
            role_barrier();                                        // BAR6: xh holds x_in(t+1)
            TR(11)
        }
    } else {
        // ======================================================================== gather waves
        const int gid = ct;
        int last_frame = -1;
        for (int it = 0; it < a.n_steps; ++it) {
            const int t = t0 + it;
            const unsigned seq = (unsigned)t * (unsigned)PH + 1u;
            int frame = t / a.ratio;
            if (frame >= a.Tz) frame = a.Tz - 1;
            if (frame != last_frame) {   // condition projections of this frame for my 8 channels (all layers)
                last_frame = frame;
                const int per = CPB * B;
                for (int i = gid; i < (L * 2 + NS) * per; i += NCT) {
                    const int b = i % B, g8 = (i / B) % CPB, lh = i / per;
                    const int ch = bi * CPB + g8;
                    float v;
                    if (lh < 2 * L) v = a.cond[lh >> 1][((size_t)b * 2 * R + ch + (size_t)(lh & 1) * R) * a.Tz + frame];
                    else v = a.cond[L][((size_t)b * S + ch + (size_t)(lh - 2 * L) * R) * a.Tz + frame];
                    condc[i] = v;
                }
            }
            TR(0)
            for (int l = 0; l < L; ++l) {
                {   // the ks taps of the layer input: the old ones first (nothing to wait for), then the fresh one
                    const int dil = tab[l], depth = (ks - 1) * dil + 1;
                    const u64* ring = a.rings + tab[L + l];
                    const u64* src[KS];
                    unsigned tag[KS];
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
                        const int tau = t - (ks - 1 - j) * dil;
                        src[j] = (tau >= 0) ? ring + (size_t)(tau % depth) * nBR : nullptr;
                        tag[j] = (unsigned)tau + 1u;
                    }
                    gather_segs<KS - 1>(gid, fail, src, tag, KS - 1, (int)nBR, xg, 0);
                    TR(10)
                    role_barrier();                                // BAR0
                    TR(11)
                    gather_segs<1>(gid, fail, src + (KS - 1), tag + (KS - 1), 1, (int)nBR, xg + (size_t)(KS - 1) * nBR, 0);
                }
                TR(1)
                role_barrier();                                    // BAR1
                TR(2)
                {
                    const u64* src[1] = {a.ex_g};
                    const unsigned tag[1] = {seq + 2 * l};
                    gather_segs<1>(gid, fail, src, tag, 1, (int)nBR, xo, 0);
                }
                TR(3)
                role_barrier();                                    // BAR2
                TR(4)
            }
            {
                const u64* src[1] = {a.ex_s};
                const unsigned tag[1] = {seq + 2 * L};
                gather_segs<1>(gid, fail, src, tag, 1, B * S, xg, 1);
            }
            TR(5)
            role_barrier();                                        // BAR3
            {
                const u64* src[1] = {a.ex_h};
                const unsigned tag[1] = {seq + 2 * L + 1};
                gather_segs<1>(gid, fail, src, tag, 1, B * S, xo, 1);
            }
            TR(6)
            role_barrier();                                        // BAR4
            {
                const u64* src[1] = {a.ex_l};
                const unsigned tag[1] = {seq + 2 * L + 2};
                gather_segs<1>(gid, fail, src, tag, 1, B * Q, xg, 0);
            }
            TR(7)
            role_barrier();                                        // BAR5
            decode_rows(a, bi, tid, t, it, xg, xh);
            TR(8)
            if (*reinterpret_cast<volatile int*>(fail)) break;
            role_barrier();                                        // BAR6
            TR(9)
        }
    }
    TR_DUMP
    // ---------------- save state (workgroup 0) / report a timeout
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(fail)) {
        if (tid == 0) atomicExch(a.state + 1, 1);
        return;
    }
    if (bi == 0) {
        for (int i = tid; i < B * a.pre_k; i += 512) a.xhist[i] = xh[i];
        if (tid == 0) a.state[0] = t0 + a.n_steps;
    }
}

// ------------------------------------------------------------------------------------------------
// one-time re-blocking of the model's variables: every compute thread's share becomes contiguous

// gate kernel [ks][R][2R] of one layer -> dst[bi][q][ct][4]; element e = q*4+r = (tap*RL + i)*2 + h
__global__ void pack_gate_kernel(const float* __restrict__ src, int ks, int R, int ngq, float* __restrict__ dst) {
    const int RL = R / 32, nwg = R / CPB;
    const size_t n = (size_t)nwg * ngq * NCT * 4;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(x & 3), ct = (int)((x >> 2) % NCT), q = (int)((x / (4 * NCT)) % ngq), bi = (int)(x / ((size_t)4 * NCT * ngq));
        const int e = q * 4 + r, h = e & 1, i = (e >> 1) % RL, j = (e >> 1) / RL;
        const int c = bi * CPB + (ct >> 5), k = (ct & 31) + 32 * i;
        dst[x] = (j < ks) ? src[((size_t)j * R + k) * 2 * R + c + h * R] : 0.0f;
    }
}

// out kernel [R][ld] (skip | residual side by side) of one layer -> dst[bi][q][ct][4]; e = i*(nS+1) + j,
// column c + j*R for the skip columns j < nS, column S + c for the residual (j == nS)
__global__ void pack_out_kernel(const float* __restrict__ src, int ld, int R, int S, int nS, int noq, float* __restrict__ dst) {
    const int RL = R / 32, nwg = R / CPB;
    const size_t n = (size_t)nwg * noq * NCT * 4;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(x & 3), ct = (int)((x >> 2) % NCT), q = (int)((x / (4 * NCT)) % noq), bi = (int)(x / ((size_t)4 * NCT * noq));
        const int e = q * 4 + r, i = e / (nS + 1), j = e % (nS + 1);
        const int c = bi * CPB + (ct >> 5), k = (ct & 31) + 32 * i;
        dst[x] = (i < RL) ? src[(size_t)k * ld + (j < nS ? c + j * R : S + c)] : 0.0f;
    }
}

// head section: src [rows][ld], ncol columns per channel (c + j*R) -> dst[bi][e0 + e][ct], e = i*ncol + j, row kl + 32 i
__global__ void pack_head_kernel(const float* __restrict__ src, int ld, int rows, int R, int ncol, float* __restrict__ dst,
                                 int e0, int nhead) {
    const int nwg = R / CPB, nsec = ((rows + 31) / 32) * ncol;
    const size_t n = (size_t)nwg * nsec * NCT;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) {
        const int ct = (int)(x % NCT), e = (int)((x / NCT) % nsec), bi = (int)(x / ((size_t)NCT * nsec));
        const int i = e / ncol, j = e % ncol;
        const int c = bi * CPB + (ct >> 5), k = (ct & 31) + 32 * i;
        dst[((size_t)bi * nhead + e0 + e) * NCT + ct] = (k < rows) ? src[(size_t)k * ld + c + j * R] : 0.0f;
    }
}

#define PHIPC(x)                                                                               \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) return vqw_set_error("%s failed: %s", #x, hipGetErrorString(e_)); \
    } while (0)

template <int TB>
const void* pick_kernel(int RL, int nS, int ks) {
#define ARP_CASE(rl, ns, k) \
    if (RL == rl && nS == ns && ks == k) return reinterpret_cast<const void*>(ar_persist_kernel<TB, rl, ns, k>)
    ARP_CASE(8, 2, 3); ARP_CASE(4, 2, 3); ARP_CASE(2, 2, 3); ARP_CASE(1, 2, 3); ARP_CASE(8, 1, 3); ARP_CASE(4, 4, 3);
    ARP_CASE(8, 2, 2); ARP_CASE(1, 2, 2);
#undef ARP_CASE
    return nullptr;   // other shapes run on the launch-per-phase path of ar_decode.hip
}

}  // namespace

struct ArPersist {
    vqw_ar_weights w;
    int B = 0, nS = 0, nQ = 0, nwg = 0;
    std::vector<void*> allocs;
    const float** dcond = nullptr;   // device copy of the L+1 condition pointers
    PArgs args;
    const void* kfn = nullptr;
    size_t lds_bytes = 0;
    std::vector<std::pair<void*, size_t>> zero_on_reset;
#ifdef VQW_AR_TRACE
    u64* trace = nullptr;
    int trace_steps = 0;
#endif
};

static void* pmalloc(ArPersist* h, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) return nullptr;
    h->allocs.push_back(p);
    return p;
}

bool arp_supported(const vqw_ar_weights* w, int batch) {
    const char* env = getenv("VQW_AR_PERSISTENT");
    if (env && env[0] == '0') return false;
    if (w->R < 32 || w->R > 256 || w->R % 32 || w->S % w->R || w->Q % w->R) return false;
    const int nS = w->S / w->R, nQ = w->Q / w->R;
    if (nQ < 1 || nQ > PJ || batch > PB || w->pre_k > 60 || w->kernel_size > KSMAX) return false;
    if (!pick_kernel<1>(w->R / 32, nS, w->kernel_size)) return false;
    const Carve cv = make_carve(w->n_layers, w->kernel_size, w->R, w->S, w->Q, batch, nS, nQ, w->pre_k);
    if ((size_t)cv.total * sizeof(float) > 160 * 1024) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return w->R / CPB <= cus;   // one resident workgroup per CU is what makes the spin-waits safe
}

int arp_create(ArPersist** out, const vqw_ar_weights* w, const int* dil, const float* const* gated_w,
               const float* const* gated_b, const float* const* out_w, const float* const* out_b, int batch) {
    ArPersist* h = new ArPersist();
    h->w = *w;
    h->B = batch;
    const int L = w->n_layers, R = w->R, S = w->S, Q = w->Q, ks = w->kernel_size;
    const int nS = S / R, nQ = Q / R, RL = R / 32, nwg = R / CPB;
    h->nS = nS; h->nQ = nQ; h->nwg = nwg;
    auto fail = [&](const char* m) { arp_destroy(h); return vqw_set_error("vqw_ar_decode_create(persistent): %s", m); };
    const Carve cv = make_carve(L, ks, R, S, Q, batch, nS, nQ, w->pre_k);
    PArgs& a = h->args;
    memset(&a, 0, sizeof(a));
    a.L = L; a.ks = ks; a.R = R; a.S = S; a.Q = Q; a.B = batch; a.nS = nS; a.nQ = nQ; a.pre_k = w->pre_k;
    a.ngq = (ks * RL * 2 + 3) / 4;
    a.noq = (RL * (nS + 1) + 3) / 4;
    // ---- per-layer weight columns, blocked per compute thread
    const size_t gstride = (size_t)nwg * a.ngq * NCT * 4, ostride = (size_t)nwg * a.noq * NCT * 4;
    const size_t nheadw = (size_t)nwg * cv.nhead * NCT;
    float* gw = (float*)pmalloc(h, L * gstride * sizeof(float));
    float* ow = (float*)pmalloc(h, L * ostride * sizeof(float));
    float* hwd = (float*)pmalloc(h, nheadw * sizeof(float));
    if (!gw || !ow || !hwd) return fail("hipMalloc failed");
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(pack_gate_kernel, dim3(256), dim3(256), 0, 0, gated_w[l], ks, R, a.ngq, gw + l * gstride);
        hipLaunchKernelGGL(pack_out_kernel, dim3(256), dim3(256), 0, 0, out_w[l], w->out_ld, R, S, nS, a.noq, ow + l * ostride);
    }
    int e0 = 0;
    hipLaunchKernelGGL(pack_head_kernel, dim3(256), dim3(256), 0, 0, w->post1_w, S, S, R, nS, hwd, e0, cv.nhead);
    e0 += cv.n_p1;
    hipLaunchKernelGGL(pack_head_kernel, dim3(256), dim3(256), 0, 0, w->post2_w, Q, S, R, nQ, hwd, e0, cv.nhead);
    e0 += cv.n_p2;
    hipLaunchKernelGGL(pack_head_kernel, dim3(256), dim3(256), 0, 0, w->skip0_w, S, R, R, nS, hwd, e0, cv.nhead);
    e0 += cv.n_s0;
    hipLaunchKernelGGL(pack_head_kernel, dim3(64), dim3(256), 0, 0, w->pre_w, R, w->pre_k, R, 1, hwd, e0, cv.nhead);
    a.gw = gw; a.ow = ow; a.headw = hwd;
    // ---- biases: tiny, packed on the host
    {
        std::vector<float> hb((size_t)nwg * cv.nbias, 0.0f), tmp;
        auto fetch = [&](const float* dptr, int n) -> bool {
            tmp.resize(n);
            return hipMemcpy(tmp.data(), dptr, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
        };
        for (int l = 0; l < L; ++l) {
            if (!fetch(gated_b[l], 2 * R)) return fail("hipMemcpy failed");
            for (int bi = 0; bi < nwg; ++bi)
                for (int hh = 0; hh < 2; ++hh)
                    for (int g = 0; g < CPB; ++g)
                        hb[(size_t)bi * cv.nbias + l * (nS + 3) * CPB + hh * CPB + g] = tmp[hh * R + bi * CPB + g];
            if (!fetch(out_b[l], S + R)) return fail("hipMemcpy failed");
            for (int bi = 0; bi < nwg; ++bi)
                for (int j = 0; j <= nS; ++j)
                    for (int g = 0; g < CPB; ++g)
                        hb[(size_t)bi * cv.nbias + l * (nS + 3) * CPB + (2 + j) * CPB + g] = tmp[(j < nS ? j * R : S) + bi * CPB + g];
        }
        struct { const float* p; int n, ncol, at; } hs[] = {{w->skip0_b, S, nS, 0}, {w->post1_b, S, nS, nS},
                                                           {w->post2_b, Q, nQ, 2 * nS}, {w->pre_b, R, 1, 2 * nS + nQ}};
        for (auto& s : hs) {
            if (!fetch(s.p, s.n)) return fail("hipMemcpy failed");
            for (int bi = 0; bi < nwg; ++bi)
                for (int j = 0; j < s.ncol; ++j)
                    for (int g = 0; g < CPB; ++g)
                        hb[(size_t)bi * cv.nbias + cv.bias_head + (s.at + j) * CPB + g] = tmp[j * R + bi * CPB + g];
        }
        float* db = (float*)pmalloc(h, hb.size() * sizeof(float));
        if (!db) return fail("hipMalloc failed");
        if (hipMemcpy(db, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
        a.bias = db;
    }
    // ---- rings (layer inputs == dilation queues) and exchange buffers
    {
        std::vector<int> roff(L);
        long long total = 0;
        for (int l = 0; l < L; ++l) {
            roff[l] = (int)total;
            total += (long long)((ks - 1) * dil[l] + 1) * batch * R;
            if (total > 0x7fffffffll) return fail("dilation queues too large");
        }
        int* dtab = (int*)pmalloc(h, 2 * L * sizeof(int));
        if (!dtab) return fail("hipMalloc failed");
        if (hipMemcpy(dtab, dil, L * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dtab + L, roff.data(), L * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return fail("hipMemcpy failed");
        a.dil = dtab; a.ring_off = dtab + L;
        a.rings = (u64*)pmalloc(h, (size_t)total * sizeof(u64));
        if (!a.rings) return fail("hipMalloc failed");
        h->zero_on_reset.push_back({a.rings, (size_t)total * sizeof(u64)});
    }
    struct { u64** p; size_t n; } exs[] = {{&a.ex_g, (size_t)batch * R}, {&a.ex_s, (size_t)batch * S},
                                           {&a.ex_h, (size_t)batch * S}, {&a.ex_l, (size_t)batch * Q}};
    for (auto& e : exs) {
        *e.p = (u64*)pmalloc(h, e.n * sizeof(u64));
        if (!*e.p) return fail("hipMalloc failed");
        h->zero_on_reset.push_back({*e.p, e.n * sizeof(u64)});
    }
    a.xhist = (float*)pmalloc(h, (size_t)batch * w->pre_k * sizeof(float));
    a.prev = (float*)pmalloc(h, batch * sizeof(float));
    a.state = (int*)pmalloc(h, 2 * sizeof(int));
    h->dcond = (const float**)pmalloc(h, (L + 1) * sizeof(float*));
    if (!a.xhist || !a.prev || !a.state || !h->dcond) return fail("hipMalloc failed");
    a.cond = h->dcond;
    h->zero_on_reset.push_back({a.xhist, (size_t)batch * w->pre_k * sizeof(float)});
    h->zero_on_reset.push_back({a.prev, batch * sizeof(float)});
    h->zero_on_reset.push_back({a.state, 2 * sizeof(int)});
    if (hipDeviceSynchronize() != hipSuccess) return fail("weight re-blocking failed");
    h->lds_bytes = (size_t)cv.total * sizeof(float);
    if (h->lds_bytes < 96 * 1024) h->lds_bytes = 96 * 1024;   // > half of the 160 KiB: one workgroup per CU
    if (h->lds_bytes > 160 * 1024) return fail("LDS budget exceeded");
    h->kfn = (batch <= 1) ? pick_kernel<1>(RL, nS, ks) : pick_kernel<8>(RL, nS, ks);
    if (!h->kfn) return fail("unsupported R/S combination");
    if (hipFuncSetAttribute(h->kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess)
        return fail("hipFuncSetAttribute failed");
    *out = h;
    return 0;
}

int arp_reset(ArPersist* h, hipStream_t st) {
    for (auto& z : h->zero_on_reset) PHIPC(hipMemsetAsync(z.first, 0, z.second, st));
    return 0;
}

int arp_run(ArPersist* h, const float* const* condenc, int Tz, int ratio, int n_steps, int mode, const float* uniforms,
            float* audio, int32_t* indices, float* probs_last, hipStream_t st) {
    const int L = h->w.n_layers;
    PHIPC(hipMemcpyAsync(h->dcond, condenc, (L + 1) * sizeof(float*), hipMemcpyHostToDevice, st));
    PArgs a = h->args;
    a.Tz = Tz; a.ratio = ratio; a.mode = mode; a.n_steps = n_steps;
    a.uniforms = uniforms; a.audio = audio; a.indices = indices; a.probs_last = probs_last;
#ifdef VQW_AR_TRACE
    if (!h->trace) h->trace = (u64*)pmalloc(h, (size_t)h->nwg * 8 * 16 * sizeof(u64));
    a.trace = h->trace;
    h->trace_steps = n_steps;
#endif
    void* params[] = {&a};
    PHIPC(hipLaunchKernel(h->kfn, dim3(h->nwg), dim3(512), params, h->lds_bytes, st));
    return 0;
}

int arp_error(ArPersist* h, hipStream_t st) {
    int s2[2] = {0, 0};
    if (hipMemcpyAsync(s2, h->args.state, sizeof(s2), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
#ifdef VQW_AR_TRACE
    if (h->trace && getenv("VQW_AR_TRACE_PRINT")) {
        const int nw = h->nwg * 8;
        std::vector<u64> t((size_t)nw * 16);
        (void)hipMemcpy(t.data(), h->trace, t.size() * sizeof(u64), hipMemcpyDeviceToHost);
        const int picks[4] = {0, 4, nw - 8, nw - 4};
        for (int w : picks) {
            fprintf(stderr, "[ar-trace] wave %3d (%s): us/sample by section:", w, (w & 7) < 4 ? "compute" : "gather");
            double tot = 0;
            for (int i = 0; i < 16; ++i) {
                const double us = t[(size_t)w * 16 + i] * 0.01 / h->trace_steps;
                tot += us;
                fprintf(stderr, " %d:%.1f", i, us);
            }
            fprintf(stderr, "  total %.1f\n", tot);
        }
    }
#endif
    return s2[1];
}

void arp_destroy(ArPersist* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

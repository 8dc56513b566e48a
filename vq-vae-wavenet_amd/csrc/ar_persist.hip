// Persistent fast-generation kernel for gfx950: the whole sample loop of generate.py:103-113
// (216 tf.matmul + 91 FIFOQueue ops + numpy sampling per sample in the reference,
// wavenet.py:103-172 / wavenet_ops.py:147-267 / utils.py:13-46) in ONE launch.
//
// Decomposition: R/8 workgroups (one per CU), workgroup bi owns channels 8 bi .. 8 bi + 7 of every matrix of
// the model.  Values cross workgroups by the R2 recipe of the CDNA guide (Guideline 16): ONE naturally aligned
// 8-byte granule {tag, fp32 bits} per value, written with a relaxed agent-scope atomic store (sc1, write-through)
// and polled with relaxed agent-scope atomic loads; the tag is a per-phase sequence number, so no flags, fences
// or resets are needed and nothing depends on dispatch order or XCD placement.  The per-layer exchange buffer of
// the layer input doubles as the dilation queue (ring of (k-1)d+1 time slots).
//
// ONE exchange per layer.  The reference layer is  gate_l = act(Wg_l * [cur_l taps] + cond),
// cur_{l+1} = cur_l + Wr_l gated_l + br_l  (wavenet_ops.py:212-267): two dependent matrix-vector products, i.e.
// two all-gathers per layer (~1.1 us each, measured with tools/ar_trace.py: the exchange, not the arithmetic, is
// the price of a sample).  Here the current-tap product is re-associated,
//     Wg_l^cur cur_l = Wg_l^cur cur_{l-1} + (Wg_l^cur Wr_{l-1}) gated_{l-1} + Wg_l^cur br_{l-1},
// with M_l = Wg_l^cur Wr_{l-1} and the folded bias formed once per handle in fp64, so phase l needs only the
// vectors cur_{l-1} and gated_{l-1}, which phase l-1 publishes TOGETHER: phase l = {residual + skip of layer
// l-1, gate of layer l}.  The residual chain itself keeps the reference's operation order bit for bit; only the
// gate pre-activation sees the re-association (~1e-7 relative).  A sample is L+1 phases + 3 head exchanges.
//
// A workgroup has three roles (a wave that polls behind its own weight loads waits for them, vmcnt returns in
// order):
//   waves 0-3 "compute": weight columns in registers, requested one phase ahead (the critical ones -- current
//             tap, M, residual, skip -- right behind the phase's publishes, the past-tap ones in front of the
//             barrier) as 16-byte loads from a per-thread blocked copy served by L2 / Infinity Cache; the past
//             taps' share of the dot products is formed while the exchange is in flight.  They never poll.
//   waves 4-5 "fresh":   poll the granules of the phase into LDS, meet the compute waves at a raw s_barrier.
//   waves 6-7 "history": fetch the dilation-queue taps two phases ahead (they are old news).
// Head weights, all biases and the condition projections of the current frame live in LDS.  Every spin is
// bounded by a wall-clock timeout (s_memrealtime) that sets an error word and drains the grid.
#include <string.h>

#include <type_traits>
#include <vector>

#include "ar_persist.h"

namespace {

typedef unsigned long long u64;
constexpr int PB = 4;        // max batch rows of one persistent launch (LDS budget; larger batches: several handles)
constexpr int PJ = 8;        // max Q / R
constexpr int NCT = 256;     // compute threads (waves 0-3)
#ifndef ARP_NGF
#define ARP_NGF 128
#endif
#ifndef ARP_NGH
#define ARP_NGH 128
#endif
// Role split measured on the reference stack (us per sample, 1 / 2 / 4 rows): fresh 128 lanes 67 / 100 / 127;
// 256 lanes (9 waves, one granule per lane and segment) 101 / 119 / -; 64 lanes 87 / 97 / 146.
constexpr int NGF = ARP_NGF;           // lanes of the fresh role (the waves after the 4 compute waves)
constexpr int NGH = ARP_NGH;           // lanes of the history role (the last waves)
constexpr int NTHR = 256 + NGF + NGH;   // threads of a workgroup
constexpr unsigned long long TIMEOUT_TICKS = 300000000ull;  // 3 s of s_memrealtime (100 MHz)

struct PArgs {
    int L, R, S, Q, B, nQ, pre_k;
    const float* pw;              // [L+1][nwg][NPW][256][4]  past-tap gate columns of phase p; e = (tap*RL + i)*2 + {f, g}
    const float* cw;              // [L+1][nwg][NCW][256][4]  critical columns of phase p; e = i*(5+NS) + {Wg_f, Wg_g, M_f, M_g, Wr, Ws..}
    const float* headw;           // [nwg][nhead][256]        post1 | post2 | preprocess
    const float* bias;            // [nwg][nbias]
    const int* dil;               // [L]
    const int* ring_off;          // [L] granule offset of each layer's ring
    u64* rings;                   // sum_l depth_l * B * R granules: layer inputs == dilation queues
    const float* const* cond;     // [L+1] this run's condition projections ([B][2R][Tz] ..., [B][S][Tz])
    u64 *ex_g, *ex_s, *ex_h, *ex_l;  // [2][B][R] (phase parity), [B][S], [B][S], [B][Q]
    float* xhist;                 // [B][pre_k] encoded input history (state across runs)
    float* prev;                  // [B] last decoded sample
    int* state;                   // [0] = step, [1] = error
    int Tz, ratio, mode, n_steps;
    const float* uniforms;
    float* audio;
    int* indices;
    float* probs_last;
#ifdef VQW_AR_TRACE
    u64* trace;                   // [gridDim][16 waves][16] accumulated s_memrealtime ticks (tools/ar_trace.py)
#endif
};

// One launch may carry up to PGROUP handles (independent batch slices generating side by side): grid = (nwg, n_handles),
// workgroup (x, y) works on handle y.  ONE launch, because two persistent kernels on two streams only overlap when the
// runtime happens to map the streams to different hardware queues (it did not in bench.py once a third stream existed:
// 255 instead of 134 us per step for 8 utterances).
// Placement: the grid is one-dimensional, consecutive workgroup ids go round the 8 XCDs (each with its own L2).  Default
// ("channel"): workgroup g works for handle g / nwg as its workgroup g % nwg -- the workgroups that own the SAME channels of all
// handles share an XCD.  VQW_AR_PLACE=handle: workgroup g works for handle g % n, i.e. with 8 handles x 32 workgroups every
// handle lives on ONE XCD.  Measured (round 3, tools/ar_layouts.py, 8 one-row handles x 32 workgroups = all 256 CUs):
// channel 117.7 us per step, handle 123.0 -- the exchanges are agent-scope (sc1) granules that cross the fabric wherever
// the two workgroups sit (per-XCD L2s are not coherent: there is no correct XCD-local form), so a handle gains nothing from
// having its workgroups on one XCD, while same-channel workgroups on one XCD read their weights through one L2.
constexpr int PGROUP = 8;
struct PGroup {
    PArgs h[PGROUP];
    int n;
    int by_handle;     // 1: workgroup g works for handle g % n (a handle's workgroups share XCDs); 0: for handle g / nwg (the
                       // workgroups with the SAME channels of all handles share an XCD, i.e. one L2 copy of their weights)
};

// LDS carve (in floats), shared by the host (size) and the device (offsets)
struct Carve {
    int xfresh, xpast, hx1, hx2, hw, bias, condc, xh, misc, tab, dtab, total;
    int n_p1, n_p2, n_pre, nhead, nbias, bias_head, npast;
};
__host__ __device__ inline Carve make_carve(int L, int ks, int R, int S, int Q, int B, int nS, int nQ, int pre_k, int CPB) {
    Carve c;
    const int LPC = NCT / CPB, SL = S / LPC;             // lanes per channel, rows of an S-long column per lane
    c.n_p1 = SL * nS; c.n_p2 = SL * nQ; c.n_pre = (pre_k + LPC - 1) / LPC;
    c.nhead = c.n_p1 + c.n_p2 + c.n_pre;
    c.bias_head = (L + 1) * (nS + 3) * CPB;              // per phase: folded gate bias {f, g}, residual bias, skip biases
    c.nbias = c.bias_head + (nS + nQ + 1) * CPB;         // post1, post2, preprocess
    c.npast = (ks - 1) * B * R;
    const int nh = (B * S > B * Q) ? B * S : B * Q;
    c.xfresh = 0;
    c.xpast = c.xfresh + 2 * 2 * B * R;
    c.hx1 = c.xpast + 3 * c.npast;
    c.hx2 = c.hx1 + nh;
    c.hw = c.hx2 + B * S;
    c.bias = c.hw + c.nhead * NCT;
    c.condc = c.bias + ((c.nbias + 3) & ~3);
    c.xh = c.condc + (L * 2 + nS) * CPB * B;
    c.misc = c.xh + ((B * pre_k + 3) & ~3);
    c.tab = c.misc + 64;
    c.dtab = c.tab + 2 * L;                              // mu-law decode / re-encode of every class index 0..Q
    c.total = c.dtab + 2 * (Q + 1);
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
#ifndef ARP_SPIN_SLEEP
#define ARP_SPIN_SLEEP 2
#endif
        __builtin_amdgcn_s_sleep(ARP_SPIN_SLEEP);
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

// Gather lanes (NL of them): nseg segments of n granules each -> dst[s*n + i]; src[s] == nullptr reads as zeros
// (zero-filled queues).  All first loads of a chunk are issued before the first tag is looked at.  The producers
// of a phase publish at about the same time, so once ONE late granule of this lane has shown its tag the others
// are reloaded TOGETHER (one more round trip) instead of being polled one after the other (a round trip each).
template <int NSEG, int NL, bool BATCH>   // BATCH: the reload-together strategy (measured: pays from 3 batch rows on:
                                         // 4 rows 143 -> 131 us, 8 rows 153 -> 140; one row 67.5 -> 70, two rows 95 -> 101)
__device__ __forceinline__ void gather_segs(int gid, int* fail, const u64* const* src, const unsigned* tag,
                                            int n, float* dst, int relu) {
    constexpr int CH = 4;
    for (int base = gid; base < n; base += NL * CH) {
        u64 v[NSEG][CH];
        bool late[NSEG][CH];
#pragma unroll
        for (int s = 0; s < NSEG; ++s)
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                const int idx = base + NL * m;
                v[s][m] = (src[s] && idx < n) ? __hip_atomic_load(src[s] + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
        bool any_late = false, waited = false;
#pragma unroll
        for (int s = 0; s < NSEG; ++s)
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                late[s][m] = src[s] && (base + NL * m < n) && (unsigned)(v[s][m] >> 32) != tag[s];
                any_late = any_late || late[s][m];
            }
        if (any_late && !BATCH) {          // one after the other, two polls of each in flight
#pragma unroll
            for (int s = 0; s < NSEG; ++s)
#pragma unroll
                for (int m = 0; m < CH; ++m)
                    if (late[s][m]) v[s][m] = poll_granule(fail, src[s] + base + NL * m, tag[s], v[s][m]);
        } else if (any_late) {
#pragma unroll
            for (int s = 0; s < NSEG; ++s)
#pragma unroll
                for (int m = 0; m < CH; ++m)
                    if (late[s][m] && !waited) {          // spin on the first late granule only
                        v[s][m] = poll_granule(fail, src[s] + base + NL * m, tag[s], v[s][m]);
                        late[s][m] = false;
                        waited = true;
                    }
#pragma unroll
            for (int s = 0; s < NSEG; ++s)
#pragma unroll
                for (int m = 0; m < CH; ++m)
                    if (late[s][m]) v[s][m] = __hip_atomic_load(src[s] + base + NL * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int s = 0; s < NSEG; ++s)
#pragma unroll
                for (int m = 0; m < CH; ++m)
                    if (late[s][m]) v[s][m] = poll_granule(fail, src[s] + base + NL * m, tag[s], v[s][m]);   // returns at once if fresh
        }
#pragma unroll
        for (int s = 0; s < NSEG; ++s)
#pragma unroll
            for (int m = 0; m < CH; ++m) {
                const int idx = base + NL * m;
                if (idx < n) {
                    const float x = src[s] ? __uint_as_float((unsigned)v[s][m]) : 0.0f;
                    dst[(size_t)s * n + idx] = relu ? fmaxf(x, 0.0f) : x;
                }
            }
    }
}

// Sum over the lanes of a channel with DPP only (full-rate VALU, no LDS round trip): four steps inside each row
// of 16, then row_bcast15 adds row 0's total into row 1 (and row 2's into row 3), and for a channel that owns the
// whole wave row_bcast31 adds lane 31's total into rows 2-3.  The total is valid in the channel's last lane,
// which publishes.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xF, false));
}
template <int LPC>
__device__ __forceinline__ float chan_sum(float v) {   // sum over the LPC (32 or 64) lanes of a channel; valid in its last lane
    v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);   // row_mirror
    v = dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
    if constexpr (LPC == 64) v = dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// softmax + sampling + mu-law decode of the batch rows this wave owns (row b -> wave b): utils.py:13-46,
// mu_law_ops.py:26-31.  Every workgroup does it redundantly (identical bits everywhere); xh receives
// x_in(t+1) = mu_law_encode(decoded sample)  (wavenet.py:113).
__device__ __forceinline__ void decode_rows(const PArgs& a, int bi, int tid, int t, int it, float* xg, float* xh, const float* dtab) {
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
            const float dec = dtab[idx];                   // p_mu_dec(idx), p_mu_enc(that): tabulated once per launch
            xh[b * a.pre_k + ((t + 1) % a.pre_k)] = dtab[Q + 1 + idx];
            if (bi == 0) {
                if (a.audio) a.audio[(size_t)b * a.n_steps + it] = dec;
                if (a.indices) a.indices[(size_t)b * a.n_steps + it] = idx;
                if (last) a.prev[b] = dec;
            }
        }
    }
}

#ifndef ARP_POLL_SLEEP
#define ARP_POLL_SLEEP 12
#endif
#ifdef VQW_AR_TRACE
#define TR_DECL u64 tr_acc[16] = {0}; u64 tr_t = __builtin_amdgcn_s_memrealtime();
#define TR(i) { const u64 n_ = __builtin_amdgcn_s_memrealtime(); tr_acc[i] += n_ - tr_t; tr_t = n_; }
#define TR_DUMP if (a.trace && (tid & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) a.trace[((size_t)bi * 16 + (tid >> 6)) * 16 + i_] = tr_acc[i_];
#else
#define TR_DECL
#define TR(i)
#define TR_DUMP
#endif

// TB: compile-time bound of the batch rows; CPB channels per workgroup, i.e. LPC = 256/CPB lanes per channel;
// RLT = R/LPC rows of a weight column per lane; NS = S/R; KS taps.
// (Compile-time load counts let the compiler wait with exact vmcnt values: with a run-time count it falls back to
// vmcnt(0) in front of the first use, i.e. waits for the loads it has requested a moment earlier.)
template <int TB, int RLT, int NS, int KS, int CPB>
__global__ __launch_bounds__(NTHR, 1) void ar_persist_kernel(const PGroup grp) {
    const int nhandles = grp.n, nwg = gridDim.x / nhandles;
    const PArgs& a = grp.h[grp.by_handle ? blockIdx.x % nhandles : blockIdx.x / nwg];
    extern __shared__ float lds[];
    constexpr int LPC = NCT / CPB;                        // lanes per channel
    constexpr int PUBL = LPC - 1;                         // the lane that ends up with a channel's sums and publishes them
    constexpr int SL = RLT * NS;                          // S / LPC
    constexpr int NCOL = 5 + NS;                          // critical columns per weight row: Wg f,g | M f,g | Wr | Ws..
    constexpr int NPW = ((KS - 1) * RLT * 2 + 3) / 4;     // float4 groups per thread: past taps
    constexpr int NCW = (RLT * NCOL + 3) / 4;             // float4 groups per thread: critical columns
    const int bi = grp.by_handle ? blockIdx.x / nhandles : blockIdx.x % nwg, tid = threadIdx.x;
    const int role = tid < NCT ? 0 : (tid < NCT + NGF ? 1 : 2);   // 0 compute, 1 fresh, 2 history
    const int ct = tid & (NCT - 1);
    const int cg = ct / LPC, kl = ct % LPC;
    const int c = bi * CPB + cg;                          // compute thread's channel
    const int B = a.B, R = a.R, S = a.S, Q = a.Q, L = a.L, nQ = a.nQ;
    const Carve cv = make_carve(L, KS, R, S, Q, B, NS, nQ, a.pre_k, CPB);
    float* const xfresh = lds + cv.xfresh;                // [2][{cur_{p-1}, gated_{p-1}}][B][R]
    float* const xpast = lds + cv.xpast;                  // [3][KS-1][B][R]
    float* const hx1 = lds + cv.hx1;                      // relu(skip) [B][S], later the logits [B][Q]
    float* const hx2 = lds + cv.hx2;                      // relu(h) [B][S]
    const float* const hw_p1 = lds + cv.hw;
    const float* const hw_p2 = hw_p1 + cv.n_p1 * NCT;
    const float* const hw_pre = hw_p2 + cv.n_p2 * NCT;
    const float* const bs = lds + cv.bias;
    float* const condc = lds + cv.condc;                  // [L][2][8][B], then postprocess1 [NS][8][B]
    float* const xh = lds + cv.xh;                        // [B][pre_k]
    float* const misc = lds + cv.misc;
    int* const fail = reinterpret_cast<int*>(misc + 60);
    int* const tab = reinterpret_cast<int*>(lds + cv.tab);   // [L] dilation, [L] ring offset

    const int t0 = a.state[0];
    // ---------------- one-time LDS fill
    for (int i = tid; i < cv.nhead * NCT; i += NTHR) lds[cv.hw + i] = a.headw[(size_t)bi * cv.nhead * NCT + i];
    for (int i = tid; i < cv.nbias; i += NTHR) lds[cv.bias + i] = a.bias[(size_t)bi * cv.nbias + i];
    for (int i = tid; i < L; i += NTHR) { tab[i] = a.dil[i]; tab[L + i] = a.ring_off[i]; }
    for (int i = tid; i < B * a.pre_k; i += NTHR) xh[i] = a.xhist[i];
    if (tid == 0) *fail = 0;
    for (int i = tid; i <= Q; i += NTHR) {
        const float dec = p_mu_dec((float)i);
        lds[cv.dtab + i] = dec;
        lds[cv.dtab + Q + 1 + i] = p_mu_enc(dec);
    }
    __syncthreads();
    if (tid < B) xh[tid * a.pre_k + (t0 % a.pre_k)] = p_mu_enc(a.prev[tid]);   // x_in(t0) = mu_law_encode(previous sample)
    __syncthreads();

    const int PH = L + 4;                                 // tags of a step: gated_p (p < L), skip, h, logits
    const int nBR = B * R;
    TR_DECL

    if (role == 0) {
        // ======================================================================== compute waves
        f32x4 pw[NPW], cw[NCW];
        auto load_pw = [&](int p) {
            const f32x4* q = reinterpret_cast<const f32x4*>(a.pw) + ((size_t)(p * nwg + bi) * NPW) * NCT + ct;
#pragma unroll
            for (int i = 0; i < NPW; ++i) pw[i] = q[(size_t)i * NCT];
        };
        auto load_cw = [&](int p) {
            const f32x4* q = reinterpret_cast<const f32x4*>(a.cw) + ((size_t)(p * nwg + bi) * NCW) * NCT + ct;
#pragma unroll
            for (int i = 0; i < NCW; ++i) cw[i] = q[(size_t)i * NCT];
        };
        load_pw(0);
        load_cw(0);
        float cur[TB], skipacc[NS][TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            cur[b] = 0.0f;
#pragma unroll
            for (int j = 0; j < NS; ++j) skipacc[j][b] = 0.0f;
        }
        role_barrier();                                    // BAR_INIT: the history waves' prologue is in LDS
        __builtin_amdgcn_s_setprio(3);                     // the pollers sharing my SIMD never delay the critical chain
        int p = 0, it = 0, P = 0;
        bool stop = false;
        auto phase = [&]() {
            const int t = t0 + it;
            const unsigned seq = (unsigned)t * (unsigned)PH + 1u;
            const unsigned ttag = (unsigned)t + 1u;
            if (p == 0) {
                // ---------------- preprocess: causal conv over the encoded input history, lane kl = tap kl (+LPC)
                const int depth0 = (KS - 1) * tab[0] + 1;
                u64* ring0 = a.rings + tab[L] + (size_t)(t % depth0) * nBR;
                const int tm1 = t % a.pre_k + 1;           // slot of x_in(t - (pre_k-1-j)) = (t + 1 + j) mod pre_k, no division per tap
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    if (b < B) {
                        float acc = 0.0f;
                        for (int i = 0; i < cv.n_pre; ++i) {
                            const int j = kl + LPC * i;
                            if (j < a.pre_k) {
                                int slot = tm1 + j;
                                if (slot >= a.pre_k) slot -= a.pre_k;
                                acc = fmaf(hw_pre[i * NCT + ct], xh[b * a.pre_k + slot], acc);
                            }
                        }
                        cur[b] = chan_sum<LPC>(acc) + bs[cv.bias_head + (NS + nQ) * CPB + cg];
                        if (kl == PUBL) publish(ring0 + (size_t)b * R + c, ttag, cur[b]);
#pragma unroll
                        for (int j = 0; j < NS; ++j) skipacc[j][b] = 0.0f;
                    }
                }
            }
            TR(0)
            // ---------------- the old taps' share of the gate pre-activation (while the exchange is in flight)
            float pf[TB], pg[TB];
            {
                const float* xp = xpast + (P % 3) * cv.npast;
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    pf[b] = 0.0f; pg[b] = 0.0f;
                    if (b < B && p < L) {
#pragma unroll
                        for (int j = 0; j < KS - 1; ++j) {
#pragma unroll
                            for (int i = 0; i < RLT; ++i) {
                                const float xv = xp[(j * B + b) * R + kl + LPC * i];
                                const int e = (j * RLT + i) * 2;
                                pf[b] = fmaf(pw[e >> 2][e & 3], xv, pf[b]);
                                pg[b] = fmaf(pw[(e + 1) >> 2][(e + 1) & 3], xv, pg[b]);
                            }
                        }
                        if (TB > 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_pw(p + 1 > L ? 0 : p + 1);                // a whole phase to land
            __builtin_amdgcn_sched_barrier(0);
            const float* bl = bs + p * (NS + 3) * CPB;
            float cf[TB], cgv[TB];                         // bias + condition of my channel's filter / gate row
            u64* ringp = nullptr;                          // where cur_p(t) of my channel goes
            auto cond_bias = [&]() {
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    const int bb = (b < B) ? b : 0;
                    cf[b] = bl[cg] + condc[((p * 2 + 0) * CPB + cg) * B + bb];
                    cgv[b] = bl[CPB + cg] + condc[((p * 2 + 1) * CPB + cg) * B + bb];
                }
            };
#pragma unroll
            for (int b = 0; b < TB; ++b) { cf[b] = 0.0f; cgv[b] = 0.0f; }
            if (p < L) {
                if (p > 0) cond_bias();                    // (phase 0 reads them behind the barrier: the fresh waves may be
                const int depth = (KS - 1) * tab[p] + 1;   //  loading a new frame's projections right now)
                ringp = a.rings + tab[L + p] + (size_t)(t % depth) * nBR + c;
            }
            const float brv = bl[2 * CPB + cg];
            u64* const exgp = a.ex_g + (size_t)(P & 1) * nBR + c;
            TR(1)
            role_barrier();                                // BAR_A: xfresh[P&1] = {cur_{p-1}(t), gated_{p-1}(t)}
            TR(2)
            const float* vA = xfresh + (P & 1) * 2 * nBR;
            const float* vB = (p == 0) ? vA : vA + nBR;    // phase 0: skip = linear(cur_0) rides in the Ws slot
            if (p == 0) cond_bias();
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
                    float f = pf[b], g = pg[b], r = 0.0f, sk[NS];
#pragma unroll
                    for (int j = 0; j < NS; ++j) sk[j] = 0.0f;
#pragma unroll
                    for (int i = 0; i < RLT; ++i) {
                        const float xa = vA[b * R + kl + LPC * i], xv = vB[b * R + kl + LPC * i];
                        const int e = i * NCOL;
                        f = fmaf(cw[e >> 2][e & 3], xa, f);
                        g = fmaf(cw[(e + 1) >> 2][(e + 1) & 3], xa, g);
                        f = fmaf(cw[(e + 2) >> 2][(e + 2) & 3], xv, f);
                        g = fmaf(cw[(e + 3) >> 2][(e + 3) & 3], xv, g);
                        r = fmaf(cw[(e + 4) >> 2][(e + 4) & 3], xv, r);
#pragma unroll
                        for (int j = 0; j < NS; ++j) sk[j] = fmaf(cw[(e + 5 + j) >> 2][(e + 5 + j) & 3], xv, sk[j]);
                    }
                    f = chan_sum<LPC>(f);
                    g = chan_sum<LPC>(g);
                    r = chan_sum<LPC>(r);
                    cur[b] += r + brv;                     // net = net + (residual conv + bias)  (wavenet_ops.py:266)
                    if (kl == PUBL && p < L) {
                        const float vf = f + cf[b];
                        const float vg = g + cgv[b];
                        const float th = 1.0f - 2.0f * fast_rcp(__expf(2.0f * vf) + 1.0f);   // v_rcp_f32: 1 ulp
                        const float sg = fast_rcp(1.0f + __expf(-vg));
                        publish(exgp + (size_t)b * R, seq + p, th * sg);
                        if (p > 0) publish(ringp + (size_t)b * R, ttag, cur[b]);
                    }
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        skipacc[j][b] += chan_sum<LPC>(sk[j]) + bl[(3 + j) * CPB + cg];   // skip = skip + (skip conv + bias)
                        if (p == L && kl == PUBL) publish(a.ex_s + (size_t)b * S + c + (size_t)j * R, seq + L, skipacc[j][b]);
                    }
                    if (TB > 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            TR(3)
            __builtin_amdgcn_sched_barrier(0);             // publishes first: hipcc otherwise hoists the independent loads above them
#ifdef ARP_CW_SLEEP
            __builtin_amdgcn_s_sleep(ARP_CW_SLEEP);
#endif
            load_cw(p + 1 > L ? 0 : p + 1);                // the exchange's travel time to land
            __builtin_amdgcn_sched_barrier(0);
            TR(4)
            if (p == L) {
                // ---------------- postprocess1: relu(skip) -> 1x1 + condition  (wavenet.py:152-162)
                role_barrier();                            // BAR_H1: hx1 = relu(skip) [B][S]
                TR(5)
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    if (b < B) {
#pragma unroll
                        for (int j = 0; j < NS; ++j) {
                            float s = 0.0f;
#pragma unroll
                            for (int i = 0; i < SL; ++i) s = fmaf(hw_p1[(i * NS + j) * NCT + ct], hx1[b * S + kl + LPC * i], s);
                            s = chan_sum<LPC>(s);
                            if (kl == PUBL) {
                                const float v = s + bs[cv.bias_head + j * CPB + cg] + condc[((L * 2 + j) * CPB + cg) * B + b];
                                publish(a.ex_h + (size_t)b * S + c + (size_t)j * R, seq + L + 1, v);
                            }
                        }
                    }
                }
                TR(6)
                // ---------------- postprocess2: relu(h) -> logits  (wavenet.py:165-167)
                role_barrier();                            // BAR_H2: hx2 = relu(h) [B][S]
                TR(7)
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    if (b < B) {
                        for (int j = 0; j < nQ; ++j) {
                            float s = 0.0f;
#pragma unroll
                            for (int i = 0; i < SL; ++i) s = fmaf(hw_p2[(i * nQ + j) * NCT + ct], hx2[b * S + kl + LPC * i], s);
                            s = chan_sum<LPC>(s);
                            if (kl == PUBL) publish(a.ex_l + (size_t)b * Q + c + (size_t)j * R, seq + L + 2, s + bs[cv.bias_head + (NS + j) * CPB + cg]);
                        }
                    }
                }
                TR(8)
                role_barrier();                            // BAR_H3: hx1 = logits [B][Q]
                TR(9)
                decode_rows(a, bi, tid, t, it, hx1, xh, lds + cv.dtab);
                TR(10)
                if (*reinterpret_cast<volatile int*>(fail)) stop = true;   // stable between BAR_H3 and BAR_H4: nobody polls there
                role_barrier();                            // BAR_H4: xh holds x_in(t+1)
                TR(11)
            }
            ++P;
            if (++p > L) { p = 0; ++it; }
        };
        while (it < a.n_steps && !stop) phase();
    } else {
        // ======================================================================== gather waves
        const int gid = role == 1 ? tid - NCT : tid - NCT - NGF;
        // History role: the dilation-queue taps of global phase Pq (old news, but ~1.2 us of load latency each) are
        // requested in one phase and committed to xpast[Pq % 3] in the next one, so that the loads are in flight
        // across the phase barrier instead of holding it up (a raw s_barrier does not drain vmcnt).
        constexpr int HMAX = PB * 256 / NGH;               // granules per lane and tap (B <= PB, R <= 256)
        u64 hv[KS - 1][HMAX];
        const u64* hsrc[KS - 1];
        unsigned htag[KS - 1];
        int hP = -1;
        auto hist_issue = [&](int Pq) {
            hP = -1;
            const int itq = Pq / (L + 1), q = Pq - itq * (L + 1);
            if (itq >= a.n_steps || q >= L) return;
            hP = Pq;
            const int t = t0 + itq;
            const int dil = tab[q], depth = (KS - 1) * dil + 1;
            const u64* ring = a.rings + tab[L + q];
#pragma unroll
            for (int j = 0; j < KS - 1; ++j) {
                const int tau = t - (KS - 1 - j) * dil;
                hsrc[j] = (tau >= 0) ? ring + (size_t)(tau % depth) * nBR : nullptr;
                htag[j] = (unsigned)tau + 1u;
#pragma unroll
                for (int m = 0; m < HMAX; ++m) {
                    const int idx = gid + NGH * m;
                    hv[j][m] = (hsrc[j] && idx < nBR) ? __hip_atomic_load(hsrc[j] + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
            }
        };
        auto hist_commit = [&]() {
            if (hP < 0) return;
            float* dst = xpast + (hP % 3) * cv.npast;
#pragma unroll
            for (int j = 0; j < KS - 1; ++j)
#pragma unroll
                for (int m = 0; m < HMAX; ++m) {
                    const int idx = gid + NGH * m;
                    if (idx < nBR) {
                        float x = 0.0f;
                        if (hsrc[j]) x = __uint_as_float((unsigned)poll_granule(fail, hsrc[j] + idx, htag[j], hv[j][m]));
                        dst[(size_t)j * nBR + idx] = x;
                    }
                }
            hP = -1;
        };
        if (role == 2) {
            hist_issue(0); hist_commit();
            hist_issue(1); hist_commit();
            hist_issue(2);                                 // committed in phase 0, read by the compute waves in phase 1..2
        }
        role_barrier();                                    // BAR_INIT
        int p = 0, it = 0, P = 0, last_frame = -1;
        bool stop = false;
        while (it < a.n_steps && !stop) {
            const int t = t0 + it;
            const unsigned seq = (unsigned)t * (unsigned)PH + 1u;
            const unsigned ttag = (unsigned)t + 1u;
            if (role == 1) {
                if (p == 0) {
                    int frame = t / a.ratio;
                    if (frame >= a.Tz) frame = a.Tz - 1;
                    if (frame != last_frame) {   // condition projections of this frame for my 8 channels (all layers)
                        last_frame = frame;
                        const int per = CPB * B;
                        for (int i = gid; i < (L * 2 + NS) * per; i += NGF) {
                            const int b = i % B, g8 = (i / B) % CPB, lh = i / per;
                            const int ch = bi * CPB + g8;
                            float v;
                            if (lh < 2 * L) v = a.cond[lh >> 1][((size_t)b * 2 * R + ch + (size_t)(lh & 1) * R) * a.Tz + frame];
                            else v = a.cond[L][((size_t)b * S + ch + (size_t)(lh - 2 * L) * R) * a.Tz + frame];
                            condc[i] = v;
                        }
                    }
                }
                TR(0)
                // the input vectors of phase p: cur_{p-1}(t) from its dilation queue (phase 0: cur_0), gated_{p-1}(t)
                const int lq = (p == 0) ? 0 : p - 1;
                const int depth = (KS - 1) * tab[lq] + 1;
                const u64* src[2] = {a.rings + tab[L + lq] + (size_t)(t % depth) * nBR, a.ex_g + (size_t)((P + 1) & 1) * nBR};
                const unsigned tag[2] = {ttag, seq + (unsigned)lq};
                // nothing can arrive before the producers' critical chain + the store's travel time: polling earlier only
                // queues requests in front of the compute waves' weight loads (measured: 84 -> 75 us per sample; 8..20 are equivalent, 1 and 28+ slower)
                __builtin_amdgcn_s_sleep(ARP_POLL_SLEEP);
                if (p == 0) { if (TB > 1 && B >= 3) gather_segs<1, NGF, true>(gid, fail, src, tag, nBR, xfresh + (P & 1) * 2 * nBR, 0); else gather_segs<1, NGF, false>(gid, fail, src, tag, nBR, xfresh + (P & 1) * 2 * nBR, 0); }
                else { if (TB > 1 && B >= 3) gather_segs<2, NGF, true>(gid, fail, src, tag, nBR, xfresh + (P & 1) * 2 * nBR, 0); else gather_segs<2, NGF, false>(gid, fail, src, tag, nBR, xfresh + (P & 1) * 2 * nBR, 0); }
                TR(1)
            } else {
                hist_commit();                             // target P+2, requested one phase ago
                hist_issue(P + 3);
            }
            role_barrier();                                // BAR_A
            TR(2)
            if (p == L) {
                if (role == 1) {
                    const u64* src[1] = {a.ex_s};
                    const unsigned tag[1] = {seq + (unsigned)L};
                    { if (TB > 1 && B >= 3) gather_segs<1, NGF, true>(gid, fail, src, tag, B * S, hx1, 1); else gather_segs<1, NGF, false>(gid, fail, src, tag, B * S, hx1, 1); }
                }
                TR(3)
                role_barrier();                            // BAR_H1
                if (role == 1) {
                    const u64* src[1] = {a.ex_h};
                    const unsigned tag[1] = {seq + (unsigned)L + 1u};
                    { if (TB > 1 && B >= 3) gather_segs<1, NGF, true>(gid, fail, src, tag, B * S, hx2, 1); else gather_segs<1, NGF, false>(gid, fail, src, tag, B * S, hx2, 1); }
                }
                TR(4)
                role_barrier();                            // BAR_H2
                if (role == 1) {
                    const u64* src[1] = {a.ex_l};
                    const unsigned tag[1] = {seq + (unsigned)L + 2u};
                    { if (TB > 1 && B >= 3) gather_segs<1, NGF, true>(gid, fail, src, tag, B * Q, hx1, 0); else gather_segs<1, NGF, false>(gid, fail, src, tag, B * Q, hx1, 0); }
                }
                TR(5)
                role_barrier();                            // BAR_H3
                decode_rows(a, bi, tid, t, it, hx1, xh, lds + cv.dtab);
                if (*reinterpret_cast<volatile int*>(fail)) stop = true;
                role_barrier();                            // BAR_H4
                TR(6)
            }
            ++P;
            if (++p > L) { p = 0; ++it; }
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
        for (int i = tid; i < B * a.pre_k; i += NTHR) a.xhist[i] = xh[i];
        if (tid == 0) a.state[0] = t0 + a.n_steps;
    }
}

// ------------------------------------------------------------------------------------------------
// one-time re-blocking of the model's variables: every compute thread's share becomes contiguous.
// Phase p in 0..L: gate of layer p (p < L), residual + skip of layer p-1 (p > 0; phase 0: skip = linear(cur_0)).

// past-tap gate columns: dst[bi][q][ct][4]; element e = q*4+r = (tap*RL + i)*2 + h, taps 0..ks-2, RL = R/LPC rows per lane
__global__ void pack_pw_kernel(const float* __restrict__ gw, int ks, int R, int npw, int CPB, float* __restrict__ dst) {
    const int LPC = NCT / CPB, RL = R / LPC, nwg = R / CPB;
    const size_t n = (size_t)nwg * npw * NCT * 4;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(x & 3), ct = (int)((x >> 2) % NCT), q = (int)((x / (4 * NCT)) % npw), bi = (int)(x / ((size_t)4 * NCT * npw));
        const int e = q * 4 + r, h = e & 1, i = (e >> 1) % RL, j = (e >> 1) / RL;
        const int c = bi * CPB + ct / LPC, k = ct % LPC + LPC * i;
        dst[x] = (gw && j < ks - 1) ? gw[((size_t)j * R + k) * 2 * R + c + h * R] : 0.0f;
    }
}

// critical columns: dst[bi][q][ct][4]; e = i*(5+nS) + col, weight row k = kl + LPC i:
//   col 0,1: Wg^cur[k][c + h R]                      (gw: gate kernel [ks][R][2R] of layer p, or null)
//   col 2,3: M[k][c + h R] = sum_m Wr[k][m] Wg^cur[m][c + h R]   (fp64 accumulation; needs gw and ow)
//   col 4:   Wr[k][c] = ow[k][S + c]                 (ow: [R][ld] skip | residual of layer p-1, or null)
//   col 5+j: Ws[k][c + j R] = ow[k][c + j R]         (phase 0: sw = decoder/skip kernel [R][S], ld = S)
__global__ void pack_cw_kernel(const float* __restrict__ gw, const float* __restrict__ ow, const float* __restrict__ sw,
                               int ld, int ks, int R, int S, int nS, int ncw, int CPB, float* __restrict__ dst) {
    const int LPC = NCT / CPB, RL = R / LPC, nwg = R / CPB, ncol = 5 + nS;
    const size_t n = (size_t)nwg * ncw * NCT * 4;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(x & 3), ct = (int)((x >> 2) % NCT), q = (int)((x / (4 * NCT)) % ncw), bi = (int)(x / ((size_t)4 * NCT * ncw));
        const int e = q * 4 + r, i = e / ncol, col = e % ncol;
        const int c = bi * CPB + ct / LPC, k = ct % LPC + LPC * i;
        float v = 0.0f;
        if (i < RL) {
            const float* gc = gw ? gw + (size_t)(ks - 1) * R * 2 * R : nullptr;   // current tap [R][2R]
            if (col < 2) {
                if (gc) v = gc[(size_t)k * 2 * R + c + col * R];
            } else if (col < 4) {
                if (gc && ow) {
                    double acc = 0.0;
                    for (int m = 0; m < R; ++m) acc += (double)ow[(size_t)k * ld + S + m] * (double)gc[(size_t)m * 2 * R + c + (col - 2) * R];
                    v = (float)acc;
                }
            } else if (col == 4) {
                if (ow) v = ow[(size_t)k * ld + S + c];
            } else {
                const int j = col - 5;
                if (ow) v = ow[(size_t)k * ld + c + j * R];
                else if (sw) v = sw[(size_t)k * S + c + j * R];
            }
        }
        dst[x] = v;
    }
}

// biases of phase p for every workgroup: dst[bi][p*(3+nS)*8 + slot*8 + cg]
//   slot 0,1: bg[c + h R] + sum_m br[m] Wg^cur[m][c + h R]   (folded gate bias, fp64)
//   slot 2:   br[c] = ob[S + c];  slot 3+j: bs[c + j R] = ob[c + j R]  (phase 0: sb = decoder/skip bias)
__global__ void pack_bias_kernel(const float* __restrict__ gw, const float* __restrict__ gb, const float* __restrict__ ob,
                                 const float* __restrict__ sb, int ks, int R, int S, int nS, int nbias, int p, int CPB,
                                 float* __restrict__ dst) {
    const int nwg = R / CPB, nslot = 3 + nS;
    const int n = nwg * nslot * CPB;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
        const int cg = x % CPB, slot = (x / CPB) % nslot, bi = x / (CPB * nslot);
        const int c = bi * CPB + cg;
        float v = 0.0f;
        if (slot < 2) {
            if (gw && gb) {
                double acc = gb[c + slot * R];
                if (ob) {
                    const float* gc = gw + (size_t)(ks - 1) * R * 2 * R;
                    for (int m = 0; m < R; ++m) acc += (double)ob[S + m] * (double)gc[(size_t)m * 2 * R + c + slot * R];
                }
                v = (float)acc;
            }
        } else if (slot == 2) {
            if (ob) v = ob[S + c];
        } else {
            const int j = slot - 3;
            if (ob) v = ob[c + j * R];
            else if (sb) v = sb[c + j * R];
        }
        dst[(size_t)bi * nbias + p * nslot * CPB + slot * CPB + cg] = v;
    }
}

// head section: src [rows][ld], ncol columns per channel (c + j*R) -> dst[bi][e0 + e][ct], e = i*ncol + j, row kl + LPC i
__global__ void pack_head_kernel(const float* __restrict__ src, int ld, int rows, int R, int ncol, float* __restrict__ dst,
                                 int e0, int nhead, int CPB) {
    const int LPC = NCT / CPB, nwg = R / CPB, nsec = ((rows + LPC - 1) / LPC) * ncol;
    const size_t n = (size_t)nwg * nsec * NCT;
    for (size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x; x < n; x += (size_t)gridDim.x * blockDim.x) {
        const int ct = (int)(x % NCT), e = (int)((x / NCT) % nsec), bi = (int)(x / ((size_t)NCT * nsec));
        const int i = e / ncol, j = e % ncol;
        const int c = bi * CPB + ct / LPC, k = ct % LPC + LPC * i;
        dst[((size_t)bi * nhead + e0 + e) * NCT + ct] = (k < rows) ? src[(size_t)k * ld + c + j * R] : 0.0f;
    }
}

// head biases: dst[bi][off + j*8 + cg] = src[c + j*R]
__global__ void pack_head_bias_kernel(const float* __restrict__ src, int R, int ncol, int nbias, int off, int CPB,
                                      float* __restrict__ dst) {
    const int nwg = R / CPB, n = nwg * ncol * CPB;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) {
        const int cg = x % CPB, j = (x / CPB) % ncol, bi = x / (CPB * ncol);
        dst[(size_t)bi * nbias + off + j * CPB + cg] = src[bi * CPB + cg + j * R];
    }
}

#define PHIPC(x)                                                                               \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) return vqw_set_error("%s failed: %s", #x, hipGetErrorString(e_)); \
    } while (0)

// channels per workgroup: 4 (a whole wave per channel, R/4 workgroups: half the weight stream per CU and half the
// dot-product length per lane) where R allows it and the chip has the CUs, else 8; VQW_AR_CPB overrides
int pick_cpb(int R, int cus, int want = 0) {
    int cpb = (R % 64 == 0 && R / 4 <= cus) ? 4 : 8;
    const char* env = getenv("VQW_AR_CPB");
    if (want == 4 || want == 8) cpb = want;
    else if (env && (env[0] == '4' || env[0] == '8')) cpb = env[0] - '0';
    if (cpb == 4 && R % 64) cpb = 8;
    return cpb;
}

template <int TB>
const void* pick_kernel(int R, int nS, int ks, int cpb) {
    const int RL = R / (NCT / cpb);
#define ARP_CASE(rl, ns, k, cp) \
    if (RL == rl && nS == ns && ks == k && cpb == cp) return reinterpret_cast<const void*>(ar_persist_kernel<TB, rl, ns, k, cp>)
    ARP_CASE(4, 2, 3, 4); ARP_CASE(4, 2, 2, 4); ARP_CASE(2, 2, 3, 4); ARP_CASE(1, 2, 3, 4);
    ARP_CASE(8, 2, 3, 8); ARP_CASE(4, 2, 3, 8); ARP_CASE(2, 2, 3, 8); ARP_CASE(1, 2, 3, 8); ARP_CASE(8, 1, 3, 8); ARP_CASE(4, 4, 3, 8);
    ARP_CASE(8, 2, 2, 8); ARP_CASE(1, 2, 2, 8);
#undef ARP_CASE
    return nullptr;   // other shapes run on the launch-per-phase path of ar_decode.hip
}

int device_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus;
}

}  // namespace

struct ArPersist {
    vqw_ar_weights w;
    int B = 0, nS = 0, nQ = 0, nwg = 0, cpb = 8;
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
    if (nQ < 1 || nQ > PJ || batch > PB || w->pre_k > 60 || w->n_layers < 3) return false;
    const int cus = device_cus();
    if (cus <= 0) return false;
    int cpb = pick_cpb(w->R, cus);
    if (!pick_kernel<1>(w->R, nS, w->kernel_size, cpb)) {
        cpb = 8;
        if (!pick_kernel<1>(w->R, nS, w->kernel_size, cpb)) return false;
    }
    const Carve cv = make_carve(w->n_layers, w->kernel_size, w->R, w->S, w->Q, batch, nS, nQ, w->pre_k, cpb);
    if ((size_t)cv.total * sizeof(float) > 160 * 1024) return false;
    return w->R / cpb <= cus;   // one resident workgroup per CU is what makes the spin-waits safe
}

int arp_create(ArPersist** out, const vqw_ar_weights* w, const int* dil, const float* const* gated_w,
               const float* const* gated_b, const float* const* out_w, const float* const* out_b, int batch, int cpb_want) {
    ArPersist* h = new ArPersist();
    h->w = *w;
    h->B = batch;
    const int L = w->n_layers, R = w->R, S = w->S, Q = w->Q, ks = w->kernel_size;
    const int nS = S / R, nQ = Q / R;
    int CPB = pick_cpb(R, device_cus(), cpb_want);
    if (!pick_kernel<1>(R, nS, ks, CPB)) CPB = 8;
    const int RL = R / (NCT / CPB), nwg = R / CPB;
    h->nS = nS; h->nQ = nQ; h->nwg = nwg; h->cpb = CPB;
    auto fail = [&](const char* m) { arp_destroy(h); return vqw_set_error("vqw_ar_decode_create(persistent): %s", m); };
    const Carve cv = make_carve(L, ks, R, S, Q, batch, nS, nQ, w->pre_k, CPB);
    PArgs& a = h->args;
    memset(&a, 0, sizeof(a));
    a.L = L; a.R = R; a.S = S; a.Q = Q; a.B = batch; a.nQ = nQ; a.pre_k = w->pre_k;
    const int npw = ((ks - 1) * RL * 2 + 3) / 4, ncw = (RL * (5 + nS) + 3) / 4;
    // ---- per-phase weight columns, blocked per compute thread; biases; head
    const size_t pstride = (size_t)nwg * npw * NCT * 4, cstride = (size_t)nwg * ncw * NCT * 4;
    const size_t nheadw = (size_t)nwg * cv.nhead * NCT;
    float* pw = (float*)pmalloc(h, (L + 1) * pstride * sizeof(float));
    float* cw = (float*)pmalloc(h, (L + 1) * cstride * sizeof(float));
    float* hwd = (float*)pmalloc(h, nheadw * sizeof(float));
    float* db = (float*)pmalloc(h, (size_t)nwg * cv.nbias * sizeof(float));
    if (!pw || !cw || !hwd || !db) return fail("hipMalloc failed");
    for (int p = 0; p <= L; ++p) {
        const float* gw = (p < L) ? gated_w[p] : nullptr;
        const float* gb = (p < L) ? gated_b[p] : nullptr;
        const float* ow = (p > 0) ? out_w[p - 1] : nullptr;
        const float* ob = (p > 0) ? out_b[p - 1] : nullptr;
        hipLaunchKernelGGL(pack_pw_kernel, dim3(256), dim3(256), 0, 0, gw, ks, R, npw, CPB, pw + p * pstride);
        hipLaunchKernelGGL(pack_cw_kernel, dim3(512), dim3(256), 0, 0, gw, ow, (p == 0) ? w->skip0_w : nullptr, w->out_ld, ks, R, S, nS,
                           ncw, CPB, cw + p * cstride);
        hipLaunchKernelGGL(pack_bias_kernel, dim3(8), dim3(256), 0, 0, gw, gb, ob, (p == 0) ? w->skip0_b : nullptr, ks, R, S, nS, cv.nbias, p,
                           CPB, db);
    }
    int e0 = 0;
    hipLaunchKernelGGL(pack_head_kernel, dim3(256), dim3(256), 0, 0, w->post1_w, S, S, R, nS, hwd, e0, cv.nhead, CPB);
    e0 += cv.n_p1;
    hipLaunchKernelGGL(pack_head_kernel, dim3(256), dim3(256), 0, 0, w->post2_w, Q, S, R, nQ, hwd, e0, cv.nhead, CPB);
    e0 += cv.n_p2;
    hipLaunchKernelGGL(pack_head_kernel, dim3(64), dim3(256), 0, 0, w->pre_w, R, w->pre_k, R, 1, hwd, e0, cv.nhead, CPB);
    hipLaunchKernelGGL(pack_head_bias_kernel, dim3(8), dim3(256), 0, 0, w->post1_b, R, nS, cv.nbias, cv.bias_head, CPB, db);
    hipLaunchKernelGGL(pack_head_bias_kernel, dim3(8), dim3(256), 0, 0, w->post2_b, R, nQ, cv.nbias, cv.bias_head + nS * CPB, CPB, db);
    hipLaunchKernelGGL(pack_head_bias_kernel, dim3(8), dim3(256), 0, 0, w->pre_b, R, 1, cv.nbias, cv.bias_head + (nS + nQ) * CPB, CPB, db);
    a.pw = pw; a.cw = cw; a.headw = hwd; a.bias = db;
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
    struct { u64** p; size_t n; } exs[] = {{&a.ex_g, (size_t)2 * batch * R}, {&a.ex_s, (size_t)batch * S},
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
    h->kfn = (batch <= 1) ? pick_kernel<1>(R, nS, ks, CPB) : pick_kernel<PB>(R, nS, ks, CPB);
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

int arp_workgroups(const ArPersist* h) { return h->nwg; }

bool arp_same_launch(const ArPersist* x, const ArPersist* y) {
    return x->kfn == y->kfn && x->nwg == y->nwg && x->lds_bytes == y->lds_bytes;
}

int arp_run(ArPersist* const* hs, int n, const float* const* const* condenc, int Tz, int ratio, int n_steps, int mode,
            const float* const* uniforms, float* const* audio, int32_t* const* indices, float* const* probs_last,
            hipStream_t st) {
    if (n < 1 || n > PGROUP) return vqw_set_error("vqw_ar_decode_run: 1..%d handles per launch (got %d)", PGROUP, n);
    const int cus = device_cus();
    if (n * hs[0]->nwg > cus)
        return vqw_set_error("vqw_ar_decode_run: %d handles x %d workgroups do not fit %d CUs (one resident workgroup per CU)",
                             n, hs[0]->nwg, cus);
    PGroup g;
    memset(&g, 0, sizeof(g));
    g.n = n;
    {
        const char* env = getenv("VQW_AR_PLACE");      // "handle" / "channel" (measurements: tools/ar_layouts.py)
        g.by_handle = (env && env[0] == 'h') ? 1 : 0;
    }
    for (int i = 0; i < n; ++i) {
        ArPersist* h = hs[i];
        if (!arp_same_launch(hs[0], h)) return vqw_set_error("vqw_ar_decode_run: handles of one launch must share one kernel");
        const int L = h->w.n_layers;
        PHIPC(hipMemcpyAsync(h->dcond, condenc[i], (L + 1) * sizeof(float*), hipMemcpyHostToDevice, st));
        PArgs a = h->args;
        a.Tz = Tz; a.ratio = ratio; a.mode = mode; a.n_steps = n_steps;
        a.uniforms = uniforms ? uniforms[i] : nullptr; a.audio = audio[i];
        a.indices = indices ? indices[i] : nullptr; a.probs_last = probs_last ? probs_last[i] : nullptr;
#ifdef VQW_AR_TRACE
        if (!h->trace) h->trace = (u64*)pmalloc(h, (size_t)h->nwg * 16 * 16 * sizeof(u64));
        a.trace = h->trace;
        h->trace_steps = n_steps;
#endif
        g.h[i] = a;
    }
    void* params[] = {&g};
    PHIPC(hipLaunchKernel(hs[0]->kfn, dim3(hs[0]->nwg * n), dim3(NTHR), params, hs[0]->lds_bytes, st));
    return 0;
}

int arp_error(ArPersist* h, hipStream_t st) {
    int s2[2] = {0, 0};
    if (hipMemcpyAsync(s2, h->args.state, sizeof(s2), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
#ifdef VQW_AR_TRACE
    if (h->trace && getenv("VQW_AR_TRACE_PRINT")) {
        const int nw = h->nwg * 16;
        std::vector<u64> t((size_t)nw * 16);
        (void)hipMemcpy(t.data(), h->trace, t.size() * sizeof(u64), hipMemcpyDeviceToHost);
        const int picks[3] = {0, 4, 4 + NGF / 64};
        const char* names[3] = {"compute", "fresh", "history"};
        for (int k = 0; k < 3; ++k) {
            const int w = picks[k];
            fprintf(stderr, "[ar-trace] wave %3d (%s): us/sample by section:", w, names[k]);
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

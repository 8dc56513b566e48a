// Persistent fast-generation kernel for gfx950: the whole sample loop of generate.py:103-113
// (216 tf.matmul + 91 FIFOQueue ops + numpy sampling per sample in the reference,
// wavenet.py:103-172 / wavenet_ops.py:147-267 / utils.py:13-46) in ONE launch.
//
// Decomposition: R workgroups (one per CU; R = residual channels = 256), workgroup c owns
// column c of every matrix of the model (plus columns c+R, ... of the wider ones), so a sample
// is a chain of 2L+3 phases, each = "gather the full input vector, multiply by my columns,
// publish my outputs".  The exchange follows the R2 recipe of the CDNA guide (Guideline 16):
// every value travels as ONE naturally aligned 8-byte granule {tag, fp32 bits} written with a
// relaxed agent-scope atomic store (sc1, write-through) and polled with relaxed agent-scope
// atomic loads; the tag is a per-phase sequence number, so no flags, fences or resets are needed
// and nothing depends on dispatch order or XCD placement.  The per-layer exchange buffer of the
// layer input doubles as the dilation queue (ring of 2d+1 time slots).  Each workgroup's weight
// columns are stored contiguously ("blocked" copies made once per handle) and streamed from
// L2 / Infinity Cache while the workgroup polls.  Every spin is bounded by a wall-clock timeout
// (s_memrealtime) that sets an error word and drains the grid.
#include <string.h>

#include <vector>

#include "ar_persist.h"

namespace {

typedef unsigned long long u64;
constexpr int PB = 8;        // max batch rows
constexpr int PJ = 8;        // max columns per workgroup in one phase
constexpr unsigned long long TIMEOUT_TICKS = 300000000ull;  // 3 s of s_memrealtime (100 MHz)

struct PLayer {
    const float* gw;    // [R][ks][R][2]   blocked gate kernel: (filter col c, gate col c+R)
    const float* gb;    // [R][2]
    const float* ow;    // [R][R][nS+1]    skip cols c+j*R (j<nS), then residual col c
    const float* ob;    // [R][nS+1]
    const float* cond;  // [B][2R][Tz]     this run's condition projections
    u64* ex_cur;        // [depth][B][R]   layer input granules == dilation queue
    int dil, depth;
};

struct PArgs {
    int L, ks, R, S, Q, B, nS, nQ, pre_k;
    const PLayer* layers;
    const float *prew, *preb;  // [R][pre_k], [R]
    const float *s0w, *s0b;    // [R][R][nS], [R][nS]
    const float *p1w, *p1b;    // [R][S][nS], [R][nS]
    const float *p2w, *p2b;    // [R][S][nQ], [R][nQ]
    const float* cond1;        // [B][S][Tz]
    u64 *ex_g, *ex_s, *ex_h, *ex_l;  // [B][R], [B][S], [B][S], [B][Q]
    float* xhist;              // [B][pre_k] encoded input history (state across runs)
    float* prev;               // [B] last decoded sample
    int* state;                // [0] = step, [1] = error
    int Tz, ratio, mode, n_steps;
    const float* uniforms;
    float* audio;
    int* indices;
    float* probs_last;
};

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

struct Ctx {
    int tid, lane, wv;
    float* red;      // [4][PJ*PB]
    float* res;      // [PJ*PB]
    int* fail;       // LDS flag
};

// Every thread polls the granules it is responsible for; dst[i] (LDS) receives the values.
// relu != 0 applies max(.,0) while storing.  The wall-clock timeout is only consulted every 256
// polls (s_memrealtime is an SMEM round trip of its own).  Callers __syncthreads() afterwards.
__device__ __forceinline__ u64 poll_granule(const Ctx& c, const u64* p, unsigned tag, u64 g) {
    if ((unsigned)(g >> 32) == tag) return g;
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(2);   // 255 other workgroups poll too: back off (guide: polling-cost)
        g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(g >> 32) == tag) return g;
        if ((spins & 255u) == 0 &&
            (__builtin_amdgcn_s_memrealtime() - t_start > TIMEOUT_TICKS || *reinterpret_cast<volatile int*>(c.fail))) {
            *c.fail = 1;
            return g;
        }
    }
}

__device__ __forceinline__ void gather(const Ctx& c, const u64* src, int n, unsigned tag, float* dst, int relu) {
    for (int i = c.tid; i < n; i += 256) {
        u64 g = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g = poll_granule(c, src + i, tag, g);
        const float v = __uint_as_float((unsigned)g);
        dst[i] = relu ? fmaxf(v, 0.0f) : v;
    }
}

// The ks taps of a layer input at once: all first loads are issued together (the past taps are
// already there), only what is missing gets polled.  tau < 0 reads as zero (zero-filled queues).
__device__ __forceinline__ void gather_taps(const Ctx& c, const u64* ring, int depth, int n, int ks, int dil, int t,
                                            float* dst) {
    for (int i = c.tid; i < n; i += 256) {
        u64 g[VQW_MAX_TAPS];
#pragma unroll
        for (int j = 0; j < VQW_MAX_TAPS; ++j) {
            const int tau = t - (ks - 1 - j) * dil;
            g[j] = (j < ks && tau >= 0)
                       ? __hip_atomic_load(ring + (size_t)(tau % depth) * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : 0ull;
        }
#pragma unroll
        for (int j = 0; j < VQW_MAX_TAPS; ++j) {
            if (j < ks) {
                const int tau = t - (ks - 1 - j) * dil;
                float v = 0.0f;
                if (tau >= 0) {
                    const u64 gg = poll_granule(c, ring + (size_t)(tau % depth) * n + i, (unsigned)tau + 1u, g[j]);
                    v = __uint_as_float((unsigned)gg);
                }
                dst[(size_t)j * n + i] = v;
            }
        }
    }
}

// Block-wide sums of per-thread partials v[j][b] (j < nj, b < B); results land in c.res[j*PB + b].
template <int NJ>
__device__ __forceinline__ void block_sum(const Ctx& c, float (&v)[NJ][PB], int nj, int B) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            if (j < nj && b < B) {
                float s = v[j][b];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                if (c.lane == 0) c.red[c.wv * (PJ * PB) + j * PB + b] = s;
            }
        }
    }
    __syncthreads();
    if (c.tid < NJ * PB && c.tid / PB < nj && c.tid % PB < B)
        c.res[c.tid] = (c.red[c.tid] + c.red[PJ * PB + c.tid]) + (c.red[2 * PJ * PB + c.tid] + c.red[3 * PJ * PB + c.tid]);
    __syncthreads();
}

__global__ __launch_bounds__(256, 1) void ar_persist_kernel(const PArgs a) {
    extern __shared__ float lds[];
    const int c0 = blockIdx.x, tid = threadIdx.x;
    const int B = a.B, R = a.R, S = a.S, Q = a.Q, L = a.L, ks = a.ks, nS = a.nS, nQ = a.nQ;
    // LDS carve
    float* xs = lds;                                   // [max(ks*B*R, B*S, B*Q)]
    int xmax = ks * B * R;
    if (B * S > xmax) xmax = B * S;
    if (B * Q > xmax) xmax = B * Q;
    float* red = xs + xmax;                            // [4][PJ*PB]
    float* res = red + 4 * PJ * PB;                    // [PJ*PB]
    float* xh = res + PJ * PB;                         // [B][pre_k]
    float* misc = xh + B * a.pre_k;                    // [64]
    int* fail = reinterpret_cast<int*>(misc + 60);
    Ctx cx{tid, tid & 63, tid >> 6, red, res, fail};
    if (tid == 0) *fail = 0;
    for (int i = tid; i < B * a.pre_k; i += 256) xh[i] = a.xhist[i];
    __syncthreads();

    const int t0 = a.state[0];
    const int PH = 2 * L + 4;
    float cur[PB], skipacc[4][PB], prevs[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) prevs[b] = (b < B) ? a.prev[b] : 0.0f;

    for (int it = 0; it < a.n_steps; ++it) {
        const int t = t0 + it;
        const unsigned seq = (unsigned)t * (unsigned)PH + 1u;
        const unsigned ttag = (unsigned)t + 1u;
        int frame = t / a.ratio;
        if (frame >= a.Tz) frame = a.Tz - 1;
        // ---------------- preprocess: x_in(t) = mu_law_encode(previous sample); causal k=pre_k conv, column c0
        if (tid < B) xh[tid * a.pre_k + (t % a.pre_k)] = p_mu_enc(prevs[tid]);
        __syncthreads();
        if (tid < B) {
            float acc = a.preb[c0];
            const float* w = a.prew + (size_t)c0 * a.pre_k;
            for (int j = 0; j < a.pre_k; ++j) {
                const int tau = t - (a.pre_k - 1 - j);
                const int slot = ((tau % a.pre_k) + a.pre_k) % a.pre_k;
                acc = fmaf(w[j], xh[tid * a.pre_k + slot], acc);
            }
            res[tid] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < PB; ++b) cur[b] = (b < B) ? res[b] : 0.0f;
        if (tid < B) publish(a.layers[0].ex_cur + ((size_t)(t % a.layers[0].depth) * B + tid) * R + c0, ttag, cur[tid]);
        __syncthreads();

        for (int l = 0; l < L; ++l) {
            const PLayer& ly = a.layers[l];
            // ---------------- gate phase: gather cur_l(t), cur_l(t-d), cur_l(t-2d) -> gated[c0]
            // weights first: they do not depend on the hop, so their L2 / Infinity-Cache round trip
            // overlaps the wait for the granules (R <= 256: one row per thread and tap)
            f32x2 gwr[VQW_MAX_TAPS];
            float swr[4];
            {
                const float* gw = ly.gw + (size_t)c0 * ks * R * 2;
#pragma unroll
                for (int j = 0; j < VQW_MAX_TAPS; ++j)
                    gwr[j] = (j < ks && tid < R) ? *reinterpret_cast<const f32x2*>(gw + ((size_t)j * R + tid) * 2) : f32x2{0, 0};
                const float* sw = a.s0w + (size_t)c0 * R * nS;
#pragma unroll
                for (int j = 0; j < 4; ++j) swr[j] = (l == 0 && j < nS && tid < R) ? sw[(size_t)tid * nS + j] : 0.0f;
            }
            for (int j = 0; j < ks; ++j) {   // oldest tap first: only the last one (tau = t) can still be in flight
                const int tau = t - (ks - 1 - j) * ly.dil;
                float* dst = xs + (size_t)j * B * R;
                if (tau >= 0) {
                    gather(cx, ly.ex_cur + (size_t)(tau % ly.depth) * B * R, B * R, (unsigned)tau + 1u, dst, 0);
                } else {
                    for (int i = tid; i < B * R; i += 256) dst[i] = 0.0f;   // queues start filled with zeros
                }
            }
            __syncthreads();
            {
                float acc[6][PB];   // [0]=filter, [1]=gate, [2..2+nS) = skip-start columns (layer 0 only)
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int b = 0; b < PB; ++b) acc[j][b] = 0.0f;
                if (tid < R) {
#pragma unroll
                    for (int j = 0; j < VQW_MAX_TAPS; ++j) {
                        if (j < ks) {
#pragma unroll
                            for (int b = 0; b < PB; ++b) {
                                if (b < B) {
                                    const float xv = xs[((size_t)j * B + b) * R + tid];
                                    acc[0][b] = fmaf(gwr[j][0], xv, acc[0][b]);
                                    acc[1][b] = fmaf(gwr[j][1], xv, acc[1][b]);
                                }
                            }
                        }
                    }
                    if (l == 0) {   // skip = linear(current)  (wavenet.py:127-128)
                        const float* xc = xs + (size_t)(ks - 1) * B * R;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int b = 0; b < PB; ++b)
                                if (j < nS && b < B) acc[2 + j][b] = fmaf(swr[j], xc[(size_t)b * R + tid], acc[2 + j][b]);
                    }
                }
                block_sum<6>(cx, acc, l == 0 ? 2 + nS : 2, B);
            }
            if (tid < B) {
                const int b = tid;
                const float* cb = ly.cond + ((size_t)b * 2 * R) * a.Tz + frame;
                const float vf = res[b] + ly.gb[c0 * 2] + cb[(size_t)c0 * a.Tz];
                const float vg = res[PB + b] + ly.gb[c0 * 2 + 1] + cb[(size_t)(R + c0) * a.Tz];
                const float th = 1.0f - 2.0f / (__expf(2.0f * vf) + 1.0f);
                const float sg = 1.0f / (1.0f + __expf(-vg));
                publish(a.ex_g + (size_t)b * R + c0, seq + 2 * l, th * sg);
            }
            if (l == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < PB; ++b)
                        skipacc[j][b] = (j < nS && b < B) ? res[(2 + j) * PB + b] + a.s0b[c0 * nS + j] : 0.0f;
            }
            __syncthreads();
            // ---------------- out phase: gather gated -> skip columns (private accumulators) + residual column
            const int ncol = (l + 1 < L) ? nS + 1 : nS;     // net of the last layer is unused
            float owr[5];
            {
                const float* ow = ly.ow + (size_t)c0 * R * (nS + 1);
#pragma unroll
                for (int j = 0; j < 5; ++j) owr[j] = (j < ncol && tid < R) ? ow[(size_t)tid * (nS + 1) + j] : 0.0f;
            }
            gather(cx, a.ex_g, B * R, seq + 2 * l, xs, 0);
            __syncthreads();
            {
                float acc[5][PB];
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int b = 0; b < PB; ++b) acc[j][b] = 0.0f;
                if (tid < R) {
#pragma unroll
                    for (int j = 0; j < 5; ++j)
#pragma unroll
                        for (int b = 0; b < PB; ++b)
                            if (j < ncol && b < B) acc[j][b] = owr[j] * xs[(size_t)b * R + tid];
                }
                block_sum<5>(cx, acc, ncol, B);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < PB; ++b)
                    if (j < nS && b < B) skipacc[j][b] += res[j * PB + b] + ly.ob[c0 * (nS + 1) + j];
            if (l + 1 < L) {
#pragma unroll
                for (int b = 0; b < PB; ++b)
                    if (b < B) cur[b] += res[nS * PB + b] + ly.ob[c0 * (nS + 1) + nS];
                const PLayer& nx = a.layers[l + 1];
                if (tid < B) publish(nx.ex_cur + ((size_t)(t % nx.depth) * B + tid) * R + c0, ttag, cur[tid]);
            }
            __syncthreads();
        }
        // ---------------- postprocess1: relu(skip) -> 1x1 + condition  (wavenet.py:152-162)
        if (tid < nS * B) {
            const int j = tid / B, b = tid % B;
            float v = 0.0f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int bb = 0; bb < PB; ++bb)
                    if (jj == j && bb == b) v = skipacc[jj][bb];
            publish(a.ex_s + (size_t)b * S + c0 + (size_t)j * R, seq + 2 * L, v);
        }
        float p1r[4][4];    // [row group i: k = tid + 256 i][column j]
        {
            const float* w = a.p1w + (size_t)c0 * S * nS;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = tid + 256 * i;
                    p1r[i][j] = (k < S && j < nS) ? w[(size_t)k * nS + j] : 0.0f;
                }
        }
        gather(cx, a.ex_s, B * S, seq + 2 * L, xs, 1);
        __syncthreads();
        {
            float acc[4][PB];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < PB; ++b) acc[j][b] = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = tid + 256 * i;
                if (k < S) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int b = 0; b < PB; ++b)
                            if (j < nS && b < B) acc[j][b] = fmaf(p1r[i][j], xs[(size_t)b * S + k], acc[j][b]);
                }
            }
            block_sum<4>(cx, acc, nS, B);
        }
        if (tid < nS * B) {
            const int j = tid / B, b = tid % B;
            const float v = res[j * PB + b] + a.p1b[c0 * nS + j] + a.cond1[((size_t)b * S + c0 + (size_t)j * R) * a.Tz + frame];
            publish(a.ex_h + (size_t)b * S + c0 + (size_t)j * R, seq + 2 * L + 1, v);
        }
        __syncthreads();
        // ---------------- postprocess2: relu(h) -> logits  (wavenet.py:165-167)
        float p2r[4][PJ];
        {
            const float* w = a.p2w + (size_t)c0 * S * nQ;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PJ; ++j) {
                    const int k = tid + 256 * i;
                    p2r[i][j] = (k < S && j < nQ) ? w[(size_t)k * nQ + j] : 0.0f;
                }
        }
        gather(cx, a.ex_h, B * S, seq + 2 * L + 1, xs, 1);
        __syncthreads();
        {
            float acc[PJ][PB];
#pragma unroll
            for (int j = 0; j < PJ; ++j)
#pragma unroll
                for (int b = 0; b < PB; ++b) acc[j][b] = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = tid + 256 * i;
                if (k < S) {
#pragma unroll
                    for (int j = 0; j < PJ; ++j)
#pragma unroll
                        for (int b = 0; b < PB; ++b)
                            if (j < nQ && b < B) acc[j][b] = fmaf(p2r[i][j], xs[(size_t)b * S + k], acc[j][b]);
                }
            }
            block_sum<PJ>(cx, acc, nQ, B);
        }
        if (tid < nQ * B) {
            const int j = tid / B, b = tid % B;
            publish(a.ex_l + (size_t)b * Q + c0 + (size_t)j * R, seq + 2 * L + 2, res[j * PB + b] + a.p2b[c0 * nQ + j]);
        }
        __syncthreads();
        // ---------------- softmax + decode (every workgroup redundantly: identical bits everywhere)
        gather(cx, a.ex_l, B * Q, seq + 2 * L + 2, xs, 0);
        __syncthreads();
        for (int b = 0; b < B; ++b) {
            float* lg = xs + (size_t)b * Q;
            float m = -INFINITY;
            int mi = 0x7fffffff;
            for (int q = tid; q < Q; q += 256) {
                const float v = lg[q];
                if (v > m) { m = v; mi = q; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float om = __shfl_xor(m, o);
                const int oi = __shfl_xor(mi, o);
                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
            }
            if (cx.lane == 0) { red[cx.wv] = m; reinterpret_cast<int*>(red)[8 + cx.wv] = mi; }
            __syncthreads();
            m = red[0]; mi = reinterpret_cast<int*>(red)[8];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float om = red[w];
                const int oi = reinterpret_cast<int*>(red)[8 + w];
                if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
            }
            __syncthreads();
            float s = 0.0f;
            for (int q = tid; q < Q; q += 256) {
                const float e = __expf(lg[q] - m);
                lg[q] = e;
                s += e;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (cx.lane == 0) red[cx.wv] = s;
            __syncthreads();
            s = (red[0] + red[1]) + (red[2] + red[3]);
            const float inv = 1.0f / s;
            for (int q = tid; q < Q; q += 256) {
                const float p = lg[q] * inv;
                lg[q] = p;
                if (c0 == 0 && a.probs_last && it == a.n_steps - 1) a.probs_last[(size_t)b * Q + q] = p;
            }
            __syncthreads();
            if (tid == 0) {
                int idx;
                if (a.mode == 0) {
                    idx = mi;                                  // greedy: first maximum (np.argmax)
                } else {                                       // utils.py:20-25
                    const float u = a.uniforms[(size_t)b * a.n_steps + it];
                    float cdf = 0.0f;
                    idx = 0;
                    for (int q = 0; q < Q; ++q) {
                        cdf += lg[q];
                        if (cdf < u) idx = q + 1;
                    }
                }
                const float dec = p_mu_dec((float)idx);
                misc[b] = dec;
                if (c0 == 0) {
                    if (a.audio) a.audio[(size_t)b * a.n_steps + it] = dec;
                    if (a.indices) a.indices[(size_t)b * a.n_steps + it] = idx;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int b = 0; b < PB; ++b) prevs[b] = (b < B) ? misc[b] : 0.0f;
        if (*reinterpret_cast<volatile int*>(fail)) break;
        __syncthreads();
    }
    // ---------------- save state (workgroup 0) / report a timeout
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(fail)) {
        if (tid == 0) atomicExch(a.state + 1, 1);
        return;
    }
    if (c0 == 0) {
        for (int i = tid; i < B * a.pre_k; i += 256) a.xhist[i] = xh[i];
        if (tid < B) a.prev[tid] = prevs[tid];
        if (tid == 0) a.state[0] = t0 + a.n_steps;
    }
}

// dst[c][k][j] = src[k*ld + c + off[j]]   (one workgroup's columns made contiguous)
__global__ void blocked_copy_kernel(const float* __restrict__ src, int ld, int rows, int R, int ncol,
                                    const int* __restrict__ off, float* __restrict__ dst) {
    const size_t n = (size_t)R * rows * ncol;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % ncol);
        const int k = (int)((i / ncol) % rows);
        const int c = (int)(i / ((size_t)ncol * rows));
        dst[i] = src[(size_t)k * ld + c + off[j]];
    }
}

#define PHIPC(x)                                                                               \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) return vqw_set_error("%s failed: %s", #x, hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct ArPersist {
    vqw_ar_weights w;
    std::vector<int> dil;
    int B = 0, nS = 0, nQ = 0;
    std::vector<void*> allocs;
    std::vector<PLayer> hl;
    PLayer* dl = nullptr;
    PArgs args;
    size_t lds_bytes = 0;
    std::vector<std::pair<void*, size_t>> zero_on_reset;
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
    if (w->R < 1 || w->S % w->R || w->Q % w->R) return false;
    const int nS = w->S / w->R, nQ = w->Q / w->R;
    if (nS < 1 || nS > 4 || nQ < 1 || nQ > PJ || batch > PB || w->pre_k > 60) return false;
    if (w->R > 256 || w->S > 1024 || w->kernel_size > VQW_MAX_TAPS) return false;   // one weight row per thread and row group
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return w->R <= cus;   // one resident workgroup per CU is what makes the spin-waits safe
}

int arp_create(ArPersist** out, const vqw_ar_weights* w, const int* dil, const float* const* gated_w,
               const float* const* gated_b, const float* const* out_w, const float* const* out_b, int batch) {
    ArPersist* h = new ArPersist();
    h->w = *w;
    h->B = batch;
    const int L = w->n_layers, R = w->R, S = w->S, Q = w->Q, ks = w->kernel_size;
    const int nS = S / R, nQ = Q / R;
    h->nS = nS; h->nQ = nQ;
    h->dil.assign(dil, dil + L);
    auto fail = [&](const char* m) { arp_destroy(h); return vqw_set_error("vqw_ar_decode_create(persistent): %s", m); };
    // column offset tables
    int offs[4][PJ + 1];
    offs[0][0] = 0; offs[0][1] = R;                              // gate: filter, gate
    for (int j = 0; j < nS; ++j) offs[1][j] = j * R;             // out: skip cols..., residual col
    offs[1][nS] = S;
    for (int j = 0; j < nS; ++j) offs[2][j] = j * R;             // skip0 / post1
    for (int j = 0; j < nQ; ++j) offs[3][j] = j * R;             // post2
    int* doff = (int*)pmalloc(h, sizeof(offs));
    if (!doff) return fail("hipMalloc failed");
    if (hipMemcpy(doff, offs, sizeof(offs), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
    auto blocked = [&](const float* src, int ld, int rows, int ncol, int table) -> float* {
        float* dst = (float*)pmalloc(h, (size_t)R * rows * ncol * sizeof(float));
        if (!dst) return nullptr;
        hipLaunchKernelGGL(blocked_copy_kernel, dim3(512), dim3(256), 0, 0, src, ld, rows, R, ncol, doff + table * (PJ + 1), dst);
        return dst;
    };
    h->hl.resize(L);
    for (int l = 0; l < L; ++l) {
        PLayer& p = h->hl[l];
        // gate kernel [ks][R][2R]: per tap a [R][2R] matrix -> blocked [R][ks][R][2]
        float* gw = (float*)pmalloc(h, (size_t)R * ks * R * 2 * sizeof(float));
        if (!gw) return fail("hipMalloc failed");
        for (int j = 0; j < ks; ++j) {
            float* tmp = blocked(gated_w[l] + (size_t)j * R * 2 * R, 2 * R, R, 2, 0);   // [R][R][2]
            if (!tmp) return fail("hipMalloc failed");
            // scatter tap j of every workgroup: dst[c][j][k][2] <- tmp[c][k][2]
            if (hipMemcpy2DAsync(gw + (size_t)j * R * 2, (size_t)ks * R * 2 * sizeof(float), tmp, (size_t)R * 2 * sizeof(float),
                                 (size_t)R * 2 * sizeof(float), R, hipMemcpyDeviceToDevice, 0) != hipSuccess)
                return fail("hipMemcpy2DAsync failed");
        }
        p.gw = gw;
        p.gb = blocked(gated_b[l], 0, 1, 2, 0);
        p.ow = blocked(out_w[l], w->out_ld, R, nS + 1, 1);
        p.ob = blocked(out_b[l], 0, 1, nS + 1, 1);
        if (!p.gb || !p.ow || !p.ob) return fail("hipMalloc failed");
        p.dil = dil[l];
        p.depth = (ks - 1) * dil[l] + 1;
        const size_t ring = (size_t)p.depth * batch * R * sizeof(u64);
        p.ex_cur = (u64*)pmalloc(h, ring);
        if (!p.ex_cur) return fail("hipMalloc failed");
        h->zero_on_reset.push_back({p.ex_cur, ring});
        p.cond = nullptr;
    }
    h->dl = (PLayer*)pmalloc(h, L * sizeof(PLayer));
    if (!h->dl) return fail("hipMalloc failed");
    PArgs& a = h->args;
    memset(&a, 0, sizeof(a));
    a.L = L; a.ks = ks; a.R = R; a.S = S; a.Q = Q; a.B = batch; a.nS = nS; a.nQ = nQ; a.pre_k = w->pre_k;
    a.layers = h->dl;
    // preprocess kernel [pre_k][R] -> [R][pre_k]
    {
        std::vector<int> po(w->pre_k);
        (void)po;
        float* pw = (float*)pmalloc(h, (size_t)R * w->pre_k * sizeof(float));
        if (!pw) return fail("hipMalloc failed");
        // rows = pre_k, one column per workgroup: blocked_copy with ncol = 1 gives [R][pre_k][1]
        hipLaunchKernelGGL(blocked_copy_kernel, dim3(64), dim3(256), 0, 0, w->pre_w, R, w->pre_k, R, 1, doff, pw);
        a.prew = pw;
    }
    a.preb = w->pre_b;
    a.s0w = blocked(w->skip0_w, S, R, nS, 2);
    a.s0b = blocked(w->skip0_b, 0, 1, nS, 2);
    a.p1w = blocked(w->post1_w, S, S, nS, 2);
    a.p1b = blocked(w->post1_b, 0, 1, nS, 2);
    a.p2w = blocked(w->post2_w, Q, S, nQ, 3);
    a.p2b = blocked(w->post2_b, 0, 1, nQ, 3);
    if (!a.s0w || !a.s0b || !a.p1w || !a.p1b || !a.p2w || !a.p2b) return fail("hipMalloc failed");
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
    if (!a.xhist || !a.prev || !a.state) return fail("hipMalloc failed");
    h->zero_on_reset.push_back({a.xhist, (size_t)batch * w->pre_k * sizeof(float)});
    h->zero_on_reset.push_back({a.prev, batch * sizeof(float)});
    h->zero_on_reset.push_back({a.state, 2 * sizeof(int)});
    if (hipDeviceSynchronize() != hipSuccess) return fail("weight re-blocking failed");
    int xmax = ks * batch * R;
    if (batch * S > xmax) xmax = batch * S;
    if (batch * Q > xmax) xmax = batch * Q;
    h->lds_bytes = (size_t)(xmax + 5 * PJ * PB + batch * w->pre_k + 64) * sizeof(float);
    if (h->lds_bytes < 96 * 1024) h->lds_bytes = 96 * 1024;   // > half of the 160 KiB: one workgroup per CU
    if (h->lds_bytes > 160 * 1024) return fail("LDS budget exceeded");
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(ar_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h->lds_bytes) != hipSuccess)
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
    for (int l = 0; l < L; ++l) h->hl[l].cond = condenc[l];
    PHIPC(hipMemcpyAsync(h->dl, h->hl.data(), L * sizeof(PLayer), hipMemcpyHostToDevice, st));
    PArgs a = h->args;
    a.cond1 = condenc[L];
    a.Tz = Tz; a.ratio = ratio; a.mode = mode; a.n_steps = n_steps;
    a.uniforms = uniforms; a.audio = audio; a.indices = indices; a.probs_last = probs_last;
    hipLaunchKernelGGL(ar_persist_kernel, dim3(h->w.R), dim3(256), h->lds_bytes, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vqw_set_error("vqw_ar_decode_run(persistent): launch failed: %s", hipGetErrorString(e));
    return 0;
}

int arp_error(ArPersist* h, hipStream_t st) {
    int s2[2] = {0, 0};
    if (hipMemcpyAsync(s2, h->args.state, sizeof(s2), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    return s2[1];
}

void arp_destroy(ArPersist* h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

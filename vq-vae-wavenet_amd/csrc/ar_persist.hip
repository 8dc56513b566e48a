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
        u64 g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tau = t - (ks - 1 - j) * dil;
            g[j] = (j < ks && tau >= 0)
                       ? __hip_atomic_load(ring + (size_t)(tau % depth) * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
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

// Workgroup bi owns CPB = 8 consecutive channels; thread (cg = tid/32, kl = tid%32) works for channel
// c = bi*CPB + cg on rows k = kl + 32 i, so every dot product is finished by a 32-lane shuffle
// reduction and the per-channel state (cur, skip accumulators) lives in the kl == 0 thread.
// Fewer, fatter workgroups make every all-gather cheaper (R/8 producers instead of R).
constexpr int CPB = 8;
constexpr int RLMAX = 8;     // rows per lane for R <= 256
constexpr int KSMAX = 4;     // taps of the dilated conv handled by the persistent kernel

__device__ __forceinline__ float half_sum(float v) {   // sum over the 32 lanes of a half wave
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

template <int TB>   // compile-time bound of the batch rows (register arrays stay small for B = 1)
__global__ __launch_bounds__(256, 1) void ar_persist_kernel(const PArgs a) {
    extern __shared__ float lds[];
    const int bi = blockIdx.x, tid = threadIdx.x;
    const int cg = tid >> 5, kl = tid & 31;
    const int c = bi * CPB + cg;                       // this thread's channel
    const int B = a.B, R = a.R, S = a.S, Q = a.Q, L = a.L, ks = a.ks, nS = a.nS, nQ = a.nQ;
    const int RL = R / 32;                             // rows per lane (R % 32 == 0)
    // LDS carve
    float* xs = lds;                                   // [max(ks*B*R, B*S, B*Q)]
    int xmax = ks * B * R;
    if (B * S > xmax) xmax = B * S;
    if (B * Q > xmax) xmax = B * Q;
    float* red = xs + xmax;                            // scratch of the sampler
    float* xh = red + 5 * PJ * PB;                     // [B][pre_k]
    float* misc = xh + B * a.pre_k;                    // [64]
    int* fail = reinterpret_cast<int*>(misc + 60);
    Ctx cx{tid, tid & 63, tid >> 6, red, red + 4 * PJ * PB, fail};
    if (tid == 0) *fail = 0;
    for (int i = tid; i < B * a.pre_k; i += 256) xh[i] = a.xhist[i];
    __syncthreads();

    const int t0 = a.state[0];
    const int PH = 2 * L + 4;
    float cur[TB], skipacc[4][TB], prevs[TB];
#pragma unroll
    for (int b = 0; b < TB; ++b) { prevs[b] = (b < B) ? a.prev[b] : 0.0f; cur[b] = 0.0f; }

    for (int it = 0; it < a.n_steps; ++it) {
        const int t = t0 + it;
        const unsigned seq = (unsigned)t * (unsigned)PH + 1u;
        const unsigned ttag = (unsigned)t + 1u;
        int frame = t / a.ratio;
        if (frame >= a.Tz) frame = a.Tz - 1;
        // ---------------- preprocess: x_in(t) = mu_law_encode(previous sample); causal conv, lane kl = tap kl
        if (tid < B) xh[tid * a.pre_k + (t % a.pre_k)] = p_mu_enc(prevs[tid]);
        __syncthreads();
        {
            const float* w = a.prew + (size_t)c * a.pre_k;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
                    float acc = 0.0f;
                    for (int j = kl; j < a.pre_k; j += 32) {
                        const int tau = t - (a.pre_k - 1 - j);
                        const int slot = ((tau % a.pre_k) + a.pre_k) % a.pre_k;
                        acc = fmaf(w[j], xh[b * a.pre_k + slot], acc);
                    }
                    cur[b] = half_sum(acc) + a.preb[c];
                    if (kl == 0) publish(a.layers[0].ex_cur + ((size_t)(t % a.layers[0].depth) * B + b) * R + c, ttag, cur[b]);
                }
            }
        }

        for (int l = 0; l < L; ++l) {
            const PLayer& ly = a.layers[l];
            // ---------------- gate phase: weights first (their L2 / Infinity-Cache round trip overlaps the
            // wait for the granules), then gather cur_l(t), cur_l(t-d), cur_l(t-2d)
            f32x2 gwr[KSMAX][RLMAX];
            {
                const float* gw = ly.gw + (size_t)c * ks * R * 2;
#pragma unroll
                for (int j = 0; j < KSMAX; ++j)
#pragma unroll
                    for (int i = 0; i < RLMAX; ++i)
                        gwr[j][i] = (j < ks && i < RL) ? *reinterpret_cast<const f32x2*>(gw + ((size_t)j * R + kl + 32 * i) * 2)
                                                       : f32x2{0, 0};
            }
            const float gb0 = ly.gb[c * 2], gb1 = ly.gb[c * 2 + 1];
            float cf[TB], cgv[TB];      // condition projections of this frame: off the critical path
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const float* cb = ly.cond + ((size_t)(b < B ? b : 0) * 2 * R) * a.Tz + frame;
                cf[b] = cb[(size_t)c * a.Tz];
                cgv[b] = cb[(size_t)(R + c) * a.Tz];
            }
            gather_taps(cx, ly.ex_cur, ly.depth, B * R, ks, ly.dil, t, xs);   // all first loads in flight together
            __syncthreads();
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
                    float af = 0.0f, ag = 0.0f;
#pragma unroll
                    for (int j = 0; j < KSMAX; ++j) {
                        if (j < ks) {
#pragma unroll
                            for (int i = 0; i < RLMAX; ++i) {
                                if (i < RL) {
                                    const float xv = xs[((size_t)j * B + b) * R + kl + 32 * i];
                                    af = fmaf(gwr[j][i][0], xv, af);
                                    ag = fmaf(gwr[j][i][1], xv, ag);
                                }
                            }
                        }
                    }
                    af = half_sum(af);
                    ag = half_sum(ag);
                    if (l == 0) {   // skip = linear(current)  (wavenet.py:127-128)
                        const float* xc = xs + ((size_t)(ks - 1) * B + b) * R;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float s0 = 0.0f;
                            if (j < nS) {
#pragma unroll
                                for (int i = 0; i < RLMAX; ++i)
                                    if (i < RL) s0 = fmaf(a.s0w[((size_t)c * R + kl + 32 * i) * nS + j], xc[kl + 32 * i], s0);
                                s0 = half_sum(s0);
                            }
                            skipacc[j][b] = (j < nS) ? s0 + a.s0b[c * nS + j] : 0.0f;
                        }
                    }
                    if (kl == 0) {
                        const float vf = af + gb0 + cf[b];
                        const float vg = ag + gb1 + cgv[b];
                        const float th = 1.0f - 2.0f / (__expf(2.0f * vf) + 1.0f);
                        const float sg = 1.0f / (1.0f + __expf(-vg));
                        publish(a.ex_g + (size_t)b * R + c, seq + 2 * l, th * sg);
                    }
                }
            }
            // ---------------- out phase: skip columns (private accumulators) + residual column
            const int ncol = (l + 1 < L) ? nS + 1 : nS;     // net of the last layer is unused
            float owr[5][RLMAX], obr[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) obr[j] = (j < ncol) ? ly.ob[c * (nS + 1) + j] : 0.0f;
            {
                const float* ow = ly.ow + (size_t)c * R * (nS + 1);
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int i = 0; i < RLMAX; ++i)
                        owr[j][i] = (j < ncol && i < RL) ? ow[(size_t)(kl + 32 * i) * (nS + 1) + j] : 0.0f;
            }
            __syncthreads();   // every thread is done with xs (gate inputs)
            gather(cx, a.ex_g, B * R, seq + 2 * l, xs, 0);
            __syncthreads();
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        if (j < ncol) {
                            float s = 0.0f;
#pragma unroll
                            for (int i = 0; i < RLMAX; ++i)
                                if (i < RL) s = fmaf(owr[j][i], xs[(size_t)b * R + kl + 32 * i], s);
                            s = half_sum(s) + obr[j];
                            if (j < nS) {
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj)
                                    if (jj == j) skipacc[jj][b] += s;
                            } else {
                                cur[b] += s;
                            }
                        }
                    }
                    if (l + 1 < L && kl == 0) {
                        const PLayer& nx = a.layers[l + 1];
                        publish(nx.ex_cur + ((size_t)(t % nx.depth) * B + b) * R + c, ttag, cur[b]);
                    }
                }
            }
            __syncthreads();   // xs free again
        }
        // ---------------- postprocess1: relu(skip) -> 1x1 + condition  (wavenet.py:152-162)
        if (kl == 0) {
#pragma unroll
            for (int b = 0; b < TB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (b < B && j < nS) publish(a.ex_s + (size_t)b * S + c + (size_t)j * R, seq + 2 * L, skipacc[j][b]);
        }
        gather(cx, a.ex_s, B * S, seq + 2 * L, xs, 1);
        __syncthreads();
        {
            const float* w = a.p1w + (size_t)c * S * nS;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j < nS) {
                            float s = 0.0f;
                            for (int k = kl; k < S; k += 32) s = fmaf(w[(size_t)k * nS + j], xs[(size_t)b * S + k], s);
                            s = half_sum(s);
                            if (kl == 0) {
                                const float v = s + a.p1b[c * nS + j] + a.cond1[((size_t)b * S + c + (size_t)j * R) * a.Tz + frame];
                                publish(a.ex_h + (size_t)b * S + c + (size_t)j * R, seq + 2 * L + 1, v);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---------------- postprocess2: relu(h) -> logits  (wavenet.py:165-167)
        gather(cx, a.ex_h, B * S, seq + 2 * L + 1, xs, 1);
        __syncthreads();
        {
            const float* w = a.p2w + (size_t)c * S * nQ;
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (b < B) {
                    for (int j = 0; j < nQ; ++j) {
                        float s = 0.0f;
                        for (int k = kl; k < S; k += 32) s = fmaf(w[(size_t)k * nQ + j], xs[(size_t)b * S + k], s);
                        s = half_sum(s);
                        if (kl == 0) publish(a.ex_l + (size_t)b * Q + c + (size_t)j * R, seq + 2 * L + 2, s + a.p2b[c * nQ + j]);
                    }
                }
            }
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
                if (bi == 0 && a.probs_last && it == a.n_steps - 1) a.probs_last[(size_t)b * Q + q] = p;
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
                if (bi == 0) {
                    if (a.audio) a.audio[(size_t)b * a.n_steps + it] = dec;
                    if (a.indices) a.indices[(size_t)b * a.n_steps + it] = idx;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int b = 0; b < TB; ++b) prevs[b] = (b < B) ? misc[b] : 0.0f;
        if (*reinterpret_cast<volatile int*>(fail)) break;
        __syncthreads();
    }
    // ---------------- save state (workgroup 0) / report a timeout
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(fail)) {
        if (tid == 0) atomicExch(a.state + 1, 1);
        return;
    }
    if (bi == 0) {
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
    if (w->R > 256 || w->R % 32 || w->kernel_size > KSMAX) return false;   // R/32 <= 8 weight rows per lane, 8 channels per workgroup
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return w->R / 8 <= cus;   // one resident workgroup per CU is what makes the spin-waits safe
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
    const void* kfn[4] = {reinterpret_cast<const void*>(ar_persist_kernel<1>), reinterpret_cast<const void*>(ar_persist_kernel<2>),
                          reinterpret_cast<const void*>(ar_persist_kernel<4>), reinterpret_cast<const void*>(ar_persist_kernel<8>)};
    for (const void* f : kfn)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes) != hipSuccess)
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
    const dim3 grid(h->w.R / 8), block(256);
    if (h->B <= 1) hipLaunchKernelGGL(ar_persist_kernel<1>, grid, block, h->lds_bytes, st, a);
    else if (h->B <= 2) hipLaunchKernelGGL(ar_persist_kernel<2>, grid, block, h->lds_bytes, st, a);
    else if (h->B <= 4) hipLaunchKernelGGL(ar_persist_kernel<4>, grid, block, h->lds_bytes, st, a);
    else hipLaunchKernelGGL(ar_persist_kernel<8>, grid, block, h->lds_bytes, st, a);
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

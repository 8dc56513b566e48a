// Fast autoregressive WaveNet generation for gfx950 -- replaces the per-sample
// sess.run([predictions, push_ops]) of generate.py:103-113 and the graph built by
// wavenet.py:103-172 / wavenet_ops.py:147-267 (216 tf.matmul + 91 FIFOQueue ops per sample)
// and the host-side numpy sampling of utils.py:13-46.
//
// Design (round 1): every FIFOQueue becomes a device ring buffer indexed by a DEVICE-resident
// step counter, so one sample step is a fixed sequence of kernels with no per-step host
// arguments; that sequence is captured once into a hipGraph and replayed per sample.  The
// local-condition 1x1s (fast_condition, wavenet_ops.py:198-209) only change every `ratio`
// samples and are pre-computed for the whole utterance with the MFMA conv engine.
// Matrix-vector products stream the fp32 weights (L2 / Infinity-Cache resident after the
// first step) with 16-byte loads, 16 output columns per workgroup.
#include <string.h>

#include <vector>

#include "ar_persist.h"
#include "vqw_common.h"

namespace {

constexpr int MAXB = 8;

struct ArState {  // device resident
    int step;     // samples generated since reset
    int run_base; // value of step when the current run started
    int Tz, ratio, mode, n_steps;
    const float* uniforms;
    float* audio;
    int* indices;
    float* probs_last;
};

struct GemvSeg {
    const float* w;  // [K][ldw]
    const float* x;  // [B][K] (plain) or ring base [depth][B][K]
    int K, ldw;
    int depth;       // 0 = plain vector
    int phase;       // ring slot = (step + phase) % depth
};

enum { MODE_PLAIN = 0, MODE_GATE = 1, MODE_OUT = 2 };

struct GemvArgs {
    GemvSeg seg[VQW_MAX_TAPS];
    int nseg, B, N, H, S;      // N total columns; H gate half; S skip width (MODE_OUT)
    int in_relu;
    const float* bias;
    const float* cond;         // [B][N][Tz] or null
    float* out;                // PLAIN: [B][N]; GATE: gated [B][H]; OUT: skip [B][S]
    float* cur;                // OUT: [B][N-S] residual state (in place)
    float* ring_w;             // OUT: ring of this layer [depth][B][N-S]
    int ring_depth;
    const ArState* st;
};

__device__ __forceinline__ float mu_enc(float x) {
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    const float s = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    return s * log1pf(255.0f * fabsf(x)) / 5.5451774444795623f;
}
__device__ __forceinline__ float mu_dec(float idx) {
    const float y = 2.0f * idx / 255.0f - 1.0f;
    const float s = (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f);
    return s * (powf(256.0f, fabsf(y)) - 1.0f) / 255.0f;
}

// wavenet.py:113-124: x = mu_law_encode(input_t); fast_conv1d(k=pre_k, dilation 1, Cin=1).
__global__ void ar_pre_kernel(const ArState* st, const float* prev, float* xring, const float* pre_w,
                              const float* pre_b, float* cur, int pre_k, int R) {
    __shared__ float taps[64];
    const int b = blockIdx.x;
    const int step = st->step;
    float* ring = xring + (size_t)b * pre_k;
    const float x_new = mu_enc(prev[b]);
    if (threadIdx.x < pre_k) {
        const int j = threadIdx.x;                      // tap j multiplies x(t - (pre_k-1-j))
        const int tau = step - (pre_k - 1 - j);
        const int slot = ((tau % pre_k) + pre_k) % pre_k;
        taps[j] = (j == pre_k - 1) ? x_new : ring[slot];
        if (j == pre_k - 1) ring[slot] = x_new;         // push (the slot of t - pre_k, no longer needed)
    }
    __syncthreads();
    for (int o = threadIdx.x; o < R; o += blockDim.x) {
        float acc = pre_b[o];
#pragma unroll 8
        for (int j = 0; j < pre_k; ++j) acc = fmaf(pre_w[(size_t)j * R + o], taps[j], acc);
        cur[(size_t)b * R + o] = acc;
    }
}

// out[b][n] = sum_seg sum_k w[k][n] x[b][k]  (+bias +cond), 16 columns per block.
template <int MODE>
__global__ __launch_bounds__(256) void ar_gemv_kernel(const GemvArgs a) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    const int kl = tid >> 2, cq = tid & 3;
    const int step = a.st->step;
    const int B = a.B;
    // column base of this thread's 4 columns
    int col;
    if (MODE == MODE_GATE) {
        const int g0 = blockIdx.x * 8;
        col = (cq < 2) ? g0 + 4 * cq : a.H + g0 + 4 * (cq - 2);
    } else {
        col = blockIdx.x * 16 + 4 * cq;
    }
    const bool colok = (MODE == MODE_GATE) ? (blockIdx.x * 8 + 4 * (cq & 1) < a.H) : (col < a.N);

    float acc[MAXB][4];
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[b][e] = 0.0f;

    // The per-sample chain is latency-bound: stage the input vectors of ALL segments behind one
    // barrier, then issue every weight load of a segment before the first one is consumed
    // (K <= 256 per segment in the default model: 4 loads in flight per thread and segment).
    float* xs = sm;  // [seg][B][K]
    {
        int off = 0;
        for (int sgi = 0; sgi < a.nseg; ++sgi) {
            const GemvSeg& sg = a.seg[sgi];
            const float* xp = sg.x;
            if (sg.depth > 0) xp += (size_t)((step + sg.phase) % sg.depth) * B * sg.K;
            for (int i = tid; i < B * sg.K; i += 256) {
                const float v = xp[i];
                xs[off + i] = a.in_relu ? fmaxf(v, 0.0f) : v;
            }
            off += B * sg.K;
        }
    }
    __syncthreads();
    if (colok) {
        int off = 0;
        for (int sgi = 0; sgi < a.nseg; ++sgi) {
            const GemvSeg& sg = a.seg[sgi];
            const float* wc = sg.w + col;
            for (int k0 = 0; k0 < sg.K; k0 += 256) {
                f32x4 w4[4];
                int kc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + kl + 64 * u;
                    kc[u] = min(kk, sg.K - 1);
                    w4[u] = *reinterpret_cast<const f32x4*>(wc + (size_t)kc[u] * sg.ldw);
                    if (kk >= sg.K) w4[u] = f32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int b = 0; b < MAXB; ++b) {
                        if (b < B) {
                            const float xv = xs[off + b * sg.K + kc[u]];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[b][e] = fmaf(w4[u][e], xv, acc[b][e]);
                        }
                    }
                }
            }
            off += B * sg.K;
        }
    }
    // reduce over the 64 k-lanes: lanes of one wave hold kl = 16w..16w+15 (4 lanes each)
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        if (b < B) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[b][e];
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                acc[b][e] = v;
            }
        }
    }
    __syncthreads();
    float* red = sm;  // [4 waves][B][16]
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < 4) {
#pragma unroll
        for (int b = 0; b < MAXB; ++b)
            if (b < B)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[(wv * B + b) * 16 + 4 * lane + e] = acc[b][e];
    }
    __syncthreads();
    const bool worker = tid < 16 * B;  // one thread per (batch row, result column)
    const int b = worker ? tid / 16 : 0, c = tid % 16;
    float v = 0.0f;
    if (worker)
        v = red[(0 * B + b) * 16 + c] + red[(1 * B + b) * 16 + c] + red[(2 * B + b) * 16 + c] +
            red[(3 * B + b) * 16 + c];
    // global column of result slot c
    int n;
    if (MODE == MODE_GATE) n = (c < 8) ? blockIdx.x * 8 + c : a.H + blockIdx.x * 8 + (c - 8);
    else n = blockIdx.x * 16 + c;
    const bool ok = worker && ((MODE == MODE_GATE) ? (blockIdx.x * 8 + (c & 7) < a.H) : (n < a.N));
    if (ok) {
        if (a.bias) v += a.bias[n];
        if (a.cond) {
            int frame = step / a.st->ratio;
            if (frame >= a.st->Tz) frame = a.st->Tz - 1;
            v += a.cond[((size_t)b * a.N + n) * a.st->Tz + frame];
        }
    }
    if (MODE == MODE_GATE) {
        __syncthreads();  // every thread of the block reaches both barriers
        if (worker) red[b * 16 + c] = v;
        __syncthreads();
        if (c < 8 && ok) {
            const float vf = red[b * 16 + c], vg = red[b * 16 + c + 8];
            const float th = 1.0f - 2.0f / (__expf(2.0f * vf) + 1.0f);
            const float sg = 1.0f / (1.0f + __expf(-vg));
            a.out[(size_t)b * a.H + n] = th * sg;
        }
    } else if (MODE == MODE_OUT) {
        if (ok) {
            if (n < a.S) {
                a.out[(size_t)b * a.S + n] += v;  // skip += skip_out   (wavenet.py:142)
            } else {
                const int R = a.N - a.S, r = n - a.S;
                const float old = a.cur[(size_t)b * R + r];
                if (a.ring_w) a.ring_w[((size_t)(step % a.ring_depth) * B + b) * R + r] = old;  // push
                a.cur[(size_t)b * R + r] = old + v;  // current += res_out (wavenet.py:143)
            }
        }
    } else {
        if (ok) a.out[(size_t)b * a.N + n] = v;
    }
}

// softmax (wavenet.py:171) + utils.decode (utils.py:30-46) + mu_law_decode_np; advances the step.
__global__ __launch_bounds__(256) void ar_sample_kernel(ArState* st, const float* logits, float* probs,
                                                        float* prev, int B, int Q) {
    __shared__ float sp[1024];
    __shared__ float redv[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int step = st->step;
    const int i_out = step - st->run_base;
    for (int b = 0; b < B; ++b) {
        const float* lg = logits + (size_t)b * Q;
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
        __syncthreads();
        if (lane == 0) { redv[wv] = m; redi[wv] = mi; }
        __syncthreads();
        m = redv[0]; mi = redi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (redv[w] > m || (redv[w] == m && redi[w] < mi)) { m = redv[w]; mi = redi[w]; }
        float s = 0.0f;
        for (int q = tid; q < Q; q += 256) {
            const float e = __expf(lg[q] - m);
            sp[q] = e;
            s += e;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        __syncthreads();
        if (lane == 0) redv[wv] = s;
        __syncthreads();
        s = (redv[0] + redv[1]) + (redv[2] + redv[3]);
        const float inv = 1.0f / s;
        for (int q = tid; q < Q; q += 256) {
            const float p = sp[q] * inv;
            sp[q] = p;
            probs[(size_t)b * Q + q] = p;
            if (st->probs_last) st->probs_last[(size_t)b * Q + q] = p;
        }
        __syncthreads();
        if (tid == 0) {
            int idx;
            if (st->mode == 0) {
                idx = mi;  // greedy: np.argmax (first maximum)
            } else {
                // utils.py:20-25: cdf = cumsum(pdf) (sequential fp32), searchsorted(cdf, u, 'left')
                const float u = st->uniforms[(size_t)b * st->n_steps + i_out];
                float c = 0.0f;
                idx = 0;
                for (int q = 0; q < Q; ++q) {
                    c += sp[q];
                    if (c < u) idx = q + 1;
                }
            }
            const float dec = mu_dec((float)idx);
            if (st->audio) st->audio[(size_t)b * st->n_steps + i_out] = dec;
            if (st->indices) st->indices[(size_t)b * st->n_steps + i_out] = idx;
            prev[b] = dec;
        }
        __syncthreads();
    }
    if (tid == 0) st->step = step + 1;
}

}  // namespace

struct vqw_ar_decoder {
    vqw_ar_weights w;
    std::vector<int> dil;
    std::vector<const float*> gated_w, gated_b, cond_w, out_w, out_b;
    int B = 0;
    // device state
    ArState* st = nullptr;
    float *prev = nullptr, *xring = nullptr, *cur = nullptr, *gated = nullptr, *skip = nullptr,
          *h = nullptr, *logits = nullptr, *probs = nullptr;
    std::vector<float*> rings;   // per layer [depth][B][R]
    std::vector<float*> condenc; // per layer [B][2R][Tz]; last entry = postprocess1 [B][S][Tz]
    int cond_Tz_cap = 0;
    hipStream_t stream = nullptr;  // private stream (graph capture is illegal on the null stream)
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    bool use_graph = true;
    ArPersist* persist = nullptr;  // persistent single-launch generator (default model shapes)
    bool pending = false;          // a run has been enqueued and not yet waited for
    hipStream_t run_stream = nullptr;   // the stream that run was enqueued on (a group run: the first handle's)
};

namespace {

#define HIPC(x)                                                                       \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) return vqw_set_error("%s failed: %s", #x, hipGetErrorString(e_)); \
    } while (0)

int launch_step(vqw_ar_decoder* h, hipStream_t st) {
    const vqw_ar_weights& w = h->w;
    const int B = h->B, R = w.R, S = w.S, Q = w.Q, ks = w.kernel_size;
    hipLaunchKernelGGL(ar_pre_kernel, dim3(B), dim3(256), 0, st, h->st, h->prev, h->xring, w.pre_w, w.pre_b,
                       h->cur, w.pre_k, R);
    GemvArgs a;
    // skip = linear(current)  (wavenet.py:127-128)
    memset(&a, 0, sizeof(a));
    a.seg[0] = GemvSeg{w.skip0_w, h->cur, R, S, 0, 0};
    a.nseg = 1; a.B = B; a.N = S; a.bias = w.skip0_b; a.out = h->skip; a.st = h->st;
    hipLaunchKernelGGL((ar_gemv_kernel<MODE_PLAIN>), dim3(vqw_cdiv(S, 16)), dim3(256), B * R * sizeof(float) + 4 * MAXB * 16 * sizeof(float), st, a);
    for (int l = 0; l < w.n_layers; ++l) {
        const int d = h->dil[l];
        const int depth = (ks - 1) * d;
        // fast_gated_cnn (wavenet_ops.py:212-237)
        memset(&a, 0, sizeof(a));
        for (int j = 0; j < ks; ++j) {
            const float* wt = h->gated_w[l] + (size_t)j * R * 2 * R;
            if (j == ks - 1) a.seg[j] = GemvSeg{wt, h->cur, R, 2 * R, 0, 0};
            else a.seg[j] = GemvSeg{wt, h->rings[l], R, 2 * R, depth, j * d};  // x(t-(ks-1-j)d)
        }
        a.nseg = ks; a.B = B; a.N = 2 * R; a.H = R; a.bias = h->gated_b[l];
        a.cond = h->condenc[l]; a.out = h->gated; a.st = h->st;
        hipLaunchKernelGGL((ar_gemv_kernel<MODE_GATE>), dim3(vqw_cdiv(R, 8)), dim3(256), (size_t)ks * B * R * sizeof(float) + 4 * MAXB * 16 * sizeof(float), st, a);
        // skip / residual linears + accumulation + queue push (wavenet_ops.py:261-265, wavenet.py:142-143)
        memset(&a, 0, sizeof(a));
        a.seg[0] = GemvSeg{h->out_w[l], h->gated, R, w.out_ld, 0, 0};
        a.nseg = 1; a.B = B; a.N = S + R; a.S = S; a.bias = h->out_b[l];
        a.out = h->skip; a.cur = h->cur; a.ring_w = h->rings[l]; a.ring_depth = depth; a.st = h->st;
        hipLaunchKernelGGL((ar_gemv_kernel<MODE_OUT>), dim3(vqw_cdiv(S + R, 16)), dim3(256), B * R * sizeof(float) + 4 * MAXB * 16 * sizeof(float), st, a);
    }
    // postprocess1 (wavenet.py:152-162)
    memset(&a, 0, sizeof(a));
    a.seg[0] = GemvSeg{w.post1_w, h->skip, S, S, 0, 0};
    a.nseg = 1; a.B = B; a.N = S; a.in_relu = 1; a.bias = w.post1_b; a.cond = h->condenc[w.n_layers];
    a.out = h->h; a.st = h->st;
    hipLaunchKernelGGL((ar_gemv_kernel<MODE_PLAIN>), dim3(vqw_cdiv(S, 16)), dim3(256), B * S * sizeof(float) + 4 * MAXB * 16 * sizeof(float), st, a);
    // postprocess2 (wavenet.py:165-167)
    memset(&a, 0, sizeof(a));
    a.seg[0] = GemvSeg{w.post2_w, h->h, S, Q, 0, 0};
    a.nseg = 1; a.B = B; a.N = Q; a.in_relu = 1; a.bias = w.post2_b; a.out = h->logits; a.st = h->st;
    hipLaunchKernelGGL((ar_gemv_kernel<MODE_PLAIN>), dim3(vqw_cdiv(Q, 16)), dim3(256), B * S * sizeof(float) + 4 * MAXB * 16 * sizeof(float), st, a);
    hipLaunchKernelGGL(ar_sample_kernel, dim3(1), dim3(256), 0, st, h->st, h->logits, h->probs, h->prev, B, Q);
    VQW_LAUNCH_CHECK("vqw_ar_decode_run(step)");
    return 0;
}

void free_all(vqw_ar_decoder* h) {
    if (!h) return;
    if (h->persist) arp_destroy(h->persist);
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    for (float* p : h->rings) (void)hipFree(p);
    for (float* p : h->condenc) (void)hipFree(p);
    float* bufs[] = {reinterpret_cast<float*>(h->st), h->prev, h->xring, h->cur, h->gated, h->skip, h->h, h->logits, h->probs};
    for (float* p : bufs) (void)hipFree(p);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_out) (void)hipEventDestroy(h->ev_out);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

}  // namespace

extern "C" int vqw_ar_decode_create(vqw_ar_decoder** out, const vqw_ar_weights* w, int batch) {
    return vqw_ar_decode_create_ex(out, w, batch, 0);
}

extern "C" int vqw_ar_decode_create_ex(vqw_ar_decoder** out, const vqw_ar_weights* w, int batch, int channels_per_workgroup) {
    VQW_CHECK(out && w, "vqw_ar_decode_create: null pointer");
    VQW_CHECK(channels_per_workgroup == 0 || channels_per_workgroup == 4 || channels_per_workgroup == 8,
              "vqw_ar_decode_create_ex: channels_per_workgroup is 0 (auto), 4 or 8 (got %d)", channels_per_workgroup);
    VQW_CHECK(batch >= 1 && batch <= MAXB, "vqw_ar_decode_create: batch=%d must be in 1..%d", batch, MAXB);
    VQW_CHECK(w->n_layers >= 1 && w->kernel_size >= 2 && w->kernel_size <= VQW_MAX_TAPS, "vqw_ar_decode_create: bad layer config");
    VQW_CHECK(w->R % 16 == 0 && w->S % 16 == 0 && w->Q % 4 == 0 && w->Cc % 16 == 0, "vqw_ar_decode_create: R,S,Cc must be multiples of 16, Q of 4");
    VQW_CHECK(w->Q <= 1024 && w->pre_k <= 64, "vqw_ar_decode_create: Q must be <= 1024 and pre_k <= 64");
    VQW_CHECK(w->pre_k >= 1 && w->dilations && w->gated_w && w->gated_b && w->cond_w && w->out_w && w->out_b, "vqw_ar_decode_create: null weight table");
    vqw_ar_decoder* h = new vqw_ar_decoder();
    h->w = *w;
    h->B = batch;
    const int L = w->n_layers;
    h->dil.assign(w->dilations, w->dilations + L);
    h->gated_w.assign(w->gated_w, w->gated_w + L);
    h->gated_b.assign(w->gated_b, w->gated_b + L);
    h->cond_w.assign(w->cond_w, w->cond_w + L);
    h->out_w.assign(w->out_w, w->out_w + L);
    h->out_b.assign(w->out_b, w->out_b + L);
    h->w.dilations = nullptr; h->w.gated_w = nullptr; h->w.gated_b = nullptr; h->w.cond_w = nullptr;
    h->w.out_w = nullptr; h->w.out_b = nullptr;
    const char* env = getenv("VQW_AR_GRAPH");
    h->use_graph = !(env && env[0] == '0');
#define AR_ALLOC(ptr, n)                                                               \
    do {                                                                               \
        hipError_t e_ = hipMalloc((void**)&(ptr), (n) * sizeof(float));                \
        if (e_ != hipSuccess) { free_all(h); return vqw_set_error("vqw_ar_decode_create: hipMalloc failed: %s", hipGetErrorString(e_)); } \
    } while (0)
    const size_t B = batch;
    if (hipMalloc((void**)&h->st, sizeof(ArState)) != hipSuccess) { free_all(h); return vqw_set_error("vqw_ar_decode_create: hipMalloc failed"); }
    AR_ALLOC(h->prev, B);
    AR_ALLOC(h->xring, B * w->pre_k);
    AR_ALLOC(h->cur, B * w->R);
    AR_ALLOC(h->gated, B * w->R);
    AR_ALLOC(h->skip, B * w->S);
    AR_ALLOC(h->h, B * w->S);
    AR_ALLOC(h->logits, B * w->Q);
    AR_ALLOC(h->probs, B * w->Q);
    h->rings.assign(L, nullptr);
    for (int l = 0; l < L; ++l) AR_ALLOC(h->rings[l], (size_t)(w->kernel_size - 1) * h->dil[l] * B * w->R);
    h->condenc.assign(L + 1, nullptr);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming) != hipSuccess) {
        free_all(h);
        return vqw_set_error("vqw_ar_decode_create: stream/event creation failed");
    }
    if (arp_supported(w, batch)) {
        const int rc = arp_create(&h->persist, w, h->dil.data(), h->gated_w.data(), h->gated_b.data(), h->out_w.data(),
                                  h->out_b.data(), batch, channels_per_workgroup);
        if (rc) { free_all(h); return rc; }
    }
    *out = h;
    return vqw_ar_decode_reset(h, nullptr);
}

extern "C" int vqw_ar_decode_reset(vqw_ar_decoder* h, vqw_stream_t s) {
    VQW_CHECK(h, "vqw_ar_decode_reset: null handle");
    if (h->pending) {
        const int rcw = vqw_ar_decode_wait(h);
        if (rcw) return rcw;
    }
    hipStream_t st = (hipStream_t)s;
    const size_t B = h->B;
    if (h->persist) {
        const int rc = arp_reset(h->persist, st);
        if (rc) return rc;
    }
    HIPC(hipMemsetAsync(h->st, 0, sizeof(ArState), st));
    HIPC(hipMemsetAsync(h->prev, 0, B * sizeof(float), st));       // audio = zeros (generate.py:103)
    HIPC(hipMemsetAsync(h->xring, 0, B * h->w.pre_k * sizeof(float), st));
    for (int l = 0; l < h->w.n_layers; ++l)
        HIPC(hipMemsetAsync(h->rings[l], 0, (size_t)(h->w.kernel_size - 1) * h->dil[l] * B * h->w.R * sizeof(float), st));
    return 0;
}

extern "C" int vqw_ar_decode_wait(vqw_ar_decoder* h) {
    VQW_CHECK(h, "vqw_ar_decode_wait: null handle");
    if (!h->pending) return 0;
    h->pending = false;
    hipStream_t st = h->run_stream ? h->run_stream : h->stream;
    if (h->persist) {
        const int rc = arp_error(h->persist, st);
        if (rc) return vqw_set_error("vqw_ar_decode_run: persistent kernel %s", rc > 0 ? "timed out waiting for a workgroup" : "failed");
        return 0;
    }
    HIPC(hipStreamSynchronize(st));
    return 0;
}

extern "C" int vqw_ar_decode_run(vqw_ar_decoder* h, const float* encoding, int Tz, int ratio, int n_steps,
                                 int mode, const float* uniforms, float* audio, int32_t* indices,
                                 float* probs_last, vqw_stream_t s) {
    const int rc = vqw_ar_decode_run_async(h, encoding, Tz, ratio, n_steps, mode, uniforms, audio, indices, probs_last, s);
    if (rc) return rc;
    return vqw_ar_decode_wait(h);
}

extern "C" int vqw_ar_decode_workgroups(const vqw_ar_decoder* h) {
    return (h && h->persist) ? arp_workgroups(h->persist) : 0;
}

namespace {

// (re)allocate + compute the per-frame condition projections of one handle for this utterance, on `st`
int project_condition(vqw_ar_decoder* h, const float* encoding, int Tz, hipStream_t st) {
    const vqw_ar_weights& w = h->w;
    const int L = w.n_layers, B = h->B;
    if (Tz > h->cond_Tz_cap) {
        HIPC(hipStreamSynchronize(st));
        if (h->run_stream && h->run_stream != st) HIPC(hipStreamSynchronize(h->run_stream));
        for (int l = 0; l <= L; ++l) {
            if (h->condenc[l]) (void)hipFree(h->condenc[l]);
            const size_t n = (size_t)B * (l < L ? 2 * w.R : w.S) * Tz;
            HIPC(hipMalloc((void**)&h->condenc[l], n * sizeof(float)));
        }
        h->cond_Tz_cap = Tz;
        if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
        if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    }
    for (int l = 0; l <= L; ++l) {
        vqw_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.B = B; d.T_in = Tz; d.T_out = Tz; d.C0 = w.Cc; d.ntaps = 1; d.in_stride = 1;
        d.M = (l < L) ? 2 * w.R : w.S;
        d.ldw = (l < L) ? w.cond_ld : w.post1_cond_ld;
        d.epilogue = VQW_EPI_STORE;
        d.x0 = encoding; d.w = (l < L) ? h->cond_w[l] : w.post1_cond_w; d.out0 = h->condenc[l];
        const int rc = vqw_conv_gemm(&d, st);
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

extern "C" int vqw_ar_decode_run_async(vqw_ar_decoder* h, const float* encoding, int Tz, int ratio, int n_steps,
                                       int mode, const float* uniforms, float* audio, int32_t* indices,
                                       float* probs_last, vqw_stream_t s) {
    if (h && h->persist)
        return vqw_ar_decode_run_group_async(&h, 1, &encoding, Tz, ratio, n_steps, mode, uniforms ? &uniforms : nullptr, &audio,
                                             indices ? &indices : nullptr, probs_last ? &probs_last : nullptr, s);
    VQW_CHECK(h && encoding, "vqw_ar_decode_run: null pointer");
    if (h->pending) {
        const int rcw = vqw_ar_decode_wait(h);
        if (rcw) return rcw;
    }
    VQW_CHECK(Tz > 0 && ratio > 0 && n_steps > 0, "vqw_ar_decode_run: bad Tz/ratio/n_steps");
    VQW_CHECK(mode == 0 || (mode == 1 && uniforms), "vqw_ar_decode_run: mode must be 0 (greedy) or 1 (sample, needs uniforms)");
    hipStream_t user = (hipStream_t)s;
    hipStream_t st = h->stream;
    h->run_stream = st;
    HIPC(hipEventRecord(h->ev_in, user));
    HIPC(hipStreamWaitEvent(st, h->ev_in, 0));
    {
        const int rc = project_condition(h, encoding, Tz, st);
        if (rc) return rc;
    }
    // run parameters -> device state (step / run_base live on the device)
    ArState hs;
    memset(&hs, 0, sizeof(hs));
    hs.Tz = Tz; hs.ratio = ratio; hs.mode = mode; hs.n_steps = n_steps;
    hs.uniforms = uniforms; hs.audio = audio; hs.indices = indices; hs.probs_last = probs_last;
    // copy everything except `step`; run_base := step is done by a tiny device-side copy
    HIPC(hipMemcpyAsync(reinterpret_cast<char*>(h->st) + offsetof(ArState, Tz), reinterpret_cast<char*>(&hs) + offsetof(ArState, Tz),
                        sizeof(ArState) - offsetof(ArState, Tz), hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(&h->st->run_base, &h->st->step, sizeof(int), hipMemcpyDeviceToDevice, st));
    HIPC(hipStreamSynchronize(st));  // `hs` is a stack object; also orders the capture below

    if (h->use_graph && !h->gexec) {
        HIPC(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int rc = launch_step(h, st);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(st, &g);
        if (rc) return rc;
        if (e != hipSuccess) return vqw_set_error("vqw_ar_decode_run: graph capture failed: %s", hipGetErrorString(e));
        h->graph = g;
        HIPC(hipGraphInstantiate(&h->gexec, h->graph, nullptr, nullptr, 0));
    }
    for (int i = 0; i < n_steps; ++i) {
        if (h->gexec) {
            HIPC(hipGraphLaunch(h->gexec, st));
        } else {
            const int rc = launch_step(h, st);
            if (rc) return rc;
        }
    }
    h->pending = true;
    return 0;
}

extern "C" int vqw_ar_decode_run_group_async(vqw_ar_decoder* const* hs, int n, const float* const* encoding, int Tz,
                                             int ratio, int n_steps, int mode, const float* const* uniforms,
                                             float* const* audio, int32_t* const* indices,
                                             float* const* probs_last, vqw_stream_t s) {
    VQW_CHECK(hs && encoding && audio && n >= 1 && n <= 8, "vqw_ar_decode_run_group: null pointer or n=%d outside 1..8", n);
    VQW_CHECK(Tz > 0 && ratio > 0 && n_steps > 0, "vqw_ar_decode_run: bad Tz/ratio/n_steps");
    VQW_CHECK(mode == 0 || (mode == 1 && uniforms), "vqw_ar_decode_run: mode must be 0 (greedy) or 1 (sample, needs uniforms)");
    for (int i = 0; i < n; ++i) {
        VQW_CHECK(hs[i] && encoding[i] && audio[i] && (mode == 0 || uniforms[i]), "vqw_ar_decode_run_group: null pointer (handle %d)", i);
        VQW_CHECK(hs[i]->persist && arp_same_launch(hs[0]->persist, hs[i]->persist),
                  "vqw_ar_decode_run_group: every handle must run the same persistent kernel (start the others one by one)");
        for (int j = 0; j < i; ++j) VQW_CHECK(hs[j] != hs[i], "vqw_ar_decode_run_group: handle %d given twice", i);
        if (hs[i]->pending) {
            const int rcw = vqw_ar_decode_wait(hs[i]);
            if (rcw) return rcw;
        }
    }
    // everything is enqueued on the FIRST handle's stream, ordered after the work already in `s`; the caller's stream is
    // not made to wait (vqw_ar_decode_wait blocks the host instead and reports spin-wait timeouts)
    hipStream_t st = hs[0]->stream;
    HIPC(hipEventRecord(hs[0]->ev_in, (hipStream_t)s));
    HIPC(hipStreamWaitEvent(st, hs[0]->ev_in, 0));
    ArPersist* ps[8];
    const float* const* conds[8];
    for (int i = 0; i < n; ++i) {
        const int rc = project_condition(hs[i], encoding[i], Tz, st);
        if (rc) return rc;
        ps[i] = hs[i]->persist;
        conds[i] = hs[i]->condenc.data();
    }
    const int rc = arp_run(ps, n, conds, Tz, ratio, n_steps, mode, uniforms, audio, indices, probs_last, st);
    if (rc) return rc;
    for (int i = 0; i < n; ++i) {
        hs[i]->pending = true;
        hs[i]->run_stream = st;
    }
    return 0;
}

extern "C" int vqw_ar_decode_destroy(vqw_ar_decoder* h) {
    if (!h) return 0;
    if (h->pending) (void)vqw_ar_decode_wait(h);   // a group run lives on another handle's stream
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_all(h);
    return 0;
}

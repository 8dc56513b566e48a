/* CPU oracle, plain C -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Independent restatement of the bit-exact items of the hot path:
 *   - mu_law_encode to int   (reference mu_law_ops.py:5-15)
 *   - mu_law_decode          (reference mu_law_ops.py:26-31)
 *   - VQ nearest codebook    (reference model.py:57-74), direct-form distance,
 *     summation order d = 0..D-1 sequential with separately rounded multiply
 *     and add (compile with -ffp-contract=off), lowest index on ties.
 *   - causal dilated conv    (reference wavenet_ops.py:59-90), plain loops, used
 *     as a second opinion on the torch restatement.
 * PARITY UNPINNED against real TensorFlow: TF is not available in this image and the
 * reference ships no golden vectors.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* mu_law_ops.py:6-11, every operation rounded to fp32, trunc cast. */
void oracle_mu_law_encode_i32(const float* x, int32_t* y, size_t n) {
    const float mu = 255.0f;
    const float l1pmu = log1pf(mu);
    for (size_t i = 0; i < n; ++i) {
        float v = x[i];
        v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
        float s = (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f);
        float c = s * log1pf(mu * fabsf(v)) / l1pmu;
        float q = (c + 1.0f) / 2.0f * mu + 0.5f;
        y[i] = (int32_t)q;
    }
}

void oracle_mu_law_encode_f32(const float* x, float* y, size_t n) {
    const float mu = 255.0f;
    const float l1pmu = log1pf(mu);
    for (size_t i = 0; i < n; ++i) {
        float v = x[i];
        v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
        float s = (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f);
        y[i] = s * log1pf(mu * fabsf(v)) / l1pmu;
    }
}

/* mu_law_ops.py:26-31 */
void oracle_mu_law_decode_f32(const float* idx, float* x, size_t n) {
    const float mu = 255.0f;
    for (size_t i = 0; i < n; ++i) {
        float y = 2.0f * idx[i] / mu - 1.0f;
        float s = (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f);
        x[i] = s * (powf(1.0f + mu, fabsf(y)) - 1.0f) / mu;
    }
}

/* model.py:60-73.  z [rows][D], emb [K][D] -> idx int64 [rows], e_k, z_q [rows][D],
 * mind [rows] = the winning distance. */
void oracle_vq_nearest(const float* z, const float* emb, int64_t* idx, float* e_k,
                       float* z_q, float* mind, int rows, int K, int D) {
    for (int r = 0; r < rows; ++r) {
        const float* zr = z + (size_t)r * D;
        float best = INFINITY;
        int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float* e = emb + (size_t)k * D;
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) {
                float diff = zr[d] - e[d];
                float sq = diff * diff;
                acc = acc + sq;
            }
            if (acc < best) { best = acc; bi = k; }
        }
        idx[r] = bi;
        mind[r] = best;
        for (int d = 0; d < D; ++d) {
            float ek = emb[(size_t)bi * D + d];
            e_k[(size_t)r * D + d] = ek;
            float t = ek - zr[d];
            z_q[(size_t)r * D + d] = zr[d] + t;
        }
    }
}

/* wavenet_ops.py:59-90 on channels-last data: x [B][T][Cin], w [k][Cin][Cout],
 * y [B][Tout][Cout], Tout = ceil(T/stride) for the causal left pad d*(k-1). */
void oracle_conv1d_v2(const float* x, const float* w, const float* bias, float* y,
                      int B, int T, int Cin, int Cout, int k, int dil, int stride) {
    int Tout = (T + stride - 1) / stride;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < Tout; ++t)
            for (int o = 0; o < Cout; ++o) {
                double acc = bias ? bias[o] : 0.0;
                for (int j = 0; j < k; ++j) {
                    int ti = stride * t - (k - 1 - j) * dil;
                    if (ti < 0 || ti >= T) continue;
                    const float* xr = x + ((size_t)b * T + ti) * Cin;
                    const float* wr = w + (size_t)j * Cin * Cout + o;
                    for (int c = 0; c < Cin; ++c) acc += (double)xr[c] * (double)wr[(size_t)c * Cout];
                }
                y[((size_t)b * Tout + t) * Cout + o] = (float)acc;
            }
}

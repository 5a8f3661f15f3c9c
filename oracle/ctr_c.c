/* CPU ORACLE (test infrastructure, NOT the product): plain-C restatement of the DeepFM_v2 forward
 * (reference TFRecModel/src/com/sparrowrecsys/offline/tensorflow/DeepFM_v2.py:98-155), OpenMP over samples.
 * Two uses, both outside the product path: tests cross-check it against the numpy oracle (oracle/ctr_oracle.py,
 * deepfm_v2_forward), and bench.py's cpu_baseline leg times it on the host cores ("port": TensorFlow cannot be
 * installed here, so the reference's own CPU forward cannot be timed).
 *
 *   first  = Dense(1)(one-hot indicators) + Dense(1)(numerics)                      DeepFM_v2.py:98-104
 *   v_f    = Dense(K)(embedding_f), v_num = Dense(K)(numerics)                      DeepFM_v2.py:106-120
 *   fm     = (sum_f v_f)^2 - sum_f v_f^2                                            DeepFM_v2.py:147-152
 *   deep   = relu(Dense(H1)(relu(Dense(H0)(flatten(stack(v))))))                    DeepFM_v2.py:124-126
 *   out    = sigmoid(Dense(1)(concat[first, fm, deep]))                             DeepFM_v2.py:154-155
 * ids: [B][F] int32 in stack order, -1 = missing / out of vocabulary (zero embedding, no first-order term).
 * fp32 arithmetic; sums run in index order (the numpy oracle's BLAS sums differently: tests allow 2e-6). */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define MAXF 16
#define MAXK 64
#define MAXH 256

void deepfm_v2_forward_c(int32_t B, int32_t F, int32_t D, int32_t K, int32_t NN, int32_t H0, int32_t H1,
                         const int32_t* ids, const float* dense,
                         const float* const* tables,   /* F x [V_f][D]                  */
                         const float* const* fo,       /* F x [V_f] first-order weights */
                         const float* const* Wp,       /* F+1 x [D or NN][K]            */
                         const float* const* bp,       /* F+1 x [K]                     */
                         float fo_bias, const float* fo_num_w, float fo_num_b,
                         const float* W0, const float* b0,   /* [(F+1) K][H0], [H0] */
                         const float* W1, const float* b1,   /* [H0][H1], [H1]      */
                         const float* head_w, float head_b,  /* [1 + K + H1]        */
                         float* out, int32_t threads) {
    if (F > MAXF || K > MAXK || H0 > MAXH || H1 > MAXH) return;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int32_t m = 0; m < B; ++m) {
        float v[(MAXF + 1) * MAXK];
        float h0[MAXH], h1[MAXH];
        const int32_t* idr = ids + (size_t)m * F;
        const float* num = dense + (size_t)m * NN;
        float first = fo_bias + fo_num_b;
        for (int i = 0; i < NN; ++i) first += num[i] * fo_num_w[i];
        for (int f = 0; f < F; ++f) {
            float* vf = v + f * K;
            for (int k = 0; k < K; ++k) vf[k] = bp[f][k];
            const int32_t id = idr[f];
            if (id >= 0) {
                first += fo[f][id];
                const float* e = tables[f] + (size_t)id * D;
                for (int d = 0; d < D; ++d) {
                    const float x = e[d];
                    const float* w = Wp[f] + (size_t)d * K;
                    for (int k = 0; k < K; ++k) vf[k] += x * w[k];
                }
            }
        }
        {
            float* vf = v + F * K;
            for (int k = 0; k < K; ++k) vf[k] = bp[F][k];
            for (int i = 0; i < NN; ++i) {
                const float x = num[i];
                const float* w = Wp[F] + (size_t)i * K;
                for (int k = 0; k < K; ++k) vf[k] += x * w[k];
            }
        }
        float z = head_b + head_w[0] * first;
        for (int k = 0; k < K; ++k) {
            float s = 0.f, s2 = 0.f;
            for (int f = 0; f <= F; ++f) { const float x = v[f * K + k]; s += x; s2 += x * x; }
            z += head_w[1 + k] * (s * s - s2);
        }
        const int KD = (F + 1) * K;
        for (int n = 0; n < H0; ++n) h0[n] = b0[n];
        for (int i = 0; i < KD; ++i) {
            const float x = v[i];
            const float* w = W0 + (size_t)i * H0;
            for (int n = 0; n < H0; ++n) h0[n] += x * w[n];
        }
        for (int n = 0; n < H0; ++n) h0[n] = h0[n] > 0.f ? h0[n] : 0.f;
        for (int n = 0; n < H1; ++n) h1[n] = b1[n];
        for (int i = 0; i < H0; ++i) {
            const float x = h0[i];
            const float* w = W1 + (size_t)i * H1;
            for (int n = 0; n < H1; ++n) h1[n] += x * w[n];
        }
        for (int n = 0; n < H1; ++n) z += head_w[1 + K + n] * (h1[n] > 0.f ? h1[n] : 0.f);
        out[m] = z >= 0.f ? 1.0f / (1.0f + expf(-z)) : expf(z) / (1.0f + expf(z));
    }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * DIN forward (reference DIN.py:95-167), same role as above: cross-check of the numpy oracle's din_forward and the
 * cpu_baseline leg of `bench.py --workload din_c3`.
 *   h_t = E[hist_t], c = E[cand] (one shared Embedding; id 0 is an ordinary row: the mask has no numeric effect)
 *   a_t = sigmoid(Dense(1)(PReLU_t(Dense(H)([h_t - c, h_t, c, h_t * c]))))            DIN.py:139-151
 *   pooled = sum_t a_t h_t                                                             DIN.py:152-158
 *   x = concat(segments) -> Dense(N0) PReLU -> Dense(N1) PReLU -> Dense(1, sigmoid)    DIN.py:161-167
 * The concat is described by `seg`: n_seg x {dst, len, kind, a, b}: kind 0 = numeric column a of `dense`;
 * kind 1 = row ids[b] of extra table a (zero row when the id is < 0); kind 2 = pooled; kind 3 = candidate row. */
#define DIN_MAXD 64
#define DIN_MAXH 64
#define DIN_MAXX 512

void din_forward_c(int32_t B, int32_t F, int32_t ND, int32_t T, int32_t D, int32_t H, int32_t hist_col, int32_t cand_col,
                   const int32_t* ids, const float* dense, const float* table /*[V][D]*/,
                   const float* Wa /*[4D][H]*/, const float* ba /*[H]*/, const float* alpha /*[T][H]*/, const float* w2 /*[H]*/,
                   float b2, int32_t n_seg, const int32_t* seg, const float* const* extra /* tables [V_i][D] */,
                   int32_t X, int32_t N0, int32_t N1, const float* W0, const float* b0, const float* a0, const float* W1,
                   const float* b1, const float* a1, const float* hw, float hb, float* out, int32_t threads) {
    if (D > DIN_MAXD || H > DIN_MAXH || X > DIN_MAXX || N0 > DIN_MAXX || N1 > DIN_MAXX) return;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int32_t m = 0; m < B; ++m) {
        const int32_t* idr = ids + (size_t)m * F;
        const float* c = table + (size_t)idr[cand_col] * D;
        float pooled[DIN_MAXD], cw[DIN_MAXH], u[DIN_MAXH], x[DIN_MAXX], y0[DIN_MAXX], y1[DIN_MAXX];
        for (int d = 0; d < D; ++d) pooled[d] = 0.f;
        /* the c-only part of the activation unit's first layer is the same for every slot */
        for (int n = 0; n < H; ++n) cw[n] = ba[n];
        for (int d = 0; d < D; ++d) {
            const float x1 = -c[d], x3 = c[d];
            const float* r0 = Wa + (size_t)d * H;
            const float* r2 = Wa + (size_t)(2 * D + d) * H;
            for (int n = 0; n < H; ++n) cw[n] += x1 * r0[n] + x3 * r2[n];
        }
        for (int t = 0; t < T; ++t) {
            const float* h = table + (size_t)idr[hist_col + t] * D;
            for (int n = 0; n < H; ++n) u[n] = cw[n];
            for (int d = 0; d < D; ++d) {
                const float hd = h[d], hc = h[d] * c[d];
                const float* r0 = Wa + (size_t)d * H;
                const float* r1 = Wa + (size_t)(D + d) * H;
                const float* r3 = Wa + (size_t)(3 * D + d) * H;
                for (int n = 0; n < H; ++n) u[n] += hd * (r0[n] + r1[n]) + hc * r3[n];
            }
            float s = b2;
            const float* al = alpha + (size_t)t * H;
            for (int n = 0; n < H; ++n) s += w2[n] * (u[n] > 0.f ? u[n] : al[n] * u[n]);
            const float a = s >= 0.f ? 1.0f / (1.0f + expf(-s)) : expf(s) / (1.0f + expf(s));
            for (int d = 0; d < D; ++d) pooled[d] += a * h[d];
        }
        for (int i = 0; i < n_seg; ++i) {
            const int32_t* sg = seg + 5 * i;
            float* dst = x + sg[0];
            if (sg[2] == 0) dst[0] = dense[(size_t)m * ND + sg[3]];
            else if (sg[2] == 1) {
                const int32_t id = idr[sg[4]];
                for (int d = 0; d < sg[1]; ++d) dst[d] = id >= 0 ? extra[sg[3]][(size_t)id * D + d] : 0.f;
            } else {
                const float* src = sg[2] == 2 ? pooled : c;
                for (int d = 0; d < sg[1]; ++d) dst[d] = src[d];
            }
        }
        for (int n = 0; n < N0; ++n) y0[n] = b0[n];
        for (int i = 0; i < X; ++i) {
            const float xi = x[i];
            const float* w = W0 + (size_t)i * N0;
            for (int n = 0; n < N0; ++n) y0[n] += xi * w[n];
        }
        for (int n = 0; n < N0; ++n) y0[n] = y0[n] > 0.f ? y0[n] : a0[n] * y0[n];
        for (int n = 0; n < N1; ++n) y1[n] = b1[n];
        for (int i = 0; i < N0; ++i) {
            const float xi = y0[i];
            const float* w = W1 + (size_t)i * N1;
            for (int n = 0; n < N1; ++n) y1[n] += xi * w[n];
        }
        float z = hb;
        for (int n = 0; n < N1; ++n) z += hw[n] * (y1[n] > 0.f ? y1[n] : a1[n] * y1[n]);
        out[m] = z >= 0.f ? 1.0f / (1.0f + expf(-z)) : expf(z) / (1.0f + expf(z));
    }
}

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

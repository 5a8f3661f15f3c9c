/* CPU ORACLE (test infrastructure, NOT the product): a second, independent restatement -- plain C -- of the reference's
 * embedding similarity, Embedding.calculateSimilarity (online/model/Embedding.java:33-47) as called by
 * RecForYouProcess.calculateEmbSimilarScore (RecForYouProcess.java:100-105).  It exists to cross-check the numpy
 * restatement (oracle/emb_rank_oracle.py) bit for bit; only tests/ load it.
 *
 * Java semantics restated: `embVector.get(i) * other.get(i)` multiplies two floats in float (JLS 15.17: binary numeric
 * promotion of float x float is float; Java has no excess precision since strictfp became the default), the product is
 * widened and added to a double; the loop runs in index order; Math.sqrt is correctly rounded; one double division.
 * Build WITHOUT contraction or excess precision: gcc -O2 -ffp-contract=off -msse2 -mfpmath=sse (see oracle/Makefile).
 *
 * PARITY PIN STATUS: "parity unpinned" -- no JVM here and no vector in the reference; two restatements agreeing is a
 * consistency check of the restated arithmetic, not a pin. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* scores[q*C + c] for queries [Q][D], items [N][D], candidates [Q][C] (row index, < 0 or >= N = no embedding);
 * item_has / query_has may be NULL (all present). */
void emb_rank_scores(const float* item_emb, const uint8_t* item_has, int32_t n_items, int32_t D, const float* query_emb,
                     const uint8_t* query_has, int32_t Q, const int32_t* cand, int32_t C, double* scores) {
    for (int32_t q = 0; q < Q; ++q) {
        const volatile float* a = query_emb + (size_t)q * D;
        for (int32_t c = 0; c < C; ++c) {
            const int32_t id = cand[(size_t)q * C + c];
            double s = -1.0;
            if ((!query_has || query_has[q]) && id >= 0 && id < n_items && (!item_has || item_has[id])) {
                const volatile float* b = item_emb + (size_t)id * D;
                double dot = 0.0, d1 = 0.0, d2 = 0.0;
                for (int32_t i = 0; i < D; ++i) {
                    const float ab = a[i] * b[i], aa = a[i] * a[i], bb = b[i] * b[i];   /* float products */
                    dot += (double)ab;
                    d1 += (double)aa;
                    d2 += (double)bb;
                }
                s = dot / (sqrt(d1) * sqrt(d2));
            }
            scores[(size_t)q * C + c] = s;
        }
    }
}

// k_emb_rank.h -- the reference's "emb" ranker on the GPU: cosine similarity of a query
// embedding (a user, or a movie for the similar-movie page) against a list of candidate movies, then the candidates in
// descending score order.  Reference: RecForYouProcess.java:69-92 (ranker, case "emb"), :100-105
// (calculateEmbSimilarScore), SimilarMovieProcess.java:121-136,167-172 and Embedding.java:33-47 (calculateSimilarity).
// Included inside sparrow_hip.hip's anonymous namespace.
//
// Bit-exact restatement of the Java arithmetic: the three sums take FLOAT products (Float * Float is a float
// multiplication in Java) accumulated into DOUBLEs in index order, the score is dot / (sqrt(n1) * sqrt(n2)) in
// double; -1.0 when either side has no embedding; 0/0 -> NaN for an all-zero vector, as in Java.  No fused
// multiply-add anywhere (explicit __fmul_rn / __dadd_rn), correctly rounded sqrt and division.
//
// One workgroup per query: scores go to LDS as order-preserving 64-bit keys, a bitonic network over (key, position)
// sorts them the way `sorted(comparingByValue(reverseOrder()))` does -- Double.compareTo order: NaN above +inf,
// 0.0 above -0.0 -- with ties kept in candidate order (Java's HashMap iteration leaves ties unspecified).
// HBM-bound integer/byte style work: 4 B id + D*4 B row in, 8 B score + 4 B rank out per candidate.

#define ER_THREADS 256
#define ER_MAX_SORT 4096

__device__ __forceinline__ unsigned long long er_key(double s) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    if (s != s) return ~0ull;                                              // every NaN is the same, greatest value
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// P = padded (power of two) sort length, 0 = no ranking wanted
__global__ __launch_bounds__(ER_THREADS) void k_emb_rank(const float* __restrict__ item_emb, const unsigned char* __restrict__ item_has,
                                                         int n_items, int D, int item_stride,
                                                         const float* __restrict__ query_emb, const unsigned char* __restrict__ query_has,
                                                         int query_stride, const int* __restrict__ cand, int C, int P,
                                                         double* __restrict__ scores, int* __restrict__ order) {
    extern __shared__ unsigned long long er_smem[];
    unsigned long long* key = er_smem;                                     // [P]
    int* pos = reinterpret_cast<int*>(er_smem + P);                        // [P]
    float* qv = reinterpret_cast<float*>(pos + P);                         // [D]
    const int u = blockIdx.x, tid = threadIdx.x;
    const bool q_ok = query_has ? query_has[u] != 0 : true;
    for (int i = tid; i < D; i += ER_THREADS) qv[i] = query_emb[(size_t)u * query_stride + i];
    __syncthreads();
    // the query's squared norm: same order in every thread, so every thread holds the same double
    double n1 = 0.0;
    for (int i = 0; i < D; ++i) n1 = __dadd_rn(n1, (double)__fmul_rn(qv[i], qv[i]));
    const double r1 = __dsqrt_rn(n1);
    for (int c = tid; c < (P ? P : C); c += ER_THREADS) {
        double s = -1.0;
        if (c < C) {
            const int id = cand[(size_t)u * C + c];
            const bool ok = q_ok && id >= 0 && id < n_items && (item_has ? item_has[id] != 0 : true);
            if (ok) {
                const float* row = item_emb + (size_t)id * item_stride;
                double dot = 0.0, n2 = 0.0;
                for (int i = 0; i < D; ++i) {
                    const float x = row[i];
                    dot = __dadd_rn(dot, (double)__fmul_rn(qv[i], x));
                    n2 = __dadd_rn(n2, (double)__fmul_rn(x, x));
                }
                s = __ddiv_rn(dot, __dmul_rn(r1, __dsqrt_rn(n2)));
            }
            scores[(size_t)u * C + c] = s;
        }
        if (P) { key[c] = c < C ? er_key(s) : 0ull; pos[c] = c; }         // padding: below every real key (0 is no key)
    }
    if (!P) return;
    __syncthreads();
    // bitonic network, "before" = (key greater) or (key equal and position smaller)
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += ER_THREADS) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long ki = key[i], kl = key[l];
                    const int pi = pos[i], pl = pos[l];
                    const bool i_first = ki > kl || (ki == kl && pi < pl);
                    const bool up = (i & k) == 0;                          // this run sorts "first to the front"
                    if (up ? !i_first : i_first) { key[i] = kl; key[l] = ki; pos[i] = pl; pos[l] = pi; }
                }
            }
            __syncthreads();
        }
    }
    for (int c = tid; c < C; c += ER_THREADS) order[(size_t)u * C + c] = pos[c];
}

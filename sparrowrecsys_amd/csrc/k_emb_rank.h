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
// Ranking = `sorted(comparingByValue(reverseOrder()))`: Double.compareTo order (NaN above +inf, 0.0 above -0.0), ties
// kept in candidate order (Java's HashMap iteration leaves ties unspecified).  Scores map to order-preserving 64-bit keys.
//   k_emb_rank_wave (C <= 1024, the normal path: the reference ranks 800 candidates): ONE WAVE PER QUERY, the whole bitonic network in registers -- lane l
//     holds elements l*E .. l*E+E-1 (E = P/64), so the j < E exchange steps are register-to-register and the j >= E
//     steps one cross-lane shuffle per element; no LDS traffic for the data, no barriers.  It sorts ONE 64-bit word
//     per candidate: the key's upper 52 bits with the (inverted) candidate position in the low 12.  Two keys that agree
//     in their upper 52 bits but differ below would be ordered by position instead of by value; the sorted result is
//     checked for exactly that (adjacent pairs, full keys kept in LDS) and such a query -- adversarial, ~C^2 2^-51
//     likely for unrelated scores -- is re-ranked exactly by counting.
//   k_emb_rank (generic, one workgroup per query, (key, position) pairs through LDS): 1024 < C <= 4096, scores-only calls.
// HBM-bound integer/byte style work: 4 B id + D*4 B row in, 8 B score + 4 B rank out per candidate.

#define ER_THREADS 256
#define ER_MAX_SORT 4096

__device__ __forceinline__ unsigned long long er_key(double s) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(s);
    if (s != s) return ~0ull;                                              // every NaN is the same, greatest value
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// P = padded (power of two) sort length, 0 = no ranking wanted
static __global__ __launch_bounds__(ER_THREADS) void k_emb_rank(const float* __restrict__ item_emb, const unsigned char* __restrict__ item_has,
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


// ---- one wave per query, bitonic network in registers ----
#define ERW_WAVES 4

__device__ __forceinline__ unsigned long long er_shfl_xor(unsigned long long v, int lane_mask) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, lane_mask);
    const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), lane_mask);
    return ((unsigned long long)hi << 32) | lo;
}

template <int E>                       // elements per lane; P = 64 E candidates (padded)
__global__ __launch_bounds__(ERW_WAVES * 64) void k_emb_rank_wave(const float* __restrict__ item_emb, const unsigned char* __restrict__ item_has,
                                                                  int n_items, int D, int item_stride,
                                                                  const float* __restrict__ query_emb, const unsigned char* __restrict__ query_has,
                                                                  int n_queries, int query_stride, const int* __restrict__ cand, int C,
                                                                  double* __restrict__ scores, int* __restrict__ order) {
    constexpr int P = 64 * E;
    extern __shared__ unsigned long long er_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long* fullkey = er_smem + (size_t)wave * P;                               // [P] this wave's exact keys
    float* qv = reinterpret_cast<float*>(er_smem + (size_t)ERW_WAVES * P) + (size_t)wave * D;   // [D]
    const int u = blockIdx.x * ERW_WAVES + wave;
    if (u >= n_queries) return;                                                             // whole wave leaves together
    const bool q_ok = query_has ? query_has[u] != 0 : true;
    for (int i = lane; i < D; i += 64) qv[i] = query_emb[(size_t)u * query_stride + i];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                                                     // lgkmcnt(0): qv visible to the wave
    double n1 = 0.0;
    for (int i = 0; i < D; ++i) n1 = __dadd_rn(n1, (double)__fmul_rn(qv[i], qv[i]));
    const double r1 = __dsqrt_rn(n1);
    // scores: candidate c = e*64 + lane in this phase (coalesced ids / scores); keys parked in LDS, re-read lane-major
#pragma unroll 1
    for (int e = 0; e < E; ++e) {
        const int c = e * 64 + lane;
        unsigned long long k = 0ull;                                                        // padding: below every real key
        if (c < C) {
            double s = -1.0;
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
            k = er_key(s);
        }
        fullkey[c] = k;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // composite words: element idx = lane*E + e
    unsigned long long v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = lane * E + e;
        v[e] = (fullkey[idx] & ~0xFFFull) | (unsigned long long)(4095 - idx);
    }
    // bitonic network, descending ("first" = larger word)
#pragma unroll
    for (int k = 2; k <= P; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= E) {
                const int lm = j / E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = lane * E + e;
                    const unsigned long long o = er_shfl_xor(v[e], lm);
                    const bool want_first = ((idx & j) == 0) == ((idx & k) == 0);
                    const bool mine_first = v[e] > o;
                    v[e] = (want_first == mine_first) ? v[e] : o;
                }
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) == 0) {
                        const int idx = lane * E + e;
                        const bool up = (idx & k) == 0;                                     // e's pair: (e, e + j), e is the lower index
                        const unsigned long long a = v[e], b = v[e + j];
                        const bool swap = up ? a < b : a > b;
                        v[e] = swap ? b : a;
                        v[e + j] = swap ? a : b;
                    }
                }
            }
        }
    }
    // exactness check: neighbours that agree in the upper 52 bits must already be in (full key desc, position asc) order
    bool wrong = false;
    {
        unsigned lo = (unsigned)v[0], hi = (unsigned)(v[0] >> 32);
        lo = (unsigned)__shfl_down((int)lo, 1);
        hi = (unsigned)__shfl_down((int)hi, 1);
        const unsigned long long next_lane_first = ((unsigned long long)hi << 32) | lo;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned long long a = v[e];
            const unsigned long long b = e + 1 < E ? v[e + 1 < E ? e + 1 : e] : next_lane_first;
            const bool has_b = e + 1 < E || lane < 63;
            if (has_b && ((a ^ b) >> 12) == 0) {
                const int pa = 4095 - (int)(a & 0xFFF), pb = 4095 - (int)(b & 0xFFF);
                const unsigned long long ka = fullkey[pa], kb = fullkey[pb];
                if (ka < kb) wrong = true;                                                  // equal full keys: positions already ascending
            }
        }
    }
    if (__ballot(wrong) == 0ull) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int idx = lane * E + e;
            if (idx < C) order[(size_t)u * C + idx] = 4095 - (int)(v[e] & 0xFFF);
        }
        return;
    }
    // exact re-rank by counting (rare): rank(c) = #{c' : key[c'] > key[c], or equal and c' < c}
#pragma unroll 1
    for (int e = 0; e < E; ++e) {
        const int c = e * 64 + lane;
        const unsigned long long kc = c < C ? fullkey[c] : 0ull;
        int rank = 0;
        for (int o = 0; o < C; ++o) {
            const unsigned long long ko = fullkey[o];
            rank += (ko > kc || (ko == kc && o < c)) ? 1 : 0;
        }
        if (c < C) order[(size_t)u * C + rank] = c;
    }
}

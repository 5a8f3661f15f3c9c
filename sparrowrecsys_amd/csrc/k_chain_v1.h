// k_chain_v1.h -- k_deepfm_pairs: register-chained forward of the pairwise-dot DeepFM graph (reference
// DeepFM.py:91-115; BASELINE config 2 runs it at 6 fields, emb_dim 16), one WAVE per 16 samples.
// Included inside sparrow_hip.hip's anonymous namespace, after k_din_tail.h.
//
//   score = sigmoid( sum_f w1_f[id_f]                                  first order: rows of the head kernel (DeepFM.py:97,111-113)
//                  + sum_p hw_p <E_a[id_a], E_b[id_b]>                 FM pair dots x their head weights (DeepFM.py:100-103)
//                  + hdeep . relu(W1 relu(W0 [deep embeddings, numerics] + b0) + b1) + bias )   (DeepFM.py:106-108)
//
// Lane (r = lane&15, q = lane>>4) is sample r's q-th 16-byte column slot:
//  * every field's embedding row is gathered once, 16 B per lane (4 lanes = one 64-B row); a pair dot is four
//    FMAs on the two rows' pieces, scaled by the pair's head weight and left as a per-lane partial -- the sum
//    over the four q slots rides the one cross-lane reduction at the very end;
//  * deep0 is linear in its concat, so the deep embedding columns' share is a per-id table
//    F_g[id] = W0_g^T E_g[id] (k_fold_dense_rows, built at sprk_finalize), gathered straight into deep0's
//    accumulators in the MFMA C/D layout; only the numerics (K = 8) go through the matrix pipe for deep0;
//  * deep1 (K = 64) takes relu(deep0) from the registers it sits in (C/D layout = B layout), weights
//    read from L1/L2 per task (16 KB, hot); first-order weights: lane (r,q) fetches field q's and field q+4's.
// fp32 throughout (f32 MFMA: deep1's input has data-dependent range); nothing but ids, rows and the score
// touches memory.  The plan interpreter ran this graph in 40 us per 65 536 samples.

#define V1_MAX_FIELDS 8
#define V1_MAX_DEEP 4

struct V1Run {
    int F, ND, n_num;
    int nf;                               // fields (embedding row + first-order weight each)
    int col[V1_MAX_FIELDS];               // ids column of field f
    int vocab[V1_MAX_FIELDS];
    int row_floats;                       // floats per embedding row (Dp)
    const float* table[V1_MAX_FIELDS];    // [vocab+1][Dp], last row zero
    const float* w1[V1_MAX_FIELDS];       // [vocab+1] first-order weights, last zero
    int n_deep;                           // deep embedding columns
    int deep_field[V1_MAX_DEEP];          // ... which field each is
    const float* Fdeep[V1_MAX_DEEP];      // [vocab+1][H0] folded rows
    float pw[V1_MAX_FIELDS * V1_MAX_FIELDS];   // head weight of pair (a,b), a < b, at a*V1_MAX_FIELDS + b; 0 = not a pair
    const float* wn;                      // deep0 W^T numeric columns [H0][8] (zero padded)
    const float* b0;                      // [H0]
    const float* W1;                      // deep1 W^T [H1][ld1]
    int ld1;
    const float* b1;                      // [H1]
    const float* hdeep;                   // head weights on deep1's output [H1] (zero padded)
    float head_bias;
};

// One-time (finalize) kernel: deep0's numeric columns -> [H0][8]
__global__ __launch_bounds__(256) void k_v1_pack_wn(const float* __restrict__ W0, int ldw0, int n_off, int n_num, int H0,
                                                    float* __restrict__ wn) {
    for (int i = threadIdx.x; i < H0 * 8; i += 256) {
        const int n = i >> 3, j = i & 7;
        wn[i] = j < n_num ? W0[(size_t)n * ldw0 + n_off + j] : 0.f;
    }
}

template <int NF, int NV, int H0C, int H1C, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void k_deepfm_pairs(const V1Run A, const int* __restrict__ ids,
                                                                const float* __restrict__ dense, float* __restrict__ out,
                                                                int B, int* __restrict__ err) {
    constexpr int H0 = H0C * 16;
    static_assert(NF >= 2 && NF <= V1_MAX_FIELDS && NV >= 1 && NV <= 4, "shape");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntasks = (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    bool bad = false;

    // ---- register-resident weights: deep0's numeric columns, deep1, head weights on deep1's output ----
    f32x4 rwn[H0C], rb1[H1C], rhd[H1C];
#pragma unroll
    for (int nb = 0; nb < H0C; ++nb) {
        // 16x16x4 A operand: lane (n = r, q) feeds k = 4q + s; the numeric chunk is 8 wide
        rwn[nb] = q < 2 ? ld4(A.wn + (size_t)(nb * 16 + r) * 8 + 4 * q) : zero;
    }
#pragma unroll
    for (int n1 = 0; n1 < H1C; ++n1) {
        rb1[n1] = ld4(A.b1 + n1 * 16 + 4 * q);
        rhd[n1] = ld4(A.hdeep + n1 * 16 + 4 * q);
    }

    int tk = blockIdx.x * WAVES + wave;
    int idv[NF];
    auto ld_ids = [&](int t) {
        const int m = min(t * 16 + r, B - 1);                    // rows past the end re-read the last sample, never stored
        const int* row = ids + (size_t)m * A.F;
#pragma unroll
        for (int f = 0; f < NF; ++f) idv[f] = row[A.col[f]];
    };
    if (tk < ntasks) ld_ids(tk);
    for (; tk < ntasks; tk += task_stride) {
        const int m = min(tk * 16 + r, B - 1);
        // ---- gather: numerics, every field's row piece, first-order weights, folded deep rows ----
        f32x4 xn;
        {
            const float* nrow = dense + (size_t)m * A.ND;
            const int last = A.n_num - 1;
            // slots beyond n_num hold a duplicate finite value that only ever meets zero weights
            xn.x = nrow[min(4 * q + 0, last)];
            xn.y = nrow[min(4 * q + 1, last)];
            xn.z = nrow[min(4 * q + 2, last)];
            xn.w = nrow[min(4 * q + 3, last)];
        }
        unsigned sid[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            bad |= (unsigned)(idv[f] + 1) > (unsigned)A.vocab[f];              // neither a table row nor the "missing" marker -1
            sid[f] = min((unsigned)idv[f], (unsigned)A.vocab[f]);              // -1 / out of range -> the zero row at index vocab
        }
        f32x4 x[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) x[f] = q < NV ? ld4(A.table[f] + (size_t)sid[f] * A.row_floats + 4 * q) : zero;
        float w1a, w1b = 0.f;
        {
            // first order: lane (r,q) fetches field q's weight, then field q+4's
            const float* pa = A.w1[0] + sid[0];
            if (NF > 1) pa = q == 1 ? A.w1[NF > 1 ? 1 : 0] + sid[NF > 1 ? 1 : 0] : pa;
            if (NF > 2) pa = q == 2 ? A.w1[NF > 2 ? 2 : 0] + sid[NF > 2 ? 2 : 0] : pa;
            if (NF > 3) pa = q == 3 ? A.w1[NF > 3 ? 3 : 0] + sid[NF > 3 ? 3 : 0] : pa;
            w1a = (q < NF) ? *pa : 0.f;
            if (NF > 4) {
                const float* pb = A.w1[NF > 4 ? 4 : 0] + sid[NF > 4 ? 4 : 0];
                if (NF > 5) pb = q == 1 ? A.w1[NF > 5 ? 5 : 0] + sid[NF > 5 ? 5 : 0] : pb;
                if (NF > 6) pb = q == 2 ? A.w1[NF > 6 ? 6 : 0] + sid[NF > 6 ? 6 : 0] : pb;
                if (NF > 7) pb = q == 3 ? A.w1[NF > 7 ? 7 : 0] + sid[NF > 7 ? 7 : 0] : pb;
                w1b = (q + 4 < NF) ? *pb : 0.f;
            }
        }
        f32x4 h0[H0C];
#pragma unroll
        for (int nb = 0; nb < H0C; ++nb) h0[nb] = ld4(A.b0 + nb * 16 + 4 * q);
#pragma unroll
        for (int g0 = 0; g0 < V1_MAX_DEEP; g0 += 2) {             // two folded columns = 2*H0C loads in flight at a time
            f32x4 fd[2][H0C];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                // deep_field[] is a runtime index into the compile-time sid[]: a select chain (kept opaque, or it
                // becomes a dynamically indexed private array in scratch memory)
                unsigned s = sid[0];
#pragma unroll
                for (int f = 1; f < NF; ++f) {
                    unsigned sf = sid[f];
                    asm("" : "+v"(sf));
                    s = A.deep_field[g0 + g] == f ? sf : s;
                }
                const bool on = g0 + g < A.n_deep;                // wave-uniform
                const float* frow = A.Fdeep[g0 + g] + (size_t)s * H0 + 4 * q;
#pragma unroll
                for (int nb = 0; nb < H0C; ++nb) fd[g][nb] = on ? ld4(frow + nb * 16) : zero;
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int nb = 0; nb < H0C; ++nb) h0[nb] += fd[g][nb];
        }
        if (tk + task_stride < ntasks) ld_ids(tk + task_stride);   // next task's ids fly under this task's arithmetic

        // ---- first order + pair dots (per-lane partials; the sum over q is part of the final reduction) ----
        float z = w1a + w1b;
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = a + 1; b < NF; ++b) {
                const float hw = A.pw[a * V1_MAX_FIELDS + b];       // wave-uniform (SGPR); zero when (a,b) is not a pair
                const f32x4 p = x[a] * x[b];
                z = fmaf(hw, (p.x + p.y) + (p.z + p.w), z);
            }
        // ---- deep0: numerics on the matrix pipe onto the gathered accumulators, ReLU ----
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb)
                h0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[nb][st], xn[st], h0[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < H0C; ++nb) h0[nb] = relu4_fast(h0[nb]);
        // ---- deep1 (H1C independent chains), ReLU, head weights ----
        f32x4 h1[H1C];
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) h1[n1] = rb1[n1];
        // deep1's A fragments come from L1/L2 every task: 64 more resident registers would spill.  (The offset
        // passes through a volatile asm so that the loop-invariant loads are not hoisted back into registers.)
        int w1o = r * A.ld1 + 4 * q;
        asm volatile("" : "+v"(w1o));
#pragma unroll
        for (int c = 0; c < H0C; ++c) {
            f32x4 a[H1C];
#pragma unroll
            for (int n1 = 0; n1 < H1C; ++n1) a[n1] = ld4(A.W1 + w1o + n1 * 16 * A.ld1 + 16 * c);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int n1 = 0; n1 < H1C; ++n1)
                    h1[n1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n1][st], h0[c][st], h1[n1], 0, 0, 0);
        }
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) z += dot4(rhd[n1], relu4_fast(h1[n1]));
        z = rows4_sum(z);
        const int mm = tk * 16 + r;
        if (q == 0 && mm < B) out[mm] = sigmoidf_acc(z + A.head_bias);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

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
//  * the same 16-byte piece is the B operand of v_mfma_f32_16x16x4_f32 (lane supplies k = 4q + s), so deep0 runs
//    straight on the gathered rows of the deep fields (the host orders them first) + the numerics chunk;
//    (folding the deep columns into per-id tables of deep0's 64 outputs, as the DIN tail does, was measured here
//    and lost: 2 x 256 B more per sample made the kernel memory bound at 5.9 TB/s of actual traffic)
//  * deep1 (K = 64) takes relu(deep0) from the registers it sits in (C/D layout = B layout); weights of both
//    layers are read from LDS as A fragments; first-order weights: lane (r,q) fetches field q's and field q+4's.
// deep0 on f32 MFMA; deep1 (data-dependent input range) on the f16 matrix pipe with a per-sample dynamic power-of-two
// scale and hi + lo split operands (dyn_split.h; SPRK_DYN_F16=0: f32 MFMA); nothing but ids, rows and the score
// touches memory.  The plan interpreter ran this graph in 40 us per 65 536 samples.

#define V1_MAX_FIELDS 8
#define V1_MAX_DEEP 2

struct V1Run {
    int F, ND, n_num;
    int nf;                               // fields (embedding row + first-order weight each)
    int col[V1_MAX_FIELDS];               // ids column of field f
    int vocab[V1_MAX_FIELDS];
    int row_floats;                       // floats per embedding row (Dp)
    const float* table[V1_MAX_FIELDS];    // [vocab+1][Dp], last row zero
    const float* w1[V1_MAX_FIELDS];       // [vocab+1] first-order weights, last zero
    int n_deep;                           // deep embedding columns = fields 0 .. n_deep-1 (the host orders them first)
    float pw[V1_MAX_FIELDS * V1_MAX_FIELDS];   // head weight of pair (a,b), a < b, at a*V1_MAX_FIELDS + b; 0 = not a pair
    const float* w0;                      // deep0 W^T packed [H0][16*(V1_MAX_DEEP+1)]: deep field chunks, then the numerics chunk (zero padded)
    const float* b0;                      // [H0]
    const float* W1;                      // deep1 W^T [H1][ld1]
    int ld1;
    const float* b1;                      // [H1]
    const float* hdeep;                   // head weights on deep1's output [H1] (zero padded)
    float head_bias;
    const float* w1frag;                  // DYN: deep1's W^T as split-f16 A fragments (dyn_split.h: k_dyn_pack_w), or NULL
    float inv_w1_scale;                   // DYN: 1 / their static power-of-two scale (0 = deep1 on f32 MFMA)
};

// One-time (finalize) kernel: deep0's W^T columns -> [H0][16*(V1_MAX_DEEP+1)]: chunk g < n_deep = the Dp columns of deep
// field g (at col_off[g] of the layer's input slice), chunk V1_MAX_DEEP = the numerics; everything else zero.
__global__ __launch_bounds__(256) void k_v1_pack_w0(const float* __restrict__ W0, int ldw0, int n_deep, int off0, int off1,
                                                    int Dp, int n_off, int n_num, int H0, float* __restrict__ w0) {
    const int KW = 16 * (V1_MAX_DEEP + 1);
    for (int i = threadIdx.x; i < H0 * KW; i += 256) {
        const int n = i / KW, k = i - n * KW, c = k >> 4, j = k & 15;
        float v = 0.f;
        if (c < V1_MAX_DEEP) { if (c < n_deep && j < Dp) v = W0[(size_t)n * ldw0 + (c == 0 ? off0 : off1) + j]; }
        else if (j < n_num) v = W0[(size_t)n * ldw0 + n_off + j];
        w0[i] = v;
    }
}

// Multi-batch launch (sprk_set_many_batches): launch task t is task t % ntpb of batch t / ntpb, every batch with its own buffers.
#define V1_MB 16
struct V1Many {
    const int* ids[V1_MB];
    const float* dense[V1_MB];
    float* out[V1_MB];
    int n, ntpb;
};

template <int H0C, int H1C>
struct V1Lds {
    static constexpr int H0 = H0C * 16, H1 = H1C * 16;
    static constexpr int S1 = H0 + 4;                 // deep1 W^T row stride
    static constexpr int K0 = 16 * (V1_MAX_DEEP + 1); // deep0's packed K: deep field chunks + numerics chunk
    static constexpr int S0 = K0 + 4;                 // deep0 W^T row stride
    static constexpr int off_w1 = 0;                  // [H1][S1]
    static constexpr int off_w0 = off_w1 + H1 * S1;   // [H0][S0]
    static constexpr int off_b0 = off_w0 + H0 * S0;   // [H0]
    static constexpr int off_b1 = off_b0 + H0;        // [H1]
    static constexpr int off_hd = off_b1 + H1;        // [H1]
    static constexpr int total = off_hd + H1;
    static constexpr size_t bytes = sizeof(float) * total;
};

// Task pipeline: the rows of task n+1 are in flight (in the gather registers) while task n is scored from copies;
// its ids were fetched one task earlier still.  Everything the scoring stage reads besides its operands comes from
// LDS (deep1's weights, biases): a global load there would sit behind the prefetched gather in the in-order vmcnt
// queue and drain it.
template <int NF, int NV, int H0C, int H1C, int WAVES, bool DYN, bool MB = false>
__global__ __launch_bounds__(WAVES * 64, 2) void k_deepfm_pairs(const V1Run A, const int* __restrict__ ids0,
                                                                const float* __restrict__ dense0, float* __restrict__ out0,
                                                                int B, int* __restrict__ err, const V1Many M) {
    using LD = V1Lds<H0C, H1C>;
    constexpr int H0 = H0C * 16;
    static_assert(NF >= 2 && NF <= V1_MAX_FIELDS && NV >= 1 && NV <= 4, "shape");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntasks = MB ? M.n * M.ntpb : (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    bool bad = false;
    // MB: (batch, task inside the batch) of launch task t (wave-uniform)
    auto batch_of = [&](int t, int& tl) {
        if constexpr (MB) {
            const int b = __builtin_amdgcn_readfirstlane(t / M.ntpb);
            tl = t - b * M.ntpb;
            return b;
        } else {
            tl = t;
            return 0;
        }
    };

    // ---- one-time: both layers' W^T, biases, head weights -> LDS ----
    if constexpr (DYN) {
        static_assert(H1C * (H0C / 2) * 512 <= LD::H1 * LD::S1 && H0C % 2 == 0, "fragments fit W1's region");
        for (int i = tid; i < H1C * (H0C / 2) * 512; i += WAVES * 64) smem[LD::off_w1 + i] = A.w1frag[i];
    } else {
        for (int i = tid; i < LD::H1 * LD::S1; i += WAVES * 64) {
            const int n = i / LD::S1, k = i - n * LD::S1;
            smem[LD::off_w1 + i] = k < H0 ? A.W1[(size_t)n * A.ld1 + k] : 0.f;
        }
    }
    for (int i = tid; i < H0 * LD::S0; i += WAVES * 64) {
        const int n = i / LD::S0, k = i - n * LD::S0;
        smem[LD::off_w0 + i] = k < LD::K0 ? A.w0[(size_t)n * LD::K0 + k] : 0.f;
    }
    for (int i = tid; i < H0; i += WAVES * 64) smem[LD::off_b0 + i] = A.b0[i];
    for (int i = tid; i < LD::H1; i += WAVES * 64) { smem[LD::off_b1 + i] = A.b1[i]; smem[LD::off_hd + i] = A.hdeep[i]; }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // one-time loads have landed before the pipelined loop

    // ---- gather registers (task n+1) ----
    int idv[NF];
    f32x4 gx[NF], gxn = zero;
    float gw1a = 0.f, gw1b = 0.f;
    auto ld_ids = [&](int tg) {
        int t;
        const int bi = batch_of(tg, t);
        const int* ids = MB ? M.ids[bi] : ids0;
        const int m = min(t * 16 + r, B - 1);                    // rows past the end re-read the last sample, never stored
        const int* row = ids + (size_t)m * A.F;
#pragma unroll
        for (int f = 0; f < NF; ++f) idv[f] = row[A.col[f]];
    };
    auto issue_gather = [&](int tg) {
        int t;
        const int bi = batch_of(tg, t);
        const float* dense = MB ? M.dense[bi] : dense0;
        const int m = min(t * 16 + r, B - 1);
        {
            const float* nrow = dense + (size_t)m * A.ND;
            const int last = A.n_num - 1;
            // slots beyond n_num hold a duplicate finite value that only ever meets zero weights
            gxn.x = nrow[min(4 * q + 0, last)];
            gxn.y = nrow[min(4 * q + 1, last)];
            gxn.z = nrow[min(4 * q + 2, last)];
            gxn.w = nrow[min(4 * q + 3, last)];
        }
        unsigned sid[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            bad |= (unsigned)(idv[f] + 1) > (unsigned)A.vocab[f];              // neither a table row nor the "missing" marker -1
            sid[f] = min((unsigned)idv[f], (unsigned)A.vocab[f]);              // -1 / out of range -> the zero row at index vocab
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) gx[f] = q < NV ? ld4(A.table[f] + (size_t)sid[f] * A.row_floats + 4 * q) : zero;
        {
            // first order: lane (r,q) fetches field q's weight, then field q+4's
            const float* pa = A.w1[0] + sid[0];
            if (NF > 1) pa = q == 1 ? A.w1[NF > 1 ? 1 : 0] + sid[NF > 1 ? 1 : 0] : pa;
            if (NF > 2) pa = q == 2 ? A.w1[NF > 2 ? 2 : 0] + sid[NF > 2 ? 2 : 0] : pa;
            if (NF > 3) pa = q == 3 ? A.w1[NF > 3 ? 3 : 0] + sid[NF > 3 ? 3 : 0] : pa;
            gw1a = (q < NF) ? *pa : 0.f;
            if (NF > 4) {
                const float* pb = A.w1[NF > 4 ? 4 : 0] + sid[NF > 4 ? 4 : 0];
                if (NF > 5) pb = q == 1 ? A.w1[NF > 5 ? 5 : 0] + sid[NF > 5 ? 5 : 0] : pb;
                if (NF > 6) pb = q == 2 ? A.w1[NF > 6 ? 6 : 0] + sid[NF > 6 ? 6 : 0] : pb;
                if (NF > 7) pb = q == 3 ? A.w1[NF > 7 ? 7 : 0] + sid[NF > 7 ? 7 : 0] : pb;
                gw1b = (q + 4 < NF) ? *pb : 0.f;
            }
        }
    };

    int tk = blockIdx.x * WAVES + wave;
    if (tk < ntasks) {
        ld_ids(tk);
        issue_gather(tk);
        if (tk + task_stride < ntasks) ld_ids(tk + task_stride);
    }
    for (; tk < ntasks; tk += task_stride) {
        // ---- hand-off: this task's operands out of the gather registers ----
        f32x4 x[NF], h0[H0C];
#pragma unroll
        for (int f = 0; f < NF; ++f) x[f] = gx[f];
#pragma unroll
        for (int nb = 0; nb < H0C; ++nb) h0[nb] = ld4(smem + LD::off_b0 + nb * 16 + 4 * q);
        const f32x4 xn = gxn;
        float z = gw1a + gw1b;
        if (tk + task_stride < ntasks) {                          // next task's rows fly under this task's arithmetic
            issue_gather(tk + task_stride);
            if (tk + 2 * task_stride < ntasks) ld_ids(tk + 2 * task_stride);
        }
        // ---- pair dots (per-lane partials; the sum over q is part of the final reduction) ----
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = a + 1; b < NF; ++b) {
                const float hw = A.pw[a * V1_MAX_FIELDS + b];       // wave-uniform (SGPR); zero when (a,b) is not a pair
                const f32x4 p = x[a] * x[b];
                z = fmaf(hw, (p.x + p.y) + (p.z + p.w), z);
            }
        // ---- deep0 (DeepFM.py:106-107): the deep fields' row pieces and the numerics as B operands, A fragments from
        //      LDS (the offset passes through a volatile asm so that the loop-invariant reads are not hoisted) ----
        int w0o = LD::off_w0 + r * LD::S0 + 4 * q;
        asm volatile("" : "+v"(w0o));
#pragma unroll
        for (int c = 0; c <= V1_MAX_DEEP; ++c) {
            if (c < V1_MAX_DEEP && c >= A.n_deep) continue;       // wave-uniform
            const f32x4 bop = c < V1_MAX_DEEP ? x[c < NF ? c : 0] : xn;
            f32x4 a[H0C];
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb) {
                a[nb] = ld4(smem + w0o + nb * 16 * LD::S0 + 16 * c);
                if (c == V1_MAX_DEEP && q >= 2) a[nb] = zero;     // the numeric chunk is 8 wide: k = 4q + s < 8
            }
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int nb = 0; nb < H0C; ++nb)
                    h0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb][st], bop[st], h0[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < H0C; ++nb) h0[nb] = relu4_fast(h0[nb]);
        // ---- deep1 (H1C independent chains; A fragments from LDS), ReLU, head weights ----
        f32x4 h1[H1C];
        if constexpr (DYN) {
            // f16 matrix pipe with a per-sample power-of-two scale of relu(deep0) (dyn_split.h; see k_din_tail's fc1)
            float mx = 0.f;
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, __builtin_fabsf(h0[nb][j]));
            mx = rows4_max(mx);
            float scale, inv;
            dyn_scale(mx, A.inv_w1_scale, scale, inv);
            f32x4 acc[H1C];
#pragma unroll
            for (int n1 = 0; n1 < H1C; ++n1) acc[n1] = zero;
            int wfo = LD::off_w1 + (r * 4 + q) * 4;               // this lane's 16 bytes inside a 1-KB fragment
            asm volatile("" : "+v"(wfo));                         // (keeps the loop-invariant LDS reads inside the task loop)
#pragma unroll
            for (int b = 0; b < H0C / 2; ++b) {
                din_f16x8 bh, bl;
                dyn_split8(h0[2 * b], h0[2 * b + 1], scale, bh, bl);
#pragma unroll
                for (int n1 = 0; n1 < H1C; ++n1) {
                    const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(smem + wfo + ((n1 * (H0C / 2) + b) * 2 + 0) * 256));
                    const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(smem + wfo + ((n1 * (H0C / 2) + b) * 2 + 1) * 256));
                    acc[n1] = mfma_f16(ah, bh, acc[n1]);
                    acc[n1] = mfma_f16(ah, bl, acc[n1]);
                    acc[n1] = mfma_f16(al, bh, acc[n1]);
                }
            }
#pragma unroll
            for (int n1 = 0; n1 < H1C; ++n1) h1[n1] = acc[n1] * inv + ld4(smem + LD::off_b1 + n1 * 16 + 4 * q);
        } else {
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) h1[n1] = ld4(smem + LD::off_b1 + n1 * 16 + 4 * q);
        // (the offset passes through a volatile asm so that the loop-invariant LDS reads are not hoisted into 64 registers)
        int w1o = LD::off_w1 + r * LD::S1 + 4 * q;
        asm volatile("" : "+v"(w1o));
#pragma unroll
        for (int c = 0; c < H0C; ++c) {
            f32x4 a[H1C];
#pragma unroll
            for (int n1 = 0; n1 < H1C; ++n1) a[n1] = ld4(smem + w1o + n1 * 16 * LD::S1 + 16 * c);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int n1 = 0; n1 < H1C; ++n1)
                    h1[n1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n1][st], h0[c][st], h1[n1], 0, 0, 0);
        }
        }
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) z += dot4(ld4(smem + LD::off_hd + n1 * 16 + 4 * q), relu4_fast(h1[n1]));
        z = rows4_sum(z);
        int tl;
        const int bo = batch_of(tk, tl);
        float* out = MB ? M.out[bo] : out0;
        const int mm = tl * 16 + r;
        if (q == 0 && mm < B) out[mm] = sigmoidf_acc(z + A.head_bias);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

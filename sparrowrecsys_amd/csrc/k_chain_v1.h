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
// deep0's embedding columns (K = 32: the two deep fields' rows) and deep1 (K = 64) run on the f16 matrix pipe with a per-sample
// dynamic power-of-two scale and hi + lo split operands (dyn_split.h; SPRK_DYN_F16=0: f32 MFMA); the numerics (up to 67 000
// next to values of order 1) stay on f32 MFMA, K = 8 as two steps (k = q + 4s); nothing but ids, rows and the score touches
// memory.  Round 2: the prologue no longer serialises weight staging -> ids -> rows (ids of a wave's first two tasks are
// requested first, the weight image arrives by LDS-DMA inside their latency, and at two tasks per wave -- B = 65 536 on a full
// chip -- both gathers are issued before the first scoring stage).  The plan interpreter ran this graph in 40 us per 65 536
// samples, round 1's version of this kernel in 15.8.

#ifndef V1_XP
#define V1_XP 0                           // ablation builds (scripts/r06, WRONG RESULTS, timing only): 1 one ids load per lane instead of NF, 2 the rows of fields 3.. not
#endif                                    // requested, 4 no second first-order load, 8 the deep part's own rows not requested, 16 no numerics loads
#define V1_MAX_FIELDS 8
#define V1_MAX_DEEP 2
#define V1_MAX_ROWS (V1_MAX_FIELDS + V1_MAX_DEEP)

struct V1Run {
    int F, ND, n_num;
    int nf;                               // fields (embedding row + first-order weight each)
    int col[V1_MAX_FIELDS];               // ids column of field f
    int vocab[V1_MAX_FIELDS];
    int row_floats;                       // floats per embedding row (Dp)
    const float* table[V1_MAX_ROWS];      // [vocab+1][Dp], last row zero; entries nf.. = the deep part's own tables (sep)
    const float* w1[V1_MAX_FIELDS];       // [vocab+1] first-order weights, last zero
    int n_deep;                           // deep embedding columns: looked up with the ids of fields 0 .. n_deep-1 (the host orders them first)
    // sep = 1: the deep part reads its OWN tables (DeepFM.py:106: DenseFeatures(deep_feature_columns) creates its own
    // movieId / userId embedding variables, distinct from the FM part's DeepFM.py:91-92) = rows nf .. nf+n_deep-1; sep = 0: tied
    // tables, the deep columns are fields 0 .. n_deep-1's rows themselves
    int sep;
    // sep, narrow rows: the deep row rides in its field's 128-byte line -- `pack` = its byte offset there (0 = rows of its own).
    // Rows of <= 12 floats: {E 48 B | . | w1 at 64 | . | E_deep at 80}; 16 floats: {E 64 B | E_deep 64 B} and the first-order
    // weights of these fields in the compact array w1c (an L2-resident 4-byte gather instead of a third line per deep field)
    int pack;
    const float* w1c;                     // pack == 64: first-order weights of fields 0 .. n_deep-1, [vocab+1] each, back to back
    unsigned w1cbase[V1_MAX_DEEP];
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
    const float* w0frag;                  // DYN: deep0's embedding columns [H0][32] as split-f16 A fragments, or NULL
    float inv_w0_scale;
    const float* image;                   // the LDS image (k_v1_pack_image), staged by LDS-DMA
    // narrow rows (emb_dim <= 16): ONE derived table for all fields, 128-byte rows {E[<=16] | w1 | 0..} -- an id's embedding row
    // and its first-order weight share a cache line (one fabric request per (sample, field) instead of two: PMC had the
    // kernel at 38.7 MB per 65 536 samples against 30.4 MB of algorithmic bytes), one SGPR base + 32-bit byte offsets
    float e_scale, e_inv;                 // static split scale of the deep fields' rows and 1 / (e_scale * w0 scale); 0 = per sample
    const float* tab;                     // NULL: gather from table[] / w1[] (wide rows)
    unsigned rowbase[V1_MAX_ROWS];        // first row of field f (or deep table nf + d) in tab
};

// One-time (finalize) kernel: rows of one field of the derived table
// (w1 == NULL: no first-order weight in the row; deep != NULL: the deep part's row of the same id packed at float deep_off)
static __global__ __launch_bounds__(256) void k_v1_build_rows(const float* __restrict__ table, int Dp, const float* __restrict__ w1,
                                                       long long rows, float* __restrict__ out, const float* __restrict__ deep, int deep_off) {
    const long long total = rows * 32;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i >> 5;
        const int c = (int)(i & 31);
        float x = c < Dp ? table[v * Dp + c] : ((c == 16 && w1) ? w1[v] : 0.f);
        if (deep && c >= deep_off && c - deep_off < Dp) x = deep[v * Dp + (c - deep_off)];
        out[i] = x;
    }
}

// One-time (finalize) kernel: deep0's W^T columns -> [H0][KW], KW = 16 * (V1_MAX_DEEP * PC + 1), PC = 16-float chunks per embedding
// row: chunk c < V1_MAX_DEEP * PC = columns [16 (c % PC), +16) of deep field c / PC (at col_off of the layer's input slice), the
// last chunk = the numerics; everything else zero.
static __global__ __launch_bounds__(256) void k_v1_pack_w0(const float* __restrict__ W0, int ldw0, int n_deep, int off0, int off1,
                                                    int Dp, int n_off, int n_num, int H0, int PC, float* __restrict__ w0) {
    const int KW = 16 * (V1_MAX_DEEP * PC + 1);
    for (int i = threadIdx.x; i < H0 * KW; i += 256) {
        const int n = i / KW, k = i - n * KW, c = k >> 4, j = k & 15;
        float v = 0.f;
        if (c < V1_MAX_DEEP * PC) {
            const int f = c / PC, col = 16 * (c - f * PC) + j;
            if (f < n_deep && col < Dp) v = W0[(size_t)n * ldw0 + (f == 0 ? off0 : off1) + col];
        } else if (j < n_num) {
            v = W0[(size_t)n * ldw0 + n_off + j];
        }
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

template <int H0C, int H1C, int PC = 1>
struct V1Lds {
    static constexpr int H0 = H0C * 16, H1 = H1C * 16;
    static constexpr int S1 = H0 + 4;                 // deep1 W^T row stride (f32 path)
    static constexpr int KE = 32 * PC;                // deep0's embedding columns: two deep fields x PC chunks of 16
    static constexpr int SE = KE + 4;                 // deep0 embedding-part W^T row stride (f32 path)
    static constexpr int off_w1 = 0;                  // DYN: H1C*(H0C/2)*512 fragment floats; else [H1][S1]
    static constexpr int off_w0e = off_w1 + H1 * S1;  // DYN: H0C*PC*512 fragment floats (PC K = 32 blocks); else [H0][SE]
    static constexpr int off_w0n = off_w0e + H0 * SE; // [H0][8] numerics columns
    static constexpr int off_b0 = off_w0n + H0 * 8;   // [H0]
    static constexpr int off_b1 = off_b0 + H0;        // [H1]
    static constexpr int off_hd = off_b1 + H1;        // [H1]
    static constexpr int total = off_hd + H1;
    static constexpr int total_pad = (total + 255) & ~255;
    static constexpr size_t bytes = sizeof(float) * total_pad;
    static_assert(H1C * (H0C / 2) * 512 <= H1 * S1 && H0C * PC * 512 <= H0 * SE && H0C % 2 == 0, "fragments fit their regions");
};

// One-time (finalize) kernel: the LDS image.  w0 = k_v1_pack_w0's [H0][48] (deep field chunks, numerics chunk).
template <int H0C, int H1C, int PC>
__global__ __launch_bounds__(256) void k_v1_pack_image(const V1Run A, float* __restrict__ img) {
    using LD = V1Lds<H0C, H1C, PC>;
    const int tid = threadIdx.x, KW = 16 * (V1_MAX_DEEP * PC + 1);
    for (int i = tid; i < LD::total_pad; i += 256) img[i] = 0.f;
    __syncthreads();
    if (A.w1frag) {
        for (int i = tid; i < H1C * (H0C / 2) * 512; i += 256) img[LD::off_w1 + i] = A.w1frag[i];
    } else {
        for (int i = tid; i < LD::H1 * LD::S1; i += 256) {
            const int n = i / LD::S1, k = i - n * LD::S1;
            img[LD::off_w1 + i] = k < LD::H0 ? A.W1[(size_t)n * A.ld1 + k] : 0.f;
        }
    }
    if (A.w0frag) {
        for (int i = tid; i < H0C * PC * 512; i += 256) img[LD::off_w0e + i] = A.w0frag[i];
    } else {
        for (int i = tid; i < LD::H0 * LD::SE; i += 256) {
            const int n = i / LD::SE, k = i - n * LD::SE;
            img[LD::off_w0e + i] = k < LD::KE ? A.w0[(size_t)n * KW + k] : 0.f;
        }
    }
    for (int i = tid; i < LD::H0 * 8; i += 256) img[LD::off_w0n + i] = A.w0[(size_t)(i >> 3) * KW + 16 * V1_MAX_DEEP * PC + (i & 7)];
    for (int i = tid; i < LD::H0; i += 256) img[LD::off_b0 + i] = A.b0[i];
    for (int i = tid; i < LD::H1; i += 256) { img[LD::off_b1 + i] = A.b1[i]; img[LD::off_hd + i] = A.hdeep[i]; }
}

template <int NR, int PC>
struct V1Set {
    f32x4 x[NR][PC];                      // every field's (and separate deep table's) row pieces: elements 16 pc + 4q .. +3
    float xa, xb;                         // numerics q and q + 4
    float w1a, w1b;                       // first-order weights fetched by this lane
};

// NV = 16-byte pieces per embedding row (Dp / 4): up to 4 = one piece per lane (emb_dim <= 16), 16 = four per lane (emb_dim 64,
// BASELINE config 4: the 256-byte rows of the 27 M-row table are gathered whole -- the fold does not apply to pair dots).
// SEP: the deep part's own tables are gathered as rows NF, NF + 1 (V1Run::sep).  ONE: one task per wave, no loop -- the
// strict one-batch launch at four waves per SIMD (see k_chain_v2j1.h for the measurements behind this shape).
template <int NF, int NV, int H0C, int H1C, int WAVES, bool DYN, bool MB, bool SEP, bool ONE = false>
__device__ __forceinline__ void v1_body(const V1Run& A, const int* __restrict__ ids0, const float* __restrict__ dense0,
                                        float* __restrict__ out0, int B, int* __restrict__ err, const V1Many* __restrict__ Mp) {
    constexpr int PC = (NV + 3) / 4;
    constexpr int NR = NF + (SEP ? V1_MAX_DEEP : 0);
    constexpr int DB = SEP ? NF : 0;                                // first deep row
    using LD = V1Lds<H0C, H1C, PC>;
    using Set = V1Set<NR, PC>;
    static_assert(!(ONE && MB) && !(ONE && PC > 1), "one-task shape: one batch, narrow rows");
    static_assert(NF >= 2 && NF <= V1_MAX_FIELDS && NV >= 1 && (NV <= 4 || NV % 4 == 0) && NV <= 16, "shape");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m_ntpb = MB ? Mp->ntpb : 0;
    const int ntasks = MB ? Mp->n * m_ntpb : (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    bool bad = false;
    // MB: (batch, task inside the batch) of launch task t (wave-uniform)
    auto batch_of = [&](int t, int& tl) {
        if constexpr (MB) {
            const int b = __builtin_amdgcn_readfirstlane(t / m_ntpb);
            tl = t - b * m_ntpb;
            return b;
        } else {
            tl = t;
            return 0;
        }
    };
    auto ld_ids = [&](int tg, int (&idv)[NF]) {
        int t;
        const int bi = batch_of(tg, t);
        const int* ids = MB ? Mp->ids[bi] : ids0;
        const int m = min(t * 16 + r, B - 1);                    // rows past the end re-read the last sample, never stored
        const int* row = ids + (size_t)m * A.F;
#pragma unroll
        for (int f = 0; f < NF; ++f) idv[f] = ((V1_XP & 1) && f > 0) ? (int)((unsigned)idv[0] % (unsigned)A.vocab[f]) : row[A.col[f]];
    };
    auto issue_gather = [&](int tg, const int (&idv)[NF], Set& S) {
        int t;
        const int bi = batch_of(tg, t);
        const float* dense = MB ? Mp->dense[bi] : dense0;
        const int m = min(t * 16 + r, B - 1);
        {
            const float* nrow = dense + (size_t)m * A.ND;
            const int last = A.n_num - 1;
            // slots beyond n_num hold a duplicate finite value that only ever meets zero weights
            S.xa = (V1_XP & 16) ? (float)q : nrow[min(q, last)];
            S.xb = (V1_XP & 16) ? (float)q : nrow[min(q + 4, last)];
        }
        unsigned sid[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            bad |= (unsigned)(idv[f] + 1) > (unsigned)A.vocab[f];              // neither a table row nor the "missing" marker -1
            sid[f] = min((unsigned)idv[f], (unsigned)A.vocab[f]);              // -1 / out of range -> the zero row at index vocab
        }
        if ((ONE && NV == 4) || (PC == 1 && A.tab)) {                          // (wave-uniform) narrow rows: the derived {E | w1} table.  ONE is only dispatched with it (host_setup_pairs.h): no second path,
                                                                  // whose join in front of the scoring stage made hipcc's waitcnt pass put a vmcnt(0) there (build/sparrow.s, round 5)
            const char* tb = reinterpret_cast<const char*>(A.tab);
            unsigned ro[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) ro[f] = (sid[f] + A.rowbase[f]) * 128u;
#pragma unroll
            for (int f = 0; f < NF; ++f) S.x[f][0] = ((V1_XP & 2) && f >= 3) ? f32x4{(float)ro[f], 0.f, 0.f, 0.f} : (q < NV ? *reinterpret_cast<const f32x4*>(tb + (ro[f] + 16u * q)) : zero);
            if constexpr (SEP) {
#pragma unroll
                for (int d = 0; d < V1_MAX_DEEP; ++d) {
                    S.x[NF + d][0] = zero;
                    if (d < A.n_deep) {                           // (wave-uniform)
                        const unsigned rd = A.pack ? ro[d] + (unsigned)A.pack : (sid[d] + A.rowbase[NF + d]) * 128u;
                        if (q < NV) S.x[NF + d][0] = (V1_XP & 8) ? f32x4{(float)rd, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(tb + (rd + 16u * q));
                    }
                }
            }
            // first order: lane (r,q) fetches field q's weight, then field q+4's -- float 16 of the row it has just asked for
            unsigned oa = ro[0], ob = ro[NF > 4 ? 4 : 0];
            if (NF > 1) oa = q == 1 ? ro[NF > 1 ? 1 : 0] : oa;
            if (NF > 2) oa = q == 2 ? ro[NF > 2 ? 2 : 0] : oa;
            if (NF > 3) oa = q == 3 ? ro[NF > 3 ? 3 : 0] : oa;
            if (NF > 5) ob = q == 1 ? ro[NF > 5 ? 5 : 0] : ob;
            if (NF > 6) ob = q == 2 ? ro[NF > 6 ? 6 : 0] : ob;
            if (NF > 7) ob = q == 3 ? ro[NF > 7 ? 7 : 0] : ob;
            const float* pa = reinterpret_cast<const float*>(tb + (oa + 64u));
            if (A.w1c) {                                          // (wave-uniform) fields 0 .. n_deep-1: the compact array
                const unsigned ci = (q == 1 ? sid[1] + A.w1cbase[1] : sid[0] + A.w1cbase[0]);
                pa = q < A.n_deep ? A.w1c + ci : pa;
            }
            S.w1a = (q < NF) ? *pa : 0.f;
            S.w1b = (NF > 4 && q + 4 < NF && !(V1_XP & 4)) ? *reinterpret_cast<const float*>(tb + (ob + 64u)) : 0.f;
            return;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int pc = 0; pc < PC; ++pc)
                S.x[f][pc] = 4 * pc + q < NV ? ld4(A.table[f] + (size_t)sid[f] * A.row_floats + 16 * pc + 4 * q) : zero;
        if constexpr (SEP) {
#pragma unroll
            for (int d = 0; d < V1_MAX_DEEP; ++d)
#pragma unroll
                for (int pc = 0; pc < PC; ++pc)
                    S.x[NF + d][pc] = (d < A.n_deep && 4 * pc + q < NV) ? ld4(A.table[NF + d] + (size_t)sid[d] * A.row_floats + 16 * pc + 4 * q) : zero;
        }
        {
            // first order: lane (r,q) fetches field q's weight, then field q+4's
            const float* pa = A.w1[0] + sid[0];
            if (NF > 1) pa = q == 1 ? A.w1[NF > 1 ? 1 : 0] + sid[NF > 1 ? 1 : 0] : pa;
            if (NF > 2) pa = q == 2 ? A.w1[NF > 2 ? 2 : 0] + sid[NF > 2 ? 2 : 0] : pa;
            if (NF > 3) pa = q == 3 ? A.w1[NF > 3 ? 3 : 0] + sid[NF > 3 ? 3 : 0] : pa;
            S.w1a = (q < NF) ? *pa : 0.f;
            S.w1b = 0.f;
            if (NF > 4) {
                const float* pb = A.w1[NF > 4 ? 4 : 0] + sid[NF > 4 ? 4 : 0];
                if (NF > 5) pb = q == 1 ? A.w1[NF > 5 ? 5 : 0] + sid[NF > 5 ? 5 : 0] : pb;
                if (NF > 6) pb = q == 2 ? A.w1[NF > 6 ? 6 : 0] + sid[NF > 6 ? 6 : 0] : pb;
                if (NF > 7) pb = q == 3 ? A.w1[NF > 7 ? 7 : 0] + sid[NF > 7 ? 7 : 0] : pb;
                S.w1b = (q + 4 < NF) ? *pb : 0.f;
            }
        }
    };
    float rna[H0C], rnb[H0C];             // numerics' A operands: rows (nb*16 + r) of W0[:, numerics]^T, columns q and q + 4
    auto compute = [&](const Set& S) -> float {
#pragma clang fp contract(off)
        // ---- deep0 (DeepFM.py:106-107), the part that needs NO gathered row: bias + numerics on f32 MFMA.  [r5] In FRONT of the pair dots:
        //      the numerics are the first loads a task issues, the rows the last; with the pair dots first (rounds 2-4) hipcc's vmcnt(0) sat in
        //      front of all 44 MFMAs and of every LDS read of the stage (scripts/r04/isa_wait_sequence.py), i.e. nothing ran under the rows'
        //      flight.  z's own chain (first-order + pair dots, then the head's fmas) is unchanged: same bits. ----
        //      Only in the one-task shape with full-width rows (config 2): measured 10.81 -> 10.63 us there, but 20.7 -> 21.6 us in the looped
        //      kernel of config 4 and worse for DeepFM.py's 3-piece rows (profiles/r05/experiments/r05_11).
        constexpr bool NUM_FIRST = ONE && NV == 4;
        f32x4 h0[H0C];
        auto deep0_numerics = [&]() {
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb) h0[nb] = ld4(smem + LD::off_b0 + nb * 16 + 4 * q);
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb) h0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rna[nb], S.xa, h0[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb) h0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rnb[nb], S.xb, h0[nb], 0, 0, 0);
        };
        if constexpr (NUM_FIRST) deep0_numerics();
        // ---- pair dots (per-lane partials over this lane's 4 columns; the sum over q is part of the final reduction).  Only the
        //      real pairs (their head weight is wave-uniform: an SGPR test), two packed multiplies + two packed FMAs per pair and
        //      16-float chunk (round 2: all NF (NF - 1) / 2 products, 8 scalar instructions each) ----
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 pa = {0.f, 0.f}, pb = {0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int b = a + 1; b < NF; ++b) {
                const float hw = A.pw[a * V1_MAX_FIELDS + b];       // wave-uniform (SGPR); zero when (a,b) is not a pair
                if (hw == 0.f) continue;
                const f32x2 hw2 = {hw, hw};
#pragma unroll
                for (int pc = 0; pc < PC; ++pc) {
                    const f32x4 xa = S.x[a][pc], xb = S.x[b][pc];
                    pa = __builtin_elementwise_fma(f32x2{xa.x, xa.y} * f32x2{xb.x, xb.y}, hw2, pa);
                    pb = __builtin_elementwise_fma(f32x2{xa.z, xa.w} * f32x2{xb.z, xb.w}, hw2, pb);
                }
            }
        float z = (S.w1a + S.w1b) + ((pa[0] + pa[1]) + (pb[0] + pb[1]));
        if constexpr (!NUM_FIRST) deep0_numerics();
        // the deep fields' chunks in K order: chunk c = field c / PC, pieces 16 (c % PC) + 4q; K block b = chunks 2b, 2b + 1
        f32x4 ec[2 * PC];
#pragma unroll
        for (int c = 0; c < 2 * PC; ++c) ec[c] = (c / PC == 0 || A.n_deep > 1) ? S.x[DB + c / PC][c % PC] : zero;
        if constexpr (DYN) {
            // the deep fields' rows come out of tables whose max |E| is known at finalize: ONE static power-of-two scale puts it
            // in [2^14, 2^15) (e_scale; refused by the dynamic-range guard for tables with outlier rows) and spares the
            // per-sample maximum, its cross-lane reduction and the scale arithmetic; e_scale = 0: per-sample scale
            float scale = A.e_scale, inv = A.e_inv;
            if (A.e_scale == 0.f) {                               // wave-uniform
                float mx = 0.f;
#pragma unroll
                for (int c = 0; c < 2 * PC; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, __builtin_fabsf(ec[c][j]));
                mx = rows4_max(mx);
                dyn_scale(mx, A.inv_w0_scale, scale, inv);
            }
            int wfo = LD::off_w0e + lane * 4;                     // this lane's 16 bytes inside a 1-KB fragment (k_dyn_pack_w: lane order)
            asm volatile("" : "+v"(wfo));                         // (keeps the loop-invariant LDS reads inside the task loop)
            f32x4 acc[H0C];
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb) acc[nb] = zero;
#pragma unroll
            for (int b = 0; b < PC; ++b) {
                din_f16x8 bh, bl;
                dyn_split8(ec[2 * b], ec[2 * b + 1], scale, bh, bl);
#pragma unroll
                for (int nb = 0; nb < H0C; ++nb) {
                    const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(smem + wfo + ((nb * PC + b) * 2 + 0) * 256));
                    const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(smem + wfo + ((nb * PC + b) * 2 + 1) * 256));
                    acc[nb] = mfma_f16(ah, bh, acc[nb]);
                    acc[nb] = mfma_f16(ah, bl, acc[nb]);
                    acc[nb] = mfma_f16(al, bh, acc[nb]);
                }
            }
#pragma unroll
            for (int nb = 0; nb < H0C; ++nb) h0[nb] = acc[nb] * inv + h0[nb];
        } else {
            int w0o = LD::off_w0e + r * LD::SE + 4 * q;
            asm volatile("" : "+v"(w0o));
#pragma unroll
            for (int c = 0; c < 2 * PC; ++c) {
                f32x4 a[H0C];
#pragma unroll
                for (int nb = 0; nb < H0C; ++nb) a[nb] = ld4(smem + w0o + nb * 16 * LD::SE + 16 * c);
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int nb = 0; nb < H0C; ++nb)
                        h0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb][st], ec[c][st], h0[nb], 0, 0, 0);
            }
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
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, h0[nb][j]);
            mx = rows4_max(mx);
            float scale, inv;
            dyn_scale(mx, A.inv_w1_scale, scale, inv);
            f32x4 acc[H1C];
#pragma unroll
            for (int n1 = 0; n1 < H1C; ++n1) acc[n1] = zero;
            int wfo = LD::off_w1 + lane * 4;
            asm volatile("" : "+v"(wfo));
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
        for (int n1 = 0; n1 < H1C; ++n1) {
            const f32x4 hd = ld4(smem + LD::off_hd + n1 * 16 + 4 * q), hr = relu4_fast(h1[n1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) z = fmaf(hd[j], hr[j], z);
        }
        z = rows4_sum(z);
        return sigmoidf_acc(z + A.head_bias);
    };
    auto store = [&](int tg, float score) {
        int tl;
        const int bo = batch_of(tg, tl);
        float* out = MB ? Mp->out[bo] : out0;
        const int mm = tl * 16 + r;
        if (q == 0 && mm < B) out[mm] = score;
    };

    // ---- prologue: ids of this wave's first two tasks, then the weight image by LDS-DMA inside their latency ----
    int tA = blockIdx.x * WAVES + wave, tB = tA + task_stride;
    int idA[NF], idB[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { idA[f] = -1; idB[f] = -1; }
    if (tA < ntasks) ld_ids(tA, idA);
    if (!ONE && tB < ntasks) ld_ids(tB, idB);
#pragma unroll 1
    for (int c = wave; c < LD::total_pad / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(A.image + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
    Set SA, SB;
    if constexpr (ONE) {
        // the wave's only task.  [r4] The workgroup meets with its ids and DMA pieces in, BEFORE the rows are requested (k_chain_v2j1.h:
        // behind the requests every wave waited for the slowest issuer of the workgroup)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (tA < ntasks) issue_gather(tA, idA, SA);
#pragma unroll
        for (int nb = 0; nb < H0C; ++nb) {
            rna[nb] = smem[LD::off_w0n + (nb * 16 + r) * 8 + q];
            rnb[nb] = smem[LD::off_w0n + (nb * 16 + r) * 8 + q + 4];
        }
        if (tA < ntasks) store(tA, compute(SA));
    } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): ids and this wave's DMA pieces
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nb = 0; nb < H0C; ++nb) {
        rna[nb] = smem[LD::off_w0n + (nb * 16 + r) * 8 + q];
        rnb[nb] = smem[LD::off_w0n + (nb * 16 + r) * 8 + q + 4];
    }
    if (tA >= ntasks) {
        // a wave without work leaves after the barrier
    } else if (PC == 1 && ntasks <= 2 * task_stride) {
        // at most two tasks per wave (B <= 65 536 on a full chip): both gathers in flight before the first scoring stage
        // (narrow rows only: two sets of wide rows would not fit the register budget of 2 waves per SIMD)
        issue_gather(tA, idA, SA);
        if (tB < ntasks) issue_gather(tB, idB, SB);
        store(tA, compute(SA));
        if (tB < ntasks) store(tB, compute(SB));
    } else {
        issue_gather(tA, idA, SA);
        for (int tk = tA; tk < ntasks; tk += task_stride) {
            const Set cur = SA;
            if (tk + task_stride < ntasks) {                      // next task's rows fly under this task's arithmetic
                issue_gather(tk + task_stride, idB, SA);
                if (tk + 2 * task_stride < ntasks) ld_ids(tk + 2 * task_stride, idB);
            }
            store(tk, compute(cur));
        }
    }
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

template <int NF, int NV, int H0C, int H1C, int WAVES, bool DYN, bool SEP>
__global__ __launch_bounds__(WAVES * 64, 2) void k_deepfm_pairs(const V1Run A, const int* __restrict__ ids, const float* __restrict__ dense,
                                                                float* __restrict__ out, int B, int* __restrict__ err) {
    v1_body<NF, NV, H0C, H1C, WAVES, DYN, false, SEP>(A, ids, dense, out, B, err, nullptr);
}
template <int NF, int NV, int H0C, int H1C, int WAVES, bool DYN, bool SEP>
__global__ __launch_bounds__(WAVES * 64, 2) void k_deepfm_pairs_many(const V1Run A, const V1Many M, int B, int* __restrict__ err) {
    v1_body<NF, NV, H0C, H1C, WAVES, DYN, true, SEP>(A, nullptr, nullptr, nullptr, B, err, &M);
}
// one task per wave, four waves per SIMD: the strict one-batch launch (narrow rows, split-f16 form)
template <int NF, int NV, int H0C, int H1C, int WAVES, bool SEP>
__global__ __launch_bounds__(WAVES * 64, 4) void k_deepfm_pairs1(const V1Run A, const int* __restrict__ ids, const float* __restrict__ dense,
                                                                 float* __restrict__ out, int B, int* __restrict__ err) {
    v1_body<NF, NV, H0C, H1C, WAVES, true, false, SEP, true>(A, ids, dense, out, B, err, nullptr);
}

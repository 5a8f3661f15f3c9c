// k_rows_chain.h -- k_rows_chain: the "everything that depends on one id is one table row" forward, for graphs whose
// first layers are LINEAR in each field's embedding.  Included inside sparrow_hip.hip's anonymous namespace, after
// k_chain_v2j.h (whose lane mapping, task pipeline and staging helpers it shares).
//
// DeepFM_v2.py:98-155 (the reference's literal shape: Dense(64) projections, which k_deepfm_v2_joint's one-chunk row
// layout does not cover) and NeuralCF.py:45-53 (the model the Jetty server calls) both have this form:
//
//     score = sigmoid( c + sum_f s_f[id_f] + fn.x + hfm.(S*S - pn*pn) + hd.relu(W1 relu(hpre) + b1) )
//     S     = sum_f P_f[id_f] + pn                  pn   = Wn x + bn           (FM sum; absent for NeuralCF: KPC = 0)
//     hpre  = sum_f Q_f[id_f] + M x + c0            Q_f  = W0_f^T P_f          (first hidden layer, linear in every field)
//
// with, per field f and id v (built once at sprk_finalize by k_rows_build):
//     P_f[v] = Wp_f^T E_f[v] + bp_f                 the field's Dense projection        (DeepFM_v2.py:113-116)
//     Q_f[v] = W0[:, f-th block]^T P_f[v]           its share of deep0's pre-activation (DeepFM_v2.py:124-125; NeuralCF.py:48-49
//                                                   with P = E: the concat feeds Dense(10) directly)
//     s_f[v] = h0w*w1_f[v] - hfm.P_f[v]^2           first-order weight + its share of the FM sum of squares
// and M = W0[:, numeric block]^T Wn, c0 = b0 + W0[:, numeric block]^T bn folded on the host.  Rows {P | Q} of the big
// fields live in ONE global buffer (RB bytes per row, a multiple of 64 so a row never straddles more lines than it must),
// their scalars in a compact float array that stays L2-resident; the rows {P | Q | s} of small-vocabulary fields (the
// 19-entry genre lists) live in LDS.  Per 16-sample task a wave issues G_BIG*(KPC+H0C) 16-byte row gathers (+1 scalar
// gather) straight into the registers the scoring stage consumes: the sums are plain fp32 VALU adds in the C/D layout of
// the MFMA that follows (lane (r,q) holds elements 16*nb + 4q .. 4q+3 of sample r), so nothing is converted, split or
// staged.  The matrix pipe only sees what is genuinely per-sample: Wn x and M x (K = 8 numerics: two 16x16x4 steps per
// 16 outputs, k = q + 4s so that steps 2..3 of a K = 16 chunk never exist) and deep1 (K = H0).
//
// Arithmetic: exact fp32 throughout (f32 MFMA + VALU); only the association differs from the reference's (a row's P and Q
// are rounded once at finalize).
//
// UNF ("unfolded" big rows, round 3).  A row {P | Q} of the literal DeepFM_v2 is 96 floats = 384 bytes for an embedding of 10
// floats: the fold trades bytes for flops, and PMC showed the strict launch moving 1.61x its algorithmic bytes at ~5 TB/s, i.e.
// bound by the traffic the fold created.  With UNF the big fields' table holds the RAW embedding, pre-split into f16 hi / lo
// halves with a static power-of-two scale (16 padded values: 32 + 32 bytes, ONE 64-byte row per id instead of six 64-byte
// pieces), and {sum_f P_f | sum_f Q_f} = [A_0 | A_1] [E_0 ; E_1] is computed per task on the matrix pipe: the two big fields'
// 16 values are the K = 32 of one v_mfma_f32_16x16x32_f16 (lane (r, q) gathers the 8 halfs k = 8q .. 8q+7 of sample r: field
// q >> 1, half q & 1), the (KPC + H0C) static A fragments A = {Wp^T | (Wp W0_f)^T} (hi and lo) come out of the LDS image, three
// products per 16 outputs (hi.hi + lo.hi + hi.lo: 22 significand bits), and the result lands in the very C/D layout the folded
// rows were summed in.  The projections' biases move into constants (cP in the image, cQ folded into c0); per-id scalars are
// unchanged.  Per sample: 2 x (64 + 4) bytes instead of 2 x (384 + 4).

#define RC_MAX_BIG 3
#define RC_MAX_SMALL 3
#define RC_MB 64

struct RowsRun {
    int F, ND, n_num;
    int big_col[RC_MAX_BIG];              // ids column of big field b
    int big_vocab[RC_MAX_BIG];
    unsigned big_rowbase[RC_MAX_BIG];     // first row of field b inside `rows` (its block has vocab + 1 rows, the last one = "no id")
    unsigned big_scal[RC_MAX_BIG];        // first float of field b's scalars inside `scal` (vocab + 1 floats)
    int s_col[RC_MAX_SMALL];              // ids columns of the LDS-resident fields
    int s_vocab[RC_MAX_SMALL];
    int s_off[RC_MAX_SMALL];              // float offset of small field f's rows inside the small block
    int small_floats;                     // size of the small block (multiple of 256 floats = one LDS-DMA piece per wave)
    const float* rows;                    // big fields' rows {P[KP] | Q[H0] | pad}, RB bytes each
    const float* scal;                    // big fields' per-id scalars
    const float* small;                   // small fields' rows {P | Q | s | pad}, SS floats each (device image of the LDS block)
    float bias;                           // every constant term of the logit
    int flags;                            // 1 = ids/dense not 16-byte aligned: stage element-wise
    float unscale;                        // UNF: 1 / (row scale * fragment scale)
};
struct RowsMany {                         // several batches per launch (sprk_set_many_batches), see V2JMany
    const int* ids[RC_MB];
    const float* dense[RC_MB];
    float* out[RC_MB];
    int n, ntpb;
};

template <int KPC, int H0C, int H1C, bool HASNUM, bool UNF = false>
struct RowsLds {
    static constexpr int KP = KPC * 16, H0 = H0C * 16, H1 = H1C * 16;
    static constexpr int SS = KP + H0 + 4;            // floats per small-field row ((SS / 4) odd: LDS banks)
    static constexpr int RB = UNF ? 64 : (((KP + H0) * 4 + 63) & ~63);   // bytes per big-field row (UNF: 16 hi + 16 lo halfs)
    static constexpr int SN = 12;                     // row stride of the K = 8 numeric matrices
    static constexpr int S1 = H0 + 4;                 // deep1 W^T row stride
    static constexpr int off_wn = 0;                  // Wn^T [KP][SN]           (HASNUM)
    static constexpr int off_bn = off_wn + (HASNUM ? KP * SN : 0);   // bn [KP]
    static constexpr int off_m = off_bn + (HASNUM ? KP : 0);         // M [H0][SN]
    static constexpr int off_c0 = off_m + (HASNUM ? H0 * SN : 0);    // c0 [H0]
    static constexpr int off_w1 = off_c0 + H0;        // W1^T [H1][S1]
    static constexpr int off_b1 = off_w1 + H1 * S1;
    static constexpr int off_hfm = off_b1 + H1;       // [KP]
    static constexpr int off_hd = off_hfm + KP;       // [H1]
    static constexpr int off_fn = off_hd + H1;        // [8]
    static constexpr int off_cp = off_fn + 8;         // UNF: sum of the big fields' projection biases [KP]
    static constexpr int off_af = (off_cp + (UNF ? KP : 0) + 3) & ~3;   // UNF: A fragments, (KPC + H0C) x {hi, lo} x 256 floats
    static constexpr int total = off_af + (UNF ? (KPC + H0C) * 2 * 256 : 0);
    static constexpr int total_pad = (total + 255) & ~255;
    static constexpr int stage_floats = 256;          // per-wave ids / numerics slot
    static_assert((SS / 4) % 2 == 1, "small-row stride must be an odd number of 16-byte slots");
};

// One-time (finalize) kernel, one wave per table row v:
//   src  = KP > 0 ? P = bp + Wp^T E[v]  (k order of k_v2_fold: within each 16-chunk s outer, q inner)  :  E[v]
//   Q[m] = sum_n W0t[m][col0 + n] * src[n]  (+ qbias[m] when given)
//   s    = h0w * w1[v] - sum_n hfm[n] * P[n]^2            (0 when w1 == NULL)
//   out[v*out_stride ..] = {P (KP) | Q (H0) | (scal_in_row ? s : nothing)}, scal_out[v] = s when scal_out != NULL
static __global__ __launch_bounds__(256) void k_rows_build(const float* __restrict__ table, int row_floats, long long rows,
                                                    const float* __restrict__ Wp, int ldp, const float* __restrict__ bp, int KP,
                                                    const float* __restrict__ W0t, int ld0, int col0, int H0, int nsrc,
                                                    const float* __restrict__ qbias,
                                                    const float* __restrict__ w1, const float* __restrict__ hfm, int n_hfm, float h0w,
                                                    float* __restrict__ out, int out_stride, float* __restrict__ scal_out,
                                                    int scal_in_row) {
    __shared__ float sP[4][64];
    const int wv = threadIdx.x >> 6, n = threadIdx.x & 63;
    for (long long v = (long long)blockIdx.x * 4 + wv; v < rows; v += (long long)gridDim.x * 4) {
        const float* x = table + v * row_floats;
        float p = 0.f;
        if (KP > 0) {
            if (n < KP) {
                const float* w = Wp + (size_t)n * ldp;
                p = bp ? bp[n] : 0.f;
                for (int c = 0; c < row_floats; c += 16)
                    for (int s = 0; s < 4; ++s)
                        for (int q = 0; q < 4; ++q) {
                            const int k = c + 4 * q + s;
                            if (k < row_floats) p = fmaf(w[k], x[k], p);
                        }
            }
        } else if (n < nsrc) {
            p = x[n];
        }
        sP[wv][n] = p;                                          // one wave: LDS operations complete in issue order
        float sqw = (KP > 0 && n < n_hfm && n < KP) ? hfm[n] * p * p : 0.f;
        for (int d = 32; d >= 1; d >>= 1) sqw += __shfl_xor(sqw, d);
        float* o = out ? out + v * out_stride : nullptr;  // (out == NULL: scalars only -- the UNF tables hold raw rows)
        if (o && n < KP) o[n] = p;
        for (int m = n; o && m < H0; m += 64) {
            const float* w = W0t + (size_t)m * ld0 + col0;
            float acc = qbias ? qbias[m] : 0.f;
            for (int k = 0; k < nsrc; ++k) acc = fmaf(w[k], sP[wv][k], acc);
            o[KP + m] = acc;
        }
        const float s = w1 ? h0w * w1[v] - sqw : 0.f;
        if (n == 0) {
            if (scal_out) scal_out[v] = s;
            if (o && scal_in_row) o[KP + H0] = s;
        }
    }
}

// UNF, one-time: raw embedding rows -> {W hi halfs | W lo halfs} of E * scale, W = 16 or 32 (values beyond row_floats: 0)
static __global__ __launch_bounds__(256) void k_rows_unf_split(const float* __restrict__ table, int row_floats, long long rows, float scale,
                                                        _Float16* __restrict__ out, int W = 16) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * W; i += (long long)gridDim.x * 256) {
        const long long v = i / W;
        const int k = (int)(i - v * W);
        const float x = k < row_floats ? table[v * row_floats + k] * scale : 0.f;
        const _Float16 hi = (_Float16)x;
        out[v * 2 * W + k] = hi;
        out[v * 2 * W + W + k] = (_Float16)(x - (float)hi);
    }
}
typedef _Float16 rows_f16x8 __attribute__((ext_vector_type(8)));
template <int G_BIG, int KPC, int H0C>
struct RowsSetUnf {
    rows_f16x8 ehi, elo;                                  // this lane's 8 halfs of the K = 32 operand (field q >> 1, half q & 1)
    int so[RC_MAX_SMALL];
    float xa, xb;
    float sc;
};

template <int G_BIG, int KPC, int H0C>
struct RowsSet {
    f32x4 xp[G_BIG > 0 ? G_BIG : 1][KPC > 0 ? KPC : 1];   // big fields' P pieces
    f32x4 xq[G_BIG > 0 ? G_BIG : 1][H0C];                 // ... Q pieces
    int so[RC_MAX_SMALL];                                 // small fields: LDS float offset of this sample's row
    float xa, xb;                                         // numerics q and q + 4 of this sample
    float sc;                                             // per-id scalar fetched by this lane
};

// a.b + acc as an explicit fma chain: the scoring stage is inlined at several call sites (two-task fast path, task loop,
// one- and many-batch kernels) and must round identically in all of them, so nothing is left to -ffp-contract
__device__ __forceinline__ float dot4_fma(f32x4 a, f32x4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    return fmaf(a.w, b.w, acc);
}

// ONE: one task per wave, no loop -- the strict one-batch launch at up to four waves per SIMD (k_chain_v2j1.h has the measurements)
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM, int WAVES, bool MB, bool ONE = false, bool UNF = false>
__device__ __forceinline__ void rows_chain_body(const RowsRun& A, const int* __restrict__ ids, const float* __restrict__ dense,
                                                float* __restrict__ out, int B, int* __restrict__ err,
                                                const float* __restrict__ image, const RowsMany* __restrict__ Mp) {
    using LD = RowsLds<KPC, H0C, H1C, HASNUM, UNF>;
    using Set = typename std::conditional<UNF, RowsSetUnf<G_BIG, KPC, H0C>, RowsSet<G_BIG, KPC, H0C>>::type;
    static_assert(!UNF || (G_BIG <= 2 && KPC > 0), "UNF: two big fields are one K = 32 block");
    constexpr int KP = LD::KP, H0 = LD::H0;
    constexpr unsigned RB = LD::RB;
    constexpr bool HASFM = KPC > 0;
    static_assert(G_BIG >= 1 && G_BIG <= RC_MAX_BIG && NJF >= 0 && NJF <= RC_MAX_SMALL, "field split");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m_n = MB ? Mp->n : 1, m_ntpb = MB ? Mp->ntpb : 0;
    const int ntasks = MB ? m_n * m_ntpb : (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    const int wave_global = blockIdx.x * WAVES + wave;
    auto batch_of = [&](int tk, int& tl) {
        if constexpr (MB) {
            const int b = __builtin_amdgcn_readfirstlane(tk / m_ntpb);
            tl = tk - b * m_ntpb;
            return b;
        } else {
            tl = tk;
            return 0;
        }
    };
    float* stage = smem + LD::total_pad + wave * LD::stage_floats;
    const float* small_s = smem + LD::total_pad + WAVES * LD::stage_floats;
    const float* wq = smem + 4 * q;
    bool bad = false;
    const bool aligned = !(A.flags & 1);
    auto clampt = [&](int tk) { return tk < ntasks ? tk : ntasks - 1; };

    auto ld_raw = [&](int tkg, f32x4& raw) {
        int tk;
        const int bi = batch_of(tkg, tk);
        const int* ids_b = MB ? Mp->ids[bi] : ids;
        const float* dense_b = MB ? Mp->dense[bi] : dense;
        if (aligned && tk * 16 + 16 <= B) {                       // wave-uniform
            const bool isid = lane < 32 || !HASNUM;
            const int j = lane < 32 ? lane : lane - 32;
            const int n4 = 4 * (isid ? A.F : A.ND);
            const float* src = isid ? reinterpret_cast<const float*>(ids_b) + (size_t)tk * 16 * A.F
                                    : dense_b + (size_t)tk * 16 * A.ND;
            raw = ld4(src + 4 * (j < n4 ? j : 0));
        }
    };
    auto gather = [&](int tkg, const f32x4& raw, Set& S) {
        int tk;
        const int bi = batch_of(tkg, tk);
        const int* ids_b = MB ? Mp->ids[bi] : ids;
        const float* dense_b = MB ? Mp->dense[bi] : dense;
        if (aligned && tk * 16 + 16 <= B) {
            const bool isid = lane < 32;
            const int j = isid ? lane : lane - 32;
            const int n4 = 4 * (isid ? A.F : (HASNUM ? A.ND : 0));
            if (j < n4) st4(stage + (isid ? 0 : 128) + 4 * j, raw);
        } else {
            stage_task_slow(stage, ids_b, dense_b, A.F, HASNUM ? A.ND : 0, tk, B, lane);
        }
        const int* sid_row = reinterpret_cast<const int*>(stage) + r * A.F;
        unsigned sid[G_BIG], rid[G_BIG];
#pragma unroll
        for (int b = 0; b < G_BIG; ++b) {
            const int id = sid_row[A.big_col[b]];
            bad |= (unsigned)(id + 1) > (unsigned)A.big_vocab[b];
            rid[b] = min((unsigned)id, (unsigned)A.big_vocab[b]);            // -1 / out of range -> the "no id" row at index vocab
            sid[b] = rid[b] + A.big_rowbase[b];
        }
#pragma unroll
        for (int f = 0; f < NJF; ++f) {
            const int id = sid_row[A.s_col[f]];
            bad |= (unsigned)(id + 1) > (unsigned)A.s_vocab[f];
            S.so[f] = A.s_off[f] + (int)min((unsigned)id, (unsigned)A.s_vocab[f]) * LD::SS;
        }
        if constexpr (HASNUM) {
            const float* nrow = stage + 128 + r * A.ND;
            const int last = A.n_num - 1;
            // slots beyond n_num hold a duplicate finite value that only ever meets zero weights
            S.xa = nrow[min(q, last)];
            S.xb = nrow[min(q + 4, last)];
        }
        const char* tb = reinterpret_cast<const char*>(A.rows);
        if constexpr (UNF) {
            // lane (r, q): field q >> 1 (a lane beyond the last field reads field 0's all-zero "no id" row), halfs 8 (q & 1) ..
            unsigned sq = sid[0];
            if (G_BIG > 1) sq = (q >> 1) ? sid[G_BIG > 1 ? 1 : 0] : sq;
            else sq = (q >> 1) ? (unsigned)A.big_vocab[0] + A.big_rowbase[0] : sq;
            const size_t ro = (size_t)sq * RB + 16u * (q & 1);
            S.ehi = *reinterpret_cast<const rows_f16x8*>(tb + ro);
            S.elo = *reinterpret_cast<const rows_f16x8*>(tb + ro + 32u);
        } else {
#pragma unroll
        for (int b = 0; b < G_BIG; ++b) {
            const size_t ro = (size_t)sid[b] * RB + 16u * q;
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) S.xp[b][nb] = *reinterpret_cast<const f32x4*>(tb + ro + 64u * nb);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) S.xq[b][n0] = *reinterpret_cast<const f32x4*>(tb + ro + 4u * KP + 64u * n0);
        }
        }
        if constexpr (HASFM) {
            // per-id logit terms of the big fields: lane (r,q) fetches big field q's scalar
            unsigned s0 = rid[0] + A.big_scal[0], s1 = rid[G_BIG > 1 ? 1 : 0] + A.big_scal[G_BIG > 1 ? 1 : 0],
                     s2 = rid[G_BIG > 2 ? 2 : 0] + A.big_scal[G_BIG > 2 ? 2 : 0];
            asm("" : "+v"(s0), "+v"(s1), "+v"(s2));           // (keeps the select chain out of a private array, see k_chain_v2j.h)
            unsigned so = s0;
            if (G_BIG > 1) so = q == 1 ? s1 : so;
            if (G_BIG > 2) so = q == 2 ? s2 : so;
            S.sc = A.scal[so];
        }
    };

    // ---- register-resident weights (filled once, after the image barrier): the MFMA A operands.  Biases and output
    //      weights are read from the LDS image where they are used (one ds_read_b128 each per task): at KPC = 4 keeping
    //      them in registers as well spills (two gather sets of G_BIG * (KPC + H0C) float4 are live across the scoring) ----
    float rwna[HASFM && HASNUM ? KPC : 1], rwnb[HASFM && HASNUM ? KPC : 1];   // Wn^T rows (nb*16 + r), columns q and q + 4
    float rma[H0C], rmb[H0C];                                                 // M rows (n0*16 + r), columns q and q + 4
    f32x4 rW1[H1C][H0C];
    float rfa = 0.f, rfb = 0.f;
    auto load_weights = [&]() {
        if constexpr (HASNUM) {
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) {
                rwna[nb] = smem[LD::off_wn + (nb * 16 + r) * LD::SN + q];
                rwnb[nb] = smem[LD::off_wn + (nb * 16 + r) * LD::SN + q + 4];
            }
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) {
                rma[n0] = smem[LD::off_m + (n0 * 16 + r) * LD::SN + q];
                rmb[n0] = smem[LD::off_m + (n0 * 16 + r) * LD::SN + q + 4];
            }
            rfa = smem[LD::off_fn + q];
            rfb = smem[LD::off_fn + q + 4];
        }
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1)
#pragma unroll
            for (int j = 0; j < H0C; ++j) rW1[n1][j] = ld4(wq + LD::off_w1 + (n1 * 16 + r) * LD::S1 + 16 * j);
    };

    // ---- scoring stage ----
    auto compute = [&](const Set& S) -> float {
#pragma clang fp contract(off)
        float zz = 0.f;
        // per-sample linear parts on the matrix pipe: pn = Wn x + bn, hpre = M x + c0 (K = 8: steps k = q, k = q + 4)
        f32x4 pn[HASFM ? KPC : 1], hp[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) hp[n0] = ld4(wq + LD::off_c0 + n0 * 16);
        if constexpr (HASNUM) {
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) {
                pn[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rwna[nb], S.xa, ld4(wq + LD::off_bn + nb * 16), 0, 0, 0);
            }
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) hp[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rma[n0], S.xa, hp[n0], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) pn[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rwnb[nb], S.xb, pn[nb], 0, 0, 0);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) hp[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rmb[n0], S.xb, hp[n0], 0, 0, 0);
            // numeric first-order partial: every numeric once per sample (lane q holds numerics q and q + 4)
            zz = fmaf(rfb, S.xb, rfa * S.xa);
        } else if constexpr (HASFM) {
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) pn[nb] = zero;
        }
        if constexpr (HASFM) zz += (q < G_BIG) ? S.sc : 0.f;
        // field sums: big fields from the gathered registers, small fields from their LDS rows
        f32x4 s[HASFM ? KPC : 1];
        f32x4 hq[H0C];
        if constexpr (UNF) {
            // {sum_f P_f | sum_f Q_f} = [A_0 | A_1] [E_0 ; E_1]: three split-f16 products per 16 outputs, fragments from the image
            const float* af = smem + LD::off_af + 4 * lane;
#pragma unroll
            for (int nb = 0; nb < KPC + H0C; ++nb) {
                const rows_f16x8 ahi = __builtin_bit_cast(rows_f16x8, ld4(af + (2 * nb) * 256));
                const rows_f16x8 alo = __builtin_bit_cast(rows_f16x8, ld4(af + (2 * nb + 1) * 256));
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, S.ehi, zero, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, S.elo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, S.ehi, acc, 0, 0, 0);
                const f32x4 v = f32x4{acc.x * A.unscale, acc.y * A.unscale, acc.z * A.unscale, acc.w * A.unscale};
                if (nb < KPC) s[nb < KPC ? nb : 0] = ld4(wq + LD::off_cp + (nb < KPC ? nb : 0) * 16) + v;
                else hq[nb >= KPC ? nb - KPC : 0] = v;
            }
        } else {
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) {
            s[nb] = S.xp[0][nb];
#pragma unroll
            for (int b = 1; b < G_BIG; ++b) s[nb] += S.xp[b][nb];
        }
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) {
            hq[n0] = S.xq[0][n0];
#pragma unroll
            for (int b = 1; b < G_BIG; ++b) hq[n0] += S.xq[b][n0];
        }
        }
        float ssc = 0.f;
#pragma unroll
        for (int f = 0; f < NJF; ++f) {
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) s[nb] += ld4(small_s + S.so[f] + 16 * nb + 4 * q);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) hq[n0] += ld4(small_s + S.so[f] + KP + 16 * n0 + 4 * q);
            ssc += small_s[S.so[f] + KP + H0];
        }
        if (NJF > 0) zz += (q == 3) ? ssc : 0.f;
        float z = zz;
        if constexpr (HASFM) {
            // FM cross (DeepFM_v2.py:147-152): sum_n hfm[n] (S_n^2 - sum_g P_g[n]^2); the fields' squares sit in their
            // scalars, the numeric group's are subtracted here
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) {
                const f32x4 t = s[nb] + pn[nb], u = pn[nb] * pn[nb];
                const f32x4 d = f32x4{fmaf(t.x, t.x, -u.x), fmaf(t.y, t.y, -u.y), fmaf(t.z, t.z, -u.z), fmaf(t.w, t.w, -u.w)};
                z = dot4_fma(ld4(wq + LD::off_hfm + nb * 16), d, z);
            }
        }
        f32x4 h0[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = relu4_fast(hp[n0] + hq[n0]);
        // second hidden layer + output weights; two chains (even / odd K step)
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) {
            f32x4 e = ld4(wq + LD::off_b1 + n1 * 16), o = zero;
#pragma unroll
            for (int j = 0; j < H0C; ++j) {
                e = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].x, h0[j].x, e, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].y, h0[j].y, o, 0, 0, 0);
                e = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].z, h0[j].z, e, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].w, h0[j].w, o, 0, 0, 0);
            }
            z = dot4_fma(ld4(wq + LD::off_hd + n1 * 16), relu4_fast(e + o), z);
        }
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        return sigmoidf_fast(z + A.bias);
    };
    auto store = [&](int tkg, float score) {
        int tk;
        const int bi = batch_of(tkg, tk);
        float* out_b = MB ? Mp->out[bi] : out;
        const int m = tk * 16 + r;
        if (q == 0 && m < B) out_b[m] = score;
    };

    // ---- prologue / task pipeline: as k_deepfm_v2_joint ----
    Set SA, SB;
    f32x4 rawA = zero, rawB = zero;
    int tA = wave_global, tB = wave_global + task_stride;
    if (ntasks > 0) {
        if (!ONE || tA < ntasks) ld_raw(clampt(tA), rawA);
        if (!ONE) ld_raw(clampt(tB), rawB);
    }
#pragma unroll 1
    for (int c = wave; c < LD::total_pad / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
    if constexpr (NJF > 0) {
#pragma unroll 1
        for (int c = wave; c < A.small_floats / 256; c += WAVES)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(A.small + c * 256 + lane * 4),
                (__attribute__((address_space(3))) void*)(smem + LD::total_pad + WAVES * LD::stage_floats + c * 256), 16, 0, 0);
    }
    constexpr int NG = (UNF ? 2 : G_BIG * (KPC + H0C)) + (HASFM ? 1 : 0);   // VMEM loads per gather
    // s_waitcnt vmcnt(NG), lgkmcnt / expcnt untouched: vmcnt is a 6-bit field split over bits [3:0] and [15:14]
    constexpr int WAIT_NG = 0x0F70 | (NG & 15) | ((NG >> 4) << 14);
    static_assert(NG < 64, "vmcnt field");
    if constexpr (ONE) {
        // the wave's only task: rows requested before the barrier; in-order retirement => "at most NG outstanding" = DMA landed
        if (tA < ntasks) {
            gather(tA, rawA, SA);
            __builtin_amdgcn_s_waitcnt(WAIT_NG);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        __builtin_amdgcn_s_barrier();
        load_weights();
        if (tA < ntasks) store(tA, compute(SA));
    } else {
    constexpr bool FAST2 = NG < 16;                                   // (round 2's encoding: vmcnt < 16)
    const bool two = FAST2 && tA < ntasks && ntasks <= 2 * task_stride;
    if (two) {
        gather(tA, rawA, SA);
        __builtin_amdgcn_s_waitcnt(0x0F70 | (FAST2 ? NG : 0));        // s_waitcnt vmcnt(NG): ids + DMA landed, A's rows may fly
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    load_weights();
    if (tA >= ntasks) {
        // a wave without work leaves after the barrier
    } else if (two) {
        gather(clampt(tB), rawB, SB);
        store(tA, compute(SA));
        if (tB < ntasks) store(tB, compute(SB));
    } else {
        gather(tA, rawA, SA);
        ld_raw(clampt(tA + 2 * task_stride), rawA);
        gather(clampt(tB), rawB, SB);
        ld_raw(clampt(tB + 2 * task_stride), rawB);
        for (;;) {
            store(tA, compute(SA));
            tA += 2 * task_stride;
            gather(clampt(tA), rawA, SA);
            ld_raw(clampt(tA + 2 * task_stride), rawA);
            if (tB >= ntasks) break;
            store(tB, compute(SB));
            tB += 2 * task_stride;
            gather(clampt(tB), rawB, SB);
            ld_raw(clampt(tB + 2 * task_stride), rawB);
            if (tA >= ntasks) break;
        }
    }
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM, int WAVES, bool UNF = false>
__global__ __launch_bounds__(WAVES * 64, 2) void k_rows_chain(const RowsRun A, const int* __restrict__ ids,
                                                              const float* __restrict__ dense, float* __restrict__ out, int B,
                                                              int* __restrict__ err, const float* __restrict__ image) {
    rows_chain_body<KPC, H0C, H1C, G_BIG, NJF, HASNUM, WAVES, false, false, UNF>(A, ids, dense, out, B, err, image, nullptr);
}
// one task per wave (strict one-batch launch): register budget of three waves per SIMD (one gather set instead of two)
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM, int WAVES, bool UNF = false>
__global__ __launch_bounds__(WAVES * 64, UNF ? 4 : 3) void k_rows_chain1(const RowsRun A, const int* __restrict__ ids,
                                                               const float* __restrict__ dense, float* __restrict__ out, int B,
                                                               int* __restrict__ err, const float* __restrict__ image) {
    rows_chain_body<KPC, H0C, H1C, G_BIG, NJF, HASNUM, WAVES, false, true, UNF>(A, ids, dense, out, B, err, image, nullptr);
}
// several batches per launch: the per-batch pointer table travels in the kernel arguments (the kernarg segment IS the
// cheapest transport for 1.5 KB that change every launch: +0.05 us of host time per launch measured by
// scripts/ubench/launch_floor.hip, no device-side cost); only this instantiation carries it
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM, int WAVES, bool UNF = false>
__global__ __launch_bounds__(WAVES * 64, 2) void k_rows_chain_many(const RowsRun A, const RowsMany M, int B,
                                                                   int* __restrict__ err, const float* __restrict__ image) {
    rows_chain_body<KPC, H0C, H1C, G_BIG, NJF, HASNUM, WAVES, true, false, UNF>(A, nullptr, nullptr, nullptr, B, err, image, &M);
}

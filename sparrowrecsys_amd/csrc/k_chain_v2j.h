// k_chain_v2j.h -- k_deepfm_v2_joint: the register-chained DeepFM_v2 forward (k_chain_v2.h, FOLD + REG)
// with the SMALL-vocabulary fields folded one step further and kept in LDS.  Included inside
// sparrow_hip.hip's anonymous namespace, after k_chain_v2.h.
//
// Everything field g contributes to a score is a function of its id alone (DeepFM_v2.py:106-126,147-155):
//     P_g[id]                       its projected embedding  -> FM sum S = sum_g P_g and deep0's input
//     W0_g^T P_g[id]                its share of deep0's pre-activation (deep0 is linear in the concat)
//     h0w*w1_g[id] - hfm.P_g[id]^2  its first-order weight and its share of the FM sum of squares
// For a field with a handful of rows (the 19-entry genre vocabularies) all of that is a tiny table:
// k_v2_fold_small builds, once at sprk_finalize, rows {P (KP) | W0_g^T P (H0) | row scalar} for every id of
// every small field (3 x 20 rows x 208 B = 12.5 KB at BASELINE config 2; deep0's bias rides in the first
// field's rows), the kernel stages them in LDS next to the weight image, and a sample's small fields cost
// three ds_read_b128 per field at scoring time -- no HBM/L2 gather, no MFMA: deep0 shrinks from 7 K-chunks
// to 4, the per-sample gather from 6 rows + 2 row scalars to 3 rows + 1.
// (A first version kept ONE joint table over the tuple of small ids in global memory -- 20^3 rows x 256 B =
// 2 MB, a single gather per sample.  PMC showed what that costs: every XCD's 4 MB L2 streams 3 MB of
// big-table lines per launch and keeps evicting the joint table, +77 k line fetches = +10 MB of fabric
// traffic per 65 536-sample launch, 26 % above the algorithmic bytes.  In LDS the small fields cost no
// memory traffic beyond a 12.5 KB broadcast read per workgroup.)
//
// HALF: the big fields' share of deep0 and of the FM sum runs on v_mfma_f32_16x16x32_f16 with SPLIT
// operands instead of f32 MFMA.  On gfx950 the f32-input MFMA issues at the f32 vector rate and shares the
// SIMD's vector ALU with every VALU instruction (measured, profiles/r01/ubench_*); the f16 MFMA runs 16x
// faster per FLOP and overlaps VALU work.  A value x (scaled by a power of two so that max|x| lands near
// 2^15) is stored as two halfs hi = f16(x), lo = f16(x - hi): hi + lo carries 22 significand bits.  The
// folded row of a big field holds, per 16-byte piece q, [hi(P[4q..4q+3]) | lo(P[4q..4q+3])] -- the same
// 64 bytes per row as the 16 floats it replaces, gathered by the same load -- and IS the B operand of the
// K = 32 instruction (8 halfs per lane).  Against A = [Whi | Whi] it yields Whi.(hi + lo), against
// A = [Wlo | Wlo] the correction Wlo.(hi + lo): two MFMAs per (field, 16 outputs) for the full fp32-class
// product (every partial product is exact in the f32 accumulator), one more against a 0/1 selection
// matrix adds hi + lo into the FM sum.  16 samples: 15 f16 MFMAs (16 cycles each, VALU keeps issuing)
// instead of 24 f32 MFMAs (32 cycles each, VALU blocked) + 12 VALU adds.  Error of the split: <= 2^-21
// relative per operand, i.e. fp32 class (tests hold the same tolerances as the f32 path).
//
// Lane mapping, task pipeline and the "pure MFMA stream" scoring stage are those of k_deepfm_v2_chain
// (see k_chain_v2.h): lane (r = lane&15, q = lane>>4) is sample r's q-th 16-byte column slot.

#define V2J_MAX_BIG 4
#define V2J_MAX_JF 3
#define V2J_SS 52                         // floats per small-field row: 16 + 32 + 1, padded so that (SS/4) is odd (LDS banks)

struct V2JRun {
    int F, ND, n_num;
    int big_col[V2J_MAX_BIG];             // ids column of big field b
    int big_vocab[V2J_MAX_BIG];
    unsigned big_rowbase[V2J_MAX_BIG];    // first row of field b inside tab0 ([KP+16]-float rows, vocab+1 rows per field)
    int big_grp[V2J_MAX_BIG];             // its group index in the model's field order (selects the W0 K-chunk)
    int j_col[V2J_MAX_JF];                // ids columns of the small (LDS-resident) fields
    int j_vocab[V2J_MAX_JF];
    int s_off[V2J_MAX_JF];                // float offset of small field f's rows inside the small-table block
    int small_floats;                     // size of the small-table block (multiple of 256 floats = one LDS-DMA piece per wave)
    int wf_off;                           // HALF: float offset, inside the small-table block, of the numerics' fold Wf [H0][8]
    const float* tab0;                    // folded rows of the big fields {P | row scalar | 0..}
    const float* small;                   // small fields' rows {P | W0^T P (+ b0 in field 0) | row scalar | 0 0 0}, V2J_SS floats each
    float h0w, fo_bias, head_bias;
    int flags;                            // 1 = ids/dense not 16-byte aligned: stage element-wise
    // HALF: tab0 rows hold split halfs of P * p_scale; the deep0 weights of the big fields are split after
    // multiplying by w_scale (powers of two chosen at finalize from the tables' / kernel's max |.|)
    float w_scale, unscale_h, unscale_s;  // 2^sW, 2^-(sP+sW), 2^-sP
};

// Multi-batch launch (sprk_forward_many with sprk_set_many_batches > 1): one launch scores up to V2J_MB batches of B rows,
// each with its own ids / dense / out buffers; task t of the launch is task t % ntpb of batch t / ntpb.  A dependent launch
// chain costs ~3.3 us per launch on this stack, more than a third of a 65 536-row forward; one launch per 16 (up to 64)
// batches spends it once.  Results are bit-identical to per-batch launches (samples are independent).
#define V2J_MB 64
struct V2JMany {
    const int* ids[V2J_MB];
    const float* dense[V2J_MB];
    float* out[V2J_MB];
    int n;                                // batches in this launch
    int ntpb;                             // tasks per batch = ceil(B / 16)
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// x*scale -> (hi, lo) halfs with hi + lo == x*scale to 22 bits
__device__ __forceinline__ void split_half4(f32x4 w, float scale, f16x4& hi, f16x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = w[j] * scale;
        const _Float16 h = (_Float16)x;
        hi[j] = h;
        lo[j] = (_Float16)(x - (float)h);
    }
}

// One-time (finalize) kernels for HALF.
// max |x| over the first `ncols` floats of each row -> *out (bits of a non-negative float, atomicMax as uint)
static __global__ __launch_bounds__(256) void k_v2_absmax(const float* __restrict__ rows, long long nrows, int row_floats,
                                                   int ncols, unsigned* __restrict__ out) {
    float m = 0.f;
    const long long total = nrows * ncols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / ncols;
        const int c = (int)(i - v * ncols);
        const float a = fabsf(rows[v * row_floats + c]);
        m = (a > m || a != a) ? a : m;                            // NaN propagates (the host then refuses HALF)
    }
    for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_xor(m, d); m = (o > m || o != o) ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
// How many non-zero entries sit more than 2^20 below max |x| (thresh = max * 2^-20) -> out[0], non-zero entries -> out[1].
// With the static scale that puts max |x| in [2^14, 2^15) such an entry's lo half is an f16 subnormal (resolution 2^-24 of
// the scaled value): it keeps fewer than ~20 significand bits, so a table full of them is not fp32-class on split f16.
static __global__ __launch_bounds__(256) void k_v2_count_small(const float* __restrict__ rows, long long nrows, int row_floats, int ncols,
                                                        float thresh, unsigned long long* __restrict__ out) {
    unsigned small = 0, nz = 0;
    const long long total = nrows * ncols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / ncols;
        const int c = (int)(i - v * ncols);
        const float a = fabsf(rows[v * row_floats + c]);
        nz += a > 0.f;
        small += (a > 0.f && a < thresh);
    }
    for (int d = 32; d >= 1; d >>= 1) { small += __shfl_xor(small, d); nz += __shfl_xor(nz, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, (unsigned long long)small); atomicAdd(out + 1, (unsigned long long)nz); }
}
// folded fp32 rows {P[16] | scalar | 0..} -> split rows {[hi4|lo4] x 4 | scalar | 0..} of P * scale
static __global__ __launch_bounds__(256) void k_v2_split_rows(const float* __restrict__ src, float* __restrict__ dst,
                                                       long long nrows, float scale) {
    const long long total = nrows * 8;                           // 8 sixteen-byte pieces per 128-byte row
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i >> 3;
        const int pc = (int)(i & 7);
        const f32x4 in = ld4(src + v * 32 + 4 * pc);
        if (pc < 4) {
            f16x4 hi, lo;
            split_half4(in, scale, hi, lo);
            f16x8 o = {hi[0], hi[1], hi[2], hi[3], lo[0], lo[1], lo[2], lo[3]};
            *reinterpret_cast<f16x8*>(dst + v * 32 + 4 * pc) = o;
        } else {
            st4(dst + v * 32 + 4 * pc, in);
        }
    }
}

// One-time (finalize) kernel: LDS rows of ONE small field from its folded rows (k_v2_fold output):
// out[v] = { P[v] (KP) | (add_b0 ? b0 : 0) + W0_g^T P[v] (H0) | row scalar | 0.. }, one wave per row.
static __global__ __launch_bounds__(256) void k_v2_fold_small(const float* __restrict__ folded, int KP, int H0, int grp,
                                                       const float* __restrict__ W0, int ldw0,   // deep0 W^T [H0][ldw0]
                                                       const float* __restrict__ b0, int add_b0,
                                                       float* __restrict__ out, int rows) {
    const int FS = KP + 16;
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < rows; v += gridDim.x * 4) {
        const int col = threadIdx.x & 63;
        const float* prow = folded + (size_t)v * FS;
        float acc = 0.f;
        if (col < KP) {
            acc = prow[col];
        } else if (col < KP + H0) {
            const int m = col - KP;
            acc = add_b0 ? b0[m] : 0.f;
            const float* w = W0 + (size_t)m * ldw0 + grp * KP;
            for (int k = 0; k < KP; ++k) acc = fmaf(w[k], prow[k], acc);
        } else if (col == KP + H0) {
            acc = prow[KP];
        }
        if (col < V2J_SS) out[(size_t)v * V2J_SS + col] = acc;
    }
}

// One-time (finalize) kernel for HALF: the numeric group folded through deep0.  deep0 is linear in its input and the
// numeric group's projection pn = Wn x + bn is linear in the (at most 8) numerics x (DeepFM_v2.py:118-125), so its share
// of deep0's pre-activation is (W0n Wn) x + W0n bn:
//     wf[m][k]  = sum_n W0[m][num_off + n] * Wn[n][k]         [H0][8], zero for k >= n_num  -> two K = 4 MFMA steps per
//                                                              16 outputs (k = q + 4s) instead of four on a dependent pn
//     rows0[v][KP + m] += sum_n W0[m][num_off + n] * bn[n]      the constant rides with b0 in the first small field's rows
// (double accumulation, one rounding.)  pn itself is still formed in the kernel -- the FM sum needs it.
static __global__ __launch_bounds__(256) void k_v2j_fold_num(const float* __restrict__ W0, int ldw0, int num_off,
                                                      const float* __restrict__ Wn, int ldn, const float* __restrict__ bn,
                                                      int n_num, int KP, int H0, float* __restrict__ wf,
                                                      float* __restrict__ rows0, int nrows0) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H0 * 8; i += gridDim.x * 256) {
        const int m = i >> 3, k = i & 7;
        double acc = 0.0;
        if (k < n_num)
            for (int n = 0; n < KP; ++n) acc += (double)W0[(size_t)m * ldw0 + num_off + n] * (double)Wn[(size_t)n * ldn + k];
        wf[i] = (float)acc;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nrows0 * H0; i += gridDim.x * 256) {
        const int v = i / H0, m = i - v * H0;
        double acc = 0.0;
        for (int n = 0; n < KP; ++n) acc += (double)W0[(size_t)m * ldw0 + num_off + n] * (double)bn[n];
        rows0[(size_t)v * V2J_SS + KP + m] = (float)((double)rows0[(size_t)v * V2J_SS + KP + m] + acc);
    }
}

// The VALU arithmetic of the scoring stage is written with explicit fused multiply-adds under `fp contract(off)`: hipcc's
// default (contract = fast) decides per instantiation which a*b + c pairs become one v_fma, so two instantiations of the
// same source -- the looped kernel and its one-task-per-wave twin (k_chain_v2j1.h), or two field splits -- could differ in
// the last bit.  With the contraction pinned they are bit-identical by construction.
__device__ __forceinline__ f32x4 fma4s(f32x4 a, float b, f32x4 c) {
    return f32x4{__builtin_fmaf(a.x, b, c.x), __builtin_fmaf(a.y, b, c.y), __builtin_fmaf(a.z, b, c.z), __builtin_fmaf(a.w, b, c.w)};
}
__device__ __forceinline__ float dot4f(f32x4 a, f32x4 b) {
    return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)));
}
// s^2 - p^2 per element (FM cross minus the numeric group's own squares)
__device__ __forceinline__ f32x4 sq_diff4(f32x4 s, f32x4 p) {
    return f32x4{__builtin_fmaf(s.x, s.x, -(p.x * p.x)), __builtin_fmaf(s.y, s.y, -(p.y * p.y)), __builtin_fmaf(s.z, s.z, -(p.z * p.z)),
                 __builtin_fmaf(s.w, s.w, -(p.w * p.w))};
}

// everything one 16-sample task needs from memory, in the (r,q) lane layout
template <int G_BIG, int NJF>
struct V2JSet {
    f32x4 x[G_BIG];       // big fields' row pieces: 4 floats of P, or [hi4 | lo4] halfs (HALF)
    int so[NJF];          // small fields: LDS float offset of this sample's row
    f32x4 xn;             // numerics: x[4q .. 4q+3]; HALF: {x[q], x[q+4]} only (K = 8 as two MFMA steps)
    float w1a;            // per-id logit terms fetched by this lane
};

// Task pipeline: TWO gather sets per wave (A, B), each the direct operand of its scoring stage.  A wave's
// first two tasks are gathered back to back at kernel entry, so at B = 65 536 (4 096 tasks on 2 048
// resident waves: exactly two per wave) every row of the batch is requested as soon as its ids have
// landed and the memory system streams without a bubble; afterwards the loop alternates
// score(A), gather(A'), score(B), gather(B') with ids fetched two tasks ahead, i.e. one gather is
// always in flight under a scoring stage.  Loads are unconditional (task indices are clamped, the tail
// re-gathers a valid task and drops the result) so that the in-order vmcnt waits stay exact.
template <int G_BIG, int NJF, int KPC, int H0C, int H1C, int WAVES, bool HALF, bool MB>
__device__ __forceinline__ void v2j_body(const V2JRun& A, const int* __restrict__ ids, const float* __restrict__ dense,
                                         float* __restrict__ out, int B, int* __restrict__ err,
                                         const float* __restrict__ image, const V2JMany* __restrict__ Mp) {
    constexpr int G_EMB = G_BIG + NJF;
    using LD = V2Lds<G_EMB, 4, KPC, H0C, H1C, true>;          // the weight image is the FOLD image of the whole model
    using Set = V2JSet<G_BIG, NJF>;
    constexpr int KP = LD::KP, H0 = H0C * 16;
    constexpr unsigned RB = (KP + 16) * 4;                    // bytes per folded row
    static_assert(KP + H0 + 1 <= V2J_SS, "small-field row layout");
    constexpr int NKR = HALF ? 1 : (G_BIG + 1) * KPC;         // deep0 K chunks on f32 MFMA: numerics (+ the big fields unless HALF)
    static_assert(KPC == 1, "row layout: one 16-float chunk of projections per field");
    static_assert(G_BIG >= 1 && G_BIG <= 3 && NJF >= 1 && NJF <= V2J_MAX_JF, "field split");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m_ntpb = MB ? Mp->ntpb : 0;
    const int ntasks = MB ? Mp->n * m_ntpb : (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    const int wave_global = blockIdx.x * WAVES + wave;
    // MB: (batch, task inside the batch) of launch task tk (wave-uniform), and that batch's buffers
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
    const float* small_s = smem + LD::total_pad + WAVES * LD::stage_floats;   // small fields' rows
    const float* wq = smem + 4 * q;
    bool bad = false;
    const bool aligned = !(A.flags & 1);
    auto clampt = [&](int tk) { return tk < ntasks ? tk : ntasks - 1; };
    const int* const ids0 = ids;
    const float* const dense0 = dense;

    // ---- gather stage ----
    //   ld_raw : the task's contiguous ids / numerics blocks, one 16-B load per lane
    //   gather : VGPR -> wave-private LDS slot -> the (r,q) lanes that need them, then the row gathers
    auto ld_raw = [&](int tkg, f32x4& raw) {
        int tk;
        const int bi = batch_of(tkg, tk);
        const int* ids_b = MB ? Mp->ids[bi] : ids;
        const float* dense_b = MB ? Mp->dense[bi] : dense;
        if (aligned && tk * 16 + 16 <= B) {                       // wave-uniform
            const bool isid = lane < 32;
            const int j = isid ? lane : lane - 32;
            const int n4 = 4 * (isid ? A.F : A.ND);
            const float* src = isid ? reinterpret_cast<const float*>(ids_b) + (size_t)tk * 16 * A.F
                                    : dense_b + (size_t)tk * 16 * A.ND;
            raw = ld4(src + 4 * (j < n4 ? j : 0));
        }
    };
    auto gather = [&](int tkg, const f32x4& raw, Set& S) {
        int tk;
        const int bi = batch_of(tkg, tk);
        const int* ids = MB ? Mp->ids[bi] : ids0;
        const float* dense = MB ? Mp->dense[bi] : dense0;
        if (aligned && tk * 16 + 16 <= B) {
            const bool isid = lane < 32;
            const int j = isid ? lane : lane - 32;
            const int n4 = 4 * (isid ? A.F : A.ND);
            if (j < n4) st4(stage + (isid ? 0 : 128) + 4 * j, raw);
        } else {
            stage_task_slow(stage, ids, dense, A.F, A.ND, tk, B, lane);
        }
        // one wave: LDS operations complete in issue order, no barrier needed
        const int* sid_row = reinterpret_cast<const int*>(stage) + r * A.F;
        unsigned sid[G_BIG];
#pragma unroll
        for (int b = 0; b < G_BIG; ++b) {
            const int id = sid_row[A.big_col[b]];
            bad |= (unsigned)(id + 1) > (unsigned)A.big_vocab[b];             // neither a table row nor the "missing" marker -1
            sid[b] = min((unsigned)id, (unsigned)A.big_vocab[b]) + A.big_rowbase[b];   // -1 / out of range -> the zero row at index vocab
        }
#pragma unroll
        for (int f = 0; f < NJF; ++f) {
            const int id = sid_row[A.j_col[f]];
            bad |= (unsigned)(id + 1) > (unsigned)A.j_vocab[f];
            S.so[f] = A.s_off[f] + (int)min((unsigned)id, (unsigned)A.j_vocab[f]) * V2J_SS;    // -1 -> the "missing" row at index vocab
        }
        {
            const float* nrow = stage + 128 + r * A.ND;
            const int c0 = 4 * q, last = A.n_num - 1;
            // lane slots beyond n_num hold a duplicate finite value that only ever meets zero weights
            if constexpr (HALF) {
                S.xn.x = nrow[min(q, last)];
                S.xn.y = nrow[min(q + 4, last)];
            } else {
                S.xn.x = nrow[min(c0 + 0, last)];
                S.xn.y = nrow[min(c0 + 1, last)];
                S.xn.z = nrow[min(c0 + 2, last)];
                S.xn.w = nrow[min(c0 + 3, last)];
            }
        }
        const char* tb = reinterpret_cast<const char*>(A.tab0);
#pragma unroll
        for (int b = 0; b < G_BIG; ++b) S.x[b] = *reinterpret_cast<const f32x4*>(tb + (sid[b] * RB + 16u * q));
        // per-id logit terms of the big fields: lane (r,q) fetches big field q's row scalar
        {
            // (the offsets pass through an empty asm: left visible, the select chain over q is turned into a
            // dynamically indexed private array -- scratch memory traffic in the gather)
            unsigned s0 = sid[0] * RB, s1 = sid[G_BIG > 1 ? 1 : 0] * RB, s2 = sid[G_BIG > 2 ? 2 : 0] * RB;
            asm("" : "+v"(s0), "+v"(s1), "+v"(s2));
            unsigned so = s0;
            if (G_BIG > 1) so = q == 1 ? s1 : so;
            if (G_BIG > 2) so = q == 2 ? s2 : so;
            S.w1a = *reinterpret_cast<const float*>(tb + (so + 4u * KP));
        }
    };

    // ---- register-resident weights (filled once, after the image barrier) ----
    f32x4 rW0[H0C][NKR], rW1[H1C][H0C];
    f32x4 rwn[KPC], rbpn[KPC], rb1[H1C], rhfm[KPC], rhd[H1C];
    f32x4 rfn = zero;
    float rwn8[2] = {0.f, 0.f}, rfn8[2] = {0.f, 0.f}, rwf[H0C][2];   // HALF: numerics with k = q + 4s (Wn, first order, Wf)
    f16x8 hWa[HALF ? G_BIG : 1][H0C], hWb[HALF ? G_BIG : 1][H0C];   // HALF: [Whi|Whi], [Wlo|Wlo] fragments of the big fields
    f16x8 hSel = {0, 0, 0, 0, 0, 0, 0, 0};                         // HALF: 0/1 selection A operand: D[n] = hi[n] + lo[n]
    auto load_weights = [&]() {
        const float* w0r = wq + LD::off_w0 + r * LD::S0;
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) {
            if constexpr (HALF) {
                rwf[n0][0] = small_s[A.wf_off + (n0 * 16 + r) * 8 + q];              // numerics folded through deep0
                rwf[n0][1] = small_s[A.wf_off + (n0 * 16 + r) * 8 + q + 4];
            } else {
                rW0[n0][0] = ld4(w0r + n0 * 16 * LD::S0 + 16 * G_EMB);                 // numeric chunk
            }
#pragma unroll
            for (int b = 0; b < G_BIG; ++b) {
                const f32x4 w = ld4(w0r + n0 * 16 * LD::S0 + 16 * A.big_grp[b]);
                if constexpr (HALF) {
                    f16x4 hi, lo;
                    split_half4(w, A.w_scale, hi, lo);
                    hWa[b][n0] = f16x8{hi[0], hi[1], hi[2], hi[3], hi[0], hi[1], hi[2], hi[3]};
                    hWb[b][n0] = f16x8{lo[0], lo[1], lo[2], lo[3], lo[0], lo[1], lo[2], lo[3]};
                } else {
                    rW0[n0][(HALF ? 0 : 1 + b)] = w;
                }
            }
        }
        if constexpr (HALF) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hSel[e] = (4 * q + (e & 3) == r) ? (_Float16)1.0f : (_Float16)0.0f;
        }
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) {
#pragma unroll
            for (int j = 0; j < H0C; ++j) rW1[n1][j] = ld4(wq + LD::off_w1 + (n1 * 16 + r) * LD::S1 + 16 * j);
            rb1[n1] = ld4(wq + LD::off_b1 + n1 * 16);
            rhd[n1] = ld4(wq + LD::off_hd + n1 * 16);
        }
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) {
            rwn[nb] = ld4(wq + LD::off_wn + (nb * 16 + r) * LD::SN);
            rbpn[nb] = ld4(wq + LD::off_bp + G_EMB * KP + nb * 16);
            rhfm[nb] = ld4(wq + LD::off_hfm + nb * 16);
        }
        rfn = ld4(smem + LD::off_fn + 4 * (q & 1));
        if constexpr (HALF) {
            rwn8[0] = smem[LD::off_wn + r * LD::SN + q];
            rwn8[1] = smem[LD::off_wn + r * LD::SN + q + 4];
            rfn8[0] = smem[LD::off_fn + q];
            rfn8[1] = smem[LD::off_fn + q + 4];
        }
    };

    // ---- scoring stage ----
    auto compute = [&](const Set& S) -> float {
#pragma clang fp contract(off)
        const f32x4 pnum = S.xn;
        // numeric group's Dense projection (DeepFM_v2.py:118-120): two chains (even / odd K step)
        f32x4 pn;
        if constexpr (HALF) {
            const f32x4 e = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn8[0], pnum.x, rbpn[0], 0, 0, 0);
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn8[1], pnum.y, zero, 0, 0, 0);
            pn = e + o;
        } else {
            f32x4 e = rbpn[0], o = zero;
            e = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[0].x, pnum.x, e, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[0].y, pnum.y, o, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[0].z, pnum.z, e, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[0].w, pnum.w, o, 0, 0, 0);
            pn = e + o;
        }
        // this lane's share of the per-id logit terms + numeric first-order partial (rfn = h0w * fo_num weights)
        float zz = ((q < G_BIG) ? S.w1a : 0.f) + (HALF ? __builtin_fmaf(rfn8[1], pnum.y, rfn8[0] * pnum.x) : ((q < 2) ? dot4f(rfn, pnum) : 0.f));
        // small fields, from their LDS rows: P -> FM sum, W0^T P (+ b0) -> deep0's accumulators, row scalars (one q row adds them)
        f32x4 sp = ld4(small_s + S.so[0] + 4 * q), sq[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) sq[n0] = ld4(small_s + S.so[0] + KP + 16 * n0 + 4 * q);
        float ssc = small_s[S.so[0] + KP + H0];
#pragma unroll
        for (int f = 1; f < NJF; ++f) {
            sp += ld4(small_s + S.so[f] + 4 * q);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) sq[n0] += ld4(small_s + S.so[f] + KP + 16 * n0 + 4 * q);
            ssc += small_s[S.so[f] + KP + H0];
        }
        zz += (q == 3) ? ssc : 0.f;
        // deep0 (DeepFM_v2.py:124-125): accumulators start at b0 + sum W0 P of the small fields
        f32x4 hA[H0C], hB[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) { hA[n0] = sq[n0]; hB[n0] = zero; }
        f32x4 s = sp + pn;                                        // FM sum: small fields + numerics ...
        if constexpr (HALF) {
            // big fields on the f16 matrix pipe: per field [Whi|Whi].x and [Wlo|Wlo].x for both n-blocks and the
            // selection matrix for the FM sum
            f32x4 aFa[H0C], aFb[H0C], aS = zero;                  // 2*H0C + 1 independent accumulator chains
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) { aFa[n0] = zero; aFb[n0] = zero; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < G_BIG; ++b) {
                const f16x8 xb = __builtin_bit_cast(f16x8, S.x[b]);
#pragma unroll
                for (int n0 = 0; n0 < H0C; ++n0) aFa[n0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hWa[b][n0], xb, aFa[n0], 0, 0, 0);
                aS = __builtin_amdgcn_mfma_f32_16x16x32_f16(hSel, xb, aS, 0, 0, 0);
#pragma unroll
                for (int n0 = 0; n0 < H0C; ++n0) aFb[n0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hWb[b][n0], xb, aFb[n0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                // keep the round-robin order: the next use of a chain is 5 MFMAs away
            }
            // numerics' share on f32 MFMA, folded through their projection (k_v2j_fold_num): K = 8, independent of pn
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int n0 = 0; n0 < H0C; ++n0)
                    hA[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rwf[n0][st], st ? pnum.y : pnum.x, hA[n0], 0, 0, 0);
            s = fma4s(aS, A.unscale_s, s);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) hB[n0] = (aFa[n0] + aFb[n0]) * A.unscale_h;
        } else {
#pragma unroll
            for (int b = 0; b < G_BIG; ++b) s += S.x[b];          // ... + big fields
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NKR; c += 2) {
                const bool hb = c + 1 < NKR;
                const f32x4 pa = c == 0 ? pn : S.x[c > 0 ? c - 1 : 0];
                const f32x4 pb = !hb ? zero : S.x[hb ? c : 0];
#pragma unroll
                for (int st = 0; st < 4; ++st) {
#pragma unroll
                    for (int n0 = 0; n0 < H0C; ++n0)
                        hA[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rW0[n0][c][st], pa[st], hA[n0], 0, 0, 0);
                    if (hb) {
#pragma unroll
                        for (int n0 = 0; n0 < H0C; ++n0)
                            hB[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rW0[n0][hb ? c + 1 : c][st], pb[st], hB[n0], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        f32x4 h0[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = relu4_fast(hA[n0] + hB[n0]);
        // FM cross (DeepFM_v2.py:147-152): sum_n hfm[n] (S_n^2 - sum_g P_g[n]^2); the fields' squares are in the
        // row scalars, the numeric group's are subtracted here
        float z = dot4f(rhfm[0], sq_diff4(s, pn));
        // deep1: Dense(relu) (DeepFM_v2.py:126) + output weights; two chains (even / odd K step)
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) {
            f32x4 e = rb1[n1], o = zero;
#pragma unroll
            for (int j = 0; j < H0C; ++j) {
                e = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].x, h0[j].x, e, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].y, h0[j].y, o, 0, 0, 0);
                e = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].z, h0[j].z, e, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[n1][j].w, h0[j].w, o, 0, 0, 0);
            }
            z += dot4f(rhd[n1], relu4_fast(e + o));
        }
        // output layer: concat([first, fm, deep]) . w + b -> sigmoid (DeepFM_v2.py:154-155)
        z += zz;
        // ([r5] two permlane swaps instead of two ds_bpermute round trips through the LDS queue at the very end of the task's chain; the storing
        //  lanes (row 0) get (z0 + z1) + (z2 + z3) either way: same bits)
        z = rows4_sum(z);
        return sigmoidf_fast(z + A.h0w * A.fo_bias + A.head_bias);
    };
    auto store = [&](int tkg, float score) {
        int tk;
        const int bi = batch_of(tkg, tk);
        float* out_b = MB ? Mp->out[bi] : out;
        const int m = tk * 16 + r;
        if (q == 0 && m < B) out_b[m] = score;
    };

    // ---- prologue: ids of the first two tasks and the weight image are requested together (the image
    //      is L2-hot and lands inside the ids' memory latency), one barrier, weights -> registers, then
    //      both gather sets back to back.  Straight-line code up to the loop: every s_waitcnt vmcnt the
    //      compiler places is exact, so score(A) starts when A's rows are in, B's still streaming. ----
    Set SA, SB;
    f32x4 rawA = zero, rawB = zero;
    int tA = wave_global, tB = wave_global + task_stride;
    if (ntasks > 0) {
        ld_raw(clampt(tA), rawA);
        ld_raw(clampt(tB), rawB);
    }
    // weight image -> LDS by LDS-DMA (no VGPRs, no ds_write pass): 1-KB pieces, wave w takes w, w+WAVES, ...
#pragma unroll 1
    for (int c = wave; c < LD::total_pad / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
#pragma unroll 1
    for (int c = wave; c < A.small_floats / 256; c += WAVES)          // ... and the small fields' rows
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(A.small + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + LD::total_pad + WAVES * LD::stage_floats + c * 256), 16, 0, 0);
    // At most two tasks per wave (B <= 65 536 on a full chip): no loop, hence no loop-carried loads -- the
    // compiler counts every outstanding load exactly and score(A) waits for A's rows only, running while
    // B's rows are still streaming in.  (At a loop header hipcc falls back to waiting for ALL outstanding
    // loads before the first use of a loop-carried one.)  Gather A is issued BEFORE the image barrier: it
    // needs the ids only, and vmcnt retires in order, so "at most NG loads outstanding" means this wave's
    // DMA pieces (older) have landed while its NG row loads (younger) may still fly; gather B follows the
    // barrier (issuing it before as well measured the same, 9.2 us, at 12 more live registers).
    const bool two = tA < ntasks && ntasks <= 2 * task_stride;        // workgroup-uniform except for idle waves
    constexpr int NG = G_BIG + 1;                                     // VMEM loads per gather: rows + row scalar
    static_assert(NG < 16, "s_waitcnt immediate below encodes vmcnt < 16");
    if (two) {
        gather(tA, rawA, SA);
        __builtin_amdgcn_s_waitcnt(0x0F70 | NG);                      // s_waitcnt vmcnt(NG)
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);                           // s_waitcnt vmcnt(0): ids and DMA
    }
    __builtin_amdgcn_s_barrier();                             // every wave's pieces of the image are in LDS
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
            gather(clampt(tA), rawA, SA);                     // (past the end: re-gathers the last task, never scored)
            ld_raw(clampt(tA + 2 * task_stride), rawA);
            if (tB >= ntasks) break;
            store(tB, compute(SB));
            tB += 2 * task_stride;
            gather(clampt(tB), rawB, SB);
            ld_raw(clampt(tB + 2 * task_stride), rawB);
            if (tA >= ntasks) break;
        }
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

// One batch per launch: no per-batch pointer table in the kernel arguments at all (round 1 passed an unused 1.5 KB V2JMany
// by value on every launch).
template <int G_BIG, int NJF, int KPC, int H0C, int H1C, int WAVES, bool HALF>
__global__ __launch_bounds__(WAVES * 64, 2) void k_deepfm_v2_joint(const V2JRun A, const int* __restrict__ ids,
                                                                   const float* __restrict__ dense, float* __restrict__ out, int B,
                                                                   int* __restrict__ err, const float* __restrict__ image) {
    v2j_body<G_BIG, NJF, KPC, H0C, H1C, WAVES, HALF, false>(A, ids, dense, out, B, err, image, nullptr);
}
// Several batches per launch (sprk_set_many_batches): the table travels in the kernarg segment (+0.05 us of host time per
// launch, no device-side cost: scripts/ubench/launch_floor.hip).
template <int G_BIG, int NJF, int KPC, int H0C, int H1C, int WAVES, bool HALF>
__global__ __launch_bounds__(WAVES * 64, 2) void k_deepfm_v2_joint_many(const V2JRun A, const V2JMany M, int B, int* __restrict__ err,
                                                                        const float* __restrict__ image) {
    v2j_body<G_BIG, NJF, KPC, H0C, H1C, WAVES, HALF, true>(A, nullptr, nullptr, nullptr, B, err, image, &M);
}

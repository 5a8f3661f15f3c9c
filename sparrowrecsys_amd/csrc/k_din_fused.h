// k_din_fused.h -- k_din_fused: the whole DIN forward (reference DIN.py:132-167) in ONE launch: activation unit + weighted sum
// pooling over the history (16 samples per MFMA tile, the weights as the static operand -- k_din_cols.h's formulation), and the
// tail MLP concat([user profile, pooled history, candidate, context]) -> Dense(128) PReLU -> Dense(64) PReLU -> Dense(1, sigmoid)
// as the EPILOGUE of the same wave: the pooled vectors never leave the registers, there is no second launch.
// Included after k_din_cols.h / k_din_tail.h / dyn_split.h.
//
// What round 4 measured before writing it (scripts/ubench/issue_rates.hip, profiles/r04/ubench_issue_rates.log):
//  * k_din_attn_cols is NOT memory bound: ids drawn from a 1 024-row window (every row an L1 / L2 hit) 31.7 us, uniform ids 33.5 us.
//  * 1, 2 and 4 waves per task measure alike (2 or 4 waves per SIMD): a SIMD spends ~680 cycles per (16 samples, slot) however
//    many waves share it -- the kernel is bound by the SUM of its instructions' issue times.  On gfx950 the f16 matrix pipe and
//    the VALU overlap only partly (12 MFMAs + 60 v_fma per step: 320 cycles, against 210 for the MFMAs alone and 190 for the
//    v_fma alone), so every VALU instruction removed is time removed.
//  * per instruction and SIMD (wave64): v_fma_f32 3.1 cycles, v_pk_fma_f32 6.0 (NOT a gain over two v_fma), v_fma_mix_f32 4.6,
//    v_fma_mixlo/hi_f16 8.7 (!), v_pk_mul/fma_f16 4.4, v_exp / v_rcp 8.3, v_permlane16/32_swap 8.2, v_cvt_pk_f16_f32 4.3,
//    v_mfma_f32_16x16x32_f16 17.6.  k_din_attn_cols spent 139 of its ~340 VALU cycles per slot in sixteen v_fma_mix{lo,hi}_f16
//    (forming and splitting h * c) and 61 in a cross-row sum + sigmoid that all four 16-lane rows compute redundantly.
// What changed against k_din_attn_cols, per (16 samples, slot):
//  * h * c * sP as PACKED f16 arithmetic on the halfs that are already in the registers: with h = hh + hl (the table's split) and
//    c' = c kappa = ch + cl (split once per task), ph = hh ch (v_pk_mul_f16), r = fma(hh, ch, -ph) (the EXACT rounding error of
//    ph), pl = fma(hl, ch, fma(hh, cl, r)): four packed instructions per two elements, 70 cycles instead of 139, the same 22
//    significand bits (only hl cl, 2^-22 relative, is dropped); all of it compiler-visible (no asm, no hand-placed hazard nops).
//  * slots are scored in PAIRS: the two slots' partial logits are reduced over the four 16-lane rows by ONE transposing
//    v_permlane16_swap + one v_permlane32_swap (row 0 / 2: slot a, row 1 / 3: slot b; same summation order as rows4_sum), ONE
//    sigmoid serves both, one more swap hands every lane both weights: 12 instructions per pair instead of 24, half the
//    transcendentals.
//  * eight waves per workgroup, ONE workgroup per CU (2 waves per SIMD -- measured equal to 4), which buys 256 VGPRs: the
//    quarter sums stay in registers (k_din_attn_cols parked them in LDS), rows are requested a PAIR ahead into a ring of four
//    register sets, and the LDS has room for the tail's weights (54 KB) next to the coefficient tables.
//  * the first four slots' rows are requested in the prologue's own round trip (the ids are all they need), not behind the staging
//    barrier.
//  * TAIL: the wave that owns a task's pooled vectors (time slice 0) runs the tail for its sixteen samples after the slot loop: fc0's
//    four embedding columns as FOLDED rows F_g[id] = W_g^T E_g[id] (k_din_tail.h; 4 x 512 bytes per sample straight into the
//    accumulators' layout, all 32 loads of a lane in ONE round trip -- the loop's registers are free by then), the numerics on f32
//    MFMA (K = 8 in two steps, fc0's bias in the free eighth slot), the pooled history on the f16 pipe from the registers it was
//    accumulated in (per-sample dynamic scale; A fragments in the attention's own k order), fc1 on split-f16 (dyn_split.h), PReLU,
//    Dense(1), sigmoid.  The splits use v_cvt_pk_f16_f32 (gfx950) instead of v_fma_mixlo/hi_f16.
// What was measured and NOT kept (profiles/r04/; all on one box): the slot loop as a software pipeline over pairs (the
// matrix half of pair i + 1 in one block with the VALU half of pair i): 42.5 us per fused launch against 42.2; the tail's rows
// gathered in the PROLOGUE and held in registers through the loop: 43.7 against 41.7 -- they are 67 MB per batch, a third of the
// history's bytes, and cost bandwidth wherever they are put; the fused form at several batches per launch: 39.2 us per batch against
// 34.0 for the two-launch pipeline, whose tail kernel shares the CUs with the next group's attention kernel.
// What bounds the slot loop (scripts/ubench/gather_pattern.hip): the same rows gathered by a kernel that does NOTHING else take
// 30.0 us per 32 768 x 50 slots in this lane layout (a quad of consecutive lanes touches four different rows) and 25.6 us fully
// coalesced -- 7.0 / 8.2 TB/s out of L2 + Infinity Cache; the ablated loop without its loads takes 26.4 us.  The stage is within
// 10 % of both its memory and its issue bound at once; what the fusion buys is the second launch, not a faster loop.
// Launch-shape invariance is kept: the pooled sum is always (q0 + q1) + (q2 + q3) of the history's quarters (a quarter is a
// multiple of four slots, so a trip of the slot loop never straddles one), whatever `ts` is, and the tail is one wave's fixed instruction sequence.

#define DF_WAVES 8
#define DF_MB 16
#ifndef DF_LATE_IMAGE
#define DF_LATE_IMAGE 1                   // [r5] one batch per launch: the tail's image (88 KB per workgroup) and the raw tail rows are requested BEHIND the first
#endif                                    // trip of the slot loop, not in front of it (below)
struct DinFusedRun {
    // ---- activation unit + pooling (as DinColsRun) ----
    int T, F, hist_col, cand_col, Dp, vocab;
    float b2, acc_scale, unscale, inv_h_scale, kappa;
    const float* tsplit;  // [vocab][KP floats]: per q group [hi(EL halfs) | lo(EL halfs)] of E * sH (k_din_split_table)
    const float* vc;      // [vocab][32]
    const float* frag;    // A fragments of W12 sA, W4 s4 (k_din_cols_pack)
    const float* coef;    // [2][64][36] PReLU . Dense(1) coefficient tables (k_din_cols_coef)
    int ts, ts_log2, ql, idp;
    // ---- tail (DIN.py:161-167); unused by the TAIL = false instantiations ----
    int ND, n_cols, n_num;
    int col[DT_MAX_COLS], tvocab[DT_MAX_COLS];
    const float* Ftab[DT_MAX_COLS];       // [vocab][128] folded rows of fc0 (fold_first_dense)
    float head_bias, inv_w1_scale, inv_w0p_scale;
    // UNF: columns unf_g[0 .. n_unf) of the tail's column list arrive as RAW rows Etab (k_din_tail.h: [hi 32 halfs | lo 32 halfs] * e_scale,
    // 128 bytes per id, an all-zero row at index vocab) and meet their A fragments (image + total_pad) on the matrix pipe; the other
    // columns stay folded rows (col[0 .. n_cols)).  n_unf = 0: every column folded.
    // [r5] emb_dim <= 16 (KC = 1, DIN.py as written): n_unf = 2 means BOTH K = 32 blocks of k_din_tail's emb_dim <= 16 form -- columns (0, 1) and
    // (2, 3) of the tail's column list, rows of [16 hi halfs | 16 lo halfs] = 64 bytes per id; lane (r, q) takes column 2 pb + (q >> 1), half
    // q & 1 -- i.e. EVERY embedding column is a raw row (n_cols = 0 folded ones, n_ucols of the four slots exist).
    int n_unf, ucol[4], uvocab[4], n_ucols;   // (col / tvocab / Ftab / n_cols above: the FOLDED columns only, these: the raw-row columns)
    const _Float16* Etab[4];
    float e_unscale;
    int b0_slot;                          // fc0's bias rides in this (free) numeric slot against a constant 1; -1: added by the VALU
    const float* image;                   // DinFusedImg, built once by k_din_fused_pack
};
struct DinFusedMany {
    const int* ids[DF_MB];
    const float* dense[DF_MB];
    float* out[DF_MB];                    // TAIL: scores [B]; else pooled vectors [B][Dp]
    int n;
};
struct DinFusedOne {};
template <bool MB> struct DinFusedArg { typedef DinFusedOne type; };
template <> struct DinFusedArg<true> { typedef DinFusedMany type; };

// The tail's LDS image (floats).  N0 = 128, N1 = 64 (DIN.py:163-166).
struct DinFusedImg {
    static constexpr int N0C = 8, N1C = 4, N0 = 128, N1 = 64;
    static constexpr int off_w0p = 0;                               // [N0C][hi 256 | lo 256]: fc0's pooled columns, k = EL q + e
    static constexpr int off_wn = off_w0p + N0C * 512;              // [N0C][2 steps][64 lanes]: fc0's numeric columns, f32
    static constexpr int off_w1 = off_wn + N0C * 2 * 64;            // [N1C][N0C / 2][hi 256 | lo 256]: fc1 (k_dyn_pack_w's K-block layout)
    static constexpr int off_b0 = off_w1 + N1C * (N0C / 2) * 512;   // b0[128] a0[128] b1[64] a1[64] hw[64]
    static constexpr int off_a0 = off_b0 + N0, off_b1 = off_a0 + N0, off_a1 = off_b1 + N1, off_hw = off_a1 + N1;
    static constexpr int total = off_hw + N1;
    static constexpr int total_pad = (total + 255) & ~255;
    // UNF (emb_dim 17..32): up to two LARGE-vocabulary embedding columns of fc0 (DIN: userId, the candidate's movieId) as raw split
    // rows on the matrix pipe instead of folded rows -- their A fragments [2 columns][N0C][hi 256 | lo 256] behind the image
    static constexpr int unf_floats = 2 * N0C * 512;
    // LDS-DMA pieces (256 floats) of image + UNF fragments, the same number for each of the eight waves (k_din_fused's waits count them)
    static constexpr int pieces_per_wave = ((total_pad + unf_floats) / 256 + 7) / 8;
    static constexpr int dma_floats = pieces_per_wave * 8 * 256;
};
constexpr int DF_COEF_FLOATS = 2 * 64 * 36;

// One-time (finalize) kernel: the tail image.  W0: fc0's W^T [128][ldw0] (folded columns zeroed by fold_first_dense), pooled
// columns at p_off (Dp of them), numerics at n_off; w1frag: fc1's split fragments (make_dyn_fragments); EL = 4 KC.
static __global__ __launch_bounds__(256) void k_din_fused_pack(const float* __restrict__ W0, int ldw0, int p_off, int Dp, int n_off, int n_num,
                                                        float w0p_scale, int EL, const float* __restrict__ b0, const float* __restrict__ a0,
                                                        const float* __restrict__ w1frag, const float* __restrict__ b1,
                                                        const float* __restrict__ a1, const float* __restrict__ hw, int n_hw,
                                                        float* __restrict__ img, const float* __restrict__ w0efrag, int n_unf, int unf_g0, int unf_g1, int nblk) {
    using IM = DinFusedImg;
    const int tid = threadIdx.x;
    _Float16* wp = reinterpret_cast<_Float16*>(img + IM::off_w0p);
    for (int i = tid; i < IM::N0C * 2 * 512; i += 256) {            // halfs: [nb][hi | lo][lane][8]
        const int e = i & 7, lane = (i >> 3) & 63, hl = (i >> 9) & 1, nb = i >> 10;
        const int r = lane & 15, q = lane >> 4;
        const int d = EL * q + e;
        float x = 0.f;
        if (e < EL && d < Dp) x = W0[(size_t)(nb * 16 + r) * ldw0 + p_off + d] * w0p_scale;
        const _Float16 hi = (_Float16)x;
        wp[i] = hl ? (_Float16)(x - (float)hi) : hi;
    }
    for (int i = tid; i < IM::N0C * 2 * 64; i += 256) {
        const int lane = i & 63, s = (i >> 6) & 1, nb = i >> 7;
        const int r = lane & 15, q = lane >> 4, k = q + 4 * s;
        // numeric slot n_num (when there is one) carries fc0's bias: the kernel feeds it a constant 1
        img[IM::off_wn + i] = k < n_num ? W0[(size_t)(nb * 16 + r) * ldw0 + n_off + k] : (k == n_num ? b0[nb * 16 + r] : 0.f);
    }
    for (int i = tid; i < IM::N1C * (IM::N0C / 2) * 512; i += 256) img[IM::off_w1 + i] = w1frag[i];
    for (int i = tid; i < IM::N0; i += 256) { img[IM::off_b0 + i] = b0[i]; img[IM::off_a0 + i] = a0[i]; }
    for (int i = tid; i < IM::N1; i += 256) {
        img[IM::off_b1 + i] = b1[i];
        img[IM::off_a1 + i] = a1[i];
        img[IM::off_hw + i] = i < n_hw ? hw[i] : 0.f;
    }
    for (int i = IM::total + tid; i < IM::total_pad; i += 256) img[i] = 0.f;
    // UNF: the A fragments of columns unf_g[u] out of k_din_tail's fragment buffer w0efrag ([nb][4 columns][hi 256 | lo 256] floats)
    for (int i = IM::total_pad + IM::unf_floats + tid; i < IM::dma_floats; i += 256) img[i] = 0.f;
    for (int i = tid; i < IM::unf_floats; i += 256) {
        const int w = i & 511, nb = (i >> 9) % IM::N0C, u = i / (512 * IM::N0C);
        // (nblk: the blocks per output block in k_din_tail's fragment buffer -- 4 columns for emb_dim 17..32, 2 column PAIRS for emb_dim <= 16)
        img[IM::total_pad + i] = (w0efrag && u < n_unf) ? w0efrag[((size_t)(nb * nblk + (u == 0 ? unf_g0 : unf_g1)) * 2) * 256 + w] : 0.f;
    }
}

typedef _Float16 df_h2 __attribute__((ext_vector_type(2)));
typedef float df_f2 __attribute__((ext_vector_type(2)));

// XP: ablation bits for scripts/r04 experiments (only instantiated under -DSPRK_DF_XP): 1 no MFMAs, 2 no product split, 4 no h32,
// 8 no PReLU dot, 16 no reduce / sigmoid, 32 no pooling, 64 no row loads in the loop; TAIL: 128 no folded-row gathers, 256 no fc1, 512 no
// fc0 MFMAs (numerics + pooled) -- results are garbage, the time is the point
template <int N> struct DfInt { static constexpr int value = N; };
#ifdef SPRK_DF_XP
// XP & 1024: a timeline -- every wave stamps the constant 100 MHz clock at kernel entry, loop entry, loop exit, after fc0, after fc1, exit
#define DF_TS_WAVES 4096
static __device__ unsigned long long g_df_ts[DF_TS_WAVES * 8];
#endif
template <int KC, bool MB, bool TAIL, bool ATT = false, int XP = 0>
__global__ __launch_bounds__(DF_WAVES * 64, 2) void k_din_fused(const DinFusedRun A, const int* __restrict__ ids, const float* __restrict__ dense,
                                                              float* __restrict__ out, float* __restrict__ att, int B, int* __restrict__ err,
                                                              const typename DinFusedArg<MB>::type Mm) {
#pragma clang fp contract(off)
    constexpr int EL = 4 * KC, KP = 16 * KC, HP = 32, AS = HP + 4, ROWS = 64, NP = EL / 2;
    constexpr int N0C = DinFusedImg::N0C, N1C = DinFusedImg::N1C;
    typedef _Float16 f16xe __attribute__((ext_vector_type(EL)));
    using IM = DinFusedImg;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int T = A.T;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    float* ca_s = smem;                                   // [64][AS]  w2 (1 + alpha) / 2
    float* cb_s = smem + ROWS * AS;                       // [64][AS]  w2 (1 - alpha) / 2
    float* img_s = smem + DF_COEF_FLOATS;                 // TAIL: the tail's weights
    constexpr int img_floats = TAIL ? IM::dma_floats : 0;
    int* ids_s = reinterpret_cast<int*>(smem + DF_COEF_FLOATS + img_floats) + wave * 16 * A.idp;
    // TAIL: the quarter sums stay in registers (the LDS they would park in holds the UNF fragments) and only the cross-wave combine
    // of ts > 1 goes through ONE slot per wave; attention only: two parking slots per wave in LDS, S0, S1
    constexpr bool PREG = TAIL;
    constexpr int PSL = PREG ? 1 : 2;
    constexpr bool AWL = TAIL && !MB;                     // the attention's A fragments through LDS (below)
    float* park0 = smem + DF_COEF_FLOATS + img_floats + DF_WAVES * 16 * A.idp;
    float* park_s = smem + DF_COEF_FLOATS + img_floats + DF_WAVES * 16 * A.idp + wave * PSL * 64 * EL;

    unsigned long long ts_entry = 0;
    if constexpr ((XP & 1024) != 0) ts_entry = __builtin_amdgcn_s_memrealtime();
    // ---- this wave's (task, time slice) ----
    const int ntpb = (B + 15) >> 4;
    int nb_batches = 1;
    if constexpr (MB) nb_batches = Mm.n;
    const int ntasks = nb_batches * ntpb;
    const int gw = blockIdx.x * DF_WAVES + wave;
    // MB: a PERSISTENT launch -- the tables and the tail's image are staged once, then every wave walks tasks gw, gw + (waves of the
    // grid), ... on its own: no workgroup-wide barrier per task, so the waves of a CU drift apart and one wave's round trips and
    // epilogue (matrix pipe, LDS) run under the other waves' slot loops (the fabric).  One wave per task (ts = 1).
    if constexpr (MB) {
#pragma unroll 1
        for (int c = wave; c < DF_COEF_FLOATS / 256; c += DF_WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.coef + c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
        if constexpr (TAIL) {
#pragma unroll 1
            for (int c = wave; c < IM::dma_floats / 256; c += DF_WAVES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.image + c * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void*)(img_s + c * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // (persistent: the attention's A fragments are the same for every task -- requested once, 8 KB per wave instead of per task)
    f32x4 aWr[2][4];
    if constexpr (MB) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int k = 0; k < 4; ++k) aWr[nb][k] = ld4(A.frag + ((nb * 4 + k) * 64 + lane) * 4);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(aWr[nb][k]));      // (landed before the first hidden load is requested)
    }
    const int task_stride = MB ? (int)gridDim.x * DF_WAVES : 0;
#pragma unroll 1
    for (int task = MB ? gw : (gw >> A.ts_log2); MB ? task < ntasks : true; task += task_stride) {
    const int slice = MB ? 0 : (gw & (A.ts - 1));
    const bool work = task < ntasks;                      // wave-uniform
    int bi = 0, tl = task;
    if constexpr (MB) { bi = __builtin_amdgcn_readfirstlane(task / ntpb); tl = task - bi * ntpb; }
    const int* ids_b = ids;
    const float* dense_b = dense;
    float* out_b = out;
    if constexpr (MB) { ids_b = Mm.ids[work ? bi : 0]; dense_b = Mm.dense[work ? bi : 0]; out_b = Mm.out[work ? bi : 0]; }
    const int nq = MB ? 4 : (4 >> A.ts_log2);             // quarters of this wave; a quarter is A.ql slots (a multiple of FOUR: see the slot loop)
    const int Tq = nq * A.ql;
    const int t0 = slice * Tq;
    const int nsteps = work ? max(0, min(T, t0 + Tq) - t0) : 0;
    const int m = tl * 16 + r;
    const int mc = min(m, B - 1);
    const bool tail_wave = TAIL && work && slice == 0;    // wave-uniform
    auto stamp = [&](int k) {
#ifdef SPRK_DF_XP
        if constexpr ((XP & 1024) != 0) {
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && gw < DF_TS_WAVES) { g_df_ts[gw * 8 + k] = t; if (k == 1) g_df_ts[gw * 8] = ts_entry; }
        }
#endif
    };

    // ================= prologue: TWO round trips (ids; then every row and table the first trip needs), the tail's image behind them =========
    // Everything of the second round trip is a HIDDEN load (asm, like the slot loop's): hipcc's own waits sit in front of the first
    // use, and a use between two requests is a round trip of its own -- round 4's first form had FOUR at the head of every task
    // (ids 16 bytes at a time with a wait each, the candidate's rows, the first slots' rows, and the 88 KB image in front of it all).
    // TAIL: this sample's numerics, K = 8 as two steps (k = q and q + 4); requested first, used in the epilogue
    // (hidden loads like everything else in front of the slot loop, and unconditional: a compiler-visible load left pending in
    // hipcc's books made it put a vmcnt(0) in front of the first instruction that so much as ENCODES its register -- inside the slot
    // loop, v_pk_fma_f32 v[108:109], v[142:143], ... with op_sel_hi = 0 reads v142 only, and v143 was xnb --, and "using" them behind the
    // image's pieces made it wait for all of those)
    float xna = 0.f, xnb = 0.f;
    if constexpr (TAIL) {
        const int last = max(A.n_num - 1, 0);
        const float* nrow = A.ND > 0 ? dense_b + (size_t)mc * A.ND : reinterpret_cast<const float*>(ids_b);   // (no numerics: any valid word)
        const float* pa = nrow + (A.ND > 0 ? min(q, last) : 0);
        const float* pb = nrow + (A.ND > 0 ? min(q + 4, last) : 0);
        asm volatile("global_load_dword %0, %1, off" : "=&v"(xna) : "v"(pa));
        asm volatile("global_load_dword %0, %1, off" : "=&v"(xnb) : "v"(pb));
    }
    // the task's ids block (16 consecutive rows of F ints: contiguous) -> LDS by LDS-DMA, 1-KB pieces all in flight together
    if (work) {
        const int nint = 16 * A.F;
        if (tl * 16 + 16 <= B && !((uintptr_t)ids_b & 15) && A.idp == A.F) {
            const int* src = ids_b + (size_t)tl * nint;
#pragma unroll 1
            for (int c = 0; c * 256 < nint; ++c)
                if (c * 256 + lane * 4 < nint)             // (16 F ints: a multiple of four)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void*)(ids_s + c * 256), 16, 0, 0);
        } else {
            for (int i = lane; i < nint; i += 64) {
                const int s = i / A.F, col = i - s * A.F;
                ids_s[s * A.idp + col] = ids_b[(size_t)min(tl * 16 + s, B - 1) * A.F + col];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ids are in LDS (one wave: its own LDS operations complete in issue order)
    stamp(3);
    // ---- round trip two: the candidate's row (for h * c) and vc row (the accumulators' start), the A fragments (the same eight for
    // every slot), TAIL: the UNF columns' raw split rows (128 bytes per id: lane (r,q) takes hi / lo halfs 8q .. 8q+7 of sample r's
    // row, the B operand of fc0's blocks for that column -- sixteen registers through the slot loop, but random rows that miss every
    // cache and an epilogue with nothing to hide their round trip behind), the first FOUR slots' rows ----
    unsigned maxid = 0;                                   // the largest id seen (as unsigned: a negative id is huge): ONE range check at the end
    df_h2 cch[NP], ccl[NP];                               // c' = c sH kappa of this lane's k = EL q + e, split into halfs, as pairs
    f32x4 acc_init[2], cp[KC];
    constexpr bool UNFK = TAIL;                           // ([r5] KC = 1 too: the two column-pair blocks of emb_dim <= 16)
    f32x4 er[UNFK ? 4 : 1];
    bool tbad = false;
    int tid_g[DT_MAX_COLS];                               // the tail's embedding columns' ids
#pragma unroll
    for (int g = 0; g < DT_MAX_COLS; ++g) tid_g[g] = -1;
    auto hload = [&](f32x4& dst, unsigned voff, const void* base) { asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(base)); };
    {
        const unsigned cid = work ? (unsigned)ids_s[r * A.idp + A.cand_col] : 0u;
        maxid = cid;
        const unsigned csafe = min(cid, (unsigned)A.vocab - 1u);
#pragma unroll
        for (int c = 0; c < KC; ++c) hload(cp[c], (csafe * (unsigned)KP + EL * q + 4 * c) * 4u, A.tsplit);   // (< 4 GiB: checked at finalize)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) hload(acc_init[nb], (csafe * (unsigned)HP + nb * 16 + 4 * q) * 4u, A.vc);
        if constexpr (AWL) {
            // (the eight A fragments are the same 8 KB for every wave: ONE copy per workgroup through LDS -- wave w stages fragment w
            // in the parking region, which nothing else touches before the workgroup's second meeting -- instead of 64 KB per CU
            // through the texture path in front of the first rows)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.frag + wave * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(park0 + wave * 256), 16, 0, 0);
        } else if constexpr (!MB) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int k = 0; k < 4; ++k) hload(aWr[nb][k], (unsigned)(((nb * 4 + k) * 64 + lane) * 16), A.frag);
        }
    }
    if constexpr (TAIL) {
        if (tail_wave) {
#pragma unroll
            for (int g = 0; g < DT_MAX_COLS; ++g) if (g < A.n_cols) tid_g[g] = ids_s[r * A.idp + A.col[g]];
        }
    }
    auto unf_load = [&]() {
        if constexpr (UNFK && KC == 1) {
            // emb_dim <= 16: block pb = columns 2 pb and 2 pb + 1 of the tail's list, this lane's is 2 pb + (q >> 1), its 16 + 16 bytes are
            // hi / lo halfs 8 (q & 1) .. + 7 of sample r's row.  Two tables per block, so the address is a 64-bit register pair.
            const bool raw = tail_wave && A.n_unf > 0;       // (wave-uniform)
            const bool up = (q >> 1) != 0;
            // (every field of the kernel argument is made a SCALAR before a lane chooses between two of them: left as `up ? A.x[g1] : A.x[g0]`
            //  hipcc turned the select of two kernarg loads into ONE global_load from the kernarg segment at a lane-dependent address,
            //  followed by a vmcnt(0) -- four serial round trips in the prologue, each draining the image's DMA: build/sparrow.s, round 5)
            auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
            auto scp = [](const _Float16* p) {
                const unsigned long long u = (unsigned long long)p;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
                return reinterpret_cast<const _Float16*>(((unsigned long long)hi << 32) | lo);
            };
            const int n_uc = sc(A.n_ucols), voc_z = sc(A.uvocab[0]);
            const _Float16* tab_z = scp(A.Etab[0]);
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                const int g0 = 2 * pb, g1 = 2 * pb + 1;      // (static indices into the kernel argument's arrays)
                const int col0 = sc(A.ucol[g0]), col1 = sc(A.ucol[g1]), voc0 = sc(A.uvocab[g0]), voc1 = sc(A.uvocab[g1]);
                const _Float16 *tab0 = scp(A.Etab[g0]), *tab1 = scp(A.Etab[g1]);
                const bool present = (up ? g1 : g0) < n_uc;
                const int id = raw ? ids_s[r * A.idp + (up ? col1 : col0)] : -1;
                const int voc = present ? (up ? voc1 : voc0) : voc_z;
                const _Float16* tab = present ? (up ? tab1 : tab0) : tab_z;      // an absent column: column 0's all-zero row
                const bool ok = raw && present && (unsigned)id < (unsigned)voc;
                tbad |= raw && present && !ok && id != -1;
                const char* row = raw ? reinterpret_cast<const char*>(tab) + (size_t)(ok ? id : voc) * 64 + 16 * (q & 1)
                                      : reinterpret_cast<const char*>(A.tsplit) + 16 * (q & 1);   // (no raw rows: valid bytes nobody looks at)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(er[2 * pb]) : "v"(row));
                asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=&v"(er[2 * pb + 1]) : "v"(row));
            }
        } else if constexpr (UNFK) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // (unconditional: a wave or a column without raw rows reads the first bytes of the history table, and nobody looks)
                const bool have = tail_wave && u < A.n_unf;   // (wave-uniform)
                const int id = have ? ids_s[r * A.idp + A.ucol[u]] : -1;
                const int voc = A.uvocab[u];
                const bool ok = (unsigned)id < (unsigned)voc;
                tbad |= have && !ok && id != -1;
                const void* eb = have ? (const void*)A.Etab[u] : (const void*)A.tsplit;
                const unsigned voff = have ? (unsigned)(ok ? id : voc) * 128u + 16u * q : 16u * q;   // (index vocab: the all-zero row; < 4 GiB: setup)
                hload(er[2 * u], voff, eb);
                hload(er[2 * u + 1], voff + 64u, eb);
            }
        }
    };
    // (one batch per launch: the raw rows are wanted in the epilogue only -- they go out BEHIND the image's pieces, so that the second
    // round trip in front of the first slot is the candidate's rows, the fragments' piece and the first row sets and nothing else)
    constexpr int NU = (UNFK && !MB) ? 4 : 0;
    if constexpr (UNFK && MB) unf_load();
    f32x4 rowA[KC], rowB[KC], rowC[KC], rowD[KC];
    const char* tbase = reinterpret_cast<const char*>(A.tsplit);
    const unsigned qoff = (unsigned)(EL * 4) * (unsigned)q, vmax = (unsigned)A.vocab - 1u;
    auto load = [&](int step, f32x4 (&rw)[KC]) {           // rows of slot t0 + step (clamped: steps past the end re-read the last slot)
        const int tt = nsteps > 0 ? t0 + min(step, nsteps - 1) : 0;                 // (a wave without slots: slot 0, a valid id, never used)
        const unsigned id = (unsigned)ids_s[r * A.idp + A.hist_col + tt];
        maxid = max(maxid, id);
        const unsigned voff = (min(id, vmax) << (KC == 2 ? 7 : 6)) | qoff;          // row bytes KP * 4; < 4 GiB (checked at finalize)
        if constexpr (KC == 2)
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
                         : "=&v"(rw[0]), "=&v"(rw[1]) : "v"(voff), "s"(tbase));
        else
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(rw[0]) : "v"(voff), "s"(tbase));
    };
    // [r5] LATE: round 4's prologue ended with "the second round trip moves 27 KB per wave at the moment every wave of the chip does the
    // same" (3.9 us in the stamped timeline) -- 11 KB of it this wave's share of the tail's image, wanted 25 us later.  Now the image's
    // pieces and the raw tail rows go out AFTER the first trip and land under the second; the coefficient pieces go out in FRONT of the first
    // row sets so that one vmcnt says "everything but the rows".
    constexpr bool LATE = TAIL && !MB && (DF_LATE_IMAGE != 0);
    constexpr int NCW = (DF_COEF_FLOATS / 256 + DF_WAVES - 1) / DF_WAVES;
    auto coef_dma = [&]() {
#pragma unroll
        for (int i = 0; i < NCW; ++i) {
            const int c = min(wave + DF_WAVES * i, DF_COEF_FLOATS / 256 - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.coef + c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
        }
    };
    if constexpr (LATE) coef_dma();
    load(0, rowA); load(1, rowB); load(2, rowC); load(3, rowD);   // (unconditional: no path on which these registers are anything else)
    // ---- LDS-DMA, 1-KB pieces, a COMPILE-TIME number per wave (the waits count them): the attention's coefficient tables (wave w
    // takes w, w + 8, w + 16; a piece past the end repeats the last), then TAIL: the tail's image (weights as fragments, 88 KB).
    // Nothing reads the image before the epilogue; requested first (round 4's first form) it sat in front of the ids and the first
    // rows in every queue: 6.4 us from kernel entry to the first slot against 3.4 us without a tail (profiles/r04, the stamped timeline)
    constexpr int NPW = (TAIL && !MB) ? IM::pieces_per_wave : 0;
    if constexpr (!MB && !LATE) coef_dma();
    auto image_dma = [&]() {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int c = wave + DF_WAVES * i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.image + c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(img_s + c * 256), 16, 0, 0);
        }
        unf_load();
    };
    if constexpr (TAIL && !MB && !LATE) image_dma();
    constexpr int NW0 = LATE ? 4 * KC : NPW + NU;             // younger operations at the wait below: the four row sets, or the image's pieces + raw rows
    // everything but the image pieces (and the raw rows behind them) has landed: this wave's coefficient pieces, the second round trip,
    // the first four row sets
    // (volatile statements keep their order: the empty ones tie the remaining registers to the wait in front of them)
    if constexpr (MB || AWL)
        asm volatile("s_waitcnt vmcnt(%3)" : "+v"(cp[0]), "+v"(acc_init[0]), "+v"(acc_init[1]) : "n"(NW0));
    else
        asm volatile("s_waitcnt vmcnt(%11)" : "+v"(cp[0]), "+v"(acc_init[0]), "+v"(acc_init[1]), "+v"(aWr[0][0]), "+v"(aWr[0][1]), "+v"(aWr[0][2]),
                     "+v"(aWr[0][3]), "+v"(aWr[1][0]), "+v"(aWr[1][1]), "+v"(aWr[1][2]), "+v"(aWr[1][3]) : "n"(NW0));
    if constexpr (KC == 2) asm volatile("" : "+v"(cp[KC - 1]));
    if constexpr (TAIL) asm volatile("" : "+v"(xna), "+v"(xnb));          // (tied to the wait above: the oldest loads of the task)
    if constexpr (UNFK && MB) asm volatile("" : "+v"(er[0]), "+v"(er[UNFK ? 1 : 0]), "+v"(er[UNFK ? 2 : 0]), "+v"(er[UNFK ? 3 : 0]));
    stamp(7);
    if constexpr (AWL) {
        __builtin_amdgcn_s_barrier();                     // coefficient tables and the A fragments staged by every wave
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int k = 0; k < 4; ++k) aWr[nb][k] = ld4(park0 + ((nb * 4 + k) * 64 + lane) * 4);
    }
    f16xe aW[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (KC == 2) aW[nb][k] = __builtin_bit_cast(f16xe, aWr[nb][k]);
            else { const din_f16x8 both = __builtin_bit_cast(din_f16x8, aWr[nb][k]); aW[nb][k] = f16xe{both[0], both[1], both[2], both[3]}; }
        }
    din_f16x8 eh[UNFK ? 2 : 1], el[UNFK ? 2 : 1];
    if constexpr (UNFK && MB) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { eh[u] = __builtin_bit_cast(din_f16x8, er[2 * u]); el[u] = __builtin_bit_cast(din_f16x8, er[2 * u + 1]); }
    }
    {
        f16xe chi, clo;
        unpack_halfs<KC>(cp, chi, clo);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            // kappa is a power of two: both scalings are exact (up to f16's subnormal spacing, 2^-24 absolute against products of 2^14)
            cch[j] = df_h2{(_Float16)((float)chi[2 * j] * A.kappa), (_Float16)((float)chi[2 * j + 1] * A.kappa)};
            ccl[j] = df_h2{(_Float16)((float)clo[2 * j] * A.kappa), (_Float16)((float)clo[2 * j + 1] * A.kappa)};
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc_init[nb] = acc_init[nb] * A.acc_scale;
    }
    f32x4 z0[TAIL ? N0C : 1];
    // fc0_front: z0 = numerics (f32 MFMA; fc0's bias in the free numeric slot) + the raw-row columns (split f16 MFMA) + the folded columns'
    // rows -- everything of the tail's first layer that does NOT need the pooled vector; the pooled history follows in the epilogue.
    // [r5] Measured and NOT kept (profiles/r05/experiments/r05_08): this part computed INSIDE the slot loop, wave w in front of trip 2 + w
    // (folded rows as hidden loads behind a manual vmcnt(0)), on the theory that the loop is fabric-bound and a wave's own chain has slack.
    // The stamped timeline: fc0 in the epilogue 3.6 -> 0.9 us, but the slot loop 24.2 -> 27.0 us -- the drained prefetch ring and the
    // 64 MFMAs are NOT hidden by the other seven waves; 38.7 against 37.9 us per launch.
    bool tb2_front = false;                                   // (a folded column's id outside its table)
    auto fc0_front = [&]() {
        if constexpr (TAIL) {
        // (two columns = 16 loads in flight; with raw-row columns the folded ones ARE two -- DIN.py's genre columns)
        f32x4 f[2][N0C];
        auto fold_load = [&](int g0) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                // folded rows straight into the accumulators' layout (lane (r,q): outputs 16 nb + 4q .. + 3 of sample r)
                const int id = g0 == 0 ? tid_g[g] : tid_g[2 + g];
                const int voc = g0 == 0 ? A.tvocab[g] : A.tvocab[2 + g];
                const float* tab = g0 == 0 ? A.Ftab[g] : A.Ftab[2 + g];
                const bool ok = g0 + g < A.n_cols && (unsigned)id < (unsigned)voc;
                tb2_front |= g0 + g < A.n_cols && !ok && id != -1;
                const float* frow = tab + (size_t)(ok ? id : 0) * IM::N0 + 4 * q;
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) f[g][nb] = (ok && !(XP & 128)) ? ld4(frow + nb * 16) : zero;
            }
        };
        fold_load(0);                                          // (requested first, under the matrix work below)
        // the numeric chunk: A = W0^T[n][numeric q + 4 s] (one scalar LDS read per block and step), two f32 MFMAs per 16 outputs;
        // fc0's bias rides in numeric slot b0_slot against a constant 1 (else it is added here)
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = zero;
        if constexpr (!(XP & 512)) {
            // slot n_num carries 1.0 against fc0's bias (b0_slot); slots beyond duplicate a finite value that only ever meets zero weights
            const float xa1 = q == A.b0_slot ? 1.0f : xna;
            const float xb1 = q + 4 == A.b0_slot ? 1.0f : xnb;
            const float* wn = img_s + IM::off_wn + lane;
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb) z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[(nb * 2 + 0) * 64], xa1, z0[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb) z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[(nb * 2 + 1) * 64], xb1, z0[nb], 0, 0, 0);
            if (A.b0_slot < 0) {
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) z0[nb] += ld4(img_s + IM::off_b0 + nb * 16 + 4 * q);
            }
        }
        {   // (KC = 2: up to two large-vocabulary columns; KC = 1: the two column-pair blocks -- the same fragments' layout, the same chain)
            const float* wf = img_s + IM::total_pad + lane * 4;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u < A.n_unf) {                                // (wave-uniform)
#pragma unroll
                    for (int nb = 0; nb < N0C; ++nb) {
                        const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(wf + (u * N0C + nb) * 512));
                        const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(wf + (u * N0C + nb) * 512 + 256));
                        f32x4 acc = mfma_f16(al, eh[u], zero);
                        acc = mfma_f16(ah, el[u], acc);
                        acc = mfma_f16(ah, eh[u], acc);
                        z0[nb] += acc * A.e_unscale;
                    }
                }
            }
        }
        {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) z0[nb] += f[g][nb];
            if (A.n_cols > 2) {                                   // (wave-uniform) every column folded: the second pair, a round trip of its own
                fold_load(2);
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int nb = 0; nb < N0C; ++nb) z0[nb] += f[g][nb];
            }
        }
        }
    };
    if constexpr (!MB && !AWL) __builtin_amdgcn_s_barrier();      // coefficient tables staged by every wave
    stamp(1);

    // ---- slot loop ----
    float pacc[EL];
#pragma unroll
    for (int e = 0; e < EL; ++e) pacc[e] = 0.f;
    const float one = 1.0f;
    // park(j) retires local quarter j < nq - 1 (wave-uniform j) into the wave's two LDS slots (three times per task: registers are the
    // scarcer resource of this kernel):
    //   nq = 2: S0 = q0          nq = 4: S0 = q0; S1 = S0 + q1; S0 = q2          -- the wave's LAST quarter stays in pacc, and
    //   finish: nq = 1: q        nq = 2: S0 + q1                 nq = 4: S1 + (S0 + q3)
    float* S0 = park_s;
    float* S1 = park_s + 64 * EL;
    float R0[PREG ? EL : 1], R1[PREG ? EL : 1];
    if constexpr (PREG) {
#pragma unroll
        for (int e = 0; e < EL; ++e) { R0[e] = 0.f; R1[e] = 0.f; }
    }
    auto park = [&](int j) {
        if constexpr (PREG) {
            // (selects, not two branches: hipcc merged the branches' stores into ONE store through a scratch slot with a dynamic address)
            const bool odd = (j & 1) != 0;
#pragma unroll
            for (int e = 0; e < EL; ++e) {
                const float sm = R0[e] + pacc[e];
                R1[e] = odd ? sm : R1[e];
                R0[e] = odd ? R0[e] : pacc[e];
                pacc[e] = 0.f;
            }
        } else {
        if ((j & 1) == 0) {
#pragma unroll
            for (int e = 0; e < EL; ++e) S0[e * 64 + lane] = pacc[e];
        } else {
#pragma unroll
            for (int e = 0; e < EL; ++e) S1[e * 64 + lane] = S0[e * 64 + lane] + pacc[e];
        }
#pragma unroll
        for (int e = 0; e < EL; ++e) pacc[e] = 0.f;
        }
    };
    // ---- a pair of slots, first half: u = W12 h + W4 (h * c) + vc[cand] of both slots on the matrix pipe; h back to f32 ----
    // PReLU(alpha[t][n]) -> Dense(1) (DIN.py:150-151) as ca u + cb |u| summed over this lane's eight units; the coefficient rows are
    // wave-uniform LDS addresses (broadcast)
    auto prelu_dot = [&](int t, const f32x4 (&uu)[2]) -> float {
        if constexpr (XP & 8) return uu[0][0] + uu[1][1];
        const int tc = min(t, ROWS - 1);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x4 ca = ld4(ca_s + tc * AS + nb * 16 + 4 * q);
            const f32x4 cb = ld4(cb_s + tc * AS + nb * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sa = fmaf(ca[j], uu[nb][j], sa);
                sb = fmaf(cb[j], __builtin_fabsf(uu[nb][j]), sb);
            }
        }
        return sa + sb;
    };
    // (slot a's partial logit is formed here, under slot b's MFMAs; slot b's units travel to the second half)
    auto mfma_part = [&](int step, const f32x4 (&ra)[KC], const f32x4 (&rb)[KC], float& lga, f32x4 (&ub)[2], float (&h32)[2][EL]) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const f32x4 (&rw)[KC] = x ? rb : ra;
            f16xe bh, bl;
            unpack_halfs<KC>(rw, bh, bl);
            f32x4 acc[2] = {acc_init[0], acc_init[1]};
            // W12 . (hi + lo): independent of the product below -- the matrix pipe works while the VALU forms h * c
            if constexpr (!(XP & 1)) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][0], bh, acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][0], bl, acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][1], bh, acc[nb]);
            }
            // h * c * sP, split: packed f16 arithmetic on the halfs (header)
            f16xe qh = bh, ql = bl;
            if constexpr (!(XP & 2)) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const df_h2 hh = {bh[2 * j], bh[2 * j + 1]}, hl = {bl[2 * j], bl[2 * j + 1]};
                const df_h2 ph = hh * cch[j];
                df_h2 pl = __builtin_elementwise_fma(hh, cch[j], -ph);
                pl = __builtin_elementwise_fma(hh, ccl[j], pl);
                pl = __builtin_elementwise_fma(hl, cch[j], pl);
                qh[2 * j] = ph[0]; qh[2 * j + 1] = ph[1];
                ql[2 * j] = pl[0]; ql[2 * j + 1] = pl[1];
            }
            }
            if constexpr (!(XP & 1)) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][2], qh, acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][2], ql, acc[nb]);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][3], qh, acc[nb]);
            } else {
                acc[0][0] += (float)qh[0] + (float)ql[1];
                acc[1][1] += (float)bh[0] + (float)bl[1];
            }
            if (x == 0) lga = prelu_dot(t0 + step, acc);
            else { ub[0] = acc[0]; ub[1] = acc[1]; }
            // h back to f32 (sH units): one v_fma_mix_f32 per element (inputs: loaded registers; outputs go to compiler-visible VALU)
#pragma unroll
            for (int e = 0; e < EL; ++e) {
                const float hp = KC == 2 ? rw[0][e >> 1] : rw[0][e >> 1];
                const float lp = KC == 2 ? rw[KC - 1][e >> 1] : rw[0][2 + (e >> 1)];
                if constexpr (XP & 4) h32[x][e] = hp;
                else h32[x][e] = (e & 1) ? halfs_sum<true>(one, hp, lp) : halfs_sum<false>(one, hp, lp);
            }
        }
    };
    // ---- second half: slot b's partial logit, the pair's two partial logits reduced over the four 16-lane rows by a transposing
    // swap ((row0 + row1) + (row2 + row3), rows4_sum's order), ONE sigmoid, one swap to hand both weights to every row, weighted
    // sum pooling (DIN.py:152-158): this lane owns pooled[r][EL*q + e] ----
    auto finish = [&](int step, float lga, const f32x4 (&ub)[2], const float (&h32)[2][EL]) {
        float lg[2];
        lg[0] = lga;
        lg[1] = prelu_dot(t0 + step + 1, ub);
        float wa, wb;
        if constexpr (XP & 16) { wa = lg[0]; wb = lg[1]; } else {
        float xx = lg[0], yy = lg[1];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(xx), "+v"(yy));   // xx = [a0 b0 a2 b2], yy = [a1 b1 a3 b3]
        xx += yy;                                                                     // [a01 b01 a23 b23]
        yy = xx;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xx), "+v"(yy));   // xx = [a01 b01 a01 b01], yy = [a23 b23 a23 b23]
        const float z = xx + yy;                                                      // rows 0, 2: slot a; rows 1, 3: slot b
        wa = sigmoidf_fast(z * A.unscale + A.b2);                                     // PReLU is positively homogeneous: unscale the logit
        wb = wa;
        // (s_nop 1 also covers the one wait state a transcendental's result needs before a non-transcendental VALU reads it)
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(wa), "+v"(wb));   // wa = [a a a a], wb = [b b b b]
        }
        wa = step < nsteps ? wa : 0.f;                                                // (a pair past the end: the pipeline's last, unused, round)
        wb = step + 1 < nsteps ? wb : 0.f;                                            // the padding slot of an odd history
        if constexpr (ATT) {
            if (att && q == 0 && m < B) {
                if (step < nsteps) att[(size_t)m * T + t0 + step] = wa;
                if (step + 1 < nsteps) att[(size_t)m * T + t0 + step + 1] = wb;
            }
        }
        if constexpr (XP & 32) {
            pacc[0] += wa + wb;
#pragma unroll
            for (int e = 0; e < EL; ++e) pacc[1] += h32[0][e] + h32[1][e];
        } else {
#pragma unroll
        for (int e = 0; e < EL; ++e) pacc[e] = fmaf(wa, h32[0][e], pacc[e]);
#pragma unroll
        for (int e = 0; e < EL; ++e) pacc[e] = fmaf(wb, h32[1][e], pacc[e]);
        }
    };
    {
        // at most N younger SETS outstanding => this set has landed (vmcnt retires in order; a set is KC loads).  No "memory" clobber:
        // the statements are volatile (kept in order among themselves) and tied to their registers; nothing else in the loop loads from
        // global memory, and stores only add YOUNGER operations -- so the compiler may move LDS reads and arithmetic across them
        // (X: younger operations that are NOT row sets -- the image pieces during the first trip)
        auto wait3 = [&](f32x4 (&rw)[KC], auto x) {
            constexpr int X = decltype(x)::value;
            if constexpr (KC == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rw[0]), "+v"(rw[1]) : "n"(6 + X));
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rw[0]) : "n"(3 + X));
        };
        auto wait2 = [&](f32x4 (&rw)[KC], auto x) {
            constexpr int X = decltype(x)::value;
            if constexpr (KC == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rw[0]), "+v"(rw[1]) : "n"(4 + X));
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rw[0]) : "n"(2 + X));
        };
        // One trip = two pairs = four slots, straight-line: the four register sets of the ring keep their roles statically (an if / else
        // on a round's parity made hipcc merge the two arms' tails and COPY row registers with loads still in flight; a loop per quarter
        // inside a loop over the quarters made it copy them at the inner loop's entry -- scripts/r04/check_din_fused_isa.py walks the ISA
        // of every instantiation for exactly that).  A quarter is a multiple of four slots, so a trip never straddles one; a history
        // that ends inside a trip leaves slots whose weights are forced to zero.  Rows are requested two pairs ahead, as soon as the
        // matrix half has consumed the set.  (A software-pipelined form -- the matrix half of pair i + 1 in one block with the VALU half
        // of pair i -- measured 0.3 us SLOWER on the same box, 42.5 against 42.2 us per fused launch: profiles/r04)
        f32x4 u0[2];
        float l0 = 0.f;
        float h0[2][EL];
        int jq = 0, qnext = A.ql;                             // current local quarter, the first slot of the next one (wave-uniform)
        auto trip = [&](int step, auto x) {
            if constexpr (!(XP & 64)) { wait3(rowA, x); wait2(rowB, x); }
            mfma_part(step, rowA, rowB, l0, u0, h0);
            if constexpr (!(XP & 64)) { load(step + 4, rowA); load(step + 5, rowB); }
            finish(step, l0, u0, h0);
            if constexpr (!(XP & 64)) { wait3(rowC, x); wait2(rowD, x); }
            mfma_part(step + 2, rowC, rowD, l0, u0, h0);
            if constexpr (!(XP & 64)) { load(step + 6, rowC); load(step + 7, rowD); }
            finish(step + 2, l0, u0, h0);
        };
        int step = 0;
        if constexpr (TAIL && !MB) {
            // the first trip runs under the image's DMA (its pieces are younger than the four row sets it waits for), by EVERY wave
            // (one without slots works on zero weights), then the wave's pieces have landed -- at most the four re-requested row
            // sets stay in flight -- and the workgroup meets once more: from here on the image is readable by all of it
            if constexpr (LATE) {
                trip(0, DfInt<0>{});
                image_dma();                                      // (younger than the four row sets trip 0 re-requested)
                if (4 < nsteps && 4 == qnext) { park(jq); ++jq; qnext += A.ql; }
                trip(4, DfInt<NPW + NU>{});                       // (a wave with at most four slots: zero weights)
            } else {
                trip(0, DfInt<NPW + NU>{});
            }
            if constexpr (UNFK)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(er[0]), "+v"(er[UNFK ? 1 : 0]), "+v"(er[UNFK ? 2 : 0]), "+v"(er[UNFK ? 3 : 0]) : "n"(4 * KC));
            else
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * KC));
#pragma unroll
            for (int u = 0; u < (UNFK ? 2 : 0); ++u) { eh[u] = __builtin_bit_cast(din_f16x8, er[2 * u]); el[u] = __builtin_bit_cast(din_f16x8, er[2 * u + 1]); }
            __builtin_amdgcn_s_barrier();
            step = LATE ? 8 : 4;
        }
        for (; step < nsteps; step += 4) {
            if (step == qnext) { park(jq); ++jq; qnext += A.ql; }
            trip(step, DfInt<0>{});
        }
        for (; jq < nq - 1; ++jq) park(jq);                    // quarters without slots hold exact zeros: retired the same way
        // nothing may still be in flight towards these registers when they are reused
        if constexpr (KC == 2)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(rowA[0]), "+v"(rowA[1]), "+v"(rowB[0]), "+v"(rowB[1]), "+v"(rowC[0]), "+v"(rowC[1]),
                         "+v"(rowD[0]), "+v"(rowD[1]));
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(rowA[0]), "+v"(rowB[0]), "+v"(rowC[0]), "+v"(rowD[0]));
    }
    stamp(2);
    float res[EL];
    if (nq == 1) {
#pragma unroll
        for (int e = 0; e < EL; ++e) res[e] = pacc[e];
    } else if (nq == 2) {
#pragma unroll
        for (int e = 0; e < EL; ++e) res[e] = (PREG ? R0[PREG ? e : 0] : S0[e * 64 + lane]) + pacc[e];
    } else {
#pragma unroll
        for (int e = 0; e < EL; ++e) res[e] = (PREG ? R1[PREG ? e : 0] : S1[e * 64 + lane]) + ((PREG ? R0[PREG ? e : 0] : S0[e * 64 + lane]) + pacc[e]);
    }
    // ---- several waves per task: their results through LDS, slice 0 sums in the fixed order ----
    if (!MB && A.ts > 1) {                                // (a persistent launch: one wave per task)
        if (slice != 0) {
#pragma unroll
            for (int e = 0; e < EL; ++e) S0[e * 64 + lane] = res[e];           // (a wave's S0 is free by now)
        }
        __syncthreads();
        if (slice == 0) {
            const float* P = park_s + PSL * 64 * EL;         // the next wave's S0
            if (A.ts == 2) {
#pragma unroll
                for (int e = 0; e < EL; ++e) res[e] = res[e] + P[e * 64 + lane];
            } else {
#pragma unroll
                for (int e = 0; e < EL; ++e)
                    res[e] = (res[e] + P[e * 64 + lane]) + (P[PSL * 64 * EL + e * 64 + lane] + P[2 * PSL * 64 * EL + e * 64 + lane]);
            }
        }
    }
    const bool bad = (work && maxid >= (unsigned)A.vocab) || tbad;
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
    if (!(work && slice == 0)) { if constexpr (MB) continue; else return; }

    if constexpr (!TAIL) {
        if (m < B) {
            float* prow = out_b + (size_t)m * A.Dp + EL * q;
            if (A.Dp == KP) {                                     // (wave-uniform) full-width rows: this lane's EL floats as 16-byte stores
#pragma unroll
                for (int c = 0; c < KC; ++c)
                    st4(prow + 4 * c, f32x4{res[4 * c] * A.inv_h_scale, res[4 * c + 1] * A.inv_h_scale, res[4 * c + 2] * A.inv_h_scale, res[4 * c + 3] * A.inv_h_scale});
            } else {
#pragma unroll
                for (int e = 0; e < EL; ++e)
                    if (EL * q + e < A.Dp) prow[e] = res[e] * A.inv_h_scale;
            }
        }
        if constexpr (MB) continue; else return;
    } else {
        // ================= the tail (DIN.py:161-167) for this wave's sixteen samples: registers and LDS only =================
        // fc0 = [numerics (f32 MFMA) + bias + raw-row columns (split f16 MFMA) + folded rows] (fc0_front) + the pooled history (split f16 MFMA)
        fc0_front();
        bool tb2 = tb2_front;
        // the pooled history on the f16 pipe, from the registers it was accumulated in (k = EL q + e): per-sample dynamic scale
        // (DIN's attention weights are not normalised), hi / lo split, three products per 16 outputs
        if constexpr (XP & 512) { z0[0][0] += res[0]; } else {
            float xp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xp[e] = e < EL ? res[e < EL ? e : 0] * A.inv_h_scale : 0.f;
            float mx = 0.f;
#pragma unroll
            for (int e = 0; e < EL; ++e) mx = fmaxf(mx, __builtin_fabsf(xp[e]));
            mx = rows4_max(mx);
            float scale, inv;
            dyn_scale(mx, A.inv_w0p_scale, scale, inv);
            din_f16x8 bh, bl;
            dyn_split8(f32x4{xp[0], xp[1], xp[2], xp[3]}, f32x4{xp[4], xp[5], xp[6], xp[7]}, scale, bh, bl);
            const float* wf = img_s + IM::off_w0p + lane * 4;
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb) {
                const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(wf + nb * 512));
                const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(wf + nb * 512 + 256));
                f32x4 acc = mfma_f16(al, bh, zero);
                acc = mfma_f16(ah, bl, acc);
                acc = mfma_f16(ah, bh, acc);
                z0[nb] += acc * inv;
            }
        }
        if (__ballot(tb2) != 0 && lane == 0) atomicOr(err, 1);
        if constexpr ((XP & 1024) != 0) { if (z0[0][0] + z0[7][3] == 123.456f) stamp(7); }   // (the stamp below waits for fc0's results)
        stamp(4);
        // PReLU(alpha0) (DIN.py:164)
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) {
            const f32x4 al = ld4(img_s + IM::off_a0 + nb * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float uu = z0[nb][j];
                z0[nb][j] = fmaf(al[j], neg1_fast(uu), relu1_fast(uu));
            }
        }
        // fc1 on split f16 (dyn_split.h's K-block layout: chunks 2b, 2b + 1 of h1 are block b's operand as they sit in the registers)
        f32x4 z1[N1C];
        {
            float mx = 0.f;
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, __builtin_fabsf(z0[nb][j]));
            mx = rows4_max(mx);
            float scale, inv;
            dyn_scale(mx, A.inv_w1_scale, scale, inv);
            f32x4 acc[N1C];
#pragma unroll
            for (int n1 = 0; n1 < N1C; ++n1) acc[n1] = zero;
            const float* wf = img_s + IM::off_w1 + lane * 4;                 // this lane's 16 bytes inside a 1-KB fragment (k_dyn_pack_w: lane order)
#pragma unroll
            for (int b = 0; b < ((XP & 256) ? 0 : N0C / 2); ++b) {
                din_f16x8 bh, bl;
                dyn_split8(z0[2 * b], z0[2 * b + 1], scale, bh, bl);
#pragma unroll
                for (int n1 = 0; n1 < N1C; ++n1) {
                    const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(wf + ((n1 * (N0C / 2) + b) * 2 + 0) * 256));
                    const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(wf + ((n1 * (N0C / 2) + b) * 2 + 1) * 256));
                    acc[n1] = mfma_f16(ah, bh, acc[n1]);
                    acc[n1] = mfma_f16(ah, bl, acc[n1]);
                    acc[n1] = mfma_f16(al, bh, acc[n1]);
                }
            }
#pragma unroll
            for (int n1 = 0; n1 < N1C; ++n1) z1[n1] = acc[n1] * inv + ld4(img_s + IM::off_b1 + n1 * 16 + 4 * q);
        }
        if constexpr ((XP & 1024) != 0) { if (z1[0][0] + z1[3][3] == 123.456f) stamp(7); }
        stamp(5);
        // PReLU(alpha1) (DIN.py:166) -> Dense(1) -> sigmoid (DIN.py:167)
        float z = 0.f;
#pragma unroll
        for (int n1 = 0; n1 < N1C; ++n1) {
            const f32x4 al = ld4(img_s + IM::off_a1 + n1 * 16 + 4 * q);
            const f32x4 hw = ld4(img_s + IM::off_hw + n1 * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float uu = z1[n1][j];
                const float h2 = fmaf(al[j], neg1_fast(uu), relu1_fast(uu));
                z = fmaf(hw[j], h2, z);
            }
        }
        z = rows4_sum(z);
        if (q == 0 && m < B) out_b[m] = sigmoidf_acc(z + A.head_bias);
        stamp(6);
    }
    if constexpr (!MB) break;
    }   // (tasks of this wave)
}

// k_din_attn.h -- DIN activation unit + weighted sum pooling (reference DIN.py:132-158), one WAVE per
// sample.  Included inside sparrow_hip.hip's anonymous namespace.
//
// The reference feeds a = [h-c, h, c, h*c] (4D wide) of every (sample, history slot) through
// Dense(hidden) -> PReLU -> Dense(1, sigmoid).  With the Dense kernel split in its four row blocks
// W = [W1; W2; W3; W4] that first layer is
//
//     u[b,t,:] = (W1+W2)^T h[b,t] + W4^T (h[b,t] * c[b]) + (W3-W1)^T c[b] + bias
//              = A_b h[b,t] + vc[cand_b]      with  A_b = W12 + W4 diag(c_b)   (hidden x D, per sample)
//                                                   vc[id] = (W3-W1)^T E[id] + bias   (a table, per id)
//
// so the per-(b,t) contraction is K = D instead of 4D (a quarter of the FLOPs), the c-only term is one
// extra 4*hidden-byte row gather per SAMPLE from a table built once at sprk_finalize (k_din_prep),
// and the [B,T,4D] activation-unit input never exists.  (fp32 throughout; only the association of the
// sums differs from the reference's einsum, i.e. rounding-level differences.)
//
// HALF: the K = D contraction runs on the f16 matrix pipe with split operands (see k_chain_v2j.h): both
// A_b and the history rows are bounded by quantities known at sprk_finalize (max|E|, max|W12|, max|W4|),
// so each is scaled by a power of two into f16 range and stored as hi + lo halfs (22 significand bits);
// Ahi.Bhi + Ahi.Blo + Alo.Bhi accumulate in f32 (every partial product is exact there).  Per 16 rows that
// is 6 v_mfma_f32_16x16x32_f16 (16 cycles each, VALU keeps issuing) + 16 conversion VALU instead of
// 16 v_mfma_f32_16x16x4_f32 (32 cycles each on the SIMD's vector ALU).  PReLU is positively homogeneous,
// so the power-of-two unscale is applied once, to the attention logit.
//
// Mapping: a wave owns one sample at a time.  Its T history rows are gathered ONCE from the table
// (16-B pieces, whole 128-B rows per 8 lanes) into a wave-private LDS tile; the attention logits run on
// v_mfma_f32_16x16x4_f32 with A = A_b (built in registers from the resident W12 / W4 fragments and the
// candidate row), B = 16 history rows per group read from LDS, C initialised with vc[cand] -- two groups
// in flight = four independent accumulator chains.  PReLU(alpha[t][n]) / Dense(1) / sigmoid finish in
// registers + two cross-lane adds, and the pooled vector sum_t w[t] h[t] is accumulated from the very
// B-operand registers the MFMAs just consumed (lane (r,q) holds 4*KC floats of row t = 16g + r: one FMA
// each with w[t]), then reduced over the 16 r-lanes with DPP adds -- each row is read from memory once
// and from LDS once.  No workgroup barrier after the one-time alpha
// staging: waves run independently, the next sample's ids are prefetched during the current one.

struct DinRun {
    int T, F, hist_col, cand_col, Dp, vocab;
    float b2;
    const float* table;   // [vocab][Dp]
    const float* w12;     // [HC*16][KC*16]  (W1+W2)^T, zero padded
    const float* w4;      // [HC*16][KC*16]  W4^T, zero padded
    const float* vc;      // [vocab][HC*16]  (W3-W1)^T E[id] + bias
    const float* alpha;   // [T][HC*16]
    const float* w2;      // [HC*16]
    // HALF: w12 / w4 are pre-multiplied by a_scale; history rows are multiplied by h_scale when they are split
    float h_scale, acc_scale, unscale;    // 2^sH, 2^(sA+sH), 2^-(sA+sH)
    const float* tsplit;  // HALF: the table pre-split, rows of KP*4 bytes: per q group [hi(EL halfs) | lo(EL halfs)] of E * h_scale
    float inv_h_scale;    // 2^-sH
};

// One-time (finalize) kernel for HALF: E[v][d] * scale -> hi/lo halfs in the lane layout of the f16 MFMA's B operand:
// q group g = d / EL holds [hi(EL) | lo(EL)], EL = KP / 4 elements per lane.
static __global__ __launch_bounds__(256) void k_din_split_table(const float* __restrict__ table, long long vocab, int Dp, int KP,
                                                         float scale, _Float16* __restrict__ out) {
    const int EL = KP / 4;
    const long long total = vocab * KP;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / KP;
        const int d = (int)(i - v * KP);
        const float x = d < Dp ? table[v * Dp + d] * scale : 0.f;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        const int g = d / EL, e = d - g * EL;
        _Float16* row = out + v * (2 * KP);
        row[g * 2 * EL + e] = hi;
        row[g * 2 * EL + EL + e] = lo;
    }
}

// One-time (finalize) kernels.
static __global__ __launch_bounds__(256) void k_din_prep_w(const float* __restrict__ W, int hidden, int Dp, int KP,
                                                    float scale, float* __restrict__ w12, float* __restrict__ w4) {
    // W: [hidden][4*Dp] in [h-c | h | c | h*c] blocks (sprk_din.w_slot)
    const int total = hidden * KP;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int n = i / KP, k = i - n * KP;
        float a = 0.f, b = 0.f;
        if (k < Dp) {
            const float* w = W + (size_t)n * 4 * Dp;
            a = w[k] + w[Dp + k];
            b = w[3 * Dp + k];
        }
        w12[i] = a * scale;
        w4[i] = b * scale;
    }
}
static __global__ __launch_bounds__(256) void k_din_prep_vc(const float* __restrict__ W, const float* __restrict__ bias,
                                                     const float* __restrict__ table, int hidden, int Dp,
                                                     long long vocab, float* __restrict__ vc) {
    // vc[v][n] = bias[n] + sum_k (W3[n][k] - W1[n][k]) * E[v][k]
    const long long total = vocab * hidden;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / hidden;
        const int n = (int)(i - v * hidden);
        const float* w = W + (size_t)n * 4 * Dp;
        const float* e = table + v * Dp;
        float acc = bias[n];
        for (int k = 0; k < Dp; ++k) acc = fmaf(w[2 * Dp + k] - w[k], e[k], acc);
        vc[i] = acc;
    }
}

// sum over the 16 lanes of a DPP row (lanes 16q .. 16q+15), result in every lane of the row:
// quad_perm xor 1, xor 2, then row_half_mirror and row_mirror (after two steps a quad's lanes are equal)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// Reduce-scatter over the 16 lanes of a DPP row: every lane comes in with EL partial sums v[0..EL), lane r < EL goes out
// with the row's total of ONE element, e_out(r) -- a halving butterfly (xor 1, 2 by DPP quad permutes, xor 4, 8 by
// ds_bpermute): 2 EL + 6 instructions instead of the 8 EL of EL full row sums.
template <int EL>
__device__ __forceinline__ float row16_reduce_scatter(const float (&v)[EL], int r, int& e_out) {
    static_assert(EL == 4 || EL == 8, "4 or 8 partials per lane");
    const bool b0 = r & 1, b1 = r & 2, b2 = r & 4;
    float a[EL / 2];
#pragma unroll
    for (int i = 0; i < EL / 2; ++i) {                           // xor 1: odd lanes keep the upper half
        const float send = b0 ? v[i] : v[i + EL / 2], keep = b0 ? v[i + EL / 2] : v[i];
        a[i] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
    }
    float b[EL / 4];
#pragma unroll
    for (int i = 0; i < EL / 4; ++i) {                           // xor 2
        const float send = b1 ? a[i] : a[i + EL / 4], keep = b1 ? a[i + EL / 4] : a[i];
        b[i] = keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x4E, 0xF, 0xF, true));
    }
    float t;
    if constexpr (EL == 8) {                                     // xor 4
        const float send = b2 ? b[0] : b[1], keep = b2 ? b[1] : b[0];
        t = keep + __shfl_xor(send, 4);
        e_out = (b0 ? 4 : 0) | (b1 ? 2 : 0) | (b2 ? 1 : 0);
    } else {
        t = b[0] + __shfl_xor(b[0], 4);
        e_out = (b0 ? 2 : 0) | (b1 ? 1 : 0);
    }
    return t + __shfl_xor(t, 8);
}

// sum over the four 16-lane rows (q = 0..3) of a wave, result in every lane: v_permlane16_swap /
// v_permlane32_swap (gfx950) exchange rows inside the VALU, no LDS round trip as ds_bpermute would take
// (rows4_sum: k_chain_v2.h)

// f16 MFMA over a lane's EL = 4 or 8 operand halfs (K = 16 or 32)
typedef _Float16 din_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 din_f16x8 __attribute__((ext_vector_type(8)));
// (x0..x3 | y0..y3) * scale -> packed hi / lo halfs; defined in dyn_split.h (which follows this file and uses its typedefs)
__device__ __forceinline__ void dyn_split8(f32x4 x, f32x4 y, float scale, din_f16x8& hi, din_f16x8& lo);
__device__ __forceinline__ f32x4 mfma_f16(din_f16x4 a, din_f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_f16(din_f16x8 a, din_f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// d = a * f16(lo / hi half of the dword `packed`) + c in one VALU instruction (v_fma_mix_f32: per-source f32 / f16
// selection; hipcc 7.2 emits v_cvt_f32_f16 + v_fma for fmaf(a, (float)half, c) here)
__device__ __forceinline__ float fma_mix_lo(float a, float packed, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(packed), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(float a, float packed, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(a), "v"(packed), "v"(c));
    return d;
}

// KC 16-byte pieces holding [hi(EL) | lo(EL)] halfs (EL = 4*KC) -> the two operand vectors
template <int KC>
__device__ __forceinline__ void unpack_halfs(const f32x4* x, _Float16 __attribute__((ext_vector_type(4 * KC)))& hi,
                                             _Float16 __attribute__((ext_vector_type(4 * KC)))& lo) {
    if constexpr (KC == 2) {
        hi = __builtin_bit_cast(din_f16x8, x[0]);
        lo = __builtin_bit_cast(din_f16x8, x[1]);
    } else {
        const din_f16x8 both = __builtin_bit_cast(din_f16x8, x[0]);
        hi = din_f16x4{both[0], both[1], both[2], both[3]};
        lo = din_f16x4{both[4], both[5], both[6], both[7]};
    }
}

template <int KC, int HC, int WPB = 4>
struct DinLds {
    static constexpr int KP = KC * 16, HP = HC * 16;
    static constexpr int hs = KP + 4;           // history-row stride (floats): 16-B aligned, (hs/4) odd
    static constexpr int as = HP + 4;           // alpha-row stride
    static constexpr int rows = 64;             // T <= 64, padded to whole 16-row groups
    // 16 waves per workgroup (4 per SIMD): the tiles must leave room for the tables in 160 KB -- 56 rows each (T <= 56); the
    // last 16-row group then reads 8 rows of the NEXT wave's tile (finite values, their attention weight is forced to 0), the
    // last wave's into a zeroed slack
    static constexpr int trows = WPB == 16 ? 56 : 64;
    static constexpr int alpha_floats = 2 * rows * as;      // two coefficient tables (see the epilogue)
    static constexpr int w_floats = 2 * HC * KC * 256;       // W12 / W4 fragments in lane order: [w12 | w4][nb][c][lane] float4
    static constexpr int wave_floats = trows * hs;           // Hs tile
    static constexpr int slack_floats = (rows - trows) * hs;
    static constexpr size_t bytes = sizeof(float) * (alpha_floats + w_floats + WPB * wave_floats + slack_floats);
};

// WPB waves per workgroup: 4 (two workgroups per CU, 2 waves per SIMD: round 1) or 12 (ONE workgroup per CU, 3 waves per SIMD).
// The kernel is bound by its per-sample dependency chains (LDS tile -> MFMA chain -> PReLU/Dense(1) sum -> cross-lane sum ->
// sigmoid -> pooling), not by VALU issue (cutting 14 % of the VALU instructions changed nothing), MFMA (13 % busy) or memory
// (2.8 TB/s at the fabric): the lever is more waves per SIMD, and what stood in the way was 220 VGPRs -- the resident W12 / W4
// fragments now live in LDS (8 ds_read_b128 per sample) and the PReLU coefficient rows are read after the MFMAs, not before.
// Several batches per launch (sprk_forward_many with sprk_set_many_batches > 1): sample s of the launch is sample s % B of batch
// s / B, every batch with its own ids / pooled buffers.  (Round 1's version of this was withdrawn over wrong pooled sums in
// some lanes of one instantiation; the cause -- a transcendental's result read by a v_fma_mix inside an asm statement without
// the wait state, invisible to the compiler's hazard recognizer -- is guarded where the attention weight is formed, see the
// epilogue and DESIGN.md section 8.)  A wave tracks (batch, row) of its samples with scalar adds, no division per sample.
#define DIN_ATTN_MB 16
struct DinAttnMany {
    const int* ids[DIN_ATTN_MB];
    float* pooled[DIN_ATTN_MB];
    int n;                                // batches in this launch (each of B samples)
};

struct DinAttnOne {};                     // the one-batch instantiation carries no pointer table in its kernel arguments
template <bool MB> struct DinAttnArg { typedef DinAttnOne type; };
template <> struct DinAttnArg<true> { typedef DinAttnMany type; };

// (ONE __global__ template for both forms: wrapping the body in a device function that two kernels call changed the inliner's
// decisions -- arrays indexed in the unrolled gather loops landed in scratch, 24-40 bytes per lane.)
template <int KC, int HC, int NP, bool HALF, int WPB = 4, bool MB = false>
__global__ __launch_bounds__(WPB * 64, WPB / 4 == 1 ? 2 : WPB / 4) void k_din_attn(const DinRun A, const int* __restrict__ ids,
                                                     float* __restrict__ pooled, float* __restrict__ att, int B,
                                                     int* __restrict__ err, const typename DinAttnArg<MB>::type Mm) {
    using LD = DinLds<KC, HC, WPB>;
    constexpr int KP = LD::KP, HP = LD::HP, hs = LD::hs, as = LD::as;
    static_assert(NP >= 1 && NP <= 8 && KC <= 2, "the 8-pass row gather covers 64 rows only for rows of <= 8 pieces");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int T = A.T, F = A.F, Dp = A.Dp;
    const int NV = HALF ? KP >> 2 : Dp >> 2;            // 16-B pieces per gathered row (HALF: the pre-split row is KP*4 bytes)
    const int RPP = 64 / NV;                             // rows per gather pass
    const int lrow = lane / NV, piece = lane < RPP * NV ? lane - lrow * NV : 0;
    const int G = (T + 15) >> 4;                         // 16-row groups
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    float* alpha_s = smem;
    float* wfrag_s = smem + LD::alpha_floats;
    float* Hs = smem + LD::alpha_floats + LD::w_floats + wave * LD::wave_floats;
    // first row element of this lane's c-th 4-float operand piece: the f32 MFMA steps through k = 16c + 4q + s,
    // the f16 MFMA takes EL = 4*KC consecutive elements k = EL*q .. EL*q + EL-1 per lane
    constexpr int EL = 4 * KC;
    auto kof = [&](int c) { return HALF ? EL * q + 4 * c : 16 * c + 4 * q; };
    typedef _Float16 f16xe __attribute__((ext_vector_type(EL)));

    // ---- one-time: zero the wave tile (padding columns / rows stay zero for ever), stage alpha ----
    for (int i = lane; i < LD::wave_floats; i += 64) Hs[i] = 0.f;
    if (LD::slack_floats > 0 && wave == WPB - 1)
        for (int i = lane; i < LD::slack_floats; i += 64) Hs[LD::wave_floats + i] = 0.f;
    // PF: the rows of sample n+1 are requested (into registers) while sample n is scored.  The 16-wave form cannot afford
    // the 28 registers that keeps alive (128 per wave at 4 waves per SIMD): it requests a sample's rows when it starts on the
    // sample and leaves the latency to the other three waves of its SIMD; ids are still fetched a sample ahead.
    constexpr bool PF = WPB != 16;
    // PReLU(alpha) followed by the Dense(1) weight w2, as two coefficients per (slot t, unit n):
    //   w2 (max(u,0) + alpha min(u,0)) = ca u + cb |u|,  ca = w2 (1 + alpha) / 2,  cb = w2 (1 - alpha) / 2
    // (max(u,0) = (u + |u|)/2, min(u,0) = (u - |u|)/2): two FMAs per element, |u| is a free source modifier
    float* cb_s = alpha_s + LD::rows * as;
    for (int i = tid; i < LD::rows * as; i += WPB * 64) {
        const int t = i / as, n = i - t * as;
        const bool ok = t < T && n < HP;
        const float al = ok ? A.alpha[(size_t)t * HP + n] : 0.f;
        const float w2 = ok ? A.w2[n] : 0.f;
        alpha_s[i] = 0.5f * w2 * (1.0f + al);
        cb_s[i] = 0.5f * w2 * (1.0f - al);
    }
    // W12 / W4 fragments: lane (r,q) needs W[n = nb*16 + r][kof(c) .. +3]; wave 0 lays them out in LDS in lane order, every
    // sample reads them back with conflict-free ds_read_b128 (they were 32 resident registers in round 1)
    if (wave == 0) {
#pragma unroll
        for (int nb = 0; nb < HC; ++nb)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                f32x4 w4v = ld4(A.w4 + (size_t)(nb * 16 + r) * KP + kof(c));
                if constexpr (HALF && KC == 2) w4v = w4v * A.inv_h_scale;           // meets the candidate's SCALED halfs (c * 2^sH)
                st4(wfrag_s + ((nb * KC + c) * 64 + lane) * 4, ld4(A.w12 + (size_t)(nb * 16 + r) * KP + kof(c)));
                st4(wfrag_s + ((HC * KC + nb * KC + c) * 64 + lane) * 4, w4v);
            }
    }
    __syncthreads();
    // the one-time loads above have landed before the pipelined loop starts: otherwise the compiler,
    // which cannot count in-order vmcnt across the loop's back edge, drains the prefetched rows at the
    // first use of a weight fragment inside the loop
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // s_waitcnt vmcnt(0)

    const int stride = gridDim.x * WPB;
    int s = blockIdx.x * WPB + wave;
    int Btot = B;                             // samples of this launch
    if constexpr (MB) Btot = Mm.n * B;
    // (batch, row inside the batch) of the sample being scored and of the sample whose ids are fetched next: wave-uniform
    int cbi = 0, csl = s;
    if constexpr (MB) { cbi = __builtin_amdgcn_readfirstlane(s / B); csl = s - cbi * B; }
    int pbi = cbi, psl = csl;
#define DIN_ADVANCE(bi, sl)                                                \
    do {                                                                  \
        sl = __builtin_amdgcn_readfirstlane(sl + stride);                 \
        if constexpr (MB) {                                               \
            while (sl >= B) {                                             \
                sl = __builtin_amdgcn_readfirstlane(sl - B);              \
                bi = __builtin_amdgcn_readfirstlane(bi + 1);              \
            }                                                             \
        }                                                                 \
    } while (0)
    bool bad = false;
    // Software pipeline over this wave's samples: the rows of sample n+1 are in flight (in registers)
    // while sample n is scored from the LDS tile; its ids were fetched one sample earlier still.
    // Gather pass p covers history slots p*RPP .. p*RPP+RPP-1, NV lanes (16-B pieces) per row.
    // NP passes (compile time, >= ceil(T / RPP)): no control flow inside the gather, so every load of a
    // sample is in flight together and the compiler can count them (s_waitcnt vmcnt is in-order).
    // this lane's history slot in pass p (clamped) and whether it stores what it loaded: recomputed where used (two VALU each)
    // rather than held in 2 NP registers across the sample loop
    auto prow = [&](int p) { const int row = p * RPP + lrow; return row < T ? row : T - 1; };
    auto pok = [&](int p) { return lrow < RPP && p * RPP + lrow < T; };
    int hid[NP], cid = 0;                 // ids of the sample whose rows are issued next
    f32x4 v[NP], cvn[KC], vcn[HC];         // rows / candidate row / vc row of the sample scored next
    auto ld_ids = [&](int bi, int sl) {
        const int* idsb = ids;
        if constexpr (MB) idsb = Mm.ids[bi];
        const int* row = idsb + (size_t)sl * F;
#pragma unroll
        for (int p = 0; p < NP; ++p) hid[p] = row[A.hist_col + prow(p)];
        cid = row[A.cand_col];
    };
    auto issue_rows = [&]() {
        bad |= (unsigned)cid >= (unsigned)A.vocab;
        const unsigned csafe = (unsigned)cid < (unsigned)A.vocab ? (unsigned)cid : 0u;
        if constexpr (HALF) {
            // this lane's [hi(EL) | lo(EL)] group of the candidate's pre-split row (KC 16-byte pieces), raw
            const float* crow = A.tsplit + csafe * (unsigned)KP + EL * q;
#pragma unroll
            for (int c = 0; c < KC; ++c) cvn[c] = ld4(crow + 4 * c);
        } else {
            const float* crow = A.table + csafe * (unsigned)Dp;
#pragma unroll
            for (int c = 0; c < KC; ++c) cvn[c] = (kof(c) < Dp) ? ld4(crow + kof(c)) : zero;
        }
        const float* vrow = A.vc + csafe * (unsigned)HP;
#pragma unroll
        for (int nb = 0; nb < HC; ++nb) vcn[nb] = ld4(vrow + nb * 16 + 4 * q);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            bad |= (unsigned)hid[p] >= (unsigned)A.vocab;
            const unsigned id = (unsigned)hid[p] < (unsigned)A.vocab ? (unsigned)hid[p] : 0u;
            v[p] = HALF ? ld4(A.tsplit + (id * (unsigned)KP + 4u * (unsigned)piece))
                        : ld4(A.table + (id * (unsigned)Dp + 4u * (unsigned)piece));   // 32-bit element offsets (checked at finalize)
        }
    };
    if (s < Btot) {
        ld_ids(pbi, psl);
        if constexpr (PF) {
            issue_rows();
            DIN_ADVANCE(pbi, psl);
            if (s + stride < Btot) ld_ids(pbi, psl);
        }
    }
    for (; s < Btot; s += stride) {
        if constexpr (!PF) {
            issue_rows();                                         // this sample's rows; then the next sample's ids
            DIN_ADVANCE(pbi, psl);
            if (s + stride < Btot) ld_ids(pbi, psl);
        }
        // ---- hand-off: this sample's rows -> LDS tile, candidate-side operands -> registers ----
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (pok(p)) st4(Hs + prow(p) * hs + 4 * piece, v[p]);
        f32x4 cv[KC], acc_init[HC];
        if constexpr (HALF && KC != 2) {
            // candidate row back to f32 from its halfs: c[EL*q + e] = (hi + lo) * 2^-sH
            f16xe chi, clo;
            unpack_halfs<KC>(cvn, chi, clo);
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) cv[c][j] = ((float)chi[4 * c + j] + (float)clo[4 * c + j]) * A.inv_h_scale;
        } else if constexpr (HALF) {
#pragma unroll
            for (int c = 0; c < KC; ++c) cv[c] = cvn[c];          // KC == 2: the packed halfs themselves (piece 0 = hi, piece 1 = lo), see below
        } else {
#pragma unroll
            for (int c = 0; c < KC; ++c) cv[c] = cvn[c];
        }
#pragma unroll
        for (int nb = 0; nb < HC; ++nb) acc_init[nb] = vcn[nb];
        if (PF && s + stride < Btot) {                           // next sample's rows fly under this sample's MFMAs
            issue_rows();
            DIN_ADVANCE(pbi, psl);
            if (s + 2 * stride < Btot) ld_ids(pbi, psl);
        }
        // A_b = W12 + W4 diag(c); the weight fragments come from LDS (lane-ordered, conflict free)
        f32x4 w12f[HC][KC], w4f[HC][KC];
        {
            int wo = lane * 4;
            asm volatile("" : "+v"(wo));                          // keeps these loop-invariant reads inside the sample loop (registers)
#pragma unroll
            for (int nb = 0; nb < HC; ++nb)
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    w12f[nb][c] = ld4(wfrag_s + (nb * KC + c) * 256 + wo);
                    w4f[nb][c] = ld4(wfrag_s + (HC * KC + nb * KC + c) * 256 + wo);
                }
        }
        f32x4 Ab[HC][KC];
        if constexpr (HALF && KC == 2) {
            // straight from the candidate's packed halfs: w4s (c_hi + c_lo) + w12 as two mixed-precision FMAs per element
            // (w4s = W4 * 2^-sH, folded once per wave): no unpack, no rescale -- 32 VALU instead of 48 per sample
#pragma unroll
            for (int nb = 0; nb < HC; ++nb)
#pragma unroll
                for (int e = 0; e < EL; ++e) {
                    const float w4 = w4f[nb][e >> 2][e & 3];
                    float x = w12f[nb][e >> 2][e & 3];   // (fragments re-read from LDS just above)
                    x = (e & 1) ? fma_mix_hi(w4, cv[1][e >> 1], x) : fma_mix_lo(w4, cv[1][e >> 1], x);    // + w4s * c_lo
                    x = (e & 1) ? fma_mix_hi(w4, cv[0][e >> 1], x) : fma_mix_lo(w4, cv[0][e >> 1], x);    // + w4s * c_hi
                    Ab[nb][e >> 2][e & 3] = x;
                }
        } else {
#pragma unroll
            for (int nb = 0; nb < HC; ++nb)
#pragma unroll
                for (int c = 0; c < KC; ++c) Ab[nb][c] = w4f[nb][c] * cv[c] + w12f[nb][c];
        }
        f16xe Ahi[HC], Alo[HC];                                  // HALF: A_b * a_scale as hi + lo halfs
        if constexpr (HALF) {
#pragma unroll
            for (int nb = 0; nb < HC; ++nb) {
                if constexpr (KC == 2) {
                    // hi = f16(x), lo = f16(x - hi) with v_fma_mixlo/hi_f16 (2 VALU per element, halfs land packed; ends
                    // with the hazard guard an asm-written MFMA operand needs -- dyn_split.h)
                    dyn_split8(Ab[nb][0], Ab[nb][1], 1.0f, Ahi[nb], Alo[nb]);
                } else {
#pragma unroll
                    for (int c = 0; c < KC; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float x = Ab[nb][c][j];        // (w12 / w4 were pre-multiplied by a_scale)
                            const _Float16 hh = (_Float16)x;
                            Ahi[nb][4 * c + j] = hh;
                            Alo[nb][4 * c + j] = (_Float16)(x - (float)hh);
                        }
                }
                acc_init[nb] = acc_init[nb] * A.acc_scale;       // C operand in the accumulator's scale
            }
        }

        // ---- attention logits, two 16-row groups (= 2*HC accumulator chains) at a time ----
        f32x4 pacc[KC];                                          // this lane's share of sum_t w[t] h[t][kof(c) .. +3]
#pragma unroll
        for (int c = 0; c < KC; ++c) pacc[c] = zero;
        auto score_groups = [&](int g, auto two_tag) {
            constexpr bool TWO = decltype(two_tag)::value;
            f32x4 b0[KC], b1[KC], a0[HC], a1[HC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                // (HALF: the tile holds pre-split rows; EL*q + 4c addresses this lane's [hi | lo] group, raw)
                b0[c] = ld4(Hs + (16 * g + r) * hs + (HALF ? EL * q + 4 * c : kof(c)));
                b1[c] = TWO ? ld4(Hs + (16 * g + 16 + r) * hs + (HALF ? EL * q + 4 * c : kof(c))) : zero;
            }
#pragma unroll
            for (int nb = 0; nb < HC; ++nb) {
                a0[nb] = acc_init[nb];
                a1[nb] = acc_init[nb];
            }
            if constexpr (HALF) {
                // three products per (group, n-block): 2*HC (4*HC with TWO) accumulator chains issued round robin
                f16xe bh0, bl0, bh1, bl1;                          // the gathered bytes ARE the B operands
                unpack_halfs<KC>(b0, bh0, bl0);
                unpack_halfs<KC>(b1, bh1, bl1);
#pragma unroll
                for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
                    for (int nb = 0; nb < HC; ++nb) {
                        a0[nb] = mfma_f16(pr == 2 ? Alo[nb] : Ahi[nb], pr == 1 ? bl0 : bh0, a0[nb]);
                        if (TWO) a1[nb] = mfma_f16(pr == 2 ? Alo[nb] : Ahi[nb], pr == 1 ? bl1 : bh1, a1[nb]);
                    }
                }
            } else {
#pragma unroll
            for (int c = 0; c < KC; ++c)
#pragma unroll
                for (int st = 0; st < 4; ++st) {
#pragma unroll
                    for (int nb = 0; nb < HC; ++nb)
                        a0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[nb][c][st], b0[c][st], a0[nb], 0, 0, 0);
                    if (TWO) {
#pragma unroll
                        for (int nb = 0; nb < HC; ++nb)
                            a1[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ab[nb][c][st], b1[c][st], a1[nb], 0, 0, 0);
                    }
                }
            }
            // epilogue: PReLU(alpha[t][n]) -> Dense(1) (both in ca / cb) -> sigmoid (DIN.py:150-151); lane (r,q) holds
            // u[n = nb*16 + 4q + j] of row t = 16g + r
#pragma unroll
            for (int h = 0; h < (TWO ? 2 : 1); ++h) {
                const int t = 16 * (g + h) + r;
                // ca[t][n], cb[t][n] of this row, read here (after the MFMAs were issued) rather than held across them;
                // four independent partial sums: the 16-term chain was on the per-sample critical path
                float sum = 0.f, sum1 = 0.f;
#pragma unroll
                for (int nb = 0; nb < HC; ++nb) {
                    const f32x4 ca = ld4(alpha_s + t * as + nb * 16 + 4 * q);
                    const f32x4 cb = ld4(cb_s + t * as + nb * 16 + 4 * q);
                    const f32x4 u = h ? a1[nb] : a0[nb];
                    sum = fmaf(cb[0], __builtin_fabsf(u[0]), fmaf(ca[0], u[0], sum));
                    sum1 = fmaf(cb[1], __builtin_fabsf(u[1]), fmaf(ca[1], u[1], sum1));
                    sum = fmaf(cb[2], __builtin_fabsf(u[2]), fmaf(ca[2], u[2], sum));
                    sum1 = fmaf(cb[3], __builtin_fabsf(u[3]), fmaf(ca[3], u[3], sum1));
                }
                sum += sum1;
                float wgt = sigmoidf_fast(rows4_sum(sum) * (HALF ? A.unscale : 1.0f) + A.b2);   // PReLU is positively homogeneous
                // HAZARD GUARD.  wgt comes out of v_rcp_f32, a transcendental: on gfx940+ a non-transcendental VALU instruction
                // that reads a transcendental's result needs ONE wait state, which hipcc inserts for the consumers it can see
                // (the s_nop 0 between v_exp and v_add in sigmoidf_fast) -- but the first consumer here is the v_fma_mix_f32
                // inside an asm statement, which the hazard recognizer does not look into.  In the several-batches-per-launch
                // instantiation the scheduler put that v_fma_mix straight behind the v_rcp: the first pooled element of every
                // lane was accumulated with a stale weight (columns 0, 4, 8, ... of every pooled vector wrong, everything else
                // exact) -- round 1's withdrawn multi-batch kernel showed the same picture.  One statement that owns wgt and
                // carries the wait state closes it wherever the scheduler puts the pooling.
                if constexpr (HALF) asm volatile("s_nop 0" : "+v"(wgt));
                if constexpr (LD::trows < LD::rows) wgt = t < T ? wgt : 0.f;       // rows of the neighbouring tile: no weight
                if constexpr (!MB) { if (q == 0 && att && t < T) att[(size_t)s * T + t] = wgt; }
                // weighted sum pooling (DIN.py:152-158): rows past T are all-zero in the tile, so they add nothing
                if constexpr (HALF) {
                    // w[t] * (hi + lo): two mixed-precision FMAs per element straight from the packed halfs
                    // (element e = 4c + j: KC == 2: hi in dword e/2 of piece 0, lo in dword e/2 of piece 1;
                    //  KC == 1: hi in dword e/2, lo in dword 2 + e/2 of the one piece)
                    const f32x4* bb = h ? b1 : b0;
#pragma unroll
                    for (int e = 0; e < EL; ++e) {
                        const float hw = KC == 2 ? bb[0][e >> 1] : bb[0][e >> 1];
                        const float lw = KC == 2 ? bb[KC - 1][e >> 1] : bb[0][2 + (e >> 1)];
                        float acc = pacc[e >> 2][e & 3];
                        acc = (e & 1) ? fma_mix_hi(wgt, hw, acc) : fma_mix_lo(wgt, hw, acc);
                        acc = (e & 1) ? fma_mix_hi(wgt, lw, acc) : fma_mix_lo(wgt, lw, acc);
                        pacc[e >> 2][e & 3] = acc;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < KC; ++c) pacc[c] += wgt * (h ? b1[c] : b0[c]);
                }
            }
        };
        {
            int g = 0;
            for (; g + 1 < G; g += 2) score_groups(g, std::true_type{});
            if (g < G) score_groups(g, std::false_type{});
        }

        // ---- reduce the pooled partials over the 16 r-lanes of each q row: lane r < EL ends up with element e(r) ----
        {
            float pv[EL];
#pragma unroll
            for (int e = 0; e < EL; ++e) pv[e] = pacc[e >> 2][e & 3];
            int e_mine;
            const float tot = row16_reduce_scatter<EL>(pv, r, e_mine);
            const int k = kof(e_mine >> 2) + (e_mine & 3);
            float* pout = pooled;
            if constexpr (MB) pout = Mm.pooled[cbi];
            if (r < EL && k < Dp) pout[(size_t)csl * Dp + k] = HALF ? tot * A.inv_h_scale : tot;
        }
        DIN_ADVANCE(cbi, csl);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
#undef DIN_ADVANCE
}

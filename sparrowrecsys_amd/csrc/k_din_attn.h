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

// [r6] k_din_attn -- round 2's attention kernel, one WAVE per sample (its history rows gathered once into a wave-private LDS tile, A_b built in
// registers per sample; 4-, 12- and several-batches forms, split-f16 and f32: twenty instantiations) -- lived here until round 6.  Since round 3
// every shape the reference or BASELINE has runs on k_din_attn_cols / k_din_fused (sixteen samples per MFMA tile, the weights as the static
// operand); this kernel only took what those refuse -- attention hidden != 32, operands outside the split's range, SPRK_DIN_COLS=0 /
// SPRK_DIN_HALF=0.  Those now run the generic stage k_din_pool (k_tile_forward.h; fp32 MFMA).  What is left in this file is what the successors
// share: the algebra above, the finalize kernels that build W12 / W4 / vc and the pre-split table, the small device helpers.

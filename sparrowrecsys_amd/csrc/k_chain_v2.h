// k_chain_v2.h -- register-chained fused forward for DeepFM_v2-structured plans
// (reference graph: DeepFM_v2.py:98-155; BASELINE config 2 is this graph at F=6, D=16).
// Included inside sparrow_hip.hip's anonymous namespace.
//
// One WAVE owns 16 samples from ids to score.  Lane (r = lane&15, q = lane>>4) is sample r's q-th
// 16-byte column slot, which is at the same time
//   * the unit of the embedding-row gather (4 lanes x 16 B = one 64-B row), and
//   * the B-operand layout of v_mfma_f32_16x16x4_f32 (lane supplies B[k=4q+s][col=r] at step s), and
//   * the C/D layout of the previous layer's output (lane holds D[row=4q+j][col=r]).
// So with the weights as the A operand (W^T rows, read from LDS), embedding rows are loaded from
// HBM straight into MFMA operand registers and every layer's output feeds the next layer's MFMA
// without leaving the register file: per-field Dense projections -> FM (sum)^2 - sum(squares) ->
// Dense+ReLU -> Dense+ReLU -> output dot + sigmoid.  No activation ever touches LDS or HBM, there
// is no barrier after the one-time weight staging, and waves progress independently, so gather
// latency of one wave hides under the MFMAs of its neighbours.
//
// FOLD: a per-field Dense with no activation applied to a gathered row is itself a table:
// (table_g @ Wp_g + bp_g)[id].  When the projection is not wider than the embedding the library
// builds those projected tables once at sprk_finalize (k_v2_fold, same fp32 fmaf order as the
// in-kernel MFMA chain, so bit-identical) and the kernel gathers P_g directly: same bytes per
// sample, 24 of 92 MFMAs per 16 samples gone.
//
// Memory instructions per 16 samples: ONE coalesced 16-B/lane load brings the task's contiguous
// ids block (lanes 0..4F-1) and numerics block (lanes 32..32+4ND-1); it is re-distributed to the
// (r,q) layout through a 1-KB wave-private LDS slot.  Then G_EMB row gathers and two first-order
// gathers (lane (r,q) fetches the weight of field q, then of field q+4; the cross-q sum rides the
// output reduction).
//
// HBM traffic per sample = ids + gathered rows + first-order weights + numerics + score (the
// algorithmic minimum, 464 B at F=6, D=16).

#define V2_MAX_FIELDS 8

struct V2Args {
    int F;                                // ids row width (int32 columns)
    int ND;                               // dense row width
    int n_num;                            // numeric columns used (<= 8)
    int n_fo;                             // first-order fields
    int emb_col[V2_MAX_FIELDS];           // ids column of embedding group g
    int emb_vocab[V2_MAX_FIELDS];
    int fo_col[V2_MAX_FIELDS];            // ids column of first-order field i
    int fo_vocab[V2_MAX_FIELDS];
    const float* table[V2_MAX_FIELDS];    // [vocab+1][4*DV] padded embedding tables (last row zero)
    const float* w1[V2_MAX_FIELDS];       // [vocab+1] first-order weights (last zero)
    const float* Wp[V2_MAX_FIELDS + 1];   // projection W^T: [Kp][ldp] (group G_EMB = numerics, ldp_num)
    const float* bp[V2_MAX_FIELDS + 1];   // projection bias [Kp]
    int ldp_emb, ldp_num;
    const float* W0; const float* b0;     // deep0 W^T [H0p][(G_EMB+1)*Kp], bias [H0p]
    const float* W1; const float* b1;     // deep1 W^T [H1p][H0p], bias [H1p]
    const float* hfm; int n_hfm;          // output-layer weights on the FM vector
    const float* hdeep; int n_hdeep;      // ... on the deep vector
    const float* fo_num_w;                // Dense(1) over the numerics (first order)
    float h0w;                            // output-layer weight of the first-order scalar
    float fo_bias;                        // fo_cat bias + fo_num bias
    float head_bias;
};

template <int G_EMB, int DV, int KPC, int H0C, int H1C, bool FOLD>
struct V2Lds {
    static constexpr int G = G_EMB + 1;
    static constexpr int DPC = (DV + 3) / 4;          // 16-float chunks per embedding row
    static constexpr int KP = KPC * 16;
    static constexpr int SP = DPC * 16 + 4;           // LDS row stride of an embedding-group projection W^T
    static constexpr int SN = 16 + 4;                 // ... of the numeric group
    static constexpr int S0 = G * KP + 4;             // deep0 W^T row stride
    static constexpr int S1 = H0C * 16 + 4;           // deep1 W^T row stride
    static constexpr int off_wp = 0;                  // [G_EMB][KP][SP]   (absent when FOLD)
    static constexpr int off_wn = off_wp + (FOLD ? 0 : G_EMB * KP * SP);   // [KP][SN]
    static constexpr int off_bp = off_wn + KP * SN;   // [G][KP]
    static constexpr int off_w0 = off_bp + G * KP;    // [H0C*16][S0]
    static constexpr int off_b0 = off_w0 + H0C * 16 * S0;
    static constexpr int off_w1 = off_b0 + H0C * 16;  // [H1C*16][S1]
    static constexpr int off_b1 = off_w1 + H1C * 16 * S1;
    static constexpr int off_hfm = off_b1 + H1C * 16; // [KP]
    static constexpr int off_hd = off_hfm + KP;       // [H1C*16]
    static constexpr int off_fn = off_hd + H1C * 16;  // [8]
    static constexpr int total = off_fn + 8;          // floats in the weight image
    static constexpr int total_pad = (total + 255) & ~255;   // ... rounded up to whole 1-KB LDS-DMA pieces
    static constexpr int stage_floats = 256;          // per-wave ids/numerics slot: ids [0,128), numerics [128,256)
};

// copy a [rows][ld] global matrix into LDS [rows_pad][stride], zero-filling everything outside
__device__ __forceinline__ void stage_matrix(float* dst, int rows_pad, int stride, const float* src, int rows,
                                             int cols, int ld, int tid, int nthreads) {
    const int total = rows_pad * stride;
    for (int i = tid; i < total; i += nthreads) {
        const int r = i / stride, c = i - r * stride;
        dst[i] = (r < rows && c < cols) ? src[(size_t)r * ld + c] : 0.f;
    }
}
__device__ __forceinline__ void stage_vector(float* dst, int n_pad, const float* src, int n, int tid, int nthreads) {
    for (int i = tid; i < n_pad; i += nthreads) dst[i] = (i < n) ? src[i] : 0.f;
}

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}

// single-instruction ReLU: v_med3_f32(v, 0, BIG) (fmaxf() first canonicalises an operand it cannot prove quiet -- an MFMA result, a
// loaded value -- with a second v_max; an inline-asm v_max would hide the VALU-write -> MFMA-read hazard from the compiler).
// [r5] BIG is FINITE: hipcc 7.2 folds fmed3(v, 0, +inf) back into fmaxf(v, 0), canonicalisation included (build/sparrow.s: two v_max
// per element behind every MFMA); med3 against 3e38 stays one v_med3_f32 and is the same function for every finite v <= 3e38.
#define SPRK_RELU_BIG 3.0e38f
__device__ __forceinline__ float relu1_fast(float v) {
    return __builtin_amdgcn_fmed3f(v, 0.f, SPRK_RELU_BIG);
}
__device__ __forceinline__ float neg1_fast(float v) {                    // min(v, 0), one instruction, same reasoning
    return __builtin_amdgcn_fmed3f(v, -SPRK_RELU_BIG, 0.f);
}
__device__ __forceinline__ f32x4 relu4_fast(f32x4 v) {
    return f32x4{relu1_fast(v.x), relu1_fast(v.y), relu1_fast(v.z), relu1_fast(v.w)};
}
// sum over the four 16-lane rows of a wave, (row0 + row1) + (row2 + row3), in every lane
__device__ __forceinline__ float rows4_sum(float v) {
    // inline asm: hipcc 7.2's __builtin_amdgcn_permlane{16,32}_swap hands back its FIRST result for both
    // elements of the returned pair (the sum became x + x).  s_nop 1 = the two wait states a VALU-written
    // VGPR needs before a permlane swap reads it; the assembler inserts nothing inside asm statements.
#if defined(SPRK_NO_ASM) || defined(SPRK_NO_ASM_ROWS4)
    { const float t = v + __shfl_xor(v, 16); return t + __shfl_xor(t, 32); }
#endif
#ifdef SPRK_ASM_PAD
    { float a = v, b = v;
      asm volatile("s_nop 7\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
      a += b; b = a;
      asm volatile("s_nop 7\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
      return a + b; }
#endif
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = [v0 v0 v2 v2], b = [v1 v1 v3 v3]
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = [lo lo], b = [hi hi]
    return a + b;
}

// 1/(1+exp(-z)) on the hardware exp2/rcp units (4 VALU instructions; |err| < 3e-7 absolute on the score)
__device__ __forceinline__ float sigmoidf_fast(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-z));
}

// One-time (finalize) kernel: lays every weight out exactly as the fused kernel wants it in LDS
// (padded row strides, zero fill), so the per-launch staging is a flat copy.
template <int G_EMB, int DV, int KPC, int H0C, int H1C, bool FOLD>
__global__ __launch_bounds__(256) void k_v2_pack_image(const V2Args A, float* __restrict__ lds) {
    using LD = V2Lds<G_EMB, DV, KPC, H0C, H1C, FOLD>;
    constexpr int G = LD::G, KP = LD::KP;
    const int tid = threadIdx.x, NT = 256;
    static_assert(LD::total % 4 == 0, "LDS image must be a whole number of float4");
    if (!FOLD)
        for (int g = 0; g < G_EMB; ++g)
            stage_matrix(lds + LD::off_wp + g * KP * LD::SP, KP, LD::SP, A.Wp[g], KP, 4 * DV, A.ldp_emb, tid, NT);
    stage_matrix(lds + LD::off_wn, KP, LD::SN, A.Wp[G_EMB], KP, A.n_num, A.ldp_num, tid, NT);
    for (int g = 0; g < G; ++g) stage_vector(lds + LD::off_bp + g * KP, KP, A.bp[g], KP, tid, NT);
    stage_matrix(lds + LD::off_w0, H0C * 16, LD::S0, A.W0, H0C * 16, G * KP, G * KP, tid, NT);
    stage_vector(lds + LD::off_b0, H0C * 16, A.b0, H0C * 16, tid, NT);
    stage_matrix(lds + LD::off_w1, H1C * 16, LD::S1, A.W1, H1C * 16, H0C * 16, H0C * 16, tid, NT);
    stage_vector(lds + LD::off_b1, H1C * 16, A.b1, H1C * 16, tid, NT);
    stage_vector(lds + LD::off_hfm, KP, A.hfm, A.n_hfm, tid, NT);
    stage_vector(lds + LD::off_hd, H1C * 16, A.hdeep, A.n_hdeep, tid, NT);
    stage_vector(lds + LD::off_fn, 8, A.fo_num_w, A.n_num, tid, NT);
    if (FOLD) {                                               // FOLD kernels expect h0w * fo_num (see k_v2_fold)
        __syncthreads();
        if (tid < 8) lds[LD::off_fn + tid] *= A.h0w;
    }
}

// One-time (finalize) kernel: projected table P[v][n] = bp[n] + sum_k Wp^T[n][k] * table[v][k],
// v in [0, rows) (the last row of `table` is the all-zero "missing id" row, so P's last row is the
// bias).  The k order is the one the in-kernel MFMA chain uses (within each 16-chunk: k = 4q+s for
// s outer, q inner), one fmaf per term, so folding does not change a single bit of P_g.
// Output rows are [P (KP floats) | row scalar | 15 zero floats]: with KP = 16 one 128-byte L2 line
// holds everything the forward needs for an id.  The row scalar collects every term of the logit
// that depends on this id alone:
//     h0w * w1[v]                      first-order weight times its output-layer weight
//   - sum_n hfm[n] * P[v][n]^2         this field's share of the FM "sum of squares"
// so the kernel only has to accumulate S = sum_g P_g for the FM cross.
static __global__ __launch_bounds__(256) void k_v2_fold(const float* __restrict__ table, int row_floats,
                                                 const float* __restrict__ Wp, int ldp, const float* __restrict__ bp,
                                                 const float* __restrict__ w1, const float* __restrict__ hfm, int n_hfm,
                                                 float h0w, float* __restrict__ out, int KP, long long rows) {
    const int OS = KP + 16;
    for (long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); v < rows; v += (long long)gridDim.x * 4) {
        const int n = threadIdx.x & 63;                       // one wave per row, lane = output column
        float acc = 0.f;
        if (n < KP) {
            const float* x = table + v * row_floats;
            const float* w = Wp + (size_t)n * ldp;
            acc = bp[n];
            for (int c = 0; c < row_floats; c += 16)
                for (int s = 0; s < 4; ++s)
                    for (int q = 0; q < 4; ++q) {
                        const int k = c + 4 * q + s;
                        if (k < row_floats) acc = fmaf(w[k], x[k], acc);
                    }
        }
        float sqw = (n < n_hfm && n < KP) ? hfm[n] * acc * acc : 0.f;
        for (int d = 32; d >= 1; d >>= 1) sqw += __shfl_xor(sqw, d);
        float* o = out + v * OS;
        if (n < KP) o[n] = acc;
        else if (n == KP) o[n] = h0w * w1[v] - sqw;
        else if (n < OS) o[n] = 0.f;
    }
}

// Run-time arguments of the fused kernel (the weights travel through the packed image).
struct V2Run {
    int F, ND, n_num;
    int col[V2_MAX_FIELDS];               // ids column of field g (embedding AND first-order weight)
    int vocab[V2_MAX_FIELDS];
    unsigned fo_off[V2_MAX_FIELDS];       // not FOLD: start of field g's block inside fo_all
    const float* table[V2_MAX_FIELDS];    // not FOLD: [vocab+1][4*DV] embedding tables
    const float* fo_all;                  // not FOLD: concatenated first-order blocks, each [vocab+1] (last entry 0)
    const float* tab0;                    // FOLD: ONE buffer of [KP+16]-float rows {P | row scalar | 0..}, all fields back to back
    unsigned rowbase[V2_MAX_FIELDS];      // FOLD: first row of field g inside tab0 (its block has vocab+1 rows)
    float h0w, fo_bias, head_bias;
    int flags;                            // 1 = ids/dense not 16-byte aligned: stage element-wise
};

// cold path of the ids/numerics staging: a partial last task, or inputs that do not start on a
// 16-byte boundary -- element-wise, rows past the end of the batch clamped to the last row
static __device__ __noinline__ void stage_task_slow(float* stage, const int* __restrict__ ids, const float* __restrict__ dense,
                                             int F, int ND, int tk, int B, int lane) {
    int* si = reinterpret_cast<int*>(stage);
#pragma clang loop vectorize(disable) unroll(disable)
    for (int e = lane; e < 16 * F; e += 64) {
        const int mm = e / F, c = e - mm * F;
        const int m = min(tk * 16 + mm, B - 1);
        si[e] = ids[(size_t)m * F + c];
    }
#pragma clang loop vectorize(disable) unroll(disable)
    for (int e = lane; e < 16 * ND; e += 64) {
        const int mm = e / ND, c = e - mm * ND;
        const int m = min(tk * 16 + mm, B - 1);
        stage[128 + e] = dense[(size_t)m * ND + c];
    }
}

// [r6] k_deepfm_v2_chain -- round 1's fused DeepFM_v2 kernel (every field gathered per field, weights in registers, two tasks per wave;
// with its TRACE instantiation behind sprk_debug_set_trace) -- lived here until round 6.  Since round 3 it only ran where the joint kernels
// (k_chain_v2j.h, k_chain_v2j1.h) refused a model -- no small-vocabulary field to put into LDS, or more than three large ones -- and under the
// A/B switches SPRK_V2_JOINT=0 / SPRK_V2_FOLD=0.  Those models now go where the DeepFM_v2 graphs the joint kernels were never built for already
// went: k_rows_chain (host_setup_rows.h), then the interpreter.  What is left in this file is what the joint kernels share: the plan's
// argument block, the LDS image layout and its pack kernel, the fold kernel, the small device helpers.

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
    unsigned long long* trace;            // TRACE instantiation only: per-wave phase timestamps (sprk_debug_set_trace)
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

// diagnostics (TRACE instantiation): lane 0 of every wave stamps the shader clock (s_memtime),
// optionally after draining its loads
template <bool TRACE>
__device__ __forceinline__ void trace_stamp(unsigned long long* trace, int wave_global, int k, bool drain) {
    if constexpr (TRACE) {
        if (trace) {
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == 0) trace[(size_t)wave_global * 16 + k] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// two independent 4-step MFMA chains issued alternately (a 16x16x4 MFMA has a 40-cycle dependent
// latency but a 32-cycle issue interval: alternating chains keeps the matrix pipe full from one wave)
__device__ __forceinline__ void mfma4x2(f32x4 a0, f32x4 b0, f32x4& c0, f32x4 a1, f32x4 b1, f32x4& c1) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, c1, 0, 0, 0);
}

// Each stage has ONE call site inside a software-pipelined task loop (trip i gathers task i, then
// scores task i-1 whose rows were issued a trip earlier), which keeps the code small: at one
// 16-sample task per wave (B = 65 536 fills the chip exactly once) every instruction runs once per
// launch from a cold instruction cache.
//
// REG: register-resident weights.  The wave copies its (r,q) slice of every W^T fragment from the LDS
// image into VGPRs once (96 registers at config 2), runs at 2 waves per SIMD, and its scoring stage is
// then pure MFMA issue -- four independent accumulator chains (deep0's two n-blocks x even/odd K
// chunks) pinned in round-robin order by sched_barriers -- with no LDS traffic at all, so ONE wave
// keeps the matrix pipe full while its SIMD partner gathers.
template <int G_EMB, int DV, int KPC, int H0C, int H1C, int WAVES, bool FOLD, bool TRACE, bool REG>
__global__ __launch_bounds__(WAVES * 64, REG ? 2 : 4) void k_deepfm_v2_chain(const V2Run A, const int* __restrict__ ids,
                                                                const float* __restrict__ dense,
                                                                float* __restrict__ out, int B,
                                                                int* __restrict__ err,
                                                                const float* __restrict__ image) {
    using LD = V2Lds<G_EMB, DV, KPC, H0C, H1C, FOLD>;
    constexpr int G = LD::G, KP = LD::KP;
    constexpr int XC = FOLD ? KPC : LD::DPC;          // 16-float chunks per gathered row
    constexpr int XV = FOLD ? 4 * KPC : DV;           // float4 per gathered row
    constexpr int RS = FOLD ? KP + 16 : 4 * DV;       // floats between consecutive table rows
    constexpr int NT = WAVES * 64;
    static_assert(H0C % 2 == 0, "deep0 n-blocks are processed in interleaved pairs");
    static_assert(G_EMB >= 1 && G_EMB <= V2_MAX_FIELDS, "field count");
    static_assert(FOLD == REG, "folded tables carry the row scalars only the register-resident scoring stage understands");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntasks = (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    const int wave_global = blockIdx.x * WAVES + wave;
    float* stage = smem + LD::total_pad + wave * LD::stage_floats;
    const float* wq = smem + 4 * q;                       // this lane's 16-byte column slot of the weight image

    // ---- gather stage, consumed one loop trip after it was issued:
    //        ld_raw : the task's contiguous ids / numerics blocks, one 16-B load per lane (async)
    //        gather : VGPR -> wave-private LDS slot -> the (r,q) lanes that need them, then the
    //                 embedding rows + first-order weights                             (async)
    //      A missing (-1) or out-of-range id is redirected to the all-zero row the host appended at
    //      index `vocab`, so nothing needs a select. ----
    f32x4 raw = zero;
    f32x4 x[G_EMB][XC];
    f32x4 xn = zero;
    float w1a = 0.f, w1b = 0.f;
    bool bad = false;
    const bool aligned = !(A.flags & 1);
    auto ld_raw = [&](int tk) {
        if (aligned && tk * 16 + 16 <= B) {                       // wave-uniform
            const bool isid = lane < 32;
            const int j = isid ? lane : lane - 32;
            const int n4 = 4 * (isid ? A.F : A.ND);
            const float* src = isid ? reinterpret_cast<const float*>(ids) + (size_t)tk * 16 * A.F
                                    : dense + (size_t)tk * 16 * A.ND;
            raw = ld4(src + 4 * (j < n4 ? j : 0));
        }
    };
    auto gather = [&](int tk) {
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
        unsigned sid[G_EMB];
#pragma unroll
        for (int g = 0; g < G_EMB; ++g) {
            const int id = sid_row[A.col[g]];
            bad |= (unsigned)(id + 1) > (unsigned)A.vocab[g];      // neither a table row nor the "missing" marker -1
            sid[g] = min((unsigned)id, (unsigned)A.vocab[g]);      // -1 / out of range -> the zero row at index vocab
            if (FOLD) sid[g] += A.rowbase[g];
            if (A.flags & 16) sid[g] = FOLD ? A.rowbase[g] : 0;    // experiment: every gather hits one hot row
        }
        {
            const float* nrow = stage + 128 + r * A.ND;
            const int c0 = 4 * q, last = A.n_num - 1;
            // lane slots beyond n_num hold a duplicate finite value that only ever meets zero weights
            xn.x = nrow[min(c0 + 0, last)];
            xn.y = nrow[min(c0 + 1, last)];
            xn.z = nrow[min(c0 + 2, last)];
            xn.w = nrow[min(c0 + 3, last)];
        }
        if constexpr (FOLD) {
            // one SGPR base + 32-bit byte offsets: row (sid << 7 for KP = 16), this lane's piece q << 4
            const char* tb = reinterpret_cast<const char*>(A.tab0);
            constexpr unsigned RB = RS * 4;                        // bytes per row
#pragma unroll
            for (int g = 0; g < G_EMB; ++g)
#pragma unroll
                for (int c = 0; c < XC; ++c)
                    x[g][c] = *reinterpret_cast<const f32x4*>(tb + (sid[g] * RB + 64u * c + 16u * q));
            // row scalars: lane (r,q) fetches field q's, then field q+4's (lanes beyond the field count are dropped at hand-off)
            unsigned sa = sid[0];
            if (G_EMB > 1) sa = q == 1 ? sid[G_EMB > 1 ? 1 : 0] : sa;
            if (G_EMB > 2) sa = q == 2 ? sid[G_EMB > 2 ? 2 : 0] : sa;
            if (G_EMB > 3) sa = q == 3 ? sid[G_EMB > 3 ? 3 : 0] : sa;
            w1a = *reinterpret_cast<const float*>(tb + (sa * RB + 4u * KP));
            if (G_EMB > 4) {
                unsigned sb = sid[G_EMB > 4 ? 4 : 0];
                if (G_EMB > 5) sb = q == 1 ? sid[G_EMB > 5 ? 5 : 0] : sb;
                if (G_EMB > 6) sb = q == 2 ? sid[G_EMB > 6 ? 6 : 0] : sb;
                if (G_EMB > 7) sb = q == 3 ? sid[G_EMB > 7 ? 7 : 0] : sb;
                w1b = *reinterpret_cast<const float*>(tb + (sb * RB + 4u * KP));
            }
        } else {
#pragma unroll
            for (int g = 0; g < G_EMB; ++g) {
                const unsigned rowoff = sid[g] * (unsigned)RS;
#pragma unroll
                for (int c = 0; c < XC; ++c) {
                    // this lane's 16-byte piece of the row; pieces past the row end re-read piece 0 and
                    // only ever meet zero weights
                    const unsigned o = (4 * c + 3 < XV || 4u * c + q < (unsigned)XV) ? rowoff + 16u * c + 4u * q : rowoff;
                    x[g][c] = ld4(A.table[g] + o);
                }
            }
            // first-order weights: lane (r,q) fetches field q's, then field q+4's
            unsigned oa = A.fo_off[0] + sid[0];
            if (G_EMB > 1) oa = q == 1 ? A.fo_off[G_EMB > 1 ? 1 : 0] + sid[G_EMB > 1 ? 1 : 0] : oa;
            if (G_EMB > 2) oa = q == 2 ? A.fo_off[G_EMB > 2 ? 2 : 0] + sid[G_EMB > 2 ? 2 : 0] : oa;
            if (G_EMB > 3) oa = q == 3 ? A.fo_off[G_EMB > 3 ? 3 : 0] + sid[G_EMB > 3 ? 3 : 0] : oa;
            w1a = A.fo_all[oa];
            if (G_EMB > 4) {
                unsigned ob = A.fo_off[G_EMB > 4 ? 4 : 0] + sid[G_EMB > 4 ? 4 : 0];
                if (G_EMB > 5) ob = q == 1 ? A.fo_off[G_EMB > 5 ? 5 : 0] + sid[G_EMB > 5 ? 5 : 0] : ob;
                if (G_EMB > 6) ob = q == 2 ? A.fo_off[G_EMB > 6 ? 6 : 0] + sid[G_EMB > 6 ? 6 : 0] : ob;
                if (G_EMB > 7) ob = q == 3 ? A.fo_off[G_EMB > 7 ? 7 : 0] + sid[G_EMB > 7 ? 7 : 0] : ob;
                w1b = A.fo_all[ob];
            }
        }
    };

    // ---- compute stage operands (the gathered rows of the task being scored) ----
    f32x4 P[G_EMB][KPC];                                  // per-field projections (FOLD: the gathered rows)
    f32x4 pnum = zero;                                    // raw numerics of the task being scored
    float z1 = 0.f;                                       // first-order partial of this lane
    auto compute = [&]() -> float {
        // Every LDS read is issued at least one MFMA group (>= 128 matrix-pipe cycles) before its first
        // use, so a wave's MFMAs are back to back and the W^T fragment reads ride in their shadow.
        constexpr int NK = G * KPC;                           // 16-wide K chunks of deep0: chunk kk of group g = g*KPC+kk
        // numeric group first (its operand arrived with the ids): Dense projection (DeepFM_v2.py:118-120)
        f32x4 wn[KPC], pn[KPC];
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) {
            wn[nb] = ld4(wq + LD::off_wn + (nb * 16 + r) * LD::SN);
            pn[nb] = ld4(wq + LD::off_bp + G_EMB * KP + nb * 16);       // bias = initial accumulator
        }
        f32x4 h0[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = ld4(wq + LD::off_b0 + n0 * 16);
        const float* w0r = wq + LD::off_w0 + r * LD::S0;     // W0^T row (n0*16 + r), chunk c: w0r + n0*16*S0 + 16*c
        f32x4 a[H0C], an[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) a[n0] = ld4(w0r + n0 * 16 * LD::S0 + 16 * (G_EMB * KPC));
        const f32x4 wfn = ld4(smem + LD::off_fn + 4 * (q & 1));
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) pn[nb] = mfma4(wn[nb], pnum, pn[nb]);
        // first-order: categorical weights gathered by this lane + numeric Dense(1) partial (lanes q<2
        // hold real numerics; the packed fo_num weights are zero beyond n_num)
        float zz = z1 + ((q < 2) ? dot4(wfn, pnum) : 0.f);
        // FM cross (sum)^2 - sum(squares) (DeepFM_v2.py:147-152) and deep0 = Dense(relu) over the
        // flattened projections (DeepFM_v2.py:124-125), accumulated chunk by chunk: numerics, then fields
        f32x4 s[KPC], sq[KPC];
#pragma unroll
        for (int c = 0; c < NK; ++c) {
            const int cc = (c < KPC) ? G_EMB * KPC + c : c - KPC;     // chunk processed at step c
            const int cn = (c + 1 < KPC) ? G_EMB * KPC + c + 1 : c + 1 - KPC;
            if (c + 1 < NK) {
#pragma unroll
                for (int n0 = 0; n0 < H0C; ++n0) an[n0] = ld4(w0r + n0 * 16 * LD::S0 + 16 * cn);
            }
            const f32x4 p = (c < KPC) ? pn[c] : P[(c - KPC) / KPC][(c - KPC) % KPC];
            const int nb = cc % KPC;
            if (c < KPC) { s[nb] = p; sq[nb] = p * p; }
            else { s[nb] += p; sq[nb] += p * p; }
#pragma unroll
            for (int n0 = 0; n0 < H0C; n0 += 2) mfma4x2(a[n0], p, h0[n0], a[n0 + 1], p, h0[n0 + 1]);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) a[n0] = an[n0];
        }
        // deep1 operands, fetched under the last deep0 groups
        f32x4 w1f[H1C][H0C], acc1[H1C], hd[H1C], hfm[KPC];
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) {
            acc1[n1] = ld4(wq + LD::off_b1 + n1 * 16);
            hd[n1] = ld4(wq + LD::off_hd + n1 * 16);
#pragma unroll
            for (int j = 0; j < H0C; ++j) w1f[n1][j] = ld4(wq + LD::off_w1 + (n1 * 16 + r) * LD::S1 + 16 * j);
        }
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) hfm[nb] = ld4(wq + LD::off_hfm + nb * 16);
        float z = 0.f;
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) z += dot4(hfm[nb], s[nb] * s[nb] - sq[nb]);
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = relu4(h0[n0]);
        // deep1: Dense(relu) (DeepFM_v2.py:126) + output weights
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) {
#pragma unroll
            for (int j = 0; j < H0C; ++j) acc1[n1] = mfma4(w1f[n1][j], h0[j], acc1[n1]);
            z += dot4(hd[n1], relu4(acc1[n1]));
        }
        // output layer: concat([first, fm, deep]) . w + b -> sigmoid (DeepFM_v2.py:154-155)
        z += A.h0w * zz;
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        return sigmoidf_acc(z + A.h0w * A.fo_bias + A.head_bias);
    };

    // ---- REG: weight fragments held in registers (filled once, after the image barrier) ----
    constexpr int NKR = G * KPC;                              // deep0 K chunks
    f32x4 rW0[REG ? H0C : 1][REG ? NKR : 1], rW1[REG ? H1C : 1][REG ? H0C : 1];
    f32x4 rwn[REG ? KPC : 1], rbpn[REG ? KPC : 1], rb0[REG ? H0C : 1], rb1[REG ? H1C : 1], rhfm[REG ? KPC : 1], rhd[REG ? H1C : 1];
    f32x4 rfn = zero;
    auto load_weights = [&]() {
        if constexpr (REG) {
            const float* w0r = wq + LD::off_w0 + r * LD::S0;
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0)
#pragma unroll
                for (int c = 0; c < NKR; ++c) rW0[n0][c] = ld4(w0r + n0 * 16 * LD::S0 + 16 * c);
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
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) rb0[n0] = ld4(wq + LD::off_b0 + n0 * 16);
            rfn = ld4(smem + LD::off_fn + 4 * (q & 1));
        }
    };
    auto compute_reg = [&]() -> float {
        if constexpr (REG) {
            // numeric group's Dense projection (DeepFM_v2.py:118-120): two chains (even / odd K step)
            f32x4 pn[KPC];
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) {
                f32x4 e = rbpn[nb], o = zero;
                e = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[nb].x, pnum.x, e, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[nb].y, pnum.y, o, 0, 0, 0);
                e = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[nb].z, pnum.z, e, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn[nb].w, pnum.w, o, 0, 0, 0);
                pn[nb] = e + o;
            }
            // this lane's share of the per-id logit terms (row scalars: first order and -sum hfm P^2, see
            // k_v2_fold) + numeric first-order partial (rfn = h0w * fo_num weights, zero beyond n_num)
            float zz = z1 + ((q < 2) ? dot4(rfn, pnum) : 0.f);
            // deep0 (DeepFM_v2.py:124-125) over chunk pairs: accumulators hA (even position, starts at the
            // bias) and hB (odd position) per n-block = 2*H0C independent chains, issued round robin.
            // Every VALU instruction costs the f32 MFMA stream its issue time (they share the SIMD's
            // vector ALU), so the FM cross only accumulates S here.
            f32x4 hA[H0C], hB[H0C], s[KPC];
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) { hA[n0] = rb0[n0]; hB[n0] = zero; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NKR; c += 2) {
                // processing order: numeric chunks first, then the fields
                const int ca = (c < KPC) ? G_EMB * KPC + c : c - KPC;
                const int cb = (c + 1 < KPC) ? G_EMB * KPC + c + 1 : c + 1 - KPC;
                const bool hb = c + 1 < NKR;
                const f32x4 pa = (c < KPC) ? pn[c] : P[(c - KPC) / KPC][(c - KPC) % KPC];
                const f32x4 pb = !hb ? zero : (c + 1 < KPC) ? pn[c + 1] : P[(c + 1 - KPC) / KPC][(c + 1 - KPC) % KPC];
                {
                    const int na = ca % KPC, nbb = cb % KPC;
                    if (c < KPC) s[na] = pa; else s[na] += pa;
                    if (hb) { if (c + 1 < KPC) s[nbb] = pb; else s[nbb] += pb; }
                }
#pragma unroll
                for (int st = 0; st < 4; ++st) {
#pragma unroll
                    for (int n0 = 0; n0 < H0C; ++n0)
                        hA[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rW0[n0][ca][st], pa[st], hA[n0], 0, 0, 0);
                    if (hb) {
#pragma unroll
                        for (int n0 = 0; n0 < H0C; ++n0)
                            hB[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rW0[n0][hb ? cb : ca][st], pb[st], hB[n0], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            f32x4 h0[H0C];
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = relu4_fast(hA[n0] + hB[n0]);
            // FM cross (DeepFM_v2.py:147-152): sum_n hfm[n] (S_n^2 - sum_g P_g[n]^2); the fields' squares are in
            // the row scalars, the numeric group's are subtracted here
            float z = 0.f;
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb) z += dot4(rhfm[nb], s[nb] * s[nb] - pn[nb] * pn[nb]);
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
                z += dot4(rhd[n1], relu4_fast(e + o));
            }
            // output layer: concat([first, fm, deep]) . w + b -> sigmoid (DeepFM_v2.py:154-155)
            z += zz;
            z += __shfl_xor(z, 16);
            z += __shfl_xor(z, 32);
            return sigmoidf_fast(z + A.h0w * A.fo_bias + A.head_bias);
        } else {
            return 0.f;
        }
    };

    // ---- prologue: ids first, then the weight image (independent of the ids, so both are in flight
    //      together with the row gathers) ----
    if (A.flags & 32) return;                                 // experiment: launch cost only
    if (A.flags & 64) {                                       // experiment: static issue priorities per SIMD slot
        if (REG) { if (wave >> 2) __builtin_amdgcn_s_setprio(1); }
        else {
            const int gen = (wave >> 2) + 2 * ((blockIdx.x >> 8) & 1);
            if (gen == 1) __builtin_amdgcn_s_setprio(1);
            else if (gen == 2) __builtin_amdgcn_s_setprio(2);
            else if (gen == 3) __builtin_amdgcn_s_setprio(3);
        }
    }
    trace_stamp<TRACE>(A.trace, wave_global, 0, false);       // entry
    if (TRACE && A.trace && lane == 0) A.trace[(size_t)wave_global * 16 + 8] = wall_clock64();
    int cur = wave_global, prev = -1;
    if (cur < ntasks) ld_raw(cur);
    if (A.flags & 4) {                                        // experiment: wake the matrix pipe while the ids are in flight
        f32x4 acc = zero;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, (float)lane, acc, 0, 0, 0);
        asm volatile("" ::"v"(acc));
    }
    trace_stamp<TRACE>(A.trace, wave_global, 1, true);        // trace only: ids/numerics block landed
    bool first = true;
    for (;;) {
        const bool have_cur = cur < ntasks;                   // wave-uniform
        if (have_cur) {
            gather(cur);                                      // its ids arrived during the previous compute
            if (cur + task_stride < ntasks) ld_raw(cur + task_stride);
            if (prev == wave_global) trace_stamp<TRACE>(A.trace, wave_global, 11, false);   // second task's rows issued
        }
        if (first) {                                          // every wave of the workgroup passes here once
            trace_stamp<TRACE>(A.trace, wave_global, 2, false);   // rows issued
            // weight image -> LDS by LDS-DMA (no VGPRs, no ds_write pass): 1-KB pieces, wave w takes w, w+WAVES, ...
            // (issued after the row gathers so that the ids wait above did not have to drain it)
#pragma unroll 1
            for (int c = wave; c < LD::total_pad / 256; c += WAVES)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
                    (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
            trace_stamp<TRACE>(A.trace, wave_global, 3, false);   // image copy issued by this wave
            __syncthreads();                                      // (drains this wave's DMA and gathers first)
            trace_stamp<TRACE>(A.trace, wave_global, 4, false);   // weight image staged by the whole workgroup
            load_weights();                                       // REG: this wave's fragments -> VGPRs, once
            first = false;
        }
        // The score is stored one stage late, right after the hand-off wait below: vmcnt also counts
        // stores on gfx9, so a store issued straight after the scoring stage would sit in front of the
        // next gather's "ids landed" wait and expose its full write-acknowledge latency every task.
        float score = 0.f;
        if (prev >= 0) {
            if (A.flags & 8) {                                    // experiment: no compute stage, keep the gathered data live
                score = z1 + pnum.x;
#pragma unroll
                for (int g = 0; g < G_EMB; ++g) score += P[g][0].x + P[g][0].w;
            } else if (REG) {
                score = compute_reg();
            } else {
                score = compute();
            }
            if (prev == wave_global) trace_stamp<TRACE>(A.trace, wave_global, 6, false);                 // first task scored
            else if (prev == wave_global + task_stride) trace_stamp<TRACE>(A.trace, wave_global, 10, false);   // second
        }
        if (!have_cur) {
            const int m = prev * 16 + r;
            if (prev >= 0 && q == 0 && m < B) out[m] = score;
            break;
        }
        if (prev < 0) trace_stamp<TRACE>(A.trace, wave_global, 5, true);   // trace only: first task's rows landed
        // hand the gathered rows to the compute stage
        if (FOLD) {
#pragma unroll
            for (int g = 0; g < G_EMB; ++g)
#pragma unroll
                for (int nb = 0; nb < KPC; ++nb) P[g][nb] = x[g][nb < XC ? nb : 0];
        } else {
            // per-field Dense projections (DeepFM_v2.py:106-117): accumulators start at the bias
#pragma unroll
            for (int nb = 0; nb < KPC; ++nb)
#pragma unroll
                for (int g = 0; g < G_EMB; ++g) {
                    f32x4 acc = ld4(wq + LD::off_bp + g * KP + nb * 16);
#pragma unroll
                    for (int c = 0; c < XC; ++c)
                        acc = mfma4(ld4(wq + LD::off_wp + (g * KP + nb * 16 + r) * LD::SP + 16 * c), x[g][c], acc);
                    P[g][nb] = acc;
                }
        }
        {
            const int m = prev * 16 + r;
            if (prev >= 0 && q == 0 && m < B) out[m] = score;
        }
        pnum = xn;
        z1 = ((q < G_EMB) ? w1a : 0.f) + ((q + 4 < G_EMB) ? w1b : 0.f);
        prev = cur;
        cur += task_stride;
    }
    trace_stamp<TRACE>(A.trace, wave_global, 7, true);
    if (TRACE && A.trace && lane == 0) A.trace[(size_t)wave_global * 16 + 9] = wall_clock64();
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

// k_din_tail.h -- register-chained DIN tail (reference DIN.py:161-167): concat([user profile, pooled history,
// candidate, context]) -> Dense(128) PReLU -> Dense(64) PReLU -> Dense(1, sigmoid), one WAVE per 16 samples.
// Included inside sparrow_hip.hip's anonymous namespace, after k_din_attn.h.
//
// The first Dense is linear in the concat, so every embedding column's share is a table of its own,
// F_g[id] = W_g^T E_g[id] (fold_first_dense builds them at sprk_finalize): per sample the layer is
//     z0 = b0 + F_user[userId] + F_cand[movieId] + F_ug[userGenre1] + F_mg[movieGenre1]     four row gathers
//          + Wp^T pooled + Wn^T numerics                                                     K = D + 8 on the matrix pipe
// Lane (r = lane&15, q = lane>>4) holds z0[n = 16 nb + 4q + j] of sample r for every 16-wide block nb: the
// C/D layout of v_mfma_f32_16x16x4_f32, which is also the B-operand layout of the next layer, so the folded
// rows are gathered straight into the accumulators (16-byte pieces), fc0's remaining contraction, PReLU,
// fc1 (K = 128: its B operand IS h1's registers), PReLU, the output dot and the sigmoid all stay in
// registers.  Weights sit in LDS (pre-packed image, LDS-DMA), read as A fragments right before use.
// fc0's numerics (K = 8: two steps, k = q + 4 s) run on f32 MFMA; fc1 and, since round 3, fc0's pooled-history columns run on the
// f16 matrix pipe with a per-sample DYNAMIC power-of-two scale of their input (dyn_split.h): hidden activations and DIN's pooled
// history have data-dependent range.  Round 3 also added UNF (DinTailRun::Etab below): for emb_dim <= 16 the embedding columns are
// NOT folded -- raw split-f16 rows, two K = 32 blocks on the matrix pipe, sixteen waves per workgroup.
// The interpreter ran this tail as a chain of memory round trips at one tile per workgroup (28 us at
// B = 32 768); here every load of a task is in flight at once.

#define DT_MAX_COLS 4

struct DinTailRun {
    int F, ND, NA;                        // ids / dense / aux (pooled) row widths
    int n_cols;                           // folded embedding columns (<= DT_MAX_COLS)
    int col[DT_MAX_COLS];                 // ids column
    int vocab[DT_MAX_COLS];
    const float* Ftab[DT_MAX_COLS];       // [vocab][N0] folded rows
    int n_num;                            // numerics used (<= 8)
    float head_bias;
    float inv_w1_scale;                   // DYN: 1 / (static power-of-two scale of fc1's split weights)
    float inv_w0p_scale;                  // DYN: 1 / (scale of fc0's pooled-history columns as split-f16 fragments); 0 = f32 MFMA
    // UNF ([r3], emb_dim <= 16): the embedding columns are NOT folded through fc0 -- a folded row is N0 = 128 floats for an embedding
    // of 10, and one task gathered 32 pieces of 16 bytes per lane in two dependent rounds (k_din_tail took 21 us per 65 536 samples
    // of DIN.py's shape, twice the attention stage).  Etab[g] holds column g's RAW rows, 16 padded values pre-split into f16
    // hi | lo with a static scale (64 bytes per id, an all-zero row at index vocab); columns (0, 1) and (2, 3) are the two K = 32
    // operands of fc0's embedding part: lane (r, q) gathers 16 + 16 bytes of column 2 pb + (q >> 1), half q & 1, per block.
    const _Float16* Etab[DT_MAX_COLS];
    float e_unscale;                      // 1 / (row scale * fragment scale); 0 = folded rows
};

// Multi-batch launch (sprk_set_many_batches): launch task t is task t % ntpb of batch t / ntpb, every batch with its own
// buffers -- one LDS image load and one dispatch for up to DIN_MB batches, and waves that run more than one task.
#define DIN_MB 16
struct DinTailMany {
    const int* ids[DIN_MB];
    const float* dense[DIN_MB];
    const float* aux[DIN_MB];
    float* out[DIN_MB];
    int n, ntpb;
};

// DYN: fc1 on the f16 matrix pipe with per-sample dynamic scaling (dyn_split.h); its W1 region then holds the packed
// hi / lo A fragments ((N1/16) x (N0/32) blocks of 2 KB) instead of the f32 W^T rows.
template <int N0C, int N1C, int KPC>
struct DinTailLds {
    static constexpr int N0 = N0C * 16, N1 = N1C * 16;
    static constexpr int w1h_floats = N1C * (N0C / 2) * 512;       // DYN fragments, in floats
    static constexpr int K0 = KPC * 16 + 16;          // fc0's per-sample K: pooled chunks + one numeric chunk (8 used)
    static constexpr int S0 = K0 + 4;                 // W0^T row stride
    static constexpr int S1 = N0 + 4;                 // W1^T row stride
    static constexpr int off_w0 = 0;                  // [N0][S0]
    static constexpr int off_w1 = off_w0 + N0 * S0;   // [N1][S1]
    static constexpr int off_b0 = off_w1 + (N1 * S1 > w1h_floats ? N1 * S1 : w1h_floats);   // [N0]
    static constexpr int off_a0 = off_b0 + N0;        // [N0]
    static constexpr int off_b1 = off_a0 + N0;        // [N1]
    static constexpr int off_a1 = off_b1 + N1;        // [N1]
    static constexpr int off_hw = off_a1 + N1;        // [N1]
    // [r3] fc0's pooled-history columns as split-f16 fragments (N0C x ceil(KPC / 2) blocks of 2 KB): v_mfma_f32_16x16x4_f32 takes 32
    // cycles and holds up the SIMD's VALU issue while it runs (measured on the DIEN stage, k_dien_mfma.h); the pooled chunk(s) cost
    // 4 * KPC * N0C of them per task, the f16 form 3 * N0C * ceil(KPC / 2) of 16 cycles
    static constexpr int KB0 = (KPC + 1) / 2;
    static constexpr int w0h_floats = N0C * KB0 * 512;
    static constexpr int off_w0h = (off_hw + N1 + 3) & ~3;
    // UNF: fc0's embedding columns as {hi, lo} fragments.  emb_dim <= 16 (KPC = 1): two K = 32 blocks of two columns each, rows of
    // 16 + 16 halfs; emb_dim <= 32 (KPC = 2, BASELINE config 3): one block per column, rows of 32 + 32 halfs
    static constexpr int EPB = KPC >= 2 ? 32 : 16;
    static constexpr int NBLK = KPC >= 2 ? 4 : 2;
    static constexpr int w0e_floats = N0C * NBLK * 512;
    static constexpr int off_w0e = off_w0h + w0h_floats;
    static constexpr int total = off_w0e + w0e_floats;
    static constexpr int total_pad = (total + 255) & ~255;
    static constexpr size_t bytes = sizeof(float) * total_pad;
};

// One-time (finalize) kernel: the LDS image.  W0: the first Dense's W^T [N0][ldw0] (folded columns zeroed),
// pooled columns at p_off, numerics at n_off; W1: second Dense's W^T [N1][ldw1].
template <int N0C, int N1C, int KPC>
__global__ __launch_bounds__(256) void k_din_tail_pack(const float* __restrict__ W0, int ldw0, int p_off, int Dp, int n_off,
                                                       int n_num, const float* __restrict__ b0, const float* __restrict__ a0,
                                                       const float* __restrict__ W1, int ldw1, const float* __restrict__ b1,
                                                       const float* __restrict__ a1, const float* __restrict__ hw, int n_hw,
                                                       const float* __restrict__ w1frag, float* __restrict__ img,
                                                       const float* __restrict__ w0pfrag, const float* __restrict__ w0efrag) {
    using LD = DinTailLds<N0C, N1C, KPC>;
    const int tid = threadIdx.x;
    for (int i = tid; i < LD::w0h_floats; i += 256) img[LD::off_w0h + i] = w0pfrag ? w0pfrag[i] : 0.f;
    for (int i = tid; i < LD::w0e_floats; i += 256) img[LD::off_w0e + i] = w0efrag ? w0efrag[i] : 0.f;
    for (int i = tid; i < LD::N0 * LD::S0; i += 256) {
        const int n = i / LD::S0, k = i - n * LD::S0;
        float v = 0.f;
        if (k < KPC * 16) { if (k < Dp) v = W0[(size_t)n * ldw0 + p_off + k]; }
        else if (k - KPC * 16 < n_num) v = W0[(size_t)n * ldw0 + n_off + (k - KPC * 16)];
        img[LD::off_w0 + i] = v;
    }
    if (w1frag) {
        for (int i = tid; i < LD::w1h_floats; i += 256) img[LD::off_w1 + i] = w1frag[i];     // DYN: packed f16 fragments
    } else {
        for (int i = tid; i < LD::N1 * LD::S1; i += 256) {
            const int n = i / LD::S1, k = i - n * LD::S1;
            img[LD::off_w1 + i] = k < LD::N0 ? W1[(size_t)n * ldw1 + k] : 0.f;
        }
    }
    for (int i = tid; i < LD::N0; i += 256) { img[LD::off_b0 + i] = b0[i]; img[LD::off_a0 + i] = a0[i]; }
    for (int i = tid; i < LD::N1; i += 256) {
        img[LD::off_b1 + i] = b1[i];
        img[LD::off_a1 + i] = a1[i];
        img[LD::off_hw + i] = i < n_hw ? hw[i] : 0.f;
    }
    for (int i = LD::total + tid; i < LD::total_pad; i += 256) img[i] = 0.f;
}

// ---- the pieces of one task, shared by k_din_tail and the kernels that carry this tail as their epilogue (k_dien_fused.h) ----
// S = the tail image in LDS; lane (r, q) as above.

// UNF: the raw split rows of the embedding columns, one 16 + 16 byte piece per K = 32 block and lane
template <class LD>
__device__ __forceinline__ void din_tail_unf_gather(const DinTailRun& A, const int (&idv)[DT_MAX_COLS], int q, bool& bad,
                                                    din_f16x8 (&eh)[LD::NBLK], din_f16x8 (&el)[LD::NBLK]) {
    constexpr int NBLK = LD::NBLK;
#pragma unroll
    for (int pb = 0; pb < NBLK; ++pb) {
        // (static indices only: a lane-dependent index into the kernel argument's arrays would move them to scratch)
        const bool up = LD::EPB == 16 && (q >> 1) != 0;
        const int g0 = LD::EPB == 16 ? 2 * pb : pb, g1 = LD::EPB == 16 ? 2 * pb + 1 : pb;
        const int id = up ? idv[g1] : idv[g0];
        const bool have = (up ? g1 : g0) < A.n_cols;
        int voc = up ? A.vocab[g1] : A.vocab[g0];
        const _Float16* tab = up ? A.Etab[g1] : A.Etab[g0];
        if (!have) { voc = A.vocab[0]; tab = A.Etab[0]; }           // an absent column: column 0's all-zero row
        const bool ok = have && (unsigned)id < (unsigned)voc;
        bad |= have && !ok && id != -1;
        const char* row = reinterpret_cast<const char*>(tab) + (size_t)(ok ? id : voc) * (4 * LD::EPB) +
                          16 * (LD::EPB == 16 ? (q & 1) : q);
        eh[pb] = *reinterpret_cast<const din_f16x8*>(row);           // (a missing id / an absent column: the all-zero row)
        el[pb] = *reinterpret_cast<const din_f16x8*>(row + 2 * LD::EPB);
    }
}

// UNF: fc0's embedding part, z0 += W0e^T [rows] on the f16 matrix pipe
template <class LD, int N0C>
__device__ __forceinline__ void din_tail_unf_fc0(const DinTailRun& A, const float* S, int lane, const din_f16x8 (&eh)[LD::NBLK],
                                                 const din_f16x8 (&el)[LD::NBLK], f32x4 (&z0)[N0C]) {
    constexpr int NBLK = LD::NBLK;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wf = S + LD::off_w0e + 4 * lane;
#pragma unroll
    for (int pb = 0; pb < NBLK; ++pb)
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) {
            const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(wf + ((nb * NBLK + pb) * 2 + 0) * 256));
            const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(wf + ((nb * NBLK + pb) * 2 + 1) * 256));
            f32x4 acc = mfma_f16(al, eh[pb], zero);
            acc = mfma_f16(ah, el[pb], acc);
            acc = mfma_f16(ah, eh[pb], acc);
            z0[nb] += acc * A.e_unscale;
        }
}

// From fc0's accumulators (bias + embedding columns) to the pre-sigmoid score of sample r (every lane of the sample's four holds it):
// fc0's per-sample part (numerics, pooled history xp), PReLU, fc1, PReLU, the output dot.
template <int N0C, int N1C, int KPC, bool DYN, bool UNF>
__device__ __forceinline__ float din_tail_dense(const DinTailRun& A, const float* S, f32x4 (&z0)[N0C], const f32x4 (&xp)[KPC], float xna,
                                                float xnb, int lane, int r, int q) {
    using LD = DinTailLds<N0C, N1C, KPC>;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- fc0's per-sample part: pooled chunks, then the numeric chunk (N0C independent chains) ----
    const float* w0r = S + LD::off_w0 + r * LD::S0 + 4 * q;
    {
        // the numeric chunk: A = W0^T[n][numeric q + 4 s] (two scalar LDS reads per block), two MFMAs per 16 outputs
        const float* wn = S + LD::off_w0 + r * LD::S0 + KPC * 16 + q;
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[nb * 16 * LD::S0], xna, z0[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wn[nb * 16 * LD::S0 + 4], xnb, z0[nb], 0, 0, 0);
    }
    if (UNF || (DYN && A.inv_w0p_scale != 0.f)) {             // (wave-uniform; UNF is only set up together with the pooled fragments)
        // pooled history on the f16 pipe: per-sample dynamic scale (DIN's attention weights are not normalised), hi / lo split
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c < KPC; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, __builtin_fabsf(xp[c][j]));
        mx = rows4_max(mx);
        float scale, inv;
        dyn_scale(mx, A.inv_w0p_scale, scale, inv);
        const float* wf = S + LD::off_w0h + lane * 4;
#pragma unroll
        for (int b = 0; b < LD::KB0; ++b) {
            din_f16x8 bh, bl;
            dyn_split8(xp[2 * b], 2 * b + 1 < KPC ? xp[2 * b + 1 < KPC ? 2 * b + 1 : 0] : zero, scale, bh, bl);
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb) {
                const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(wf + ((nb * LD::KB0 + b) * 2 + 0) * 256));
                const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(wf + ((nb * LD::KB0 + b) * 2 + 1) * 256));
                f32x4 acc = mfma_f16(al, bh, zero);
                acc = mfma_f16(ah, bl, acc);
                acc = mfma_f16(ah, bh, acc);
                z0[nb] += acc * inv;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < KPC; ++c) {
            const f32x4 b = xp[c];
            f32x4 a[N0C];
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb) a[nb] = ld4(w0r + nb * 16 * LD::S0 + 16 * c);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb)
                    z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb][st], b[st], z0[nb], 0, 0, 0);
        }
    }
    // PReLU(alpha0) (DIN.py:164)
#pragma unroll
    for (int nb = 0; nb < N0C; ++nb) {
        const f32x4 al = ld4(S + LD::off_a0 + nb * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = z0[nb][j];
            z0[nb][j] = __builtin_amdgcn_fmed3f(u, 0.f, __builtin_inff()) + al[j] * __builtin_amdgcn_fmed3f(u, -__builtin_inff(), 0.f);
        }
    }
    // ---- fc1: K = N0, its B operand is h1 as it sits in the registers (N1C chains) ----
    f32x4 z1[N1C];
    if constexpr (DYN) {
        static_assert(N0C % 2 == 0, "K blocks of 32");
        // per-sample scale from max |h1| (this lane's 4*N0C values, then the sample's other three q rows)
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
        const float* wf = S + LD::off_w1 + lane * 4;              // this lane's 16 bytes inside a 1-KB fragment (k_dyn_pack_w: lane order)
#pragma unroll
        for (int b = 0; b < N0C / 2; ++b) {
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
        for (int n1 = 0; n1 < N1C; ++n1) z1[n1] = acc[n1] * inv + ld4(S + LD::off_b1 + n1 * 16 + 4 * q);
    } else {
#pragma unroll
    for (int n1 = 0; n1 < N1C; ++n1) z1[n1] = ld4(S + LD::off_b1 + n1 * 16 + 4 * q);
    const float* w1r = S + LD::off_w1 + r * LD::S1 + 4 * q;
#pragma unroll
    for (int c = 0; c < N0C; ++c) {
        f32x4 a[N1C];
#pragma unroll
        for (int n1 = 0; n1 < N1C; ++n1) a[n1] = ld4(w1r + n1 * 16 * LD::S1 + 16 * c);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int n1 = 0; n1 < N1C; ++n1)
                z1[n1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n1][st], z0[c][st], z1[n1], 0, 0, 0);
    }
    }
    // PReLU(alpha1) (DIN.py:166) -> Dense(1) -> sigmoid (DIN.py:167)
    float z = 0.f;
#pragma unroll
    for (int n1 = 0; n1 < N1C; ++n1) {
        const f32x4 al = ld4(S + LD::off_a1 + n1 * 16 + 4 * q);
        const f32x4 hw = ld4(S + LD::off_hw + n1 * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = z1[n1][j];
            const float h2 = __builtin_amdgcn_fmed3f(u, 0.f, __builtin_inff()) + al[j] * __builtin_amdgcn_fmed3f(u, -__builtin_inff(), 0.f);
            z = fmaf(hw[j], h2, z);
        }
    }
    return rows4_sum(z);
}

template <int N0C, int N1C, int KPC, int WAVES, bool DYN, bool MB = false, bool UNF = false>
__global__ __launch_bounds__(WAVES * 64, WAVES >= 16 ? 4 : 2) void k_din_tail(const DinTailRun A, const int* __restrict__ ids0,
                                                            const float* __restrict__ dense0, const float* __restrict__ aux0,
                                                            float* __restrict__ out0, int B, int* __restrict__ err,
                                                            const float* __restrict__ image, const DinTailMany M) {
    using LD = DinTailLds<N0C, N1C, KPC>;
    constexpr int N0 = LD::N0;
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

    // weight image -> LDS (1-KB LDS-DMA pieces), while the first task's ids are on their way
    int tk = blockIdx.x * WAVES + wave;
    int idv[DT_MAX_COLS];
    auto ld_ids = [&](int tg) {
        int t;
        const int bi = batch_of(tg, t);
        const int* ids = MB ? M.ids[bi] : ids0;
        const int m = min(t * 16 + r, B - 1);                    // rows past the end re-read the last sample, never stored
        const int* row = ids + (size_t)m * A.F;
#pragma unroll
        for (int g = 0; g < DT_MAX_COLS; ++g) idv[g] = g < A.n_cols ? row[A.col[g]] : -1;
    };
    if (tk < ntasks) ld_ids(tk);
#pragma unroll 1
    for (int c = wave; c < LD::total_pad / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
    __syncthreads();

    for (; tk < ntasks; tk += task_stride) {
        int tl;
        const int bi = batch_of(tk, tl);
        const float* dense = MB ? M.dense[bi] : dense0;
        const float* aux = MB ? M.aux[bi] : aux0;
        float* out = MB ? M.out[bi] : out0;
        const int m = min(tl * 16 + r, B - 1);
        // ---- per-sample operands: pooled history (aux) and numerics, in B-operand layout ----
        f32x4 xp[KPC];
        float xna, xnb;                                           // numerics q and q + 4 of this sample: K = 8 is two steps, k = q + 4 s
#pragma unroll
        for (int c = 0; c < KPC; ++c) xp[c] = (16 * c + 4 * q < A.NA) ? ld4(aux + (size_t)m * A.NA + 16 * c + 4 * q) : zero;
        {
            const float* nrow = dense + (size_t)m * A.ND;
            const int last = A.n_num - 1;
            // slots beyond n_num hold a duplicate finite value that only ever meets zero weights
            xna = nrow[min(q, last)];
            xnb = nrow[min(q + 4, last)];
        }
        // ---- folded embedding columns gathered straight into fc0's accumulators ----
        f32x4 z0[N0C];
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = ld4(smem + LD::off_b0 + nb * 16 + 4 * q);
        if constexpr (UNF) {
            static_assert(DYN && DT_MAX_COLS == 4, "two K = 32 blocks of two columns, or four of one");
            din_f16x8 eh[LD::NBLK], el[LD::NBLK];
            din_tail_unf_gather<LD>(A, idv, q, bad, eh, el);
            din_tail_unf_fc0<LD, N0C>(A, smem, lane, eh, el, z0);
        } else {
#pragma unroll
        for (int g0 = 0; g0 < DT_MAX_COLS; g0 += 2) {             // two columns = 2*N0C loads in flight at a time
            f32x4 f[2][N0C];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int id = idv[g0 + g];
                const bool ok = g0 + g < A.n_cols && (unsigned)id < (unsigned)A.vocab[g0 + g];
                bad |= g0 + g < A.n_cols && !ok && id != -1;
                const float* frow = A.Ftab[g0 + g] + (size_t)(ok ? id : 0) * N0 + 4 * q;
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) f[g][nb] = ok ? ld4(frow + nb * 16) : zero;
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) z0[nb] += f[g][nb];
        }
        }
        if (tk + task_stride < ntasks) ld_ids(tk + task_stride);   // next task's ids fly under this task's MFMAs

        const float z = din_tail_dense<N0C, N1C, KPC, DYN, UNF>(A, smem, z0, xp, xna, xnb, lane, r, q);
        const int mm = tl * 16 + r;
        if (q == 0 && mm < B) out[mm] = sigmoidf_acc(z + A.head_bias);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

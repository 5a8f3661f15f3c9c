// k_mlp_chain.h -- register-chained forward for the "DenseFeatures -> Dense -> Dense -> Dense(1, sigmoid)" graphs
// (reference EmbeddingMLP.py:72-77 and the deep part of WideNDeep.py:99-107, plus its hashed-cross wide part;
// BASELINE config 5 is the latter at emb_dim 32 with a 10 M-bucket cross table), one WAVE per 16 samples.
// Included inside sparrow_hip.hip's anonymous namespace, after k_chain_v1.h.
//
// Same construction as k_din_tail (k_din_tail.h): lane (r = lane&15, q = lane>>4) is sample r's q-th 16-byte slot;
//   z0 = b0 + sum over FOLDED embedding columns of F_g[id_g]        per-id tables of the first layer's outputs
//           (fold_first_dense; the 19-row genre columns: their 512-byte rows stay in L2)
//        + W^T [rows of the unfolded columns | numerics]             16-wide K chunks on f32 MFMA, the gathered row pieces
//                                                                    are the B operands as they arrive
//   h1 = act(z0);  z1 = b1 + W1^T h1 (K = N0, B operand = h1's registers);  h2 = act(z1)
//   score = sigmoid(hw . h2 + wide term + bias),  wide term = hc . table_x[FingerprintCat64(a, b) mod buckets]  (or its
//   indicator weight): the hash runs on the integer VALU, one 16-byte piece of the cross row per lane.
// Weights of both layers sit in LDS (pre-packed image, LDS-DMA).  f32 MFMA throughout (operands with data-dependent range).
// The tile interpreter ran config 5's shape in 305 us per 131 072 samples, a chain of gather round trips.

#define MC_MAX_ACC 8
#define MC_MAX_CHUNKS 6

struct MlpChainRun {
    int F, ND;
    int n_acc;                            // folded columns
    int acc_col[MC_MAX_ACC], acc_vocab[MC_MAX_ACC];
    const float* acc_tab[MC_MAX_ACC];     // [vocab][N0]
    int n_chunks;                         // 16-wide K chunks fed per sample (the last one is the numerics chunk when n_num > 0)
    int ch_col[MC_MAX_CHUNKS];            // ids column of the chunk's embedding column (-1: numerics chunk)
    int ch_vocab[MC_MAX_CHUNKS];
    int ch_off[MC_MAX_CHUNKS];            // float offset of the chunk inside the table row
    int ch_stride[MC_MAX_CHUNKS];         // floats per table row
    int ch_width[MC_MAX_CHUNKS];          // valid floats in this chunk (row width - ch_off, capped at 16)
    const float* ch_tab[MC_MAX_CHUNKS];   // [vocab+1][stride] (zero row at index vocab)
    int n_num;
    // wide part: 0 = none, 1 = cross rows (embedding of the hashed cross) x head weights, 2 = cross scalar (indicator weight)
    int wide_kind, wide_a, wide_b, wide_dim, wide_stride;
    long long wide_buckets;
    const float* wide_tab;
    const float* wide_w;                  // [wide_dim] head weights (kind 1)
    float head_bias;
    float inv_w1_scale;                   // DYN: 1 / static scale of the second layer's split-f16 fragments (0 = f32 MFMA)
};

template <int N0C, int N1C>
struct MlpChainLds {
    static constexpr int N0 = N0C * 16, N1 = N1C * 16;
    static constexpr int K0 = MC_MAX_CHUNKS * 16;
    static constexpr int S0 = K0 + 4, S1 = N0 + 4;
    static constexpr int off_w0 = 0;                  // [N0][S0]
    static constexpr int off_w1 = off_w0 + N0 * S0;   // [N1][S1]
    static constexpr int off_b0 = off_w1 + N1 * S1;
    static constexpr int off_a0 = off_b0 + N0;        // PReLU alpha (zeros for ReLU)
    static constexpr int off_b1 = off_a0 + N0;
    static constexpr int off_a1 = off_b1 + N1;
    static constexpr int off_hw = off_a1 + N1;
    static constexpr int total = off_hw + N1;
    static constexpr int total_pad = (total + 255) & ~255;
    static constexpr size_t bytes = sizeof(float) * total_pad;
};

// One-time (finalize) kernel: the LDS image.  col_off[c] = position of chunk c's first column inside W0's row (the layer's
// K range after the fold), col_w[c] = valid columns of the chunk.
template <int N0C, int N1C>
__global__ __launch_bounds__(256) void k_mlp_chain_pack(const float* __restrict__ W0, int ldw0, int n_chunks,
                                                        const int* __restrict__ col_off, const int* __restrict__ col_w,
                                                        const float* __restrict__ b0, const float* __restrict__ a0,
                                                        const float* __restrict__ W1, int ldw1, const float* __restrict__ b1,
                                                        const float* __restrict__ a1, const float* __restrict__ hw, int n_hw,
                                                        const float* __restrict__ w1frag, float* __restrict__ img) {
    using LD = MlpChainLds<N0C, N1C>;
    static_assert(N1C * (N0C / 2) * 512 <= LD::N1 * LD::S1 && N0C % 2 == 0, "DYN fragments fit the second layer's region");
    const int tid = threadIdx.x;
    for (int i = tid; i < LD::N0 * LD::S0; i += 256) {
        const int n = i / LD::S0, k = i - n * LD::S0, c = k >> 4, j = k & 15;
        float v = 0.f;
        if (k < LD::K0 && c < n_chunks && j < col_w[c]) v = W0[(size_t)n * ldw0 + col_off[c] + j];
        img[LD::off_w0 + i] = v;
    }
    if (w1frag) {                                     // DYN: split-f16 A fragments (dyn_split.h) instead of the f32 W^T rows
        for (int i = tid; i < LD::N1 * LD::S1; i += 256) img[LD::off_w1 + i] = i < N1C * (N0C / 2) * 512 ? w1frag[i] : 0.f;
    } else {
        for (int i = tid; i < LD::N1 * LD::S1; i += 256) {
            const int n = i / LD::S1, k = i - n * LD::S1;
            img[LD::off_w1 + i] = k < LD::N0 ? W1[(size_t)n * ldw1 + k] : 0.f;
        }
    }
    for (int i = tid; i < LD::N0; i += 256) { img[LD::off_b0 + i] = b0[i]; img[LD::off_a0 + i] = a0 ? a0[i] : 0.f; }
    for (int i = tid; i < LD::N1; i += 256) {
        img[LD::off_b1 + i] = b1[i];
        img[LD::off_a1 + i] = a1 ? a1[i] : 0.f;
        img[LD::off_hw + i] = i < n_hw ? hw[i] : 0.f;
    }
    for (int i = LD::total + tid; i < LD::total_pad; i += 256) img[i] = 0.f;
}

template <int N0C, int N1C, int WAVES, bool DYN>
__global__ __launch_bounds__(WAVES * 64, 2) void k_mlp_chain(const MlpChainRun A, const int* __restrict__ ids,
                                                             const float* __restrict__ dense, float* __restrict__ out,
                                                             int B, int* __restrict__ err, const float* __restrict__ image) {
    using LD = MlpChainLds<N0C, N1C>;
    constexpr int N0 = LD::N0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntasks = (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    bool bad = false;

#pragma unroll 1
    for (int c = wave; c < LD::total_pad / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
    __syncthreads();

    for (int tk = blockIdx.x * WAVES + wave; tk < ntasks; tk += task_stride) {
        const int m = min(tk * 16 + r, B - 1);                   // rows past the end re-read the last sample, never stored
        const int* idrow = ids + (size_t)m * A.F;
        // ---- per-sample B operands: row pieces of the unfolded columns, numerics ----
        f32x4 xb[MC_MAX_CHUNKS];
#pragma unroll
        for (int c = 0; c < MC_MAX_CHUNKS; ++c) {
            xb[c] = zero;
            if (c < A.n_chunks) {                                 // wave-uniform
                if (A.ch_col[c] >= 0) {
                    const int id = idrow[A.ch_col[c]];
                    bad |= (unsigned)(id + 1) > (unsigned)A.ch_vocab[c];
                    const unsigned sid = min((unsigned)id, (unsigned)A.ch_vocab[c]);      // -1 -> the zero row at index vocab
                    if (4 * q < A.ch_width[c]) xb[c] = ld4(A.ch_tab[c] + (size_t)sid * A.ch_stride[c] + A.ch_off[c] + 4 * q);
                } else {
                    const float* nrow = dense + (size_t)m * A.ND;
                    const int last = A.n_num - 1;
                    // slots beyond n_num hold a duplicate finite value that only ever meets zero weights
                    xb[c].x = nrow[min(4 * q + 0, last)];
                    xb[c].y = nrow[min(4 * q + 1, last)];
                    xb[c].z = nrow[min(4 * q + 2, last)];
                    xb[c].w = nrow[min(4 * q + 3, last)];
                }
            }
        }
        // ---- wide part: hashed cross -> one 16-byte piece of its row per lane (q, q+4 when the row is wider than 64 B) ----
        float zw = 0.f;
        if (A.wide_kind) {                                        // wave-uniform
            const unsigned long long bkt = cross_bucket(idrow[A.wide_a], idrow[A.wide_b], (uint64_t)A.wide_buckets);
            if (A.wide_kind == 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = 16 * h + 4 * q;
                    if (d < A.wide_dim) zw += dot4(ld4(A.wide_tab + (size_t)bkt * A.wide_stride + d), ld4(A.wide_w + d));
                }
            } else if (q == 0) {
                zw = A.wide_tab[bkt];
            }
        }
        // ---- folded columns gathered straight into the first layer's accumulators ----
        f32x4 z0[N0C];
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = ld4(smem + LD::off_b0 + nb * 16 + 4 * q);
#pragma unroll
        for (int g0 = 0; g0 < MC_MAX_ACC; g0 += 2) {              // two columns = 2*N0C loads in flight at a time
            if (g0 < A.n_acc) {                                   // wave-uniform (no break: keeps the loop fully unrolled)
            f32x4 f[2][N0C];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const bool on = g0 + g < A.n_acc;
                const int id = on ? idrow[A.acc_col[g0 + g]] : -1;
                const bool ok = on && (unsigned)id < (unsigned)A.acc_vocab[g0 + g];
                bad |= on && !ok && id != -1;
                const float* frow = A.acc_tab[on ? g0 + g : 0] + (size_t)(ok ? id : 0) * N0 + 4 * q;
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) f[g][nb] = ok ? ld4(frow + nb * 16) : zero;
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) z0[nb] += f[g][nb];
            }
        }
        // ---- first layer's per-sample part (N0C independent chains) ----
        const float* w0r = smem + LD::off_w0 + r * LD::S0 + 4 * q;
#pragma unroll
        for (int c = 0; c < MC_MAX_CHUNKS; ++c) {
            if (c < A.n_chunks) {                                 // wave-uniform
                f32x4 a[N0C];
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) a[nb] = ld4(w0r + nb * 16 * LD::S0 + 16 * c);
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int nb = 0; nb < N0C; ++nb)
                        z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb][st], xb[c][st], z0[nb], 0, 0, 0);
            }
        }
        // activation: PReLU with the layer's alpha (zeros = ReLU)
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) {
            const f32x4 al = ld4(smem + LD::off_a0 + nb * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = z0[nb][j];
                z0[nb][j] = __builtin_amdgcn_fmed3f(u, 0.f, __builtin_inff()) + al[j] * __builtin_amdgcn_fmed3f(u, -__builtin_inff(), 0.f);
            }
        }
        // ---- second layer: K = N0, B operand = h1 as it sits in the registers (N1C chains) ----
        f32x4 z1[N1C];
        if constexpr (DYN) {
            // f16 matrix pipe with a per-sample power-of-two scale of the hidden activations (dyn_split.h; k_din_tail's fc1)
            float mx = 0.f;
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, __builtin_fabsf(z0[nb][j]));
            mx = rows4_max(mx);
            float scale, inv;
            dyn_scale(mx, A.inv_w1_scale, scale, inv);
#pragma unroll
            for (int n1 = 0; n1 < N1C; ++n1) z1[n1] = zero;
            const float* wf = smem + LD::off_w1 + (r * 4 + q) * 4;       // this lane's 16 bytes inside a 1-KB fragment
#pragma unroll
            for (int b = 0; b < N0C / 2; ++b) {
                din_f16x8 bh, bl;
                dyn_split8(z0[2 * b], z0[2 * b + 1], scale, bh, bl);
#pragma unroll
                for (int n1 = 0; n1 < N1C; ++n1) {
                    const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(wf + ((n1 * (N0C / 2) + b) * 2 + 0) * 256));
                    const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(wf + ((n1 * (N0C / 2) + b) * 2 + 1) * 256));
                    z1[n1] = mfma_f16(ah, bh, z1[n1]);
                    z1[n1] = mfma_f16(ah, bl, z1[n1]);
                    z1[n1] = mfma_f16(al, bh, z1[n1]);
                }
            }
#pragma unroll
            for (int n1 = 0; n1 < N1C; ++n1) z1[n1] = z1[n1] * inv + ld4(smem + LD::off_b1 + n1 * 16 + 4 * q);
        } else {
#pragma unroll
        for (int n1 = 0; n1 < N1C; ++n1) z1[n1] = ld4(smem + LD::off_b1 + n1 * 16 + 4 * q);
        const float* w1r = smem + LD::off_w1 + r * LD::S1 + 4 * q;
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
        float z = zw;
#pragma unroll
        for (int n1 = 0; n1 < N1C; ++n1) {
            const f32x4 al = ld4(smem + LD::off_a1 + n1 * 16 + 4 * q);
            const f32x4 hw = ld4(smem + LD::off_hw + n1 * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = z1[n1][j];
                const float h2 = __builtin_amdgcn_fmed3f(u, 0.f, __builtin_inff()) + al[j] * __builtin_amdgcn_fmed3f(u, -__builtin_inff(), 0.f);
                z = fmaf(hw[j], h2, z);
            }
        }
        z = rows4_sum(z);
        const int mm = tk * 16 + r;
        if (q == 0 && mm < B) out[mm] = sigmoidf_acc(z + A.head_bias);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

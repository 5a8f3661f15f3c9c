// k_chain_v2.h -- register-chained fused forward for DeepFM_v2-structured plans
// (reference graph: DeepFM_v2.py:98-155; BASELINE config 2 is this graph at F=6, D=16).
// Included inside sparrow_hip.hip's anonymous namespace.
//
// One WAVE owns 16 samples from ids to score.  Lane (r = lane&15, q = lane>>4) is sample r's q-th
// 16-byte column slot, which is at the same time
//   * the natural unit of a coalesced embedding-row gather (4 lanes x 16 B = one 64-B row), and
//   * the B-operand layout of v_mfma_f32_16x16x4_f32 (lane supplies B[k=4q+s][col=r] at step s), and
//   * the C/D layout of the previous layer's output (lane holds D[row=4q+j][col=r]).
// So with the weights as the A operand (W^T rows, read from LDS), embedding rows are loaded from
// HBM straight into MFMA operand registers and every layer's output feeds the next layer's MFMA
// without leaving the register file: per-field Dense projections -> FM (sum)^2 - sum(squares) ->
// Dense+ReLU -> Dense+ReLU -> output dot + sigmoid.  No activation ever touches LDS or HBM, there
// is no barrier after the one-time weight staging, and waves progress independently, so gather
// latency of one wave hides under the MFMAs of its neighbours.
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
    const float* table[V2_MAX_FIELDS];    // [vocab][4*DV] padded embedding tables
    const float* w1[V2_MAX_FIELDS];       // [vocab] first-order weights
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

template <int G_EMB, int DV, int KPC, int H0C, int H1C>
struct V2Lds {
    static constexpr int G = G_EMB + 1;
    static constexpr int DPC = (DV + 3) / 4;          // 16-float chunks per embedding row
    static constexpr int KP = KPC * 16;
    static constexpr int SP = DPC * 16 + 4;           // LDS row stride of an embedding-group projection W^T
    static constexpr int SN = 16 + 4;                 // ... of the numeric group
    static constexpr int S0 = G * KP + 4;             // deep0 W^T row stride
    static constexpr int S1 = H0C * 16 + 4;           // deep1 W^T row stride
    static constexpr int off_wp = 0;                  // [G_EMB][KP][SP]
    static constexpr int off_wn = off_wp + G_EMB * KP * SP;   // [KP][SN]
    static constexpr int off_bp = off_wn + KP * SN;   // [G][KP]
    static constexpr int off_w0 = off_bp + G * KP;    // [H0C*16][S0]
    static constexpr int off_b0 = off_w0 + H0C * 16 * S0;
    static constexpr int off_w1 = off_b0 + H0C * 16;  // [H1C*16][S1]
    static constexpr int off_b1 = off_w1 + H1C * 16 * S1;
    static constexpr int off_hfm = off_b1 + H1C * 16; // [KP]
    static constexpr int off_hd = off_hfm + KP;       // [H1C*16]
    static constexpr int off_fn = off_hd + H1C * 16;  // [8]
    static constexpr int total = off_fn + 8;          // floats
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

// One-time (finalize) kernel: lays every weight out exactly as the fused kernel wants it in LDS
// (padded row strides, zero fill), so the per-launch staging is a flat copy.
template <int G_EMB, int DV, int KPC, int H0C, int H1C>
__global__ __launch_bounds__(256) void k_v2_pack_image(const V2Args A, float* __restrict__ lds) {
    using LD = V2Lds<G_EMB, DV, KPC, H0C, H1C>;
    constexpr int G = LD::G, KP = LD::KP;
    const int tid = threadIdx.x, NT = 256;
    static_assert(LD::total % 4 == 0, "LDS image must be a whole number of float4");
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
}

// Run-time arguments of the fused kernel (the weights travel through the packed image).
struct V2Run {
    int F, ND, n_num;
    int col[V2_MAX_FIELDS];               // ids column of field g (embedding AND first-order weight)
    int vocab[V2_MAX_FIELDS];
    const float* table[V2_MAX_FIELDS];    // [vocab][4*DV]
    const float* w1[V2_MAX_FIELDS];       // [vocab]
    float h0w, fo_bias, head_bias;
};

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
__device__ __forceinline__ f32x4 sel4(bool ok, f32x4 v) {
    return f32x4{ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f};
}

// HOIST = true : the LDS weight reads are loop-invariant, hipcc keeps every W^T fragment in VGPRs across
//                tasks (no LDS traffic in the loop, ~230 VGPRs -> 2 waves/SIMD).
// HOIST = false: the LDS offset is laundered through an empty asm each task, so fragments are re-read from
//                LDS where used (~128 VGPRs -> 4 waves/SIMD hide gather latency by occupancy).
template <int G_EMB, int DV, int KPC, int H0C, int H1C, int WAVES, bool HOIST>
__global__ __launch_bounds__(WAVES * 64, HOIST ? 2 : 4) void k_deepfm_v2_chain(const V2Run A, const int* __restrict__ ids,
                                                                const float* __restrict__ dense,
                                                                float* __restrict__ out, int B,
                                                                int* __restrict__ err,
                                                                const float* __restrict__ image) {
    using LD = V2Lds<G_EMB, DV, KPC, H0C, H1C>;
    constexpr int G = LD::G, DPC = LD::DPC, KP = LD::KP;
    constexpr int NT = WAVES * 64;
    static_assert(H0C % 2 == 0, "deep0 n-blocks are processed in interleaved pairs");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntasks = (B + 15) >> 4;
    const int task_stride = gridDim.x * WAVES;
    int task = blockIdx.x * WAVES + wave;

    // ---- gather in three branch-free stages, each consumed one phase after it was issued, so no
    //      wait ever sits next to the loads it guards:
    //        load_ids   : this task's ids + numerics                  (consumed by issue_rows)
    //        issue_rows : embedding rows + first-order weights        (consumed by finish_rows)
    //        finish_rows: sum the first-order weights
    //      All addressing is a 32-bit offset from a wave-uniform (SGPR) base.  A missing (-1) or
    //      out-of-range id is redirected to the all-zero row the host appended at index `vocab`, and a
    //      row past the end of the batch is clamped to the last row (its score is never stored), so
    //      nothing needs a select.  The numeric lane slots beyond n_num hold a duplicate finite value
    //      that only ever meets zero weights in the packed image. ----
    f32x4 x[G_EMB][DPC];
    f32x4 xn;
    float fo;
    unsigned long long badmask = 0;   // wave-level (SALU) OR of "id outside its table" lanes
    int idv[G_EMB];
    f32x4 nvv;
    float w1v[G_EMB];
    auto load_ids = [&](int tk) {
        int m = tk * 16 + r;
        m = m < B ? m : (B - 1);
        const unsigned ibase = (unsigned)m * (unsigned)A.F;
#pragma unroll
        for (int g = 0; g < G_EMB; ++g) idv[g] = (ids + A.col[g])[ibase];
        const unsigned dbase = (unsigned)m * (unsigned)A.ND;
        const unsigned c0 = 4 * q;
        const unsigned last = (unsigned)A.n_num - 1;
        nvv.x = dense[dbase + (c0 + 0 < last ? c0 + 0 : last)];
        nvv.y = dense[dbase + (c0 + 1 < last ? c0 + 1 : last)];
        nvv.z = dense[dbase + (c0 + 2 < last ? c0 + 2 : last)];
        nvv.w = dense[dbase + (c0 + 3 < last ? c0 + 3 : last)];
    };
    auto issue_rows = [&]() {
        xn = nvv;
#pragma unroll
        for (int g = 0; g < G_EMB; ++g) {
            const int id = idv[g];
            const bool ok = (unsigned)id < (unsigned)A.vocab[g];
            badmask |= __ballot(!ok && id != -1);
            const unsigned sid = ok ? (unsigned)id : (unsigned)A.vocab[g];     // -> the zero row
            w1v[g] = A.w1[g][sid];
            const unsigned off = sid * (unsigned)(4 * DV) + 4u * q;
#pragma unroll
            for (int c = 0; c < DPC; ++c) {
                // this lane's 16-byte piece of the row; pieces past the row end re-read piece 0 and
                // only ever meet zero weights
                const unsigned o = (4 * c + 3 < DV || 4u * c + q < (unsigned)DV) ? off + 16u * c : sid * (unsigned)(4 * DV);
                x[g][c] = ld4(A.table[g] + o);
            }
        }
    };
    auto finish_rows = [&]() {
        fo = 0.f;
#pragma unroll
        for (int g = 0; g < G_EMB; ++g) fo += w1v[g];
    };

    // ---- prologue: ids first, then the weight image (independent of the ids, so both are in flight
    //      together), then the rows ----
    const bool have0 = task < ntasks;
    if (have0) load_ids(task);
    {
        constexpr int total4 = LD::total / 4;
#pragma unroll 1
        for (int base = 0; base < total4; base += 4 * NT) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NT + tid;
                v[u] = ld4(image + 4 * (idx < total4 ? idx : 0));
            }
            if (base == 0 && have0) issue_rows();             // ids have landed; image loads still in flight
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base + u * NT + tid;
                if (idx < total4) st4(smem + 4 * idx, v[u]);
            }
        }
    }
    if (have0 && task + task_stride < ntasks) load_ids(task + task_stride);
    __syncthreads();

    for (; task < ntasks; task += task_stride) {
        const int m = task * 16 + r;
        int lds_off = 0;
        if (!HOIST) asm volatile("" : "+v"(lds_off));
        const float* lds = smem + lds_off;
        const float* wq = lds + 4 * q;                            // this lane's 16-byte column slot
        finish_rows();
        // first-order term: categorical weights (every q lane holds the same sum: count it once) +
        // numeric Dense(1) partial (only lanes q<2 hold real numerics; the packed fo_num weights
        // are zero beyond n_num)
        float z1 = (q == 0) ? fo : 0.f;
        {
            const float d = dot4(ld4(lds + LD::off_fn + 4 * (q & 1)), xn);
            z1 += (q < 2) ? d : 0.f;
        }

        // ---- per-field Dense projections (DeepFM_v2.py:106-120); accumulators start at the bias; two
        //      fields run as alternating MFMA chains; W^T fragments are fetched one step ahead ----
        f32x4 P[G][KPC];
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) {
#pragma unroll
            for (int g = 0; g < G; ++g) P[g][nb] = ld4(wq + LD::off_bp + g * KP + nb * 16);
            constexpr int NPAIR = G_EMB / 2;
            f32x4 a0 = ld4(wq + LD::off_wp + (0 * KP + nb * 16 + r) * LD::SP);
            f32x4 a1 = ld4(wq + LD::off_wp + ((G_EMB > 1 ? 1 : 0) * KP + nb * 16 + r) * LD::SP);
#pragma unroll
            for (int c = 0; c < DPC; ++c) {
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) {
                    const int g = 2 * pr;
                    // next fragment pair: (same c, next pair) or (next c, first pair)
                    const int ng = (pr + 1 < NPAIR) ? g + 2 : 0;
                    const int nc = (pr + 1 < NPAIR) ? c : c + 1;
                    f32x4 b0 = a0, b1 = a1;
                    if (nc < DPC) {
                        a0 = ld4(wq + LD::off_wp + (ng * KP + nb * 16 + r) * LD::SP + 16 * nc);
                        a1 = ld4(wq + LD::off_wp + ((ng + 1) * KP + nb * 16 + r) * LD::SP + 16 * nc);
                    }
                    mfma4x2(b0, x[g][c], P[g][nb], b1, x[g + 1][c], P[g + 1][nb]);
                }
            }
            if (G_EMB & 1) {
                constexpr int g = G_EMB - 1;
#pragma unroll
                for (int c = 0; c < DPC; ++c) {
                    const f32x4 a = ld4(wq + LD::off_wp + (g * KP + nb * 16 + r) * LD::SP + 16 * c);
                    P[g][nb] = mfma4(a, x[g][c], P[g][nb]);
                }
            }
            const f32x4 an = ld4(wq + LD::off_wn + (nb * 16 + r) * LD::SN);
            P[G_EMB][nb] = mfma4(an, xn, P[G_EMB][nb]);
        }

        // x is dead: issue the next task's row gathers (its ids arrived during the projections) and
        // the ids of the task after that; both stay in flight under FM / deep0 / deep1 and are only
        // consumed by finish_rows() at the top of the next iteration
        const float z1_keep = z1;
        if (task + task_stride < ntasks) {                    // wave-uniform
            issue_rows();
            if (task + 2 * task_stride < ntasks) load_ids(task + 2 * task_stride);
        }

        // ---- FM cross (sum)^2 - sum(squares) over the G fields (DeepFM_v2.py:147-152) + its output weights ----
        float z = 0.f;
#pragma unroll
        for (int nb = 0; nb < KPC; ++nb) {
            f32x4 s = zero, sq = zero;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                s += P[g][nb];
                sq += P[g][nb] * P[g][nb];
            }
            const f32x4 fm = s * s - sq;
            z += dot4(ld4(wq + LD::off_hfm + nb * 16), fm);
        }

        // ---- deep0: Dense(relu) over the flattened projections (DeepFM_v2.py:124-125); pairs of
        //      n-blocks as alternating chains, W^T fragments double-buffered ----
        f32x4 h0[H0C];
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = ld4(wq + LD::off_b0 + n0 * 16);
#pragma unroll
        for (int n0 = 0; n0 < H0C; n0 += 2) {
            const float* w0a = wq + LD::off_w0 + (n0 * 16 + r) * LD::S0;
            const float* w0b = wq + LD::off_w0 + ((n0 + 1) * 16 + r) * LD::S0;
            f32x4 a0 = ld4(w0a), a1 = ld4(w0b);
#pragma unroll
            for (int k = 0; k < G * KPC; ++k) {
                f32x4 b0 = a0, b1 = a1;
                if (k + 1 < G * KPC) {
                    a0 = ld4(w0a + 16 * (k + 1));
                    a1 = ld4(w0b + 16 * (k + 1));
                }
                mfma4x2(b0, P[k / KPC][k % KPC], h0[n0], b1, P[k / KPC][k % KPC], h0[n0 + 1]);
            }
        }
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = relu4(h0[n0]);

        // ---- deep1: Dense(relu) (DeepFM_v2.py:126) + output weights ----
#pragma unroll
        for (int n1 = 0; n1 < H1C; ++n1) {
            f32x4 acc = ld4(wq + LD::off_b1 + n1 * 16);
            f32x4 wv[H0C];
#pragma unroll
            for (int j = 0; j < H0C; ++j) wv[j] = ld4(wq + LD::off_w1 + (n1 * 16 + r) * LD::S1 + 16 * j);
#pragma unroll
            for (int j = 0; j < H0C; ++j) acc = mfma4(wv[j], h0[j], acc);
            z += dot4(ld4(wq + LD::off_hd + n1 * 16), relu4(acc));
        }

        // ---- output layer: concat([first, fm, deep]) . w + b -> sigmoid (DeepFM_v2.py:154-155) ----
        z += A.h0w * z1_keep;
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        if (q == 0 && m < B) out[m] = sigmoidf_acc(z + A.h0w * A.fo_bias + A.head_bias);
    }
    if (badmask != 0 && lane == 0) atomicOr(err, 1);
}

// k_dien_seq.h -- DIEN's interest-evolution stage (reference DIEN.py:163-250): shared movie Embedding -> Keras GRU over
// the T history slots (mask_zero consumed: a slot with id 0 keeps state and repeats the previous output) ->
// per-slot attention gate against the candidate -> hand-rolled AUGRU -> the final AUGRU state [B, D], written to the
// same aux buffer DIN's pooled history uses, so the tail (concat -> Dense PReLU Dense PReLU Dense sigmoid, DIEN.py:252-259)
// runs on the DIN tail kernels unchanged.  Included inside sparrow_hip.hip's anonymous namespace.
//
// The recurrence is strictly sequential in t and tiny per sample (D = 10: ~1.9 k multiply-adds per slot), so the
// mapping is ONE LANE PER SAMPLE with every state vector in registers and the 9 KB of weights in LDS, read as
// wave-uniform (broadcast) 16-byte vectors -- one LDS read feeds four FMAs of all 64 samples.  No cross-lane traffic,
// no barriers after the weight copy.  fp32 throughout.
//
// Packed weight image (built by the host, include/sparrow_hip.h documents it), strides padded to 4 floats:
//   Dq = pad4(D), N3 = pad4(3 D)
//   gru_k [D][N3] | gru_u [D][N3] | gru_b [2][N3] | att_w0 [D][H] | att_b0 [H] | att_w1 [H] | att_b1 [4]
//   | for gate in (r, z, h): in_k [D][Dq] | in_b [Dq] | hid_k [D][Dq] | out_k [D][Dq] | out_b [Dq]
//   | h0 [Dq]

// Branch-free sigmoid / tanh on v_exp_f32 + v_rcp_f32 (absolute error ~2e-7, far inside the 1e-4 bar).  The library
// tanhf / expf carry range branches; with branches in the loop body LLVM sinks the accumulate chains into the later
// blocks and keeps every loaded weight alive until then (it spilled ~650 registers per lane).
__device__ __forceinline__ float dien_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float dien_tanh(float x) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f)), 1.0f);
}

template <int D, int H>
struct DienLayout {
    static constexpr int Dq = (D + 3) & ~3, N3 = (3 * D + 3) & ~3;
    static constexpr int gru_k = 0, gru_u = gru_k + D * N3, gru_b = gru_u + D * N3;
    static constexpr int att_w0 = gru_b + 2 * N3, att_b0 = att_w0 + D * H, att_w1 = att_b0 + H, att_b1 = att_w1 + H;
    static constexpr int gate0 = att_b1 + 4;
    static constexpr int g_in_k = 0, g_in_b = D * Dq, g_hid_k = g_in_b + Dq, g_out_k = g_hid_k + D * Dq, g_out_b = g_out_k + D * Dq;
    static constexpr int gate_floats = g_out_b + Dq;
    static constexpr int h0 = gate0 + 3 * gate_floats;
    static constexpr int total = h0 + Dq;
    static constexpr int total_pad = (total + 63) & ~63;
};

struct DienRun {
    int T, F, hist_col, cand_col, Dp, vocab, NA;
    const float* table;        // [vocab][Dp]
    const float* image;        // DienLayout<D,H>::total_pad floats
};

// y[0..N) += x[0..K) . W[K][stride]   (W in LDS, wave-uniform reads).  Row i+1's weights are read while row i's FMAs run
// (explicit double buffer; the sched_barrier keeps one row of prefetch, not a whole matrix, in registers).
template <int K, int N, int STRIDE>
__device__ __forceinline__ void dien_matvec(const float* __restrict__ W, const float (&x)[K], float (&y)[N]) {
    constexpr int NV = (N + 3) / 4;
    f32x4 w[2][NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) w[0][v] = ld4(W + 4 * v);
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (i + 1 < K) {
#pragma unroll
            for (int v = 0; v < NV; ++v) w[(i + 1) & 1][v] = ld4(W + (i + 1) * STRIDE + 4 * v);
        }
        // packed FMAs (v_pk_fma_f32: two columns per instruction, x[i] broadcast); an odd last column goes alone
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 xx = {x[i], x[i]};
#pragma unroll
        for (int j = 0; j + 1 < N; j += 2) {
            f32x2 acc = {y[j], y[j + 1]};
            const f32x2 ww = {w[i & 1][j >> 2][j & 3], w[i & 1][(j + 1) >> 2][(j + 1) & 3]};
            acc = __builtin_elementwise_fma(xx, ww, acc);
            // pin this row's FMAs here: left alone, the SLP vectorizer re-bundles the accumulate chains into packed FMAs
            // placed after the LAST row, so every weight of the matrix stays live until then (hundreds of spilled registers)
            asm volatile("" : "+v"(acc));
            y[j] = acc[0];
            y[j + 1] = acc[1];
        }
        if (N & 1) {
            y[N - 1] = fmaf(x[i], w[i & 1][(N - 1) >> 2][(N - 1) & 3], y[N - 1]);
            asm volatile("" : "+v"(y[N - 1]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int D, int H>
__global__ __launch_bounds__(64) void k_dien_seq(const DienRun A, const int* __restrict__ ids, float* __restrict__ aux, int B,
                                                 int* __restrict__ err) {
    using LY = DienLayout<D, H>;
    constexpr int Dq = LY::Dq, N3 = LY::N3;
    __shared__ __attribute__((aligned(16))) float Wlds[LY::total_pad];
    const int lane = threadIdx.x;
    for (int i = lane; i < LY::total_pad; i += 64) Wlds[i] = A.image[i];
    __syncthreads();
    for (int m0 = blockIdx.x * 64; m0 < B; m0 += gridDim.x * 64) {
        const int m = min(m0 + lane, B - 1);                      // lanes past the end redo the last sample, never stored
        const int* row = ids + (size_t)m * A.F;
        bool bad = false;
        float c[D], h[D], g[D], hs[D];
        {
            int cid = row[A.cand_col];
            if (cid < 0 || cid >= A.vocab) { bad = true; cid = 0; }
            const float* cr = A.table + (size_t)cid * A.Dp;
#pragma unroll
            for (int j = 0; j < D; ++j) { c[j] = cr[j]; h[j] = 0.f; g[j] = 0.f; hs[j] = Wlds[LY::h0 + j]; }
        }
#pragma unroll 1
        for (int t = 0; t < A.T; ++t) {
            // the weights are re-read from LDS every slot: an opaque zero in the address keeps the compiler from hoisting
            // ~2 300 loop-invariant loads out of the t loop into (spilled) registers
            int wz = 0;
            asm volatile("" : "+v"(wz));
            const float* W = Wlds + wz;
            int id = row[A.hist_col + t];
            if (id < 0 || id >= A.vocab) { bad = true; id = 0; }
            // ---- GRU step (reset_after), skipped where the slot is masked (id 0) ----
            {
                const float* xr = A.table + (size_t)id * A.Dp;
                float x[D];
#pragma unroll
                for (int j = 0; j < D; ++j) x[j] = xr[j];
                float mx[3 * D], mh[3 * D];
#pragma unroll
                for (int j = 0; j < 3 * D; ++j) { mx[j] = W[LY::gru_b + j]; mh[j] = W[LY::gru_b + N3 + j]; }
                dien_matvec<D, 3 * D, N3>(W + LY::gru_k, x, mx);
                dien_matvec<D, 3 * D, N3>(W + LY::gru_u, h, mh);
                // blend by an OPAQUE all-ones / all-zeros mask: with a select on `id != 0` LLVM sinks each column's whole
                // accumulate chain into an `if (live)` block and keeps the weights alive (spilled) until there
                int live = id != 0 ? -1 : 0;
                asm("" : "+v"(live));
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float z = dien_sigmoid(mx[j] + mh[j]);
                    const float r = dien_sigmoid(mx[D + j] + mh[D + j]);
                    const float hh = dien_tanh(fmaf(r, mh[2 * D + j], mx[2 * D + j]));
                    const float hn = fmaf(z, h[j], (1.0f - z) * hh);
                    const int hb = __float_as_int(hn) & live;
                    h[j] = __int_as_float(hb | (__float_as_int(h[j]) & ~live));
                    g[j] = __int_as_float(hb | (__float_as_int(g[j]) & ~live));   // masked: the previous output again (zeros before the first)
                }
            }
            // ---- attention gate: sigmoid(Dense1(sigmoid(Dense32(g * c)))) ----
            float a;
            {
                float p[D], u[H];
#pragma unroll
                for (int j = 0; j < D; ++j) p[j] = g[j] * c[j];
#pragma unroll
                for (int j = 0; j < H; ++j) u[j] = W[LY::att_b0 + j];
                dien_matvec<D, H, H>(W + LY::att_w0, p, u);
                float s = W[LY::att_b1];
#pragma unroll
                for (int j = 0; j < H; ++j) s = fmaf(dien_sigmoid(u[j]), W[LY::att_w1 + j], s);
                a = dien_sigmoid(s);
            }
            // ---- AUGRU step ----
            {
                float rz[2][D];
#pragma unroll
                for (int gate = 0; gate < 2; ++gate) {
                    const float* G = W + LY::gate0 + gate * LY::gate_floats;
                    float pre[D], o[D];
#pragma unroll
                    for (int j = 0; j < D; ++j) { pre[j] = G[LY::g_in_b + j]; o[j] = G[LY::g_out_b + j]; }
                    dien_matvec<D, D, Dq>(G + LY::g_in_k, g, pre);
                    dien_matvec<D, D, Dq>(G + LY::g_hid_k, hs, pre);
                    dien_matvec<D, D, Dq>(G + LY::g_out_k, pre, o);
#pragma unroll
                    for (int j = 0; j < D; ++j) rz[gate][j] = dien_sigmoid(o[j]);
                }
                const float* G = W + LY::gate0 + 2 * LY::gate_floats;
                float hz[D], pre[D], o[D];
#pragma unroll
                for (int j = 0; j < D; ++j) { hz[j] = hs[j] * rz[1][j]; pre[j] = G[LY::g_in_b + j]; o[j] = G[LY::g_out_b + j]; }
                dien_matvec<D, D, Dq>(G + LY::g_in_k, g, pre);
                dien_matvec<D, D, Dq>(G + LY::g_hid_k, hz, pre);
                dien_matvec<D, D, Dq>(G + LY::g_out_k, pre, o);
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const float u = a * rz[0][j];
                    hs[j] = fmaf(u, dien_tanh(o[j]), (1.0f - u) * hs[j]);
                }
            }
        }
        if (m0 + lane < B) {
            float* o = aux + (size_t)m * A.NA;
#pragma unroll
            for (int j = 0; j < D; ++j) o[j] = hs[j];
            for (int j = D; j < A.NA; ++j) o[j] = 0.f;
        }
        if (bad) atomicOr(err, 1);
    }
}

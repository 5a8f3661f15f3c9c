// k_dien_mfma.h -- DIEN's interest-evolution stage (reference DIEN.py:163-250) on the matrix pipe: SIXTEEN SAMPLES per wave instead
// of k_dien_seq.h's one lane per sample.  Included inside sparrow_hip.hip's anonymous namespace, after k_dien_seq.h (whose weight
// image, sigmoid / tanh and DienRun it shares) and dyn_split.h.
//
// Why.  At B = 65 536 the lane-per-sample kernel is 1 024 waves -- ONE per SIMD, nothing to hide a latency behind -- each issuing
// ~7 000 dependent VALU / LDS instructions: 36.7 us for the reference's own shape (T = 5, D = 10), the slowest stage of any model
// here.  Every step of the recurrence is a handful of small Dense layers applied to per-sample vectors, i.e. the register chain
// the other tails run: a vector lives in the C/D layout of the 16x16 MFMAs -- lane (r = lane & 15, q = lane >> 4) holds features
// 4q .. 4q+3 of sample r -- two such chunks ARE one K = 32 operand of v_mfma_f32_16x16x32_f16 in dyn_split.h's permuted order
// (k_local = {4q..4q+3} U {16+4q..16+4q+3}), and the result is again in the C/D layout.  So [x_t ; h], [g ; hs] go in as ONE
// operand and "x K + h U" is one block.
//   first version: exact f32 MFMA (v_mfma_f32_16x16x4_f32, 68 per step).  Correct, and no faster than the lane kernel (34.5 vs
//   36.3 us): 32 cycles each on a pipe that shares the VALU's issue, next to 66 quarter-rate exp / rcp per step.
//   this version: split-f16 with STATIC power-of-two scales -- everything the recurrence feeds an MFMA is bounded at finalize:
//   |h|, |g| <= 1 (tanh / sigmoid mixes of a zero start), |hs| <= max(1, |h0|), |x|, |g c| <= max|E|, and a gate's hidden
//   pre-activation by its weights' absolute row sums.  36 MFMAs of 16 cycles per step (hi.hi + lo.hi + hi.lo: 22 bits) instead of
//   68 of 32; 4 096 tiles = four waves per SIMD.
// Each GRU / AUGRU gate gets its OWN 16-row output block, so z[j], r[j], h~[j] of one j meet in one lane and the gate arithmetic
// is element-wise.  Features beyond D carry zero weights and biases everywhere: they stay exactly 0 through the recurrence
// (sigmoid(0) = 0.5 multiplies a state that starts at 0, tanh(0) = 0).
//
// Fragment image (k_dien_mfma_pack builds it at sprk_finalize from the host's packed image, DienLayout<D, H>):
//   12 blocks x {hi: 64 lanes x 8 halfs | lo: same} (2 KB):   operand [chunk 0 ; chunk 1]
//      0 GRU z  [K_z ; U_z]      1 GRU r  [K_r ; U_r]      2 GRU x.h~ [K_h ; 0]      3 GRU h.h~ [0 ; U_h]          B = [x ; h]
//      4, 5 attention Dense(32) [W0 ; 0]                                                                            B = [g c ; 0]
//      6 gate R pre [in ; hid]   7 gate Z pre [in ; hid]                                                            B = [g ; hs]
//      8 gate R out [out ; 0]    9 gate Z out [out ; 0]    11 gate H out [out ; 0]                                  B = [pre ; 0]
//      10 gate H pre [in ; hid]                                                                                     B = [g ; hs z]
//   15 vectors x 16 floats (C/D layout source): GRU biases z (input + recurrent), r, x.h~, h.h~ | attention bias 2 | attention output
//      weights 2 | gate in-bias R Z H | gate out-bias R Z H | h0
//   scalars: att_b1, operand scales s_xh s_p s_gs s_pre[3], per-block un-scales [12], ok
#pragma once

#define DM_BLOCKS 12
#define DM_VECS 15
#define DM_WAVES 4
template <int D, int H>
struct DienFrag {
    static_assert(D <= 16 && H == 32, "one 16-feature chunk per vector, two attention blocks");
    static constexpr int vec0 = DM_BLOCKS * 512;
    static constexpr int sc0 = vec0 + DM_VECS * 16;               // scalars
    static constexpr int S_B1 = sc0, S_XH = sc0 + 1, S_P = sc0 + 2, S_GS = sc0 + 3, S_PRE = sc0 + 4, S_UN = sc0 + 8, S_OK = sc0 + 20;
    static constexpr int total = sc0 + 32;
    static constexpr int total_pad = (total + 255) & ~255;
    static constexpr int V_BZ = 0, V_BR = 1, V_BXH = 2, V_BRH = 3, V_AB = 4, V_AW = 6, V_GIN = 8, V_GOUT = 11, V_H0 = 14;
};

// 2^(14 - e) with bound < 2^e (bound = 0: 2^14); 0 for a non-finite bound
__device__ inline float dm_scale14(float bound) {
    if (!(bound < 3.0e38f)) return 0.f;
    int e = 0;
    if (bound > 0.f) (void)frexpf(bound, &e);
    e = 14 - e;
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
    return ldexpf(1.f, e);
}

template <int D, int H>
__global__ __launch_bounds__(256) void k_dien_mfma_pack(const float* __restrict__ img, const unsigned* __restrict__ table_absmax_bits,
                                                        float* __restrict__ out) {
    using LY = DienLayout<D, H>;
    using FR = DienFrag<D, H>;
    __shared__ float sA[DM_BLOCKS], sB[DM_BLOCKS], sc[8];
    // (chunk, block) -> weight W[k][n] of the host image, 0 outside the matrix
    auto wt = [&](int blk, int chunk, int k, int n) -> float {
        if (k >= D) return 0.f;
        if (blk < 4) {                                            // GRU [D][N3], columns z | r | h
            const int g = blk < 2 ? blk : 2;
            if (n >= D) return 0.f;
            if (blk == 2 && chunk == 1) return 0.f;
            if (blk == 3 && chunk == 0) return 0.f;
            return img[(chunk == 0 ? LY::gru_k : LY::gru_u) + k * LY::N3 + g * D + n];
        }
        if (blk < 6) return chunk == 0 ? img[LY::att_w0 + k * H + (blk - 4) * 16 + n] : 0.f;
        if (n >= D) return 0.f;
        const int g = blk == 6 || blk == 8 ? 0 : (blk == 7 || blk == 9 ? 1 : 2);
        const bool is_out = blk == 8 || blk == 9 || blk == 11;
        if (is_out && chunk == 1) return 0.f;
        const int base = LY::gate0 + g * LY::gate_floats + (is_out ? LY::g_out_k : (chunk == 0 ? LY::g_in_k : LY::g_hid_k));
        return img[base + k * LY::Dq + n];
    };
    if (threadIdx.x == 0) {
        float maxE = __uint_as_float(*table_absmax_bits);
        float hb = 1.f;
        for (int j = 0; j < D; ++j) hb = fmaxf(hb, fabsf(img[LY::h0 + j]));
        const float s_xh = dm_scale14(fmaxf(maxE, hb)), s_p = dm_scale14(maxE), s_gs = dm_scale14(hb);
        float s_pre[3];
        for (int g = 0; g < 3; ++g) {
            const int base = LY::gate0 + g * LY::gate_floats;
            float bound = 0.f;
            for (int n = 0; n < D; ++n) {
                float b = fabsf(img[base + LY::g_in_b + n]);
                for (int k = 0; k < D; ++k) b += fabsf(img[base + LY::g_in_k + k * LY::Dq + n]) + hb * fabsf(img[base + LY::g_hid_k + k * LY::Dq + n]);
                bound = fmaxf(bound, b);
            }
            s_pre[g] = dm_scale14(bound);
        }
        bool ok = s_xh > 0.f && s_p > 0.f && s_gs > 0.f && s_pre[0] > 0.f && s_pre[1] > 0.f && s_pre[2] > 0.f;
        for (int blk = 0; blk < DM_BLOCKS; ++blk) {
            float mx = 0.f;
            for (int chunk = 0; chunk < 2; ++chunk)
                for (int k = 0; k < D; ++k)
                    for (int n = 0; n < 16; ++n) mx = fmaxf(mx, fabsf(wt(blk, chunk, k, n)));
            float s = 0.f;
            if (mx < 3.0e38f) { int e = 0; if (mx > 0.f) (void)frexpf(mx, &e); e = 15 - e; e = e > 60 ? 60 : (e < -60 ? -60 : e); s = ldexpf(1.f, e); }
            ok = ok && s > 0.f;
            sA[blk] = s;
            sB[blk] = blk < 4 ? s_xh : blk < 6 ? s_p : (blk == 6 || blk == 7 || blk == 10) ? s_gs : s_pre[blk == 8 ? 0 : blk == 9 ? 1 : 2];
        }
        sc[0] = s_xh; sc[1] = s_p; sc[2] = s_gs; sc[3] = s_pre[0]; sc[4] = s_pre[1]; sc[5] = s_pre[2]; sc[6] = ok ? 1.f : 0.f;
    }
    __syncthreads();
    _Float16* oh = reinterpret_cast<_Float16*>(out);
    for (int i = threadIdx.x; i < DM_BLOCKS * 512; i += 256) {    // one hi / lo pair per iteration
        const int blk = i >> 9, lane = (i >> 3) & 63, e = i & 7, r = lane & 15, q = lane >> 4;
        const float x = wt(blk, e < 4 ? 0 : 1, 4 * q + (e & 3), r) * sA[blk];
        const _Float16 hi = (_Float16)x;
        oh[(size_t)blk * 1024 + lane * 8 + e] = hi;
        oh[(size_t)blk * 1024 + 512 + lane * 8 + e] = (_Float16)(x - (float)hi);
    }
    for (int i = threadIdx.x; i < FR::total_pad - FR::vec0; i += 256) {
        const int a = FR::vec0 + i;
        float v = 0.f;
        if (a < FR::sc0) {
            const int vec = i >> 4, j = i & 15;
            if (vec == 0) { if (j < D) v = img[LY::gru_b + j] + img[LY::gru_b + LY::N3 + j]; }
            else if (vec == 1) { if (j < D) v = img[LY::gru_b + D + j] + img[LY::gru_b + LY::N3 + D + j]; }
            else if (vec == 2) { if (j < D) v = img[LY::gru_b + 2 * D + j]; }
            else if (vec == 3) { if (j < D) v = img[LY::gru_b + LY::N3 + 2 * D + j]; }
            else if (vec < 6) v = img[LY::att_b0 + (vec - 4) * 16 + j];
            else if (vec < 8) v = img[LY::att_w1 + (vec - 6) * 16 + j];
            else if (vec < 14) { if (j < D) v = img[LY::gate0 + ((vec - 8) % 3) * LY::gate_floats + (vec < 11 ? LY::g_in_b : LY::g_out_b) + j]; }
            else if (j < D) v = img[LY::h0 + j];
        } else {
            const int s = a - FR::sc0;
            if (s == 0) v = img[LY::att_b1];
            else if (s >= 1 && s <= 3) v = sc[s - 1];
            else if (s >= 4 && s <= 6) v = sc[s - 1];
            else if (s >= 8 && s < 8 + DM_BLOCKS) v = (sA[s - 8] > 0.f && sB[s - 8] > 0.f) ? 1.f / (sA[s - 8] * sB[s - 8]) : 0.f;
            else if (s == 20) v = sc[6];
        }
        out[a] = v;
    }
}

__device__ __forceinline__ f32x4 dm_sigmoid4(f32x4 v) { return f32x4{dien_sigmoid(v.x), dien_sigmoid(v.y), dien_sigmoid(v.z), dien_sigmoid(v.w)}; }
__device__ __forceinline__ f32x4 dm_tanh4(f32x4 v) { return f32x4{dien_tanh(v.x), dien_tanh(v.y), dien_tanh(v.z), dien_tanh(v.w)}; }

template <int D, int H>
__global__ __launch_bounds__(DM_WAVES * 64, 4) void k_dien_seq_mfma(const DienRun A, const int* __restrict__ ids, float* __restrict__ aux,
                                                                    int B, int* __restrict__ err) {
    using FR = DienFrag<D, H>;
    float* W = smem;
    for (int i = threadIdx.x; i < FR::total_pad / 4; i += DM_WAVES * 64)
        st4(W + 4 * i, ld4(A.image + 4 * i));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f}, one = f32x4{1.f, 1.f, 1.f, 1.f};
    const float* fl = W + 4 * lane;                               // this lane's 16 bytes of a fragment half
    auto vec = [&](int v) { return ld4(W + FR::vec0 + v * 16 + 4 * q); };
    // block blk applied to the split operand (bh, bl): un-scaled, plus a bias vector
    auto mm = [&](int blk, din_f16x8 bh, din_f16x8 bl, f32x4 bias) {
        const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(fl + blk * 512));
        const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(fl + blk * 512 + 256));
        f32x4 acc = mfma_f16(al, bh, zero);
        acc = mfma_f16(ah, bl, acc);
        acc = mfma_f16(ah, bh, acc);
        const float un = lone_scalar(W[FR::S_UN + blk]);         // [r6] never the high dword of a pair: dyn_split.h
        return f32x4{fmaf(acc.x, un, bias.x), fmaf(acc.y, un, bias.y), fmaf(acc.z, un, bias.z), fmaf(acc.w, un, bias.w)};
    };
    const float s_xh = W[FR::S_XH], s_p = W[FR::S_P], s_gs = W[FR::S_GS];
    const float s_pre0 = W[FR::S_PRE], s_pre1 = W[FR::S_PRE + 1], s_pre2 = W[FR::S_PRE + 2];
    const bool qin = 4 * q < A.Dp;                                // this lane's four features exist in a table row
    const int ntiles = (B + 15) >> 4;
    bool bad = false;
    for (int tile = blockIdx.x * DM_WAVES + wave; tile < ntiles; tile += gridDim.x * DM_WAVES) {
        const int m = min(tile * 16 + r, B - 1);                  // rows past the end redo the last sample, never stored
        const int* row = ids + (size_t)m * A.F;
        int cid = row[A.cand_col];
        if (cid < 0 || cid >= A.vocab) { bad = true; cid = 0; }
        const f32x4 c = qin ? ld4(A.table + (size_t)cid * A.Dp + 4 * q) : zero;
        f32x4 h = zero, g = zero, hs = vec(FR::V_H0);
        int id = row[A.hist_col];
        if (id < 0 || id >= A.vocab) { bad = true; id = 0; }
        f32x4 x = qin ? ld4(A.table + (size_t)id * A.Dp + 4 * q) : zero;
#pragma unroll 1
        for (int t = 0; t < A.T; ++t) {
            const bool live = id != 0;
            const f32x4 xt = x;
            if (t + 1 < A.T) {                                    // the next slot's row flies during this step
                id = row[A.hist_col + t + 1];
                if (id < 0 || id >= A.vocab) { bad = true; id = 0; }
                x = qin ? ld4(A.table + (size_t)id * A.Dp + 4 * q) : zero;
            }
            din_f16x8 bh, bl;
            // ---- GRU step (reset_after); a masked slot (id 0) keeps the state and repeats the previous output ----
            {
                dyn_split8(xt, h, s_xh, bh, bl);
                const f32x4 z = dm_sigmoid4(mm(0, bh, bl, vec(FR::V_BZ)));
                const f32x4 rr = dm_sigmoid4(mm(1, bh, bl, vec(FR::V_BR)));
                const f32x4 xh = mm(2, bh, bl, vec(FR::V_BXH));
                const f32x4 rh = mm(3, bh, bl, vec(FR::V_BRH));
                const f32x4 hh = dm_tanh4(rr * rh + xh);
                const f32x4 hn = z * h + (one - z) * hh;
                h = live ? hn : h;
                g = live ? hn : g;
            }
            // ---- attention gate: sigmoid(Dense1(sigmoid(Dense32(g * c)))) ----
            float a;
            {
                dyn_split8(g * c, zero, s_p, bh, bl);
                const f32x4 u0 = dm_sigmoid4(mm(4, bh, bl, vec(FR::V_AB + 0)));
                const f32x4 u1 = dm_sigmoid4(mm(5, bh, bl, vec(FR::V_AB + 1)));
                const f32x4 w0 = vec(FR::V_AW + 0), w1 = vec(FR::V_AW + 1);
                float s = u0.x * w0.x;
                s = fmaf(u0.y, w0.y, s); s = fmaf(u0.z, w0.z, s); s = fmaf(u0.w, w0.w, s);
                s = fmaf(u1.x, w1.x, s); s = fmaf(u1.y, w1.y, s); s = fmaf(u1.z, w1.z, s); s = fmaf(u1.w, w1.w, s);
                a = dien_sigmoid(rows4_sum(s) + W[FR::S_B1]);
            }
            // ---- AUGRU step: every gate = out(in(g) + hid(state)) ----
            {
                dyn_split8(g, hs, s_gs, bh, bl);
                const f32x4 pre_r = mm(6, bh, bl, vec(FR::V_GIN + 0));
                const f32x4 pre_z = mm(7, bh, bl, vec(FR::V_GIN + 1));
                dyn_split8(pre_r, zero, s_pre0, bh, bl);
                const f32x4 rt = dm_sigmoid4(mm(8, bh, bl, vec(FR::V_GOUT + 0)));
                dyn_split8(pre_z, zero, s_pre1, bh, bl);
                const f32x4 zt = dm_sigmoid4(mm(9, bh, bl, vec(FR::V_GOUT + 1)));
                dyn_split8(g, hs * zt, s_gs, bh, bl);
                const f32x4 pre_h = mm(10, bh, bl, vec(FR::V_GIN + 2));
                dyn_split8(pre_h, zero, s_pre2, bh, bl);
                const f32x4 hn = dm_tanh4(mm(11, bh, bl, vec(FR::V_GOUT + 2)));
                const f32x4 u = f32x4{a, a, a, a} * rt;
                hs = u * hn + (one - u) * hs;
            }
        }
        if (tile * 16 + r < B) {
            float* o = aux + (size_t)m * A.NA;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 4 * q + j;
                if (col < A.NA) o[col] = col < D ? hs[j] : 0.f;
            }
            if (q == 3) for (int col = 16; col < A.NA; ++col) o[col] = 0.f;
        }
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

// k_dien_fused.h -- DIEN (reference DIEN.py:163-250 + the tail DIEN.py:252-262) in ONE launch: k_dien_seq_mfma's recurrence and, for the
// same sixteen samples, k_din_tail's register chain as the epilogue of the wave that evolved them.  Included after k_din_tail.h and
// k_dien_mfma.h, whose pieces it is made of (DienFrag, the mm blocks, din_tail_unf_gather / din_tail_unf_fc0 / din_tail_dense).
//
// Why.  The two launches cost 25.8 + 17 us per 65 536 samples of DIEN.py's shape: the second one pays a dependent kernel boundary
// (1.5 us), its own staging of a 101 KB image and two dependent round trips (ids -> rows) at ONE task per wave, and the final state makes a
// trip through HBM in between.  A wave of the sequence kernel ends with hs in the C/D layout -- lane (r, q) holds features 4q .. 4q+3 of
// sample r -- which IS the tail's pooled-history operand xp[0]; so the wave goes straight on:
//   * sixteen waves per workgroup, one workgroup per CU (4 waves per SIMD, 128 VGPRs: what both halves were built for), one tile per wave at
//     B = 65 536; both weight images by LDS-DMA in front of one barrier (26 + 101 KB of the CU's 160);
//   * the tail's ids are read with the tile's first loads; its raw split rows (64 bytes per embedding column) and the numerics are
//     requested in front of the LAST step of the recurrence (peeled out of the loop), so their round trip runs under ~5 us of gate
//     arithmetic instead of in front of the tail;
//   * the tail itself is the code of k_din_tail<8, 4, 1, 16, DYN, ., UNF> (the same inlined functions on the same operands): the scores are
//     the two-launch path's scores.
// Needs the split-f16 form of both halves (dien_frag, the tail's raw rows: emb_dim <= 16, DIN.py's widths 128 / 64); everything else stays
// on the two launches.
#pragma once

// WHAT FUSING IT UNCOVERED, AND ITS CAUSE ([r5] found and fenced, [r6] explained; docs/open_issue_dien_tiles.md has the whole trail).
// The first build of this kernel returned, for emb_dim 16, a few WHOLE 16-sample tiles per launch -- other tiles every run -- whose final state was
// off by ~1e-3; k_dien_seq_mfma<16, 32>, the product since round 3, did the same once four of its workgroups shared a CU (~40 of 4 096 tiles per launch
// up to 6e-5 off the fp64 oracle).  Round 5 fenced it (`s_nop 1` statements the scheduler may not move LDS reads across) without knowing why that helped.
// Round 6, on the failing build's ASSEMBLY (scripts/r06/isa_patch_build.py: wait states, probes, dumps and single-instruction replacements between
// hipcc's compile and assemble steps): every input of ONE instruction was right and its result wrong --
//     v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0]        (pre_z = acc * un + bias, un = the HIGH dword of a ds_read_b64 pair)
// came out as the bias alone in its LOW half for lanes 48..63.  Replacing that one instruction by the op_sel-free form (or two v_fma_f32) in the
// otherwise identical binary: 0 bad tiles; more wait states, full waits, a copy of the MFMA result in fresh registers: no change.  It is a property of
// gfx950 that needs no MFMA in the victim wave at all (scripts/ubench/pkfma_opsel_mfma.hip): a packed-f32 VALU op whose low result takes the high dword
// of a VGPR src1 reads that dword as 0 in the wave's last quarter while ANOTHER wave of the SIMD issues 16x16 MFMAs with 128-bit operands back to
// back.  Which is why it needed four waves per SIMD, why it moved with every change of the schedule (hipcc only forms the pair when two blocks' un
// loads end up neighbours), and why all ten failing builds and none of the clean ones carried that op_sel.  The fix is lone_scalar() (dyn_split.h) on
// the un-scale scalar, and scripts/isa/isa_pk_opsel.py over every unit and over the built library as the guard that does not depend on the compiler.
#define DNF_WAVES 16

template <int D, int H, int N0C, int N1C>
__global__ __launch_bounds__(DNF_WAVES * 64, 4) void k_dien_fused(const DienRun A, const DinTailRun TL, const int* __restrict__ ids,
                                                                  const float* __restrict__ dense, float* __restrict__ out, int B,
                                                                  int* __restrict__ err, const float* __restrict__ tail_image, float* __restrict__ dbg) {
    using FR = DienFrag<D, H>;
    using LD = DinTailLds<N0C, N1C, 1>;
    static_assert(FR::total_pad % 256 == 0 && LD::total_pad % 256 == 0, "1-KB LDS-DMA pieces");
    static_assert((FR::total_pad + LD::total_pad) * 4 <= 160 * 1024, "both images in one CU's LDS");
    float* W = smem;
    const float* S = smem + FR::total_pad;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f}, one = f32x4{1.f, 1.f, 1.f, 1.f};
    const int ntiles = (B + 15) >> 4;
    int tile = blockIdx.x * DNF_WAVES + wave;

    // the first tile's ids fly under the staging
    int idv[DT_MAX_COLS], cid = 0, id = 0;
    auto ld_ids = [&](int tl) {
        const int m = min(tl * 16 + r, B - 1);                    // rows past the end redo the last sample, never stored
        const int* row = ids + (size_t)m * A.F;
        cid = row[A.cand_col];
        id = row[A.hist_col];
#pragma unroll
        for (int g = 0; g < DT_MAX_COLS; ++g) idv[g] = g < TL.n_cols ? row[TL.col[g]] : -1;
    };
    if (tile < ntiles) ld_ids(tile);
    constexpr int C0 = FR::total_pad / 256, C1 = LD::total_pad / 256;
#pragma unroll 1
    for (int c = wave; c < C0 + C1; c += DNF_WAVES) {
        const float* src = c < C0 ? A.image + c * 256 : tail_image + (c - C0) * 256;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
    }
    __syncthreads();

    const float* fl = W + 4 * lane;                               // this lane's 16 bytes of a fragment half
    auto vec = [&](int v) { return ld4(W + FR::vec0 + v * 16 + 4 * q); };
    auto mm = [&](int blk, din_f16x8 bh, din_f16x8 bl, f32x4 bias) __attribute__((always_inline)) {
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
    bool bad = false;
    for (; tile < ntiles; tile += gridDim.x * DNF_WAVES) {
        const int m = min(tile * 16 + r, B - 1);
        const int* row = ids + (size_t)m * A.F;
        if (cid < 0 || cid >= A.vocab) { bad = true; cid = 0; }
        const f32x4 c = qin ? ld4(A.table + (size_t)cid * A.Dp + 4 * q) : zero;
        f32x4 h = zero, g = zero, hs = vec(FR::V_H0);
        if (id < 0 || id >= A.vocab) { bad = true; id = 0; }
        f32x4 x = qin ? ld4(A.table + (size_t)id * A.Dp + 4 * q) : zero;
        // one step of the recurrence (k_dien_seq_mfma's, statement for statement); every step but the last requests the next slot's row
        auto step_body = [&](const f32x4 xt, const bool live) __attribute__((always_inline)) {
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
        };
        auto step = [&](int t, auto last_c) __attribute__((always_inline)) {
            const bool live = id != 0;
            const f32x4 xt = x;
            if constexpr (!decltype(last_c)::value) {
                id = row[A.hist_col + t + 1];
                if (id < 0 || id >= A.vocab) { bad = true; id = 0; }
                x = qin ? ld4(A.table + (size_t)id * A.Dp + 4 * q) : zero;
            }
            step_body(xt, live);
        };
#pragma unroll 1
        for (int t = 0; t + 1 < A.T; ++t) step(t, std::false_type{});
        // the tail's operands fly under the last step
        din_f16x8 eh[LD::NBLK], el[LD::NBLK];
        float xna, xnb;
        din_tail_unf_gather<LD>(TL, idv, q, bad, eh, el);
        {
            const float* nrow = dense + (size_t)m * TL.ND;
            const int last = TL.n_num - 1;
            xna = nrow[min(q, last)];
            xnb = nrow[min(q + 4, last)];
        }
        step(A.T - 1, std::true_type{});
        const int next = tile + gridDim.x * DNF_WAVES;
        if (next < ntiles) ld_ids(next);                           // (a persistent launch: the next tile's ids under this tile's tail)

        // ---- the tail: the final state is its pooled-history operand (what k_dien_seq_mfma stores: features beyond D are zeros) ----
        f32x4 xp[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[0][j] = 4 * q + j < D ? hs[j] : 0.f;
        f32x4 z0[N0C];
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = ld4(S + LD::off_b0 + nb * 16 + 4 * q);
        din_tail_unf_fc0<LD, N0C>(TL, S, lane, eh, el, z0);
        const float z = din_tail_dense<N0C, N1C, 1, true, true>(TL, S, z0, xp, xna, xnb, lane, r, q);
        const int ms = tile * 16 + r;
        if (q == 0 && ms < B) out[ms] = sigmoidf_acc(z + TL.head_bias);
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

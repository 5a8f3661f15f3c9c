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

// AN OPEN ISSUE, FENCED (round 5; scripts/r05/31_dien_fused_race.sh ... 35_*, profiles/r05/experiments/r05_31 ... r05_35).
// The first build of this kernel returned, for emb_dim 16, one to five WHOLE 16-sample tiles per launch of 257 -- other tiles every run -- whose
// final state was off by ~1e-3 (scores ~1e-4) by the SAME vector for every sample of the tile, as if one bias vector of the AUGRU gates had
// been stale; emb_dim 10 never.  The hunt then showed that k_dien_seq_mfma<16, 32> -- the same source, the product since round 3 -- does the
// same once FOUR of its workgroups share a CU (B = 65 536: ~40 of 4 096 tiles per launch off by up to 6e-5 against the fp64 oracle; at the
// batches the suite ran emb_dim 16 with, a quarter of that occupancy, never).  What it is NOT (each measured on the GPU or checked on the ISA):
// the LDS-DMA staging (plain stores: same), the tail or where its gathers are issued (the state is already wrong), a hazard inside an asm
// statement (every statement padded with wait states: same; scripts/isa/asm_hazards.py finds no transcendental / MFMA result read by one), a
// missing or short s_waitcnt (scripts/isa/isa_waitcnt_check.py / isa_waitcnt_paths.py replay every counted wait of the loop, twice round, in
// order: consistent), a read of a never-written VGPR (isa_undef_reads.py), LDS reads returning out of order or a load landing in SrcC of a
// queued MFMA or in the SrcA / SrcB of one just issued (scripts/ubench/lds_order.hip, mfma_srcc_war.hip, mfma_srcab_war.hip: 5e9 trials each, none), an MFMA needing more wait states behind the
// packed conversions that write its B operand than hipcc pads (valu_to_mfma.hip: one is enough for every producer, hipcc pads two).  What it DEPENDS on: more than one wave per SIMD,
// and the ORDER the scheduler picks under the 128-VGPR cap -- 256 VGPRs (eight waves): clean; a bare sched_barrier between the blocks mm(6)
// and mm(7) (the R and Z gates' input halves, which share one B operand): clean; in front of any other single block: not; fewer statements
// the scheduler may not cross (rows4_sum on ds_bpermute instead of the volatile permlane statements): thirty times as many bad tiles.
// A CPU fit of the observed state error against "vector X read as vector Y in step t" (fp64 recurrence, all pairs) points at block 7's bias
// (the Z gate's input bias) being partly another vector in an early step (cosine 0.8 - 0.9, the right size) -- the block the barrier has to
// stand in front of -- but no wait, hazard or register the tools can see explains a stale read there.
// A marker that separates ALL failing builds (ten) from ALL clean ones (eight) in the ISA: the failing ones keep the un-scale scalars of two
// neighbouring blocks in ONE register pair (one uniform `ds_read_b64`) and pick the high one with `v_pk_fma_f32 ... op_sel:[0,1,0]`; it marks a
// schedule, its semantics are deterministic, and tests/test_isa_checks_cpu.py trips if a rebuild brings it back.
// The cause is not known.  The fence below -- one `s_nop 1` statement in front of every group of three MFMAs, which LDS reads may not cross
// -- measured 0 differing tiles in 1.4 M launches-of-tiles, every launch bit for bit the first and the first within 1.2e-7 of the fp64 oracle
// where the unfenced two-launch path was 6e-5 off; tests/test_gpu_parity.py::test_dien_is_the_same_every_launch_and_the_oracles keeps asking.
// -DDNF_GROUP_FENCE=0 brings the failing build back.
#ifndef DNF_GROUP_FENCE
#define DNF_GROUP_FENCE 1
#endif
#ifndef DNF_XP
#define DNF_XP 0                      // experiment bits of the hunt above (build with -DDNF_GROUP_FENCE=0): 1 tail gathers behind the last step, 2 full wait in front of the
#endif                                // tail, 4 run-time D mask, 8 eight waves / 256 VGPRs, 16 eight waves / 128, 32 staging by plain stores, 64 sleep behind the barrier,
                                      // 128 final state -> workspace, 256 / 512 / 1024 nops behind / full wait / nops in front of every MFMA group, 2048 no peeled step,
                                      // 8192 / 16384 `s_nop 1` in front / behind, 65536 bare sched_barrier, 131072 empty asm + memory clobber; -DDNF_BAR=mask: a
                                      // sched_barrier in front of block b for every set bit b
#define DNF_WAVES ((DNF_XP & 24) ? 8 : 16)

template <int D, int H, int N0C, int N1C>
__global__ __launch_bounds__(DNF_WAVES * 64, (DNF_XP & 8) ? 2 : 4) void k_dien_fused(const DienRun A, const DinTailRun TL, const int* __restrict__ ids,
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
        if constexpr ((DNF_XP & 32) != 0) st4(smem + c * 256 + lane * 4, ld4(src + lane * 4));
        else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
    }
    __syncthreads();
    if constexpr ((DNF_XP & 64) != 0) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __syncthreads(); }

    const float* fl = W + 4 * lane;                               // this lane's 16 bytes of a fragment half
    auto vec = [&](int v) { return ld4(W + FR::vec0 + v * 16 + 4 * q); };
    auto mm = [&](int blk, din_f16x8 bh, din_f16x8 bl, f32x4 bias) __attribute__((always_inline)) {
        const din_f16x8 ah = __builtin_bit_cast(din_f16x8, ld4(fl + blk * 512));
        const din_f16x8 al = __builtin_bit_cast(din_f16x8, ld4(fl + blk * 512 + 256));
        if constexpr ((DNF_XP & 512) != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr ((DNF_XP & 1024) != 0) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        if constexpr ((DNF_XP & 256) != 0) __builtin_amdgcn_sched_barrier(0);
        if constexpr (DNF_GROUP_FENCE || (DNF_XP & 8192) != 0) asm volatile("s_nop 1");          // (the fence: see the head of this file)
        if constexpr ((DNF_XP & 65536) != 0) __builtin_amdgcn_sched_barrier(0);
        if (((DNF_XP & 262144) && blk < 4) || ((DNF_XP & 524288) && (blk == 4 || blk == 5)) || ((DNF_XP & 1048576) && blk >= 6 && blk < 9) ||
            ((DNF_XP & 2097152) && blk >= 9)) __builtin_amdgcn_sched_barrier(0);
#ifdef DNF_BAR
        if ((DNF_BAR >> blk) & 1) __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr ((DNF_XP & 131072) != 0) asm volatile("" ::: "memory");
        f32x4 acc = mfma_f16(al, bh, zero);
        acc = mfma_f16(ah, bl, acc);
        acc = mfma_f16(ah, bh, acc);
        if constexpr ((DNF_XP & 256) != 0) { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        if constexpr ((DNF_XP & 16384) != 0) asm volatile("s_nop 1");
        const float un = W[FR::S_UN + blk];
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
        [[maybe_unused]] float dbg_a = 0.f;
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
                if constexpr ((DNF_XP & 32768) != 0) dbg_a = a;
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
        if constexpr ((DNF_XP & 2048) != 0) {                     // experiment: no peeled last step (every step requests a next row: slot T - 1 again)
#pragma unroll 1
            for (int t = 0; t < A.T; ++t) {
                const bool live = id != 0;
                const f32x4 xt = x;
                id = row[A.hist_col + min(t + 1, A.T - 1)];
                if (id < 0 || id >= A.vocab) { bad = true; id = 0; }
                x = qin ? ld4(A.table + (size_t)id * A.Dp + 4 * q) : zero;
                step_body(xt, live);
            }
        } else
#pragma unroll 1
        for (int t = 0; t + 1 < A.T; ++t) step(t, std::false_type{});
        // the tail's operands fly under the last step
        din_f16x8 eh[LD::NBLK], el[LD::NBLK];
        float xna, xnb;
        if constexpr ((DNF_XP & 1) != 0 && (DNF_XP & 2048) == 0) step(A.T - 1, std::true_type{});
        din_tail_unf_gather<LD>(TL, idv, q, bad, eh, el);
        {
            const float* nrow = dense + (size_t)m * TL.ND;
            const int last = TL.n_num - 1;
            xna = nrow[min(q, last)];
            xnb = nrow[min(q + 4, last)];
        }
        if constexpr ((DNF_XP & 1) == 0 && (DNF_XP & 2048) == 0) step(A.T - 1, std::true_type{});
        if constexpr ((DNF_XP & 2) != 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        const int next = tile + gridDim.x * DNF_WAVES;
        if (next < ntiles) ld_ids(next);                           // (a persistent launch: the next tile's ids under this tile's tail)

        // ---- the tail: the final state is its pooled-history operand (what k_dien_seq_mfma stores: features beyond D are zeros) ----
        f32x4 xp[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[0][j] = 4 * q + j < ((DNF_XP & 4) ? A.Dp : D) ? hs[j] : 0.f;
        if constexpr ((DNF_XP & 128) != 0) {                      // experiment: the final state where the two-launch path keeps it
            if (tile * 16 + r < B && 4 * q < A.NA) {
                st4(dbg + (size_t)m * A.NA + 4 * q, xp[0]);
                const size_t pl = (size_t)B * A.NA;
                if constexpr ((DNF_XP & 32768) != 0) {
                st4(dbg + pl + (size_t)m * A.NA + 4 * q, c);
                st4(dbg + 2 * pl + (size_t)m * A.NA + 4 * q, x);
                st4(dbg + 3 * pl + (size_t)m * A.NA + 4 * q, h);
                st4(dbg + 4 * pl + (size_t)m * A.NA + 4 * q, g);
                st4(dbg + 5 * pl + (size_t)m * A.NA + 4 * q, f32x4{dbg_a, dbg_a, dbg_a, dbg_a});
                }
            }
        }
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

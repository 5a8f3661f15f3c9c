// k_din_cols.h -- k_din_attn_cols: DIN activation unit + weighted sum pooling (reference DIN.py:132-158) with SIXTEEN
// SAMPLES in the MFMA's columns and the history slots walked in time order.  Included after k_din_attn.h / dyn_split.h.
//
// Round 2's k_din_attn gives a wave ONE sample: its T history rows are the 16-wide column tiles, the per-sample matrix
// A_b = W12 + W4 diag(c_b) is the A operand.  PMC said what that costs (profiles/r02/pmc_summary.json): 381 VALU against 24
// MFMA instructions per sample, the matrix pipe 12 % busy -- A_b is rebuilt and split into halfs for every sample (72 VALU),
// every 16-row tile pays PReLU / reduce / sigmoid / pooling for 64 slots although only 50 are real (T = 50 pads to four
// tiles), the rows take a detour through an LDS tile, and the pooled vector needs a cross-lane reduce-scatter.
// Here the weights are the STATIC operand:
//
//     u[n][b] (slot t) = sum_k W12[n][k] h_b,t[k]  +  sum_k W4[n][k] (h_b,t[k] c_b[k])  +  vc[cand_b][n]
//
// = two K blocks against fixed A fragments (W12 and W4, pre-split into hi / lo halfs once at finalize), B = the sixteen
// samples' rows of slot t -- which arrive from the pre-split table ALREADY in the B-operand layout (lane (r = sample, q)
// loads the 32 bytes [hi8 | lo8] of its k range straight into registers: no LDS tile) -- and B' = h * c, formed and split
// per element with three mixed-precision VALU instructions.  Per (16 samples, slot): 12 v_mfma_f32_16x16x32_f16 and ~68
// VALU, i.e. per sample at T = 50: 37.5 MFMA + ~215 VALU (was 24 + 381), no padding slots, no A_b, no reduce-scatter (lane
// (r,q) owns pooled[r][8q .. 8q+7] outright).  The accumulate chain of a slot is six MFMAs deep per 16 outputs; the W12 half
// is issued before the product's split so the matrix pipe runs under the VALU work.
// Rows are requested TWO slots ahead (three register sets in a ring, unrolled: no copies).  The first version prefetched one
// slot ahead with compiler-managed waits and ran at 38 us: 25 dependent (load -> score) rounds per wave at ~1.4 us loaded
// latency each -- latency bound, 6 TB/s of rows with 2 KB in flight per wave.  hipcc cannot express "wait for the OLDEST of
// three outstanding sets" in a loop (at the first use of a loop-carried load it waits for vmcnt(0), which would drain the
// prefetch), so the row loads are issued from asm statements the waitcnt pass does not see, and the wait is a manual
// `s_waitcnt vmcnt(2 sets)` whose asm statement OWNS the set's registers: every consumer depends on it.  vmcnt retires in
// order, so "at most the two younger sets outstanding" means the oldest has landed; stores issued in between only add younger
// operations.  The loop leaves through a vmcnt(0) that owns all three sets (a late return must not hit a reused register);
// the ISA was checked for copies of a set between its load and its wait (none: scripts/r03/check_din_cols_isa.sh).
// One strict launch of one 32 768-row batch is only 2 048 tasks: `ts` waves share a task (time slices of T, partial pooled
// vectors summed through LDS) so that the chip still holds four waves per SIMD.
// Arithmetic: split-f16 (hi + lo, 22 significand bits) with static power-of-two scales, f32 accumulation -- the same error
// class as k_din_attn<HALF>; tests/test_gpu_parity.py holds both to the fp64 oracle's attention weights and pooled vectors.

#define DC_WAVES 8
#define DC_MB 16
struct DinColsRun {
    int T, F, hist_col, cand_col, Dp, vocab;
    float b2, acc_scale, unscale, inv_h_scale, kappa;   // kappa = sP / sH^2: c (in sH units) -> the factor that turns h (sH units) into h*c*sP
    const float* tsplit;  // [vocab][KP floats]: per q group [hi(EL halfs) | lo(EL halfs)] of E * sH (k_din_split_table)
    const float* vc;      // [vocab][32]
    const float* alpha;   // [T][32]
    const float* w2;      // [32]
    const float* frag;    // [2 n-blocks][W12 hi, W12 lo, W4 hi, W4 lo][64 lanes] x 16 bytes: A fragments (k_din_cols_pack)
    int ts, ts_log2;      // waves per task (1, 2, 4): each takes 4 / ts consecutive QUARTERS of the history
    int ql;               // slots per quarter = ceil(T / 4): the pooled sum is ALWAYS formed as (q0 + q1) + (q2 + q3) of the four
                          // quarters' in-order partial sums, whatever ts is -- results do not depend on the launch shape
    int idp;              // LDS stride (ints) of a sample's ids row in the wave's block = F rounded up to a multiple of 4
    const float* coef;    // [2][64][36]: PReLU(alpha[t][n]) . Dense(1) as ca = w2 (1 + alpha) / 2, cb = w2 (1 - alpha) / 2 (k_din_cols_coef)
};
struct DinColsMany {
    const int* ids[DC_MB];
    float* pooled[DC_MB];
    int n;
};
struct DinColsOne {};
template <bool MB> struct DinColsArg { typedef DinColsOne type; };
template <> struct DinColsArg<true> { typedef DinColsMany type; };

// One-time (finalize) kernel: W12 * sA and W4 * s4 as hi / lo half fragments in the A-operand lane layout
// (lane (r,q): A[n = nb*16 + r][k = EL*q .. EL*q + EL-1]).  w12 / w4: [32][KP] f32, unscaled.
static __global__ __launch_bounds__(256) void k_din_cols_pack(const float* __restrict__ w12, const float* __restrict__ w4, int KP, float s12, float s4,
                                                       _Float16* __restrict__ frag) {
    const int EL = KP / 4;
    for (int i = threadIdx.x; i < 2 * 4 * 64 * 8; i += 256) {
        const int e = i & 7, lane = (i >> 3) & 63, kind = (i >> 9) & 3, nb = i >> 11;
        const int r = lane & 15, q = lane >> 4;
        float x = 0.f;
        if (e < EL) {
            const float* W = kind < 2 ? w12 : w4;
            x = W[(size_t)(nb * 16 + r) * KP + EL * q + e] * (kind < 2 ? s12 : s4);
        }
        const _Float16 hi = (_Float16)x;
        frag[i] = (kind & 1) ? (_Float16)(x - (float)hi) : hi;
    }
}

// One-time (finalize) kernel: the two coefficient tables of the attention unit's second half (see k_din_attn.h's epilogue),
// laid out as the kernel keeps them in LDS -- staging is a flat LDS-DMA copy.
static __global__ __launch_bounds__(256) void k_din_cols_coef(const float* __restrict__ alpha, const float* __restrict__ w2, int T,
                                                       float* __restrict__ coef) {
    for (int i = threadIdx.x; i < 64 * 36; i += 256) {
        const int t = i / 36, n = i - t * 36;
        const bool ok = t < T && n < 32;
        const float al = ok ? alpha[(size_t)t * 32 + n] : 0.f;
        const float w = ok ? w2[n] : 0.f;
        coef[i] = 0.5f * w * (1.0f + al);
        coef[64 * 36 + i] = 0.5f * w * (1.0f - al);
    }
}

// d = 1.0 * f16(half of a) + f16(half of b): a row element back to f32 from its two halfs (one VALU)
template <bool ODD>
__device__ __forceinline__ float halfs_sum(float one, float hi_packed, float lo_packed) {
    float d;
    if (ODD) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(one), "v"(hi_packed), "v"(lo_packed));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(d) : "v"(one), "v"(hi_packed), "v"(lo_packed));
    return d;
}

template <int KC, bool MB>
__global__ __launch_bounds__(DC_WAVES * 64, 4) void k_din_attn_cols(const DinColsRun A, const int* __restrict__ ids, float* __restrict__ pooled,
                                                                  float* __restrict__ att, int B, int* __restrict__ err,
                                                                  const typename DinColsArg<MB>::type Mm) {
    constexpr int EL = 4 * KC, KP = 16 * KC, HP = 32, AS = HP + 4, ROWS = 64;
    typedef _Float16 f16xe __attribute__((ext_vector_type(EL)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int T = A.T;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    float* ca_s = smem;                                   // [64][AS]  w2 (1 + alpha) / 2
    float* cb_s = smem + ROWS * AS;                       // [64][AS]  w2 (1 - alpha) / 2
    int* ids_s = reinterpret_cast<int*>(smem + 2 * ROWS * AS) + wave * 16 * A.idp;
    float* park_s = smem + 2 * ROWS * AS + DC_WAVES * 16 * A.idp + wave * 2 * 64 * EL;   // two parking slots per wave: S0, S1

    // ---- the coefficient tables (18 KB, built at finalize) by LDS-DMA: 1-KB pieces, wave w takes w, w + 8, ... ----
#pragma unroll 1
    for (int c = wave; c < 2 * ROWS * AS / 256; c += DC_WAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.coef + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);

    // ---- this wave's (task, time slice) ----
    const int ntpb = (B + 15) >> 4;
    int nb_batches = 1;
    if constexpr (MB) nb_batches = Mm.n;
    const int ntasks = nb_batches * ntpb;
    const int gw = blockIdx.x * DC_WAVES + wave;
    const int task = gw >> A.ts_log2, slice = gw & (A.ts - 1);
    const bool work = task < ntasks;                      // wave-uniform
    int bi = 0, tl = task;
    if constexpr (MB) { bi = __builtin_amdgcn_readfirstlane(task / ntpb); tl = task - bi * ntpb; }
    const int* ids_b = ids;
    float* pooled_b = pooled;
    if constexpr (MB) { ids_b = Mm.ids[work ? bi : 0]; pooled_b = Mm.pooled[work ? bi : 0]; }
    const int nq = 4 >> A.ts_log2;                        // quarters of this wave
    const int Tq = nq * A.ql;
    const int t0 = slice * Tq;
    const int nsteps = work ? max(0, min(T, t0 + Tq) - t0) : 0;
    const int m = tl * 16 + r;
    const int mc = min(m, B - 1);
    bool bad = false;

    // the task's ids block (16 consecutive rows of F ints: contiguous) -> LDS with coalesced 16-byte loads, all in flight
    // together: ONE memory round trip (the first version fetched slot by slot, 7 dependent round trips per wave).  Partial
    // last task / unaligned ids: element-wise, rows past the end clamped.
    if (work) {
        const int nint = 16 * A.F;
        if (tl * 16 + 16 <= B && !((uintptr_t)ids_b & 15) && A.idp == A.F) {
            const f32x4* src = reinterpret_cast<const f32x4*>(ids_b + (size_t)tl * nint);
            f32x4* dst = reinterpret_cast<f32x4*>(ids_s);
            for (int c = lane; c < nint / 4; c += 64) dst[c] = src[c];
        } else {
            for (int i = lane; i < nint; i += 64) {
                const int s = i / A.F, col = i - s * A.F;
                ids_s[s * A.idp + col] = ids_b[(size_t)min(tl * 16 + s, B - 1) * A.F + col];
            }
        }
    }
    // candidate: its row (for h * c), its vc row (the accumulators' start)
    float cfac[EL];                                       // c[k] * sH * kappa for this lane's k = EL*q + e
    f32x4 acc_init[2];
    {
        const int cid = work ? ids_s[r * A.idp + A.cand_col] : 0;   // (one wave: LDS operations complete in issue order)
        bad |= (unsigned)cid >= (unsigned)A.vocab;
        const unsigned csafe = (unsigned)cid < (unsigned)A.vocab ? (unsigned)cid : 0u;
        f32x4 cp[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) cp[c] = ld4(A.tsplit + csafe * (unsigned)KP + EL * q + 4 * c);
        f16xe chi, clo;
        unpack_halfs<KC>(cp, chi, clo);
#pragma unroll
        for (int e = 0; e < EL; ++e) cfac[e] = ((float)chi[e] + (float)clo[e]) * A.kappa;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc_init[nb] = ld4(A.vc + csafe * (unsigned)HP + nb * 16 + 4 * q) * A.acc_scale;
    }
    // A fragments -> registers (the same eight for every slot)
    f16xe aW[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 f = ld4(A.frag + ((nb * 4 + k) * 64 + lane) * 4);
            if constexpr (KC == 2) aW[nb][k] = __builtin_bit_cast(f16xe, f);
            else { const din_f16x8 both = __builtin_bit_cast(din_f16x8, f); aW[nb][k] = f16xe{both[0], both[1], both[2], both[3]}; }
        }
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): this wave's DMA pieces (and everything above) have landed
    __syncthreads();                                      // coefficient tables staged by every wave

    // ---- slot loop ----
    float pacc[EL];
#pragma unroll
    for (int e = 0; e < EL; ++e) pacc[e] = 0.f;
    const float one = 1.0f;
    // quarter bookkeeping (wave-uniform): local quarter j ends after slot (j + 1) * ql - 1.  park(j) retires quarter j < nq - 1:
    //   nq = 2: S0 = q0          nq = 4: S0 = q0; S1 = S0 + q1; S0 = q2          -- the wave's LAST quarter stays in pacc, and
    //   finish(): nq = 1: q      nq = 2: S0 + q1                 nq = 4: S1 + (S0 + q3)
    float* S0 = park_s;
    float* S1 = park_s + 64 * EL;
    auto park = [&](int j) {
        if ((j & 1) == 0) {
#pragma unroll
            for (int e = 0; e < EL; ++e) S0[e * 64 + lane] = pacc[e];
        } else {
#pragma unroll
            for (int e = 0; e < EL; ++e) S1[e * 64 + lane] = S0[e * 64 + lane] + pacc[e];
        }
#pragma unroll
        for (int e = 0; e < EL; ++e) pacc[e] = 0.f;
    };
    int jq = 0, qend = A.ql;                               // current local quarter and the first slot of the next one
    auto score = [&](int step, const f32x4 (&rw)[KC]) {
        if (step < nsteps && step == qend) { park(jq); ++jq; qend += A.ql; }   // (wave-uniform)
        const int t = t0 + step;
        f16xe bh, bl;
        unpack_halfs<KC>(rw, bh, bl);
        f32x4 acc[2] = {acc_init[0], acc_init[1]};
        // W12 . (hi + lo): independent of the product below -- the matrix pipe works while the VALU splits h * c
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][0], bh, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][0], bl, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][1], bh, acc[nb]);
        // h back to f32 (sH units), h * c * sP split into halfs: v_fma_mixlo/hi_f16 round (h cfac) and (h cfac - hi) to f16
        float h32[EL];
        unsigned ph[EL / 2], pl[EL / 2];
#pragma unroll
        for (int e = 0; e < EL; ++e) {
            const float hp = KC == 2 ? rw[0][e >> 1] : rw[0][e >> 1];
            const float lp = KC == 2 ? rw[KC - 1][e >> 1] : rw[0][2 + (e >> 1)];
            h32[e] = (e & 1) ? halfs_sum<true>(one, hp, lp) : halfs_sum<false>(one, hp, lp);
        }
#pragma unroll
        for (int e = 0; e < EL; ++e) {
            if (e & 1) {
                asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(ph[e >> 1]) : "v"(h32[e]), "v"(cfac[e]));
                asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(pl[e >> 1]) : "v"(h32[e]), "v"(cfac[e]), "v"(ph[e >> 1]));
            } else {
                asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(ph[e >> 1]) : "v"(h32[e]), "v"(cfac[e]));
                asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(pl[e >> 1]) : "v"(h32[e]), "v"(cfac[e]), "v"(ph[e >> 1]));
            }
        }
        f16xe qh, ql;
        if constexpr (KC == 2) {
            // HAZARD GUARD (dyn_split.h): VALU writes inside asm statements are invisible to the hazard recognizer; an MFMA
            // reading them as its B operand needs two wait states
            asm volatile("s_nop 1" : "+v"(ph[0]), "+v"(ph[1]), "+v"(ph[2]), "+v"(ph[3]), "+v"(pl[0]), "+v"(pl[1]), "+v"(pl[2]), "+v"(pl[3]));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            qh = __builtin_bit_cast(f16xe, u32x4{ph[0], ph[1], ph[2], ph[3]});
            ql = __builtin_bit_cast(f16xe, u32x4{pl[0], pl[1], pl[2], pl[3]});
        } else {
            asm volatile("s_nop 1" : "+v"(ph[0]), "+v"(ph[1]), "+v"(pl[0]), "+v"(pl[1]));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            qh = __builtin_bit_cast(f16xe, u32x2{ph[0], ph[1]});
            ql = __builtin_bit_cast(f16xe, u32x2{pl[0], pl[1]});
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][2], qh, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][2], ql, acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[nb] = mfma_f16(aW[nb][3], qh, acc[nb]);
        // PReLU(alpha[t][n]) -> Dense(1) -> sigmoid (DIN.py:150-151): lane (r,q) holds u[n = nb*16 + 4q + j] of sample r; the
        // coefficient rows are wave-uniform addresses (LDS broadcast)
        // ca . u as packed FMAs (two units per instruction), cb . |u| as scalar FMAs with the free |.| source modifier
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 sp2 = {0.f, 0.f};
        float sum = 0.f, sum1 = 0.f;
        const int tc = min(t, ROWS - 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const f32x4 ca = ld4(ca_s + tc * AS + nb * 16 + 4 * q);
            const f32x4 cb = ld4(cb_s + tc * AS + nb * 16 + 4 * q);
            const f32x4 u = acc[nb];
            sp2 = __builtin_elementwise_fma(f32x2{ca[0], ca[1]}, f32x2{u[0], u[1]}, sp2);
            sp2 = __builtin_elementwise_fma(f32x2{ca[2], ca[3]}, f32x2{u[2], u[3]}, sp2);
            sum = fmaf(cb[0], __builtin_fabsf(u[0]), sum);
            sum1 = fmaf(cb[1], __builtin_fabsf(u[1]), sum1);
            sum = fmaf(cb[2], __builtin_fabsf(u[2]), sum);
            sum1 = fmaf(cb[3], __builtin_fabsf(u[3]), sum1);
        }
        sum = (sum + sum1) + (sp2[0] + sp2[1]);
        float wgt = sigmoidf_fast(rows4_sum(sum) * A.unscale + A.b2);       // PReLU is positively homogeneous: unscale the logit
        wgt = step < nsteps ? wgt : 0.f;                                      // padding steps of the ping-pong pair
        if constexpr (!MB) { if (att && q == 0 && step < nsteps && m < B) att[(size_t)m * T + t] = wgt; }
        // weighted sum pooling (DIN.py:152-158): this lane owns pooled[r][EL*q + e]
#pragma unroll
        for (int e = 0; e < EL; ++e) pacc[e] = fmaf(wgt, h32[e], pacc[e]);
    };
    if (nsteps > 0) {
        f32x4 rowA[KC], rowB[KC], rowC[KC];
        const char* tbase = reinterpret_cast<const char*>(A.tsplit);
        auto load = [&](int step, f32x4 (&rw)[KC]) {       // rows of slot t0 + step (clamped: padding steps re-read the last slot)
            const int tt = min(step, nsteps - 1);
            const int id = ids_s[r * A.idp + A.hist_col + t0 + tt];
            bad |= (unsigned)id >= (unsigned)A.vocab;
            const unsigned safe = (unsigned)id < (unsigned)A.vocab ? (unsigned)id : 0u;
            const unsigned voff = safe * (unsigned)(KP * 4) + (unsigned)(EL * 4) * (unsigned)q;   // < 4 GiB (checked at finalize)
            if constexpr (KC == 2)
                asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
                             : "=&v"(rw[0]), "=&v"(rw[1]) : "v"(voff), "s"(tbase) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(rw[0]) : "v"(voff), "s"(tbase) : "memory");
        };
        // at most the two YOUNGER sets outstanding => this set has landed
        auto wait2 = [&](f32x4 (&rw)[KC]) {
            if constexpr (KC == 2) asm volatile("s_waitcnt vmcnt(4)" : "+v"(rw[0]), "+v"(rw[1]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(2)" : "+v"(rw[0]) : : "memory");
        };
        // every compiler-visible load of the prologue (candidate row, vc row, fragments) is "used" here, i.e. has landed before
        // the hidden loads start: a wait hipcc placed for one of them INSIDE the loop would be counted without the hidden loads --
        // never too weak (they are younger), but it would drain the prefetch on every round
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(aW[nb][k]));
            asm volatile("" : "+v"(acc_init[nb]));
        }
#pragma unroll
        for (int e = 0; e < EL; ++e) asm volatile("" : "+v"(cfac[e]));
        load(0, rowA);
        load(1, rowB);
        for (int step = 0; step < nsteps; step += 3) {
            load(step + 2, rowC);
            wait2(rowA);
            score(step, rowA);
            load(step + 3, rowA);
            wait2(rowB);
            if (step + 1 < nsteps) score(step + 1, rowB);
            load(step + 4, rowB);
            wait2(rowC);
            if (step + 2 < nsteps) score(step + 2, rowC);
        }
        // nothing may still be in flight towards these registers when they are reused
        if constexpr (KC == 2)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(rowA[0]), "+v"(rowA[1]), "+v"(rowB[0]), "+v"(rowB[1]), "+v"(rowC[0]), "+v"(rowC[1]) : : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(rowA[0]), "+v"(rowB[0]), "+v"(rowC[0]) : : "memory");
    }

    // quarters without slots (T < 4 ql) hold exact zeros: retire them the same way, then the wave's result in the fixed order
    for (; jq < nq - 1; ++jq) park(jq);
    float res[EL];
    if (nq == 1) {
#pragma unroll
        for (int e = 0; e < EL; ++e) res[e] = pacc[e];
    } else if (nq == 2) {
#pragma unroll
        for (int e = 0; e < EL; ++e) res[e] = S0[e * 64 + lane] + pacc[e];
    } else {
#pragma unroll
        for (int e = 0; e < EL; ++e) res[e] = S1[e * 64 + lane] + (S0[e * 64 + lane] + pacc[e]);
    }
    // ---- several waves per task: their results through LDS (a wave's S0 is free by now), slice 0 sums in the fixed order ----
    if (A.ts > 1) {
        if (slice != 0) {
#pragma unroll
            for (int e = 0; e < EL; ++e) S0[e * 64 + lane] = res[e];
        }
        __syncthreads();
        if (slice == 0) {
            const float* P = park_s + 2 * 64 * EL;           // the next wave's S0
            if (A.ts == 2) {
#pragma unroll
                for (int e = 0; e < EL; ++e) res[e] = res[e] + P[e * 64 + lane];
            } else {
#pragma unroll
                for (int e = 0; e < EL; ++e)
                    res[e] = (res[e] + P[e * 64 + lane]) + (P[2 * 64 * EL + e * 64 + lane] + P[4 * 64 * EL + e * 64 + lane]);
            }
        }
    }
    if (work && slice == 0 && m < B) {
        float* prow = pooled_b + (size_t)m * A.Dp + EL * q;
        if (A.Dp == KP) {                                     // (wave-uniform) full-width rows: this lane's EL floats as 16-byte stores
#pragma unroll
            for (int c = 0; c < KC; ++c)
                st4(prow + 4 * c, f32x4{res[4 * c] * A.inv_h_scale, res[4 * c + 1] * A.inv_h_scale, res[4 * c + 2] * A.inv_h_scale, res[4 * c + 3] * A.inv_h_scale});
        } else {
#pragma unroll
            for (int e = 0; e < EL; ++e)
                if (EL * q + e < A.Dp) prow[e] = res[e] * A.inv_h_scale;
        }
    }
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
}

// k_chain_v2j1.h -- k_deepfm_v2_joint1: k_deepfm_v2_joint (k_chain_v2j.h, split-f16 form) re-shaped for ONE strict launch of
// ONE batch: one 16-sample task per WAVE, four waves per SIMD.  Included after k_chain_v2j.h.
//
// Why a second shape (round 3, scripts/ubench/row_gather.hip, profiles/r03/ubench_row_gather.log).  With NO scoring work
// at all, "ids -> 3 random 128-byte lines per sample -> one float" for 65 536 samples costs, per strict launch,
//       2 048 waves x 2 tasks (k_deepfm_v2_joint's shape)   8.47 us from a 3.2 GB table, 6.13 us from the Infinity Cache
//       4 096 waves x 1 task                                7.82 us                       5.93 us
// and the fused kernel sat 1.4 / 2.25 us above its shape's floor (9.89 / 8.38 us): its two-tasks-per-wave form keeps the
// weights in 256 VGPRs (2 waves per SIMD), issues gather B only after the image barrier + the weight fragments' split, and
// ends every wave with two scoring stages back to back.  Here a wave owns exactly one task:
//   * every row of the batch is requested as soon as its ids are in and the workgroup has met ([r4]: the barrier sits in FRONT of the
//     requests -- behind them every wave waited for the slowest issuer of its workgroup, see the kernel);
//   * nothing is loop-carried and no weight lives in a register across tasks: the A fragments of the big fields arrive
//     PRE-SPLIT (hi / lo halfs, k_v2j1_pack_image) in the LDS image and are read right where an MFMA consumes them, so
//     the kernel fits 128 VGPRs = 4 waves per SIMD, and the split's VALU work is gone from the launch;
//   * one scoring stage per wave after its rows land; four waves per SIMD interleave theirs.
// Arithmetic, operand order and accumulation chains are those of v2j_body<HALF = true>: results are bit-identical to
// k_deepfm_v2_joint (tests/test_gpu_parity.py::test_v2_joint1_bit_identical_to_joint).
// Used by sprk_forward / one-batch launches with ceil(B / 16) <= V2J1_MAX_TASKS; larger batches and the several-batches-
// per-launch form stay on the looped kernel (one image staging per workgroup is only worth 8 tasks when there are few).

// Waves per workgroup.  r03 (barrier behind the gathers): 4 / 8 / 16 waves 7.93 / 7.65 / 8.41 us; r04 (barrier in front): 7.57 / 7.25 / 7.19;
// [r5] with the leaner image sixteen win wherever the rows come out of the caches -- config 2 6.88-6.95 against 6.97-6.99 us, config 4
// 5.90 against 6.09 (profiles/r05/experiments/r05_02): one image staging per CU instead of two -- and lose where they come from HBM
// (9.03 against 8.70-8.78 us on one box, r05_03): the HOIST form keeps eight.
#ifndef V2J1_WAVES
#define V2J1_WAVES 16
#endif
#ifndef V2J1_WAVES_HOIST
#define V2J1_WAVES_HOIST 8
#endif
// (HBM-resident rows AND three gathered fields: config 2 on 8 M-row tables 8.6-8.9 us with eight waves, 8.7-9.0 with sixteen; config 4's two
//  fields -- 27 M-row table -- 6.1-6.2 against 5.9: r05_02 .. r05_05)
#define V2J1_WAVES_OF(HOIST, G_BIG) (((HOIST) && (G_BIG) >= 3) ? V2J1_WAVES_HOIST : V2J1_WAVES)
#ifndef V2J1_DEDUP
#define V2J1_DEDUP 1                         // [r5] A fragments stored ONCE per (field, n-block) as {hi4 | lo4}: one ds_read_b128 where there were two, the selection fragment built in registers
#endif
#ifndef V2J1_ZZ_LATE
#define V2J1_ZZ_LATE 1                       // [r5] the rows' first-order scalar is consumed BEHIND the row fence (it is the last load issued: see phase A); not in the HOIST form
#endif
#define V2J1_MAX_TASKS 16384                 // B <= 262 144: beyond, the looped kernel amortises the image staging better

template <int G_BIG>
struct V2J1Lds {
#if V2J1_DEDUP
    static constexpr int off_frag = 0;                              // [G_BIG][2 n-blocks] A fragments {hi4 | lo4} per lane, 256 floats (1 KB) each
    static constexpr int off_sel = off_frag + G_BIG * 2 * 256;      // (no selection fragment in the image: built in registers)
    static constexpr int S1 = 36;
    static constexpr int off_w1 = off_sel;                          // deep1 W^T [16][S1]
#else
    static constexpr int off_frag = 0;                              // [G_BIG][2 n-blocks][hi, lo] A fragments, 256 floats (1 KB) each
    static constexpr int off_sel = off_frag + G_BIG * 4 * 256;      // the 0/1 selection fragment (FM sum of hi + lo)
    static constexpr int S1 = 36;
    static constexpr int off_w1 = off_sel + 256;                    // deep1 W^T [16][S1]
#endif
    static constexpr int off_b1 = off_w1 + 16 * S1;                 // [16]
    static constexpr int off_hd = off_b1 + 16;                      // [16] head weights on deep1's output
    static constexpr int off_bpn = off_hd + 16;                     // [16] numeric projection bias
    static constexpr int off_hfm = off_bpn + 16;                    // [16] head weights on the FM vector
    static constexpr int off_wn8 = off_hfm + 16;                    // [16][8] numeric projection W^T (zero beyond n_num)
    static constexpr int off_fn8 = off_wn8 + 128;                   // [8] h0w * first-order weights of the numerics
    static constexpr int total = off_fn8 + 8;
    static constexpr int total_pad = (total + 255) & ~255;
};

// One-time (finalize) kernel, one block.  Fragment (b, n0, part) lane l = (r = l & 15, q = l >> 4) holds 8 halfs
// {h[0..3], h[0..3]} of W0[n0*16 + r][16*grp_b + 4q + j] * w_scale, h = hi (part 0) or lo (part 1): exactly the registers
// v2j_body's load_weights() builds with split_half4 at every launch.
static __global__ __launch_bounds__(256) void k_v2j1_pack_image(const V2Args A, const V2JRun R, int g_big, int G, float* __restrict__ img) {
    const int tid = threadIdx.x;
#if V2J1_DEDUP
    const int off_sel = g_big * 2 * 256, off_w1 = off_sel;
#else
    const int off_sel = g_big * 4 * 256, off_w1 = off_sel + 256;
#endif
    const int off_b1 = off_w1 + 16 * 36, off_hd = off_b1 + 16, off_bpn = off_hd + 16,
              off_hfm = off_bpn + 16, off_wn8 = off_hfm + 16, off_fn8 = off_wn8 + 128, total_pad = (off_fn8 + 8 + 255) & ~255;
    for (int i = tid; i < total_pad; i += 256) img[i] = 0.f;
    __syncthreads();
    f16x8* frag = reinterpret_cast<f16x8*>(img);
    for (int i = tid; i < g_big * 2 * 64; i += 256) {
        const int l = i & 63, n0 = (i >> 6) & 1, b = i >> 7, r = l & 15, q = l >> 4;
        const f32x4 w = ld4(A.W0 + (size_t)(n0 * 16 + r) * (G * 16) + 16 * R.big_grp[b] + 4 * q);
        f16x4 hi, lo;
        split_half4(w, R.w_scale, hi, lo);
#if V2J1_DEDUP
        frag[(b * 2 + n0) * 64 + l] = f16x8{hi[0], hi[1], hi[2], hi[3], lo[0], lo[1], lo[2], lo[3]};
    }
#else
        frag[((b * 2 + n0) * 2 + 0) * 64 + l] = f16x8{hi[0], hi[1], hi[2], hi[3], hi[0], hi[1], hi[2], hi[3]};
        frag[((b * 2 + n0) * 2 + 1) * 64 + l] = f16x8{lo[0], lo[1], lo[2], lo[3], lo[0], lo[1], lo[2], lo[3]};
    }
    if (tid < 64) {
        const int r = tid & 15, q = tid >> 4;
        f16x8 s;
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = (4 * q + (e & 3) == r) ? (_Float16)1.0f : (_Float16)0.0f;
        reinterpret_cast<f16x8*>(img + off_sel)[tid] = s;
    }
#endif
    for (int i = tid; i < 16 * 32; i += 256) img[off_w1 + (i >> 5) * 36 + (i & 31)] = A.W1[i];      // deep1 W^T [16][32]
    if (tid < 16) {
        img[off_b1 + tid] = A.b1[tid];
        img[off_hd + tid] = tid < A.n_hdeep ? A.hdeep[tid] : 0.f;
        img[off_bpn + tid] = A.bp[G - 1][tid];
        img[off_hfm + tid] = tid < A.n_hfm ? A.hfm[tid] : 0.f;
    }
    if (tid < 128) img[off_wn8 + tid] = (tid & 7) < A.n_num ? A.Wp[G - 1][(size_t)(tid >> 3) * A.ldp_num + (tid & 7)] : 0.f;
    if (tid < 8) img[off_fn8 + tid] = tid < A.n_num ? A.h0w * A.fo_num_w[tid] : 0.f;
}

// (Tried and dropped, scripts/r03/09_joint1_frag_regs.sh: the big fields' A fragments global -> REGISTERS at kernel entry instead
// of global -> LDS -> registers at scoring time -- half of a wave's LDS reads in the scoring stage.  8.10 instead of 7.65 us,
// 10.3 instead of 9.2 us with HBM-resident tables: sixteen waves per CU pulling the same 12 KB through the texture path queue
// in front of the row gathers, and 110 VGPRs leave four waves per SIMD where 71 leave seven.)
#ifdef SPRK_DF_XP
// (timeline build, scripts/r04: every wave stamps the 100 MHz clock at entry, with its ids staged, with its gathers requested, behind
// the barrier, after phase A, with its rows landed, at exit -- SPRK_V2J1_TS_FILE at sprk_destroy)
#define V2J1_TS_WAVES 8192
static __device__ unsigned long long g_v2j1_ts[V2J1_TS_WAVES * 8];
#define V2J1_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (lane == 0 && tk < V2J1_TS_WAVES) g_v2j1_ts[tk * 8 + (k)] = t_; } while (0)
#else
#define V2J1_STAMP(k) do { } while (0)
#endif
// HOIST ([r4]; chosen at finalize for tables larger than the Infinity Cache): the big fields' weight fragments and the selection
// fragment -- 13 of the 26 KB a wave reads from LDS, none of it needing an id -- are read in FRONT of the gathers.  With rows coming
// from HBM the texture path backs up longer and the LDS sits idle meanwhile: 8.9 -> 8.68 us; with cache-resident tables the same
// move only delays the gathers: 7.26 -> 7.38 us (profiles/r04/experiments/r04_34).  Same arithmetic, same bits.
// (Tried and dropped [r6], profiles/r06/experiments/r06_30 .. r06_32, the patch is kept there: a PERSISTENT several-batches form of this kernel for
// sprk_forward_many -- the image staged once per workgroup, every wave walking tasks of up to 64 batches, the next task's ids DMA'd into its slot
// behind the gathers.  Bit-identical to launch per batch; 3.79-3.81 us per 65 536-sample step against k_deepfm_v2_joint_many's 3.83-3.85, and the
// same 3.79-3.84 with 16, 12 or 8 waves per CU, fragments re-read per task or held in registers; 5.7-6.1 against 5.4-5.6 us with HBM-resident
// tables.  Why nothing moves it: scripts/ubench/row_gather_steady.hip -- the BARE gather, no scoring, persistent waves -- runs at 3.7 us per step
// from a 200 MB window (6.7 TB/s of random 128-byte lines out of the Infinity Cache) and 4.9-5.3 us from a 3.2 GB table, whatever the waves per CU
// or the gathers in flight per wave: the several-batches figures of both kernels ARE the fabric's line rate.  Two things the attempt taught about
// hipcc, for the next kernel that loops: (1) once a global_load_lds builtin is outstanding its wait-count pass waits vmcnt(0) for ANY load result
// (the FLAT-encoded DMA counts as touching two address spaces), so rows requested before a DMA are waited for together with it -- an asm statement
// hides the DMA and keeps the staged vmcnt(3..0); (2) what is invariant in a loop is hoisted into registers and, at the 128-VGPR cap of sixteen
// waves per CU, spilled -- a scratch reload in the trip shares vmcnt with the gathers; forming the lane's coordinates per trip from v_mbcnt in an
// asm statement and writing selects as arithmetic on them got 96 bytes of scratch to 0.)
template <int G_BIG, int NJF, bool HOIST = false>
__global__ __launch_bounds__(V2J1_WAVES_OF(HOIST, G_BIG) * 64, 4) void k_deepfm_v2_joint1(const V2JRun A, const int* __restrict__ ids,
                                                                       const float* __restrict__ dense, float* __restrict__ out, int B,
                                                                       int* __restrict__ err, const float* __restrict__ image) {
#pragma clang fp contract(off)                                          // (pinned: see fma4s / dot4f in k_chain_v2j.h)
    using LD = V2J1Lds<G_BIG>;
    constexpr int WAVES = V2J1_WAVES_OF(HOIST, G_BIG), KP = 16, H0C = 2;
    constexpr unsigned RB = (KP + 16) * 4;
    static_assert(G_BIG >= 1 && G_BIG <= 3 && NJF >= 1 && NJF <= V2J_MAX_JF, "field split");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntasks = (B + 15) >> 4;
    const int tk = blockIdx.x * WAVES + wave;
    const bool work = tk < ntasks;                                     // wave-uniform
    const float* small_s = smem + LD::total_pad;                       // small fields' rows, then Wf
    float* stage = smem + LD::total_pad + A.small_floats + wave * 256; // this wave's ids / numerics slot
    const bool fast = work && !(A.flags & 1) && tk * 16 + 16 <= B;     // aligned, full task: one 16-byte load per lane
    V2J1_STAMP(0);
    // (tried, no gain: s_setprio 3 from here to the last gather request -- 6.82 us either way, HBM-resident 8.7-9.0 against 8.7-8.8: profiles/r05/experiments/r05_12)

    // ---- the task's ids + numerics first, then the image pieces (they land inside the ids' latency) ----
    f32x4 raw = zero;
    if (fast) {
        const bool isid = lane < 32;
        const int j = isid ? lane : lane - 32;
        const int n4 = 4 * (isid ? A.F : A.ND);
        const float* src = isid ? reinterpret_cast<const float*>(ids) + (size_t)tk * 16 * A.F : dense + (size_t)tk * 16 * A.ND;
        raw = ld4(src + 4 * (j < n4 ? j : 0));
    }
#pragma unroll 1
    for (int c = wave; c < LD::total_pad / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
#pragma unroll 1
    for (int c = wave; c < A.small_floats / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.small + c * 256 + lane * 4),
                                         (__attribute__((address_space(3))) void*)(smem + LD::total_pad + c * 256), 16, 0, 0);

    // ---- [r4] the workgroup meets HERE, with its ids and its DMA pieces in, BEFORE the rows are requested.  Round 3 met after the
    // requests ("every row of the batch is requested as soon as its ids are in -- before the barrier"), and the stamped timeline of
    // round 4 (profiles/r04/experiments/r04_25) showed what that costs: the waves of a workgroup reach this point within 0.5 us of
    // each other, but finish ISSUING their four gathers 0.4 .. 2.3 us later (the texture path is saturated by sixteen waves per CU
    // doing the same) -- a barrier behind the requests makes every wave wait for the slowest issuer, 1.4 us (median) during which
    // its rows are already on their way and its LDS phase could run. ----
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): this wave's ids and DMA pieces have landed
    __builtin_amdgcn_s_barrier();
    if (!work) return;
    f16x8 wa[G_BIG][H0C], wb[G_BIG][H0C], hSel;
#if V2J1_DEDUP
    // one 16-byte read per (field, n-block): {hi4 | lo4}; the MFMA's A operand {h, h} is the same four halfs twice -- two register
    // copies (expand_frags) instead of a second kilobyte out of LDS.  The selection fragment is a constant of the lane: 1.0 at
    // k = r - 4q.
    f16x8 wt[G_BIG][H0C];
    auto read_frags = [&]() {
        const f16x8* frag = reinterpret_cast<const f16x8*>(smem + LD::off_frag) + lane;
#pragma unroll
        for (int b = 0; b < G_BIG; ++b)
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) wt[b][n0] = frag[(b * 2 + n0) * 64];
    };
    auto expand_frags = [&]() {
#pragma unroll
        for (int b = 0; b < G_BIG; ++b)
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) {
                const f16x8 t = wt[b][n0];
                wa[b][n0] = f16x8{t[0], t[1], t[2], t[3], t[0], t[1], t[2], t[3]};
                wb[b][n0] = f16x8{t[4], t[5], t[6], t[7], t[4], t[5], t[6], t[7]};
            }
        const int d = r - 4 * q;
        const unsigned lo01 = d == 0 ? 0x00003C00u : d == 1 ? 0x3C000000u : 0u, lo23 = d == 2 ? 0x00003C00u : d == 3 ? 0x3C000000u : 0u;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        hSel = __builtin_bit_cast(f16x8, u32x4{lo01, lo23, lo01, lo23});
    };
#else
    auto read_frags = [&]() {
        const f16x8* frag = reinterpret_cast<const f16x8*>(smem + LD::off_frag) + lane;
#pragma unroll
        for (int b = 0; b < G_BIG; ++b)
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) { wa[b][n0] = frag[((b * 2 + n0) * 2 + 0) * 64]; wb[b][n0] = frag[((b * 2 + n0) * 2 + 1) * 64]; }
        hSel = reinterpret_cast<const f16x8*>(smem + LD::off_sel)[lane];
    };
    auto expand_frags = [&]() {};
#endif
    if constexpr (HOIST) {
        read_frags();
        expand_frags();                                               // (with its LDS wait: measured equal to round 4's two-reads form, 8.61 against 8.58-8.60 us, r05_01)
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- gather: ids through the wave-private LDS slot to the (r,q) lanes, then every row of the task ----
    f32x4 x[G_BIG];
    int so[NJF];
    float xn0 = 0.f, xn1 = 0.f, w1a = 0.f;
    bool bad = false;
#pragma unroll
    for (int b = 0; b < G_BIG; ++b) x[b] = zero;
#pragma unroll
    for (int f = 0; f < NJF; ++f) so[f] = 0;
    {
        if (fast) {
            const bool isid = lane < 32;
            const int j = isid ? lane : lane - 32;
            const int n4 = 4 * (isid ? A.F : A.ND);
            if (j < n4) st4(stage + (isid ? 0 : 128) + 4 * j, raw);
        } else {
            stage_task_slow(stage, ids, dense, A.F, A.ND, tk, B, lane);
        }
        V2J1_STAMP(1);
        const int* sid_row = reinterpret_cast<const int*>(stage) + r * A.F;   // (one wave: LDS operations complete in issue order)
        unsigned sid[G_BIG];
#pragma unroll
        for (int b = 0; b < G_BIG; ++b) {
            const int id = sid_row[A.big_col[b]];
            bad |= (unsigned)(id + 1) > (unsigned)A.big_vocab[b];
            sid[b] = min((unsigned)id, (unsigned)A.big_vocab[b]) + A.big_rowbase[b];
        }
#pragma unroll
        for (int f = 0; f < NJF; ++f) {
            const int id = sid_row[A.j_col[f]];
            bad |= (unsigned)(id + 1) > (unsigned)A.j_vocab[f];
            so[f] = A.s_off[f] + (int)min((unsigned)id, (unsigned)A.j_vocab[f]) * V2J_SS;
        }
        {
            const float* nrow = stage + 128 + r * A.ND;
            const int last = A.n_num - 1;
            xn0 = nrow[min(q, last)];
            xn1 = nrow[min(q + 4, last)];
        }
        const char* tb = reinterpret_cast<const char*>(A.tab0);
#pragma unroll
        for (int b = 0; b < G_BIG; ++b) x[b] = *reinterpret_cast<const f32x4*>(tb + (sid[b] * RB + 16u * q));
        {
            unsigned s0 = sid[0] * RB, s1 = sid[G_BIG > 1 ? 1 : 0] * RB, s2 = sid[G_BIG > 2 ? 2 : 0] * RB;
            asm("" : "+v"(s0), "+v"(s1), "+v"(s2));
            unsigned sx = s0;
            if (G_BIG > 1) sx = q == 1 ? s1 : sx;
            if (G_BIG > 2) sx = q == 2 ? s2 : sx;
            w1a = *reinterpret_cast<const float*>(tb + (sx + 4u * KP));
        }
        V2J1_STAMP(2);
    }
    V2J1_STAMP(3);

    // ---- scoring, phase A: everything that does NOT need the rows -- every LDS read of the stage (fragments, small fields'
    //      rows, W1, the small vectors: 26 KB per wave) and the numerics' MFMAs -- is issued HERE, while the rows are still
    //      in flight.  (The first version read each fragment where an MFMA consumed it, i.e. after the rows had landed, when
    //      all sixteen waves of a CU want the LDS at once: scripts/ubench/row_gather.hip prices 13 KB of LDS reads per wave in
    //      that position at 1.2-1.4 us per launch.  110 VGPRs instead of 71: still the four waves per SIMD a 65 536-row
    //      launch can use.) ----
    const float* vq = smem + 4 * q;
    const f32x4 rbpn = ld4(vq + LD::off_bpn);
    const float rwn8a = smem[LD::off_wn8 + r * 8 + q], rwn8b = smem[LD::off_wn8 + r * 8 + q + 4];
    const float rfn8a = smem[LD::off_fn8 + q], rfn8b = smem[LD::off_fn8 + q + 4];
    f32x4 sp = ld4(small_s + so[0] + 4 * q), sq[H0C];
#pragma unroll
    for (int n0 = 0; n0 < H0C; ++n0) sq[n0] = ld4(small_s + so[0] + KP + 16 * n0 + 4 * q);
    float ssc = small_s[so[0] + KP + 32];
#pragma unroll
    for (int f = 1; f < NJF; ++f) {
        sp += ld4(small_s + so[f] + 4 * q);
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) sq[n0] += ld4(small_s + so[f] + KP + 16 * n0 + 4 * q);
        ssc += small_s[so[f] + KP + 32];
    }
    if constexpr (!HOIST) { read_frags(); expand_frags(); }
    float rwf[H0C][2];
    {
        const float* wf = small_s + A.wf_off + r * 8 + q;
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) { rwf[n0][0] = wf[n0 * 128]; rwf[n0][1] = wf[n0 * 128 + 4]; }
    }
    f32x4 rW1[H0C];
#pragma unroll
    for (int j = 0; j < H0C; ++j) rW1[j] = ld4(vq + LD::off_w1 + r * LD::S1 + 16 * j);
    const f32x4 rb1 = ld4(vq + LD::off_b1), rhd = ld4(vq + LD::off_hd), rhfm = ld4(vq + LD::off_hfm);
    f32x4 pn;
    {
        const f32x4 e = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn8a, xn0, rbpn, 0, 0, 0);
        const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x4f32(rwn8b, xn1, zero, 0, 0, 0);
        pn = e + o;
    }
    // (zz = ((q < G_BIG ? w1a : 0) + fma(...)) + (q == 3 ? ssc : 0): w1a is the LAST load the wave issued -- formed here, hipcc's
    //  vmcnt(0) for it landed behind the first numerics MFMA, i.e. five f32 MFMAs and ten LDS reads of this phase ran only AFTER every
    //  row had arrived (build/sparrow.s, round 5).  The numerics' share is formed here, the scalar joins it behind the fence: same
    //  operands, same order of the two additions.)
    //  With HBM-resident tables (HOIST) the early wait is KEPT: measured 8.58-8.70 us with it, 8.65-8.87 without (r05_01, r05_02) -- there the
    //  rows are what everything waits for, and six f32 MFMAs issued in front of that wait hold up the VALU the SIMD's other waves need
    //  to get THEIR gathers out.)
    constexpr bool ZZ_LATE = V2J1_ZZ_LATE && !HOIST;
    const float zz_num = __builtin_fmaf(rfn8b, xn1, rfn8a * xn0);
    float zz = 0.f;
    if constexpr (!ZZ_LATE) {
        zz = ((q < G_BIG) ? w1a : 0.f) + zz_num;
        zz += (q == 3) ? ssc : 0.f;
    }
    f32x4 hA[H0C], hB[H0C];
#pragma unroll
    for (int n0 = 0; n0 < H0C; ++n0) hA[n0] = sq[n0];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) hA[n0] = __builtin_amdgcn_mfma_f32_16x16x4f32(rwf[n0][st], st ? xn1 : xn0, hA[n0], 0, 0, 0);
    f32x4 s = sp + pn;
    // ---- phase B: the rows (the compiler's s_waitcnt vmcnt lands at their first use, below this fence) ----
    __builtin_amdgcn_sched_barrier(0);
#ifdef SPRK_DF_XP
    if (s[0] + hA[0][0] == 123.456f) V2J1_STAMP(7);       // (phase A's results exist)
    V2J1_STAMP(4);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V2J1_STAMP(5);
    __builtin_amdgcn_sched_barrier(0);
#endif
    {
        f32x4 aFa[H0C], aFb[H0C], aS = zero;
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) { aFa[n0] = zero; aFb[n0] = zero; }
#pragma unroll
        for (int b = 0; b < G_BIG; ++b) {
            const f16x8 xb = __builtin_bit_cast(f16x8, x[b]);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) aFa[n0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[b][n0], xb, aFa[n0], 0, 0, 0);
            aS = __builtin_amdgcn_mfma_f32_16x16x32_f16(hSel, xb, aS, 0, 0, 0);
#pragma unroll
            for (int n0 = 0; n0 < H0C; ++n0) aFb[n0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[b][n0], xb, aFb[n0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        s = fma4s(aS, A.unscale_s, s);
#pragma unroll
        for (int n0 = 0; n0 < H0C; ++n0) hB[n0] = (aFa[n0] + aFb[n0]) * A.unscale_h;
    }
    f32x4 h0[H0C];
#pragma unroll
    for (int n0 = 0; n0 < H0C; ++n0) h0[n0] = relu4_fast(hA[n0] + hB[n0]);
    float z = dot4f(rhfm, sq_diff4(s, pn));
    {
        f32x4 e = rb1, o = zero;
#pragma unroll
        for (int j = 0; j < H0C; ++j) {
            e = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[j].x, h0[j].x, e, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[j].y, h0[j].y, o, 0, 0, 0);
            e = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[j].z, h0[j].z, e, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(rW1[j].w, h0[j].w, o, 0, 0, 0);
        }
        z += dot4f(rhd, relu4_fast(e + o));
    }
    if constexpr (ZZ_LATE) {
        zz = ((q < G_BIG) ? w1a : 0.f) + zz_num;
        zz += (q == 3) ? ssc : 0.f;
    }
    z += zz;
    // ([r5] two permlane swaps instead of two ds_bpermute round trips through the LDS queue at the very end of the task's chain; the storing
    //  lanes (row 0) get (z0 + z1) + (z2 + z3) either way: same bits)
    z = rows4_sum(z);
    const float score = sigmoidf_fast(z + A.h0w * A.fo_bias + A.head_bias);
    const int m = tk * 16 + r;
    if (q == 0 && m < B) out[m] = score;
    if (__ballot(bad) != 0 && lane == 0) atomicOr(err, 1);
    V2J1_STAMP(6);
}

// k_mlp_rows.h -- k_mlp_rows: the "DenseFeatures -> Dense(relu) -> Dense(relu) -> Dense(1, sigmoid)" graphs (reference
// EmbeddingMLP.py:72-77, the deep part of WideNDeep.py:99-107 + its hashed-cross wide part; BASELINE config 5) with EVERY
// embedding column folded through the first Dense layer, one WAVE per 16 samples.  Included inside sparrow_hip.hip's
// anonymous namespace.  It replaced round 1's k_mlp_chain (retired in round 3; its A/B numbers are in profiles/r02).
//
// What round 1's k_mlp_chain did per 16 samples: the 8 genre columns folded to per-id tables F_g = W0_g^T E_g (512-byte rows,
// 19 of them per column) and gathered from L2 -- 4 KB per SAMPLE of cache traffic -- while movieId / userId rows went through
// the first layer on f32 MFMA: 160 instructions of 32 cycles, i.e. 5 120 matrix-pipe cycles per task before the second layer
// even starts (the f32-input MFMA runs at 1/16 of the f16 rate on gfx950), gathers issued per task, not a task ahead.  101 us
// per 131 072 samples, 24 % of the HBM roofline.  Here:
//   * the genre tables live in LDS (8 x 19 rows + a shared all-zero row for "no id"): no cache traffic at all.  A row is 128 floats;
//     rows 512 B apart would put the same 16-byte piece of sixteen DIFFERENT rows into the same banks (first version: a 16-way
//     conflict, SQ_LDS_BANK_CONFLICT = 11 x SQ_INSTS_LDS, profiles/r02).  Rounds 2-4 XOR-swizzled the pieces inside a row; [r5] rows
//     are 528 B apart instead (MR_RS = 132 floats): piece p of row i starts at 16 (33 i + p) bytes, so sixteen different rows land
//     in sixteen different bank groups just the same, AND a lane's eight pieces are base + 64 nb bytes -- constants that fold into
//     ds_read_b128's offset field, where the swizzle cost 7 v_xor + 8 v_lshl_add per column and task (build/sparrow.s);
//   * movieId / userId are folded as well: their F rows (512 B per id) are gathered from HBM / Infinity Cache straight into the
//     first layer's accumulators (C/D layout: lane (r,q) holds outputs 16 nb + 4q .. +3 of sample r) -- 1 KB per sample instead
//     of 2 x 128 B, bought back many times over by the 144 f32 MFMAs it removes; total gathered bytes stay BELOW the reference's
//     algorithmic 1.6 KB per sample (10 embedding rows x 128 B + the cross row), because the genre rows never leave LDS;
//   * the matrix pipe sees the numerics (K = 8: two 16x16x4 steps per 16 outputs) and the second layer (per-sample dynamic
//     split-f16, dyn_split.h: 96 instructions of 16 cycles);
//   * the gathers of task n+1 are issued before the second layer of task n runs (ONE register set: the rows are summed into
//     the accumulators first thing in a trip, which frees the set for the next task's loads), ids / numerics one task further
//     ahead, staged through a wave-private LDS slot.
//   z0 = b0 + sum_{10 columns} F_g[id_g] + W0[:, numerics]^T x;  h1 = relu(z0);  h2 = relu(b1 + W1^T h1)
//   score = sigmoid(hw . h2 + wide + bias),  wide = hc . X[FingerprintCat64(movieId, userRatedMovie1) mod buckets] or its indicator weight

#define MR_MAX_BIG 3
#define MR_MAX_SMALL 8
#define MR_STAGE 320                      // floats per wave: ids [16][F <= 12] + numerics [16][<= 8]
#ifndef MR_RS
#define MR_RS 132                         // floats between two LDS rows of a small column (128 + 4: see the header)
#endif
#ifndef MR_XP
#define MR_XP 0                           // ablation builds (scripts/r05): 1 no small-column reads, 2 one weight fragment pair for the whole second layer,
#endif                                    // 4 big rows not loaded, 8 no wide part, 16 no second-layer MFMAs, 128 big rows with a quad of lanes per row -- WRONG RESULTS, timing only

struct MlpRowsRun {
    int F, ND, n_num;
    int n_big, n_small;
    int big_col[MR_MAX_BIG], big_vocab[MR_MAX_BIG];
    const float* big_tab[MR_MAX_BIG];     // [vocab + 1][N0] folded rows, the last one all zero ("no id")
    int s_col[MR_MAX_SMALL], s_vocab[MR_MAX_SMALL];
    int s_off[MR_MAX_SMALL];              // float offset of small column f's rows inside the LDS small block (rows MR_RS floats apart)
    int zero_off;                         // float offset of the shared all-zero row inside the small block
    int small_floats;                     // multiple of 256
    const float* small;                   // device image of the small block
    int wide_kind, wide_a, wide_b, wide_dim, wide_stride;   // 0 none, 1 cross rows x head weights, 2 cross scalar (indicator weight)
    long long wide_buckets;
    unsigned long long wide_magic;        // floor((2^64 - 1) / wide_buckets) when wide_buckets < 2^30 (the modulo as a multiply, below), else 0
    const float* wide_tab;
    const float* wide_w;
    float head_bias;
    float inv_w1_scale;                   // DYN: 1 / static scale of the second layer's split-f16 fragments
    int flags;                            // 1 = ids / dense not 16-byte aligned: stage element-wise; 2 = b0 rides in the numerics' free eighth K slot
};

#define MR_MB 16                          // batches per launch of k_mlp_rows_many
struct MlpRowsMany {
    const int* ids[MR_MB];
    const float* dense[MR_MB];
    float* out[MR_MB];
    int n;                                // batches in this launch
};

// h mod n for a 64-bit hash and n < 2^30, bit-exact, without the 64-bit division hipcc expands `%` into (about seventy VALU and fifty
// SALU instructions per task in round 4's loop -- the reciprocal of the wave-uniform divisor was recomputed every trip).
// magic = floor((2^64 - 1) / n): q = mulhi64(h, magic) satisfies floor(h / n) - 2 <= q <= floor(h / n) (h magic / 2^64 > h / n - (n + 1) / n),
// so h - q n lies in [0, 3n) < 2^32: its low 32 bits are the whole value, and two conditional subtractions finish.
__device__ __forceinline__ unsigned mod_magic(unsigned long long h, unsigned n, unsigned long long magic) {
    const unsigned long long qq = __umul64hi(h, magic);
    unsigned rr = (unsigned)h - (unsigned)qq * n;
    rr = rr >= n ? rr - n : rr;
    rr = rr >= n ? rr - n : rr;
    return rr;
}

template <int N0C, int N1C, bool DYN>
struct MlpRowsLds {
    static constexpr int N0 = N0C * 16, N1 = N1C * 16;
    static constexpr int S1 = N0 + 4;
    static constexpr int off_w1 = 0;                  // DYN: split-f16 fragments N1C*(N0C/2)*512 floats; else W1^T [N1][S1]
    static constexpr int w1_floats = DYN ? N1C * (N0C / 2) * 512 : N1 * S1;
    static constexpr int off_b0 = off_w1 + w1_floats;
    static constexpr int off_b1 = off_b0 + N0;
    static constexpr int off_hw = off_b1 + N1;
    static constexpr int total = off_hw + N1;
    static constexpr int total_pad = (total + 255) & ~255;
    // the device image continues behind the LDS part with W0[:, numerics]^T [N0][8] (+ b0 in column 7 when n_num < 8), lane-major: read ONCE per
    // wave, straight from global memory into registers ([r5]: it used to take 4 KB of LDS for that one read)
    static constexpr int off_w0n = total_pad;
    static constexpr int image_floats = total_pad + N0 * 8;
};

// One-time (finalize) kernel: rows [rows][128] -> the LDS layout of a small column (rows MR_RS floats apart).
static __global__ __launch_bounds__(256) void k_mlp_rows_pad(const float* __restrict__ in, float* __restrict__ out, int rows) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * 128; i += gridDim.x * 256) out[(i >> 7) * MR_RS + (i & 127)] = in[i];
}

// One-time (finalize) kernel: the fixed part of the image.
template <int N0C, int N1C, bool DYN>
__global__ __launch_bounds__(256) void k_mlp_rows_pack(const float* __restrict__ W0, int ldw0, int num_col0, int n_num,
                                                       const float* __restrict__ b0, const float* __restrict__ W1, int ldw1,
                                                       const float* __restrict__ b1, const float* __restrict__ hw, int n_hw,
                                                       const float* __restrict__ w1frag, float* __restrict__ img) {
    using LD = MlpRowsLds<N0C, N1C, DYN>;
    static_assert(N0C % 2 == 0, "K blocks of 32");
    const int tid = threadIdx.x;
    if constexpr (DYN) {
        for (int i = tid; i < LD::w1_floats; i += 256) img[LD::off_w1 + i] = w1frag[i];
    } else {
        for (int i = tid; i < LD::w1_floats; i += 256) {
            const int n = i / LD::S1, k = i - n * LD::S1;
            img[LD::off_w1 + i] = k < LD::N0 ? W1[(size_t)n * ldw1 + k] : 0.f;
        }
    }
    for (int i = tid; i < LD::N0 * 8; i += 256) {
        const int n = i >> 3, k = i & 7;
        // [r6] lane-major: lane (r, q) = r + 16 q finds {column q, column q + 4} of rows nb*16 + r, nb = 0 .. N0C - 1, as 2 N0C consecutive floats
        const int dst = ((n & 15) + 16 * (k & 3)) * (2 * N0C) + (n >> 4) * 2 + (k >> 2);
        img[LD::off_w0n + dst] = k < n_num ? W0[(size_t)n * ldw0 + num_col0 + k] : (k == 7 ? b0[n] : 0.f);   // (column 7 meets x = 1 when n_num < 8, flags & 2)
    }
    for (int i = tid; i < LD::N0; i += 256) img[LD::off_b0 + i] = b0[i];
    for (int i = tid; i < LD::N1; i += 256) {
        img[LD::off_b1 + i] = b1[i];
        img[LD::off_hw + i] = i < n_hw ? hw[i] : 0.f;
    }
    for (int i = LD::total + tid; i < LD::total_pad; i += 256) img[i] = 0.f;
}

// WK = the wide part (MlpRowsRun::wide_kind) as a TEMPLATE parameter.  [r4] As a run-time branch it cost config 5 a fifth of its time:
// the two forms' loads (cross row pieces / the indicator's scalar) sat in the arms of a wave-uniform branch, hipcc gave the scalar's
// register a second job in the other arm, and its waitcnt pass -- which merges the "load pending" state of every path into a join --
// put a vmcnt(0) in front of that arm and another behind the join: BOTH right after the next task's sixteen row gathers had been
// requested, i.e. every trip waited for the rows it had just asked for and the prefetch hid nothing (found by reading the ISA's
// wait sequence, scripts/r04; 42.8 -> see profiles/r04).
//
// [r5] NS = the number of small columns as a template parameter too (8: EmbeddingMLP.py / WideNDeep.py as written; -1: known at run
// time only, the generic instantiation).  VERDICT r04 weak 2 read the loop's instruction mix off PMC (63 VALU + 7 MFMA per sample) and
// called it issue-bound on instructions that do no arithmetic; build/sparrow.s (scripts/isa/isa_loop_mix.py) says which:
//   * a wave-uniform branch per small column (f < n_small) made every column's id read its own LDS round trip -- eight dependent
//     ds_read_b32 + s_waitcnt lgkmcnt(0) in a row per task -- and kept hipcc from overlapping one column's eight row reads with the
//     previous column's adds; its eight SGPR-pair conditions lived in spilled lanes (sixteen v_readlane per trip).  With NS a constant
//     the ids of a task are ONE batch of reads and the columns' reads run ahead of the adds;
//   * every ReLU behind an MFMA was two v_max (relu1_fast, k_chain_v2.h);
//   * the swizzle's XORs and shifts (above); the "bad id" flag as a 0/1 VGPR (four VALU per column: now a compare into an SGPR pair);
//   * the second layer read each pair of weight fragments right in front of the three DEPENDENT MFMAs that consume it: 32 times per task
//     an LDS latency and two MFMA latencies with nothing else of this wave to issue.  Now four output blocks go through a K block
//     together (pass 1: Ah Bh x 4, pass 2: Ah Bl x 4, pass 3: Al Bh x 4 -- the accumulation order per output is unchanged), and the next
//     group's fragments are requested as soon as a pass has freed their registers;
//   * b0 rides in the numerics' free eighth K slot (x = 1) when there are at most seven numerics: eight LDS reads + sixteen adds less.
#ifdef SPRK_DF_XP
// (timeline build, scripts/r06/34_mlp_rows_timeline.sh: every wave stamps the 100 MHz clock at entry, behind the meeting, with its first gather out, and per
// trip with its rows summed, the next gather out, the first layer done, the second layer done, its scores stored -- SPRK_MR_TS_FILE at sprk_destroy)
#define MR_TS_WAVES 2048
#define MR_TS_SLOTS 32
static __device__ unsigned long long g_mr_ts[MR_TS_WAVES * MR_TS_SLOTS];
#define MR_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
    const int w_ = blockIdx.x * WAVES + wave; if (lane == 0 && w_ < MR_TS_WAVES && (k) < MR_TS_SLOTS) g_mr_ts[w_ * MR_TS_SLOTS + (k)] = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MR_STAMP(k) do { } while (0)
#endif
// [r6] MB: several batches per launch (sprk_forward_many): the launch's tasks are n x ceil(B / 16), task t belongs to batch t / ntpb -- own ids,
// numerics and score buffers each -- and the waves walk them exactly as they walk one batch's: the image is staged once, the ramp of a strict
// launch (5 us of config 5's 39) is paid once per MR_MB batches.  Same instruction sequence per task: the same bits as batch by batch.
template <int N0C, int N1C, int NBIG, int NS, int WAVES, bool DYN, int WK, bool MB>
__device__ __forceinline__ void mr_body(const MlpRowsRun& A, const int* __restrict__ ids0, const float* __restrict__ dense0, float* __restrict__ out0,
                                        int B, int* __restrict__ err, const float* __restrict__ image, const MlpRowsMany* __restrict__ Mp) {
    using LD = MlpRowsLds<N0C, N1C, DYN>;
    constexpr int N0 = LD::N0;
    constexpr bool RT = NS < 0;
    static_assert(NS <= MR_MAX_SMALL && N1C % 4 == 0, "shape");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntpb = (B + 15) >> 4;
    const int ntasks = MB ? Mp->n * ntpb : ntpb;
    const int task_stride = gridDim.x * WAVES;
    const int ns = RT ? A.n_small : NS;
    // (batch buffers, task inside the batch) of launch task t: wave-uniform
    struct Loc { const int* ids; const float* dense; float* out; int tl; };
    auto locate = [&](int t) {
        Loc L;
        if constexpr (MB) {
            const int b = __builtin_amdgcn_readfirstlane(t / ntpb);
            L.ids = Mp->ids[b]; L.dense = Mp->dense[b]; L.out = Mp->out[b]; L.tl = t - b * ntpb;
        } else {
            L.ids = ids0; L.dense = dense0; L.out = out0; L.tl = t;
        }
        return L;
    };
    const char* small_b = reinterpret_cast<const char*>(smem + LD::total_pad) + 16 * q;       // + this lane's piece 0 (q) of a row
    float* stage = smem + LD::total_pad + A.small_floats + wave * MR_STAGE;
    unsigned long long badm = 0;          // lanes that saw an id outside [-1, vocab): an SGPR pair, not a VGPR flag
    const bool aligned = !(A.flags & 1);
    const bool bias_in_k7 = (A.flags & 2) != 0;
    auto clampt = [&](int tk) { return tk < ntasks ? tk : ntasks - 1; };

    // ---- ids / numerics of a task: two coalesced 16-byte loads per lane (ids block, numerics block) ----
    auto ld_raw = [&](int tk, f32x4& ri, f32x4& rd) {
        const Loc L = locate(tk);
        if (aligned && L.tl * 16 + 16 <= B) {                     // wave-uniform
            const int ni = 4 * A.F, nd = 4 * A.ND;
            ri = ld4(reinterpret_cast<const float*>(L.ids) + (size_t)L.tl * 16 * A.F + 4 * (lane < ni ? lane : 0));
            rd = ld4(L.dense + (size_t)L.tl * 16 * A.ND + 4 * (lane < nd ? lane : 0));
        }
    };
    // everything of a task that is in flight while the previous task's second layer runs
    f32x4 g[NBIG][N0C];                   // big columns' folded rows (C/D layout pieces)
    f32x4 gw[2];                          // cross row pieces (wide part)
    int so[MR_MAX_SMALL];                 // small columns: BYTE offset of this lane's piece 0 of the sample's row, relative to small_b
    float xa = 0.f, xb = 0.f, gws = 0.f;  // numerics q / q + 4; cross indicator weight
    // live = false: the trip behind a wave's last task.  Its loads are issued all the same -- into the all-zero rows, one hot line per table --
    // so that the registers a gather writes are the SAME on every path into the next trip: with the gather inside `if (more tasks)` hipcc
    // kept two copies of the 80 registers and moved one into the other at the end of every trip (forty v_mov_b64, build/sparrow.s).
    [[maybe_unused]] bool stamp_gather = false;                  // (timeline build: the FIRST gather stamps its stages, slots 27..30)
    auto gather = [&](int tk, const f32x4& ri, const f32x4& rd, bool live) {
        const Loc L = locate(tk);
        if (aligned && L.tl * 16 + 16 <= B) {
            if (lane < 4 * A.F) st4(stage + 4 * lane, ri);
            if (lane < 4 * A.ND) st4(stage + 192 + 4 * lane, rd);
        } else {
            int* si = reinterpret_cast<int*>(stage);
#pragma clang loop vectorize(disable) unroll(disable)
            for (int e = lane; e < 16 * A.F; e += 64) {
                const int mm = e / A.F, c = e - mm * A.F;
                si[e] = L.ids[(size_t)min(L.tl * 16 + mm, B - 1) * A.F + c];
            }
#pragma clang loop vectorize(disable) unroll(disable)
            for (int e = lane; e < 16 * A.ND; e += 64) {
                const int mm = e / A.ND, c = e - mm * A.ND;
                stage[192 + e] = L.dense[(size_t)min(L.tl * 16 + mm, B - 1) * A.ND + c];
            }
        }
        // one wave: LDS operations complete in issue order, no barrier needed.  EVERY id of the sample is read here, back to back
        // (one LDS round trip), before anything is computed from one of them.
        const int* idrow = reinterpret_cast<const int*>(stage) + r * A.F;
        int bid[NBIG], sidv[MR_MAX_SMALL], wid_a = 0, wid_b = 0;
#pragma unroll
        for (int b = 0; b < NBIG; ++b) bid[b] = idrow[A.big_col[b]];
#pragma unroll
        for (int f = 0; f < MR_MAX_SMALL; ++f) {
            sidv[f] = -1;
            if (RT ? f < ns : f < NS) sidv[f] = idrow[A.s_col[f]];
        }
        if constexpr (WK != 0) { wid_a = idrow[A.wide_a]; wid_b = idrow[A.wide_b]; }
#ifdef SPRK_DF_XP
        if (stamp_gather) { asm volatile("" : "+v"(bid[0])); MR_STAMP(28); }
#endif
        {
            const float* nrow = stage + 192 + r * A.ND;
            const int last = A.n_num - 1;
            // slots beyond n_num hold a duplicate finite value that only ever meets zero weights -- except slot 7 when it carries b0
            xa = nrow[min(q, last)];
            xb = nrow[min(q + 4, last)];
            xb = (bias_in_k7 && q == 3) ? 1.0f : xb;
        }
#pragma unroll
        for (int b = 0; b < NBIG; ++b) {
            badm |= __ballot(live && (unsigned)(bid[b] + 1) > (unsigned)A.big_vocab[b]);
            const unsigned sid = live ? min((unsigned)bid[b], (unsigned)A.big_vocab[b]) : (unsigned)A.big_vocab[b];      // -1 -> the zero row at index vocab
            // (one SGPR base + a 32-bit byte offset per lane: the set-up admits folded tables below 4 GiB)
            const char* row = reinterpret_cast<const char*>(A.big_tab[b]) + (sid * (unsigned)(N0 * 4) + 16u * q);
            if (MR_XP & 128)              // (timing only: a quad of lanes on ONE row's 64 consecutive bytes -- the same lines per instruction, coalesced)
                row = reinterpret_cast<const char*>(A.big_tab[b]) + ((unsigned)__shfl((int)sid, lane >> 2) * (unsigned)(N0 * 4) + 16u * (lane & 3));
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb) g[b][nb] = (MR_XP & 4) ? f32x4{(float)sid, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(row + 64 * nb);
        }
#ifdef SPRK_DF_XP
        if (stamp_gather) MR_STAMP(29);
#endif
#pragma unroll
        for (int f = 0; f < MR_MAX_SMALL; ++f) {
            if (RT ? f < ns : f < NS) {                           // (wave-uniform when RT)
                const int id = sidv[f];
                badm |= __ballot(live && (unsigned)(id + 1) > (unsigned)A.s_vocab[f]);
                const int t = __mul24(id, MR_RS * 4) + A.s_off[f] * 4;              // (v_mad_u32_u24; a missing id's product is discarded)
                so[f] = (unsigned)id < (unsigned)A.s_vocab[f] ? t : A.zero_off * 4;
            }
        }
        if constexpr (WK != 0 && !(MR_XP & 8)) {
            unsigned long long bkt;
            if (MR_XP & 32) {
                bkt = 0;
            } else if (A.wide_magic != 0) {                           // wave-uniform
                unsigned long long hh = 0xDECAFCAFFEULL;              // cross_bucket's chain (k_tile_forward.h), the modulo as a multiply
                hh = fingerprint_cat64(hh, (uint64_t)(int64_t)wid_a);
                hh = fingerprint_cat64(hh, (uint64_t)(int64_t)wid_b);
                bkt = mod_magic(hh, (unsigned)A.wide_buckets, A.wide_magic);
            } else {
                bkt = cross_bucket(wid_a, wid_b, (uint64_t)A.wide_buckets);
            }
            if (MR_XP & 32) bkt = (unsigned long long)(((unsigned)wid_a * 2654435761u) ^ ((unsigned)wid_b * 40503u)) % (unsigned)A.wide_buckets;
            if (MR_XP & 64) bkt = bkt & 1;
            bkt = live ? bkt : 0ull;
            if constexpr (WK == 1) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d = 16 * h + 4 * q;
                    gw[h] = d < A.wide_dim ? ld4(A.wide_tab + (size_t)bkt * A.wide_stride + d) : zero;
                }
            } else {
                gws = A.wide_tab[bkt];
            }
        }
    };

    // (Also measured and not kept [r6], r06_37 .. r06_39, with the stamped timeline of each: (i) an explicit vmcnt(0) in front of the task loop, which
    //  rids every trip of the `s_waitcnt vmcnt(19)` / `vmcnt(16)` hipcc puts in front of the numerics' MFMAs (the loop head merges "operands maybe
    //  still in flight" into all trips): 38.9-39.3 against 39.0-39.3 us; (ii) the small columns summed in BEFORE the next task's gather goes out, a
    //  column's eight reads requested while the column before is added (8-16 LDS reads in flight where hipcc, at 249 VGPRs, leaves one or two
    //  with `lgkmcnt(1)` between them): that stage shrinks from 2.6 to 1.4 us per trip, the second layer and the gather's issue grow by as much,
    //  39.6 against 39.2.  A trip's stages trade time with each other; their sum per CU does not move -- config 5 streams 157 MB of 512- and
    //  128-byte rows in 39 us, 4.0 TB/s, where the bare steady-state gather of 128-byte lines gets 4.8-5.1 TB/s out of HBM (row_gather_steady.hip).)
    // ---- prologue: first task's ids, the numerics' A operands (global -> registers) and the LDS image / small tables requested together ----
    // (Tried and dropped [r6], profiles/r06/experiments/r06_33 with the patch: the first task's rows requested BEFORE the workgroup's meeting -- a
    //  gather needs the wave's ids and its private slot, nothing of the image.  (a) gather in front of the DMA loop: config 5 40.3-40.5 us against
    //  39.3-39.6, EmbeddingMLP.py literal 25.3 against 24.1; (b) ids first, exactly npw DMA instructions per wave right behind them as asm
    //  statements, `s_waitcnt vmcnt(npw)` out of a switch -- the ids alone, the counter retires in issue order -- then the gather, then vmcnt(0)
    //  and the meeting: 40.9 against 39.8-40.2, 25.4 against 24.2.  Sixteen row requests per lane in the texture path's queue next to the 150 KB
    //  of staging slow the staging by more than the overlap gives back, and the meeting then waits for the slowest wave's ROWS: the same thing
    //  round 4's timeline showed for k_deepfm_v2_joint1's meeting.)
    f32x4 ri = zero, rd = zero;
    int tk = blockIdx.x * WAVES + wave;
    MR_STAMP(0);
    if (ntasks > 0) ld_raw(clampt(tk), ri, rd);
#pragma unroll 1
    for (int c = wave; c < LD::total_pad / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + c * 256), 16, 0, 0);
#pragma unroll 1
    for (int c = wave; c < A.small_floats / 256; c += WAVES)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(A.small + c * 256 + lane * 4),
            (__attribute__((address_space(3))) void*)(smem + LD::total_pad + c * 256), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0): this wave's DMA pieces and ids have landed
    __builtin_amdgcn_s_barrier();
    MR_STAMP(1);
    if (tk >= ntasks) return;                                     // (behind the barrier; a wave without a task flags nothing)
    MR_STAMP(27);
#ifdef SPRK_DF_XP
    stamp_gather = true;
#endif
    gather(tk, ri, rd, true);
#ifdef SPRK_DF_XP
    stamp_gather = false;
#endif
    MR_STAMP(30);
    ld_raw(clampt(tk + task_stride), ri, rd);
    MR_STAMP(2);
    // numerics' A operands: rows (nb*16 + r) of W0[:, numerics]^T, columns q and q + 4, first used by the first task's MFMAs.  [r6] Requested
    // BEHIND the first gather, as four coalesced 16-byte loads per lane (the image holds them lane-major: k_mlp_rows_pack).  Rounds 5's sixteen
    // 4-byte loads per lane sat between the meeting and the gather -- "nobody waits for them there", but a wave issues in order, and the stamped
    // timeline (profiles/r06/experiments/r06_35) has 3.3 us between the meeting and the last of them going out: sixteen waves' worth of
    // 64-lane 4-byte gathers through one texture path, in front of the rows everything else waits for.
    float rwa[N0C], rwb[N0C];
    {
        const float* wn = image + LD::off_w0n + lane * (2 * N0C);
#pragma unroll
        for (int j = 0; j < N0C / 2; ++j) {
            const f32x4 v = ld4(wn + 4 * j);
            rwa[2 * j] = v[0]; rwb[2 * j] = v[1]; rwa[2 * j + 1] = v[2]; rwb[2 * j + 1] = v[3];
        }
    }
    f32x4 wwide[2] = {zero, zero};
    if constexpr (WK == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = 16 * h + 4 * q;
            if (d < A.wide_dim) wwide[h] = ld4(A.wide_w + d);
        }
    }
    [[maybe_unused]] int trip = 0;
    for (;; tk += task_stride) {
        // ---- the task's gathered rows -> first-layer accumulators (frees the register set for the next task) ----
        f32x4 z0[N0C];
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) {
            z0[nb] = g[0][nb];
#pragma unroll
            for (int b = 1; b < NBIG; ++b) z0[nb] += g[b][nb];
        }
        float zw = 0.f;
        if constexpr (WK == 1) zw = dot4(gw[0], wwide[0]) + dot4(gw[1], wwide[1]);
        else if constexpr (WK == 2) zw = q == 0 ? gws : 0.f;
        int so_c[MR_MAX_SMALL];
#pragma unroll
        for (int f = 0; f < MR_MAX_SMALL; ++f) so_c[f] = so[f];
        const float xa_c = xa, xb_c = xb;
        // ---- next task: gathers issued now, consumed after this task's second layer; ids one task further ahead ----
        const bool more = tk + task_stride < ntasks;              // wave-uniform
        // (fence: the sums above must END the old rows' live ranges before the next gather's loads are placed.  hipcc's MachineSink moved
        //  the adds down to their first use BEHIND the gather, the loads got a second register set, and every trip ended with 32 v_mov_b64 --
        //  each behind its own vmcnt wait -- back into the first (build/sparrow.s).  The empty statements "use" the sums here.)
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) asm volatile("" : "+v"(z0[nb]));
        __builtin_amdgcn_sched_barrier(0);
        MR_STAMP(3 + 6 * trip);
        gather(clampt(tk + task_stride), ri, rd, more);
        ld_raw(clampt(tk + 2 * task_stride), ri, rd);
        MR_STAMP(4 + 6 * trip);
        // ---- small columns from LDS (a lane's piece nb of a row: + 64 nb bytes, the instruction's offset field) ----
        if (!bias_in_k7) {
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb) z0[nb] += ld4(smem + LD::off_b0 + nb * 16 + 4 * q);
        }
#pragma unroll
        for (int f = 0; f < MR_MAX_SMALL; ++f) {
            if (!(MR_XP & 1) && (RT ? f < ns : f < NS)) {
                const float* row = reinterpret_cast<const float*>(small_b + so_c[f]);
#pragma unroll
                for (int nb = 0; nb < N0C; ++nb) z0[nb] += ld4(row + 16 * nb);
            }
        }
        // ---- numerics (+ b0) on the matrix pipe ----
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rwa[nb], xa_c, z0[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(rwb[nb], xb_c, z0[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < N0C; ++nb) z0[nb] = relu4_fast(z0[nb]);
        // ---- second layer: K = N0, B operand = h1 as it sits in the registers ----
#ifdef SPRK_DF_XP
        asm volatile("" : "+v"(z0[0]), "+v"(z0[N0C - 1]));
        MR_STAMP(5 + 6 * trip);
#endif
        f32x4 z1[N1C];
        if constexpr (DYN) {
            float mx = 0.f;
#pragma unroll
            for (int nb = 0; nb < N0C; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, z0[nb][j]);       // h1 >= 0
            mx = rows4_max(mx);
            float scale, inv;
            dyn_scale(mx, A.inv_w1_scale, scale, inv);
            const float* wf = smem + LD::off_w1 + lane * 4;              // this lane's 16 bytes inside a 1-KB fragment (k_dyn_pack_w: lane order)
            constexpr int KB = N0C / 2, NG = N1C / 4;                    // K blocks of 32; groups of four output blocks
            auto frag = [&](int n1, int b, int part) {
                if (MR_XP & 2) return __builtin_bit_cast(din_f16x8, ld4(wf + part * 256 + 0 * (n1 + b)));
                return __builtin_bit_cast(din_f16x8, ld4(wf + ((n1 * KB + b) * 2 + part) * 256));
            };
            din_f16x8 ah[4], al[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { ah[j] = frag(j, 0, 0); al[j] = frag(j, 0, 1); }
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                din_f16x8 bh, bl;
                dyn_split8(z0[2 * b], z0[2 * b + 1], scale, bh, bl);
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    const int step = b * NG + gi + 1;                       // the group behind this one
                    const int nb_ = step / NG, ng_ = step - nb_ * NG;       // (compile-time after unrolling)
                    const bool more = step < KB * NG;
                    if (MR_XP & 16) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) z1[4 * gi + j] = (b == 0 ? zero : z1[4 * gi + j]) + __builtin_bit_cast(f32x4, ah[j]) + __builtin_bit_cast(f32x4, al[j]) + __builtin_bit_cast(f32x4, bh) + __builtin_bit_cast(f32x4, bl);
                        continue;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) z1[4 * gi + j] = mfma_f16(ah[j], bh, b == 0 ? zero : z1[4 * gi + j]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) z1[4 * gi + j] = mfma_f16(ah[j], bl, z1[4 * gi + j]);
                    if (more) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) ah[j] = frag(4 * ng_ + j, nb_, 0);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) z1[4 * gi + j] = mfma_f16(al[j], bh, z1[4 * gi + j]);
                    if (more) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) al[j] = frag(4 * ng_ + j, nb_, 1);
                    }
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
#ifdef SPRK_DF_XP
        asm volatile("" : "+v"(z1[0]), "+v"(z1[N1C - 1]));
        MR_STAMP(6 + 6 * trip);
#endif
        float z = zw;
#pragma unroll
        for (int n1 = 0; n1 < N1C; ++n1) {
            const f32x4 hw = ld4(smem + LD::off_hw + n1 * 16 + 4 * q);
            const f32x4 h2 = relu4_fast(z1[n1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) z = fmaf(hw[j], h2[j], z);
        }
        z = rows4_sum(z);
        const Loc Ls = locate(tk);
        const int mm = Ls.tl * 16 + r;
        if (q == 0 && mm < B) Ls.out[mm] = sigmoidf_acc(z + A.head_bias);
        MR_STAMP(7 + 6 * trip);
#ifdef SPRK_DF_XP
        ++trip;
#endif
        if (!more) break;
    }
    if (badm != 0 && lane == 0) atomicOr(err, 1);
}
template <int N0C, int N1C, int NBIG, int NS, int WAVES, bool DYN, int WK>
__global__ __launch_bounds__(WAVES * 64, 2) void k_mlp_rows(const MlpRowsRun A, const int* __restrict__ ids,
                                                            const float* __restrict__ dense, float* __restrict__ out,
                                                            int B, int* __restrict__ err, const float* __restrict__ image) {
    mr_body<N0C, N1C, NBIG, NS, WAVES, DYN, WK, false>(A, ids, dense, out, B, err, image, nullptr);
}
template <int N0C, int N1C, int NBIG, int NS, int WAVES, bool DYN, int WK>
__global__ __launch_bounds__(WAVES * 64, 2) void k_mlp_rows_many(const MlpRowsRun A, const MlpRowsMany M, int B, int* __restrict__ err,
                                                                 const float* __restrict__ image) {
    mr_body<N0C, N1C, NBIG, NS, WAVES, DYN, WK, true>(A, nullptr, nullptr, nullptr, B, err, image, &M);
}

// [r6] Measured and NOT kept (profiles/r06/experiments/r06_13, r06_14, the last form's source next to its numbers): the same
// graph with more waves per SIMD.  (a) twelve waves (3 per SIMD, 153-158 VGPRs, no spill), one task in flight per wave -- a wave requests its rows,
// reads the genre rows while they fly, goes on when they land; no second row-register set, h1 split into its operand pairs right behind the ReLU,
// the second layer in two groups of four output blocks, numerics read from global memory so that a wave's staging slot is the ids alone (with
// sixteen slots the image + genre tables + slots only just fit the CU): config 5 43.7 us against this kernel's 39.3-39.8, EmbeddingMLP.py 24.8
// against 24.2.  (b) sixteen waves at the 128-register cap: hipcc spills the row registers (load, vmcnt(0), scratch_store, eight times per
// column): 59 / 31 us.  (c) twelve waves WITH the next task's rows held through the second layer (161 live registers by count, 168 allowed):
// 18-61 dwords spill, and a spill reload shares the in-order vmcnt with the prefetched rows, so every reload waits for the gather: 79.7 / 39.8
// us.  What the eight-wave form buys with its 249 registers -- the next task's ids, rows, cross row and genre offsets in flight under a whole
// second layer -- is worth more than a third wave per SIMD; the structural change that remains is 32 samples per fragment read on 32x32x16
// MFMAs (a different C/D layout from the gather on: a new kernel, not a variant of this one).

// host_setup_v2.h -- DeepFM_v2: k_deepfm_v2_chain / _joint / _joint1 dispatch tables, plan matcher, fold + joint-table set-up.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- the DeepFM_v2 plan shapes the joint kernels are built on: folded tables (KP-wide projected rows), the LDS image of k_v2_pack_image ----
// ([r6] until round 6 this was the dispatch table of k_deepfm_v2_chain, whose kernels are retired: k_chain_v2.h)
constexpr int V2_WAVES = 8;
struct V2Variant {
    int g_emb, kpc, h0c, h1c;
    size_t lds_bytes;                 // the packed image + a staging slot per wave
    void (*pack)(const V2Args&, float*);
};
template <int G_EMB, int DV, int KPC, int H0C, int H1C, bool FOLD>
void v2_pack(const V2Args& a, float* image) {
    hipLaunchKernelGGL((k_v2_pack_image<G_EMB, DV, KPC, H0C, H1C, FOLD>), dim3(1), dim3(256), 0, 0, a, image);
}
#define V2_LDS(G_EMB, DV, KPC, H0C, H1C, FOLD) \
    (sizeof(float) * (V2Lds<G_EMB, DV, KPC, H0C, H1C, FOLD>::total_pad + V2_WAVES * V2Lds<G_EMB, DV, KPC, H0C, H1C, FOLD>::stage_floats))
#define V2_VARIANT(G_EMB) {G_EMB, 1, 2, 1, V2_LDS(G_EMB, 4, 1, 2, 1, true), &v2_pack<G_EMB, 4, 1, 2, 1, true>}
const V2Variant kV2Variants[] = {
    V2_VARIANT(6),    // BASELINE config 2: 6 fields, projection 16 (folded into the tables), deep 32-16
    V2_VARIANT(4),    // 4 fields (config 4 gathers 128-B projected rows instead of 256-B raw ones)
    V2_VARIANT(5), V2_VARIANT(3), V2_VARIANT(2),
};

// ---- dispatch table for k_deepfm_v2_joint<G_BIG, NJF, KPC, H0C, H1C, WAVES> ----
typedef void (*V2JLaunchFn)(const V2JRun&, const int*, const float*, float*, int, int*, const float*, int, size_t, hipStream_t);
typedef void (*V2JLaunchManyFn)(const V2JRun&, const V2JMany&, int, int*, const float*, int, size_t, hipStream_t);
template <int G_BIG, int NJF, bool HALF>
void v2j_launch(const V2JRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image,
                int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_deepfm_v2_joint<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), dim3(grid), dim3(V2_WAVES * 64), lds, st,
                       a, ids, dense, out, B, err, image);
}
template <int G_BIG, int NJF, bool HALF>
void v2j_launch_many(const V2JRun& a, const V2JMany& m, int B, int* err, const float* image, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_deepfm_v2_joint_many<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), dim3(grid), dim3(V2_WAVES * 64), lds, st,
                       a, m, B, err, image);
}
struct V2JVariant {
    int g_big, njf, kpc;
    bool half;                            // big fields on split-f16 MFMA
    const void* fn;
    const void* fn_many;
    V2JLaunchFn launch;
    V2JLaunchManyFn launch_many;
};
#define V2J_VARIANT(G_BIG, NJF, HALF) \
    {G_BIG, NJF, 1, HALF, reinterpret_cast<const void*>(&k_deepfm_v2_joint<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), \
     reinterpret_cast<const void*>(&k_deepfm_v2_joint_many<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), &v2j_launch<G_BIG, NJF, HALF>, \
     &v2j_launch_many<G_BIG, NJF, HALF>}
#define V2J_BOTH(G_BIG, NJF) V2J_VARIANT(G_BIG, NJF, true), V2J_VARIANT(G_BIG, NJF, false)
const V2JVariant kV2JVariants[] = {
    V2J_BOTH(3, 3),    // BASELINE config 2: movieId, userId, userRatedMovie1 + a joint table of the three genre fields
    V2J_BOTH(2, 2),    // the reference's own four fields (movieId, userId + two genres), config-4 shape
    V2J_BOTH(3, 2), V2J_BOTH(3, 1), V2J_BOTH(2, 3), V2J_BOTH(2, 1), V2J_BOTH(1, 3), V2J_BOTH(1, 2), V2J_BOTH(1, 1),
};

// k_deepfm_v2_joint1<G_BIG, NJF>: the one-task-per-wave shape of the split-f16 joint kernel (k_chain_v2j1.h)
typedef void (*V2J1LaunchFn)(const V2JRun&, const int*, const float*, float*, int, int*, const float*, int, size_t, hipStream_t);
template <int G_BIG, int NJF, bool HOIST>
void v2j1_launch(const V2JRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image, int grid,
                 size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_deepfm_v2_joint1<G_BIG, NJF, HOIST>), dim3(grid), dim3(V2J1_WAVES_OF(HOIST, G_BIG) * 64), lds, st, a, ids, dense, out, B, err, image);
}
// (fn_h / launch_h: the HOIST form, for tables larger than the Infinity Cache)
struct V2J1Variant { int g_big, njf; const void* fn; V2J1LaunchFn launch; int image_floats; const void* fn_h; V2J1LaunchFn launch_h; };
#define V2J1_VARIANT(G_BIG, NJF) \
    {G_BIG, NJF, reinterpret_cast<const void*>(&k_deepfm_v2_joint1<G_BIG, NJF, false>), &v2j1_launch<G_BIG, NJF, false>, V2J1Lds<G_BIG>::total_pad, \
     reinterpret_cast<const void*>(&k_deepfm_v2_joint1<G_BIG, NJF, true>), &v2j1_launch<G_BIG, NJF, true>}
const V2J1Variant kV2J1Variants[] = {
    V2J1_VARIANT(3, 3), V2J1_VARIANT(2, 2), V2J1_VARIANT(3, 2), V2J1_VARIANT(3, 1), V2J1_VARIANT(2, 3), V2J1_VARIANT(2, 1),
    V2J1_VARIANT(1, 3), V2J1_VARIANT(1, 2), V2J1_VARIANT(1, 1),
};

// Recognise the plan models.DeepFMv2 emits (DeepFM_v2.py graph) and fill the fused kernel's arguments.
bool match_v2_chain(sprk_engine* h) {
    const sprk_plan& p = h->plan;
    if (p.model_kind != SPRK_MODEL_DEEPFM_V2 || p.din.enabled || p.n_bufs != 2) return false;
    V2Args a;
    memset(&a, 0, sizeof(a));
    int g_emb = 0;
    while (g_emb < p.n_segs && p.segs[g_emb].kind == SPRK_SEG_ROWS) ++g_emb;
    if (g_emb < 1 || g_emb > V2_MAX_FIELDS) return false;
    if (p.n_id_cols > 8 || p.n_dense > 8 || p.n_dense < 1) return false;   // one 16-B/lane load stages a task's ids+numerics
    const int Dp = p.segs[0].row_stride;
    bool raw_over_4g = false;                 // a raw table beyond 32-bit byte offsets: fine when folded (the kernel never reads it)
    for (int g = 0; g < g_emb; ++g) {
        const sprk_seg& s = p.segs[g];
        if (s.row_stride != Dp || s.count * 4 != Dp || s.dst != g * Dp) return false;
        // the fused kernel needs the all-zero row at index vocab and (unfolded) 32-bit element offsets
        const size_t need = ((size_t)s.vocab + 1) * Dp * sizeof(float);
        if (h->slot_bytes[s.slot] < need) return false;
        if (need >= ((size_t)1 << 32)) raw_over_4g = true;
        a.emb_col[g] = s.field; a.emb_vocab[g] = s.vocab; a.table[g] = (const float*)h->slot_ptr[s.slot];
    }
    int si = g_emb;
    if (si >= p.n_segs || p.segs[si].kind != SPRK_SEG_DENSE || p.segs[si].field != 0 || p.segs[si].count > 8) return false;
    const int n_num = p.segs[si].count, num_off = p.segs[si].dst;
    ++si;
    if (si < p.n_segs && p.segs[si].kind == SPRK_SEG_ZERO) ++si;
    const int n_fo = p.n_segs - si;
    if (n_fo < 1 || n_fo > V2_MAX_FIELDS) return false;
    const int scal_off = p.segs[si].dst;
    for (int i = 0; i < n_fo; ++i) {
        const sprk_seg& s = p.segs[si + i];
        if (s.kind != SPRK_SEG_SCALAR || s.dst != scal_off + i) return false;
        if (h->slot_bytes[s.slot] < ((size_t)s.vocab + 1) * sizeof(float)) return false;
        a.fo_col[i] = s.field; a.fo_vocab[i] = s.vocab; a.w1[i] = (const float*)h->slot_ptr[s.slot];
    }
    if (p.n_ops != g_emb + 4 || p.n_taps != 4) return false;
    const int Kp = p.ops[0].N, G = g_emb + 1;
    for (int g = 0; g < g_emb; ++g) {
        const sprk_op& o = p.ops[g];
        if (o.kind != SPRK_OP_DENSE || o.act != SPRK_ACT_NONE || o.src_buf != 0 || o.src_off != g * Dp || o.K != Dp ||
            o.dst_buf != 1 || o.dst_off != g * Kp || o.N != Kp || o.ldw != Dp) return false;
        a.Wp[g] = (const float*)h->slot_ptr[o.w_slot]; a.bp[g] = (const float*)h->slot_ptr[o.b_slot];
    }
    {
        const sprk_op& o = p.ops[g_emb];
        if (o.kind != SPRK_OP_DENSE || o.act != SPRK_ACT_NONE || o.src_buf != 0 || o.src_off != num_off || o.K > 8 ||
            o.K < n_num || o.dst_buf != 1 || o.dst_off != g_emb * Kp || o.N != Kp) return false;
        a.Wp[g_emb] = (const float*)h->slot_ptr[o.w_slot]; a.bp[g_emb] = (const float*)h->slot_ptr[o.b_slot];
        a.ldp_num = o.ldw;
    }
    a.ldp_emb = Dp;
    const sprk_op& fm = p.ops[g_emb + 1];
    if (fm.kind != SPRK_OP_FM_SUMSQ || fm.src_buf != 1 || fm.src_off != 0 || fm.groups != G || fm.group_stride != Kp ||
        fm.K > Kp || fm.dst_buf != 0) return false;
    const sprk_op& d0 = p.ops[g_emb + 2];
    if (d0.kind != SPRK_OP_DENSE || d0.act != SPRK_ACT_RELU || d0.src_buf != 1 || d0.src_off != 0 || d0.K != G * Kp ||
        d0.ldw != G * Kp || d0.dst_buf != 0 || d0.dst_off != 0) return false;
    const sprk_op& d1 = p.ops[g_emb + 3];
    if (d1.kind != SPRK_OP_DENSE || d1.act != SPRK_ACT_RELU || d1.src_buf != 0 || d1.src_off != 0 || d1.K != d0.N ||
        d1.ldw != d0.N || d1.dst_buf != 1 || d1.dst_off != 0) return false;
    a.W0 = (const float*)h->slot_ptr[d0.w_slot]; a.b0 = (const float*)h->slot_ptr[d0.b_slot];
    a.W1 = (const float*)h->slot_ptr[d1.w_slot]; a.b1 = (const float*)h->slot_ptr[d1.b_slot];
    const sprk_tap &t0 = p.taps[0], &t1 = p.taps[1], &t2 = p.taps[2], &t3 = p.taps[3];
    if (t0.buf != 0 || t0.off != scal_off || t0.len != n_fo || t0.w_slot != -1) return false;
    if (t1.buf != 0 || t1.off != num_off || t1.len != n_num || t1.w_slot < 0 || t1.scale != t0.scale) return false;
    if (t2.buf != 0 || t2.off != fm.dst_off || t2.len != fm.K || t2.w_slot < 0 || t2.scale != 1.0f || t2.bias != 0.0f) return false;
    if (t3.buf != 1 || t3.off != 0 || t3.len > d1.N || t3.w_slot < 0 || t3.scale != 1.0f || t3.bias != 0.0f) return false;
    a.fo_num_w = (const float*)h->slot_ptr[t1.w_slot];
    a.hfm = (const float*)h->slot_ptr[t2.w_slot]; a.n_hfm = t2.len;
    a.hdeep = (const float*)h->slot_ptr[t3.w_slot]; a.n_hdeep = t3.len;
    a.h0w = t0.scale; a.fo_bias = t0.bias + t1.bias; a.head_bias = p.head_bias;
    a.F = p.n_id_cols; a.ND = p.n_dense; a.n_num = n_num; a.n_fo = n_fo;
    const int kpc = Kp / 16, h0c = d0.N / 16, h1c = d1.N / 16;
    // fold the per-field projections into the tables when that never widens a gathered row
    size_t total_rows = 0;
    for (int g = 0; g < g_emb; ++g) total_rows += (size_t)a.emb_vocab[g] + 1;
    // (32-bit byte offsets into ONE buffer of folded rows: needs < 4 GiB)
    const bool want_fold = Kp <= Dp && Kp + 16 <= 64 && total_rows * (size_t)(Kp + 16) * 4 < ((size_t)1 << 32) &&
                           h->tune.v2_fold;
    if (raw_over_4g && !want_fold) return false;             // e.g. BASELINE config 4's 27 M x 64 table (6.9 GB): folded rows only
    // the fused kernel reads ONE id per field for both the embedding row and the first-order
    // weight: the two field lists must be the same set of ids columns
    if (n_fo != g_emb) return false;
    {
        if (h->tune.v2_rows && !raw_over_4g) {
            h->v2 = a;
            h->rows_g_emb = g_emb;
            h->rows_from_v2 = true;
            return false;
        }
    }
    for (size_t v = 0; v < sizeof(kV2Variants) / sizeof(kV2Variants[0]); ++v) {
        const V2Variant& vv = kV2Variants[v];
        if (!want_fold) continue;                            // (the joint kernels gather folded rows only)
        if (vv.g_emb == g_emb && vv.kpc == kpc && vv.h0c == h0c && vv.h1c == h1c) {
            V2Run run;
            memset(&run, 0, sizeof(run));
            size_t fo_floats = 0;
            for (int g = 0; g < g_emb; ++g) {
                int hit = -1;
                for (int i = 0; i < n_fo; ++i)
                    if (a.fo_col[i] == a.emb_col[g] && a.fo_vocab[i] == a.emb_vocab[g]) hit = i;
                if (hit < 0) return false;
                run.col[g] = a.emb_col[g]; run.vocab[g] = a.emb_vocab[g];
                run.table[g] = a.table[g];
                run.fo_off[g] = (unsigned)fo_floats;
                fo_floats += (size_t)a.emb_vocab[g] + 1;
            }
            if (fo_floats >= ((size_t)1 << 31)) return false;
            // w1 pointers in embedding-group order
            const float* w1g[V2_MAX_FIELDS];
            for (int g = 0; g < g_emb; ++g) {
                for (int i = 0; i < n_fo; ++i)
                    if (a.fo_col[i] == a.emb_col[g] && a.fo_vocab[i] == a.emb_vocab[g]) w1g[g] = a.w1[i];
            }
            for (int g = 0; g < g_emb; ++g) a.w1[g] = w1g[g];
            run.F = a.F; run.ND = a.ND; run.n_num = a.n_num;
            run.h0w = a.h0w; run.fo_bias = a.fo_bias; run.head_bias = a.head_bias;
            h->v2run = run;
            h->v2 = a;
            h->v2_fo_floats = fo_floats;
            h->v2_variant = (int)v;
            h->v2_lds_bytes = vv.lds_bytes;
            h->rows_g_emb = g_emb;                             // (should the joint set-up refuse the model: finalize hands the parsed plan to k_rows_chain)
            h->v2_rows_ok = kpc >= 1 && kpc <= 4 && h0c >= 1 && h1c >= 1 && !raw_over_4g;
            return true;
        }
    }
    // no folded form (e.g. the reference's Dense(64) projections): the parsed plan goes to k_rows_chain
    if (kpc >= 1 && kpc <= 4 && h0c >= 1 && h1c >= 1 && !raw_over_4g) {
        h->v2 = a;
        h->rows_g_emb = g_emb;
        h->rows_from_v2 = true;
    }
    return false;
}

// Split the fields of a folded DeepFM_v2 engine into big ones (gathered per field) and a joint group of
// small-vocabulary ones (one gather per sample), build the joint table.  Leaves v2j_variant = -1 when the
// model has no small field or no instantiation fits.
int wide_dynamic_range(const float* rows, long long nrows, int row_floats, int ncols, float mx, bool* wide);
int setup_v2_joint(sprk_engine* h) {
    const V2Variant& vv = kV2Variants[h->v2_variant];
    if (vv.kpc != 1 || vv.h0c != 2 || vv.h1c != 1 || !h->tune.v2_joint) return SPRK_OK;
    const int KP = 16, H0 = 32, G = vv.g_emb;
    int big[V2_MAX_FIELDS], nbig = 0, jf[V2_MAX_FIELDS], njf = 0;
    for (int g = 0; g < G; ++g) {
        const long long v1 = (long long)h->v2run.vocab[g] + 1;
        if (v1 <= 32 && njf < V2J_MAX_JF) jf[njf++] = g;      // small enough to live in LDS
        else big[nbig++] = g;
    }
    if (njf < 1 || nbig < 1 || nbig > 3) return SPRK_OK;
    // HALF: scales from max|P| over the big fields' folded rows and max|W0|; refused for non-finite weights
    bool half = h->tune.v2_half;
    float p_scale = 1.f, w_scale = 1.f;
    if (half) {
        DevProbe d_max_probe;
        unsigned*& d_max = d_max_probe.p;
        HIP_TRY(hipMalloc((void**)&d_max, 2 * sizeof(unsigned)));
        HIP_TRY(hipMemset(d_max, 0, 2 * sizeof(unsigned)));
        for (int b = 0; b < nbig; ++b) {
            const long long rows = (long long)h->v2run.vocab[big[b]] + 1;
            long long blocks = (rows * KP + 255) / 256;
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(k_v2_absmax, dim3((unsigned)blocks), dim3(256), 0, 0,
                               h->v2_folded + (size_t)h->v2run.rowbase[big[b]] * (KP + 16), rows, KP + 16, KP, d_max);
        }
        hipLaunchKernelGGL(k_v2_absmax, dim3(8), dim3(256), 0, 0, h->v2.W0, (long long)H0, (G + 1) * KP, (G + 1) * KP, d_max + 1);
        HIP_TRY(hipGetLastError());
        unsigned bits[2];
        HIP_TRY(hipMemcpy(bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
        float mx[2];
        memcpy(mx, bits, sizeof(mx));
        for (int i = 0; i < 2; ++i) {
            if (!(mx[i] < 3.0e38f)) { half = false; break; }     // NaN / Inf in the weights: keep the f32 path
            int e = 0;
            if (mx[i] > 0.f) { (void)frexpf(mx[i], &e); e = 15 - e; }   // mx * 2^e in [2^14, 2^15)
            if (e > 60) e = 60;
            if (e < -60) e = -60;
            (i == 0 ? p_scale : w_scale) = ldexpf(1.f, e);
        }
        // an outlier row next to ordinary ones: the ordinary rows' lo halves would be subnormal -> keep the f32 variant
        for (int b = 0; half && b < nbig; ++b) {
            bool wide = false;
            if (int rcw = wide_dynamic_range(h->v2_folded + (size_t)h->v2run.rowbase[big[b]] * (KP + 16),
                                             (long long)h->v2run.vocab[big[b]] + 1, KP + 16, KP, mx[0], &wide)) return rcw;
            if (wide) half = false;
        }
        if (half) {
            bool wide = false;
            if (int rcw = wide_dynamic_range(h->v2.W0, (long long)H0, (G + 1) * KP, (G + 1) * KP, mx[1], &wide)) return rcw;
            if (wide) half = false;
        }
    }
    int variant = -1;
    for (size_t v = 0; v < sizeof(kV2JVariants) / sizeof(kV2JVariants[0]); ++v)
        if (kV2JVariants[v].g_big == nbig && kV2JVariants[v].njf == njf && kV2JVariants[v].half == half) variant = (int)v;
    if (variant < 0) return SPRK_OK;
    V2JRun& r = h->v2j_run;
    memset(&r, 0, sizeof(r));
    r.F = h->v2run.F; r.ND = h->v2run.ND; r.n_num = h->v2run.n_num;
    r.h0w = h->v2run.h0w; r.fo_bias = h->v2run.fo_bias; r.head_bias = h->v2run.head_bias;
    for (int b = 0; b < nbig; ++b) {
        r.big_col[b] = h->v2run.col[big[b]]; r.big_vocab[b] = h->v2run.vocab[big[b]];
        r.big_rowbase[b] = h->v2run.rowbase[big[b]]; r.big_grp[b] = big[b];
    }
    size_t small_floats = 0;
    for (int f = 0; f < njf; ++f) {
        r.j_col[f] = h->v2run.col[jf[f]]; r.j_vocab[f] = h->v2run.vocab[jf[f]];
        r.s_off[f] = (int)small_floats;
        small_floats += ((size_t)r.j_vocab[f] + 1) * V2J_SS;
    }
    small_floats = (small_floats + 255) & ~(size_t)255;          // whole 1-KB LDS-DMA pieces
    r.wf_off = (int)small_floats;                                // HALF: the numerics' fold through deep0, [H0][8]
    if (half) small_floats += ((size_t)H0 * 8 + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void**)&h->v2j_tab, small_floats * sizeof(float)));
    HIP_TRY(hipMemset(h->v2j_tab, 0, small_floats * sizeof(float)));
    for (int f = 0; f < njf; ++f) {
        const int rows = r.j_vocab[f] + 1;
        hipLaunchKernelGGL(k_v2_fold_small, dim3((rows + 3) / 4), dim3(256), 0, 0,
                           h->v2_folded + (size_t)h->v2run.rowbase[jf[f]] * (KP + 16), KP, H0, jf[f], h->v2.W0, (G + 1) * KP,
                           h->v2.b0, f == 0 ? 1 : 0, h->v2j_tab + r.s_off[f], rows);
    }
    if (half)
        hipLaunchKernelGGL(k_v2j_fold_num, dim3(8), dim3(256), 0, 0, h->v2.W0, (G + 1) * KP, G * KP, h->v2.Wp[G], h->v2.ldp_num,
                           h->v2.bp[G], h->v2.n_num, KP, H0, h->v2j_tab + r.wf_off, h->v2j_tab + r.s_off[0], r.j_vocab[0] + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    r.small_floats = (int)small_floats;
    r.tab0 = h->v2_folded;
    r.small = h->v2j_tab;
    r.w_scale = w_scale; r.unscale_h = 1.f / (p_scale * w_scale); r.unscale_s = 1.f / p_scale;
    if (half) {
        size_t big_rows = 0;
        for (int b = 0; b < nbig; ++b) big_rows += (size_t)r.big_vocab[b] + 1;
        if (big_rows * (KP + 16) * sizeof(float) >= ((size_t)1 << 32)) return fail(SPRK_EINVAL, "split rows exceed 32-bit offsets");
        { const int rc_ = table_alloc(h, (void**)&h->v2j_big, big_rows * (KP + 16) * sizeof(float)); if (rc_) return rc_; }
        h->derived_bytes += big_rows * (KP + 16) * sizeof(float);
        size_t base = 0;
        for (int b = 0; b < nbig; ++b) {
            const long long rows = (long long)r.big_vocab[b] + 1;
            long long nb = (rows * 8 + 255) / 256;
            if (nb > 65536) nb = 65536;
            hipLaunchKernelGGL(k_v2_split_rows, dim3((unsigned)nb), dim3(256), 0, 0,
                               h->v2_folded + (size_t)r.big_rowbase[b] * (KP + 16), h->v2j_big + base * (KP + 16), rows, p_scale);
            r.big_rowbase[b] = (unsigned)base;
            base += (size_t)rows;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        r.tab0 = h->v2j_big;
    }
    h->v2j_lds_bytes = vv.lds_bytes + small_floats * sizeof(float);
    HIP_TRY(hipFuncSetAttribute(kV2JVariants[variant].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2j_lds_bytes));
    HIP_TRY(hipFuncSetAttribute(kV2JVariants[variant].fn_many, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2j_lds_bytes));
    h->v2j_variant = variant;
    // the one-task-per-wave shape for strict one-batch launches (k_chain_v2j1.h); SPRK_V2J_ONE=0: looped kernel only
    if (half && h->tune.v2j_one) {
        for (size_t v = 0; v < sizeof(kV2J1Variants) / sizeof(kV2J1Variants[0]); ++v) {
            const V2J1Variant& ov = kV2J1Variants[v];
            if (ov.g_big != nbig || ov.njf != njf) continue;
            HIP_TRY(hipMalloc((void**)&h->v2j1_image, (size_t)ov.image_floats * sizeof(float)));
            hipLaunchKernelGGL(k_v2j1_pack_image, dim3(1), dim3(256), 0, 0, h->v2, r, nbig, G + 1, h->v2j1_image);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            // rows that cannot all sit in the 256 MB Infinity Cache come from HBM: the form that reads its weight fragments first
            h->v2j1_hoist = h->tune.v2j1_hoist >= 0 ? h->tune.v2j1_hoist != 0 : h->derived_bytes > ((size_t)256 << 20);
            h->v2j1_waves = V2J1_WAVES_OF(h->v2j1_hoist, nbig);
            h->v2j1_lds_bytes = ((size_t)ov.image_floats + small_floats + (size_t)h->v2j1_waves * 256) * sizeof(float);
            HIP_TRY(hipFuncSetAttribute(h->v2j1_hoist ? ov.fn_h : ov.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2j1_lds_bytes));
        }
    }
    return SPRK_OK;
}


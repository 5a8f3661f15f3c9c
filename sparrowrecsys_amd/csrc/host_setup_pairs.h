// host_setup_pairs.h -- pair-dot DeepFM: k_deepfm_pairs / _pairs1 dispatch table, plan matcher and set-up.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- dispatch table for k_deepfm_pairs<NF, NV, H0C, H1C, WAVES, DYN, SEP> ----
constexpr int V1_WAVES = 8;
constexpr int V1_ONE_MAX_TASKS = 16384;       // one-task-per-wave shape (k_deepfm_pairs1) up to B = 262 144
typedef void (*V1LaunchFn)(const V1Run&, const int*, const float*, float*, int, int*, int, hipStream_t);
typedef void (*V1LaunchManyFn)(const V1Run&, const V1Many&, int, int*, int, hipStream_t);
template <int NF, int NV, bool SEP>
void v1_launch(const V1Run& a, const int* ids, const float* dense, float* out, int B, int* err, int grid, hipStream_t st) {
    const size_t lds = V1Lds<4, 4, (NV + 3) / 4>::bytes;
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, true, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, ids, dense, out, B, err);
    else
        hipLaunchKernelGGL((k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, false, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, ids, dense, out, B, err);
}
// one task per wave (narrow rows, split-f16 form only): grid = ceil(tasks / waves), no cap
template <int NF, int NV, bool SEP>
void v1_launch_one(const V1Run& a, const int* ids, const float* dense, float* out, int B, int* err, int grid, hipStream_t st) {
    if constexpr (NV <= 4) {
        const size_t lds = V1Lds<4, 4, 1>::bytes;
        hipLaunchKernelGGL((k_deepfm_pairs1<NF, NV, 4, 4, V1_WAVES, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, ids, dense, out, B, err);
    }
}
template <int NF, int NV, bool SEP>
void v1_launch_many(const V1Run& a, const V1Many& m, int B, int* err, int grid, hipStream_t st) {
    const size_t lds = V1Lds<4, 4, (NV + 3) / 4>::bytes;
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, true, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, m, B, err);
    else
        hipLaunchKernelGGL((k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, false, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, m, B, err);
}
template <int NF, int NV, bool SEP>
int v1_prepare(const V1Run& r, float* img) {
    constexpr int PC = (NV + 3) / 4;
    hipLaunchKernelGGL((k_v1_pack_image<4, 4, PC>), dim3(1), dim3(256), 0, 0, r, img);
    HIP_TRY(hipGetLastError());
    const size_t lds = V1Lds<4, 4, PC>::bytes;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, true, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, false, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, true, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, false, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if constexpr (NV <= 4)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs1<NF, NV, 4, 4, V1_WAVES, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return SPRK_OK;
}
struct V1Variant { int nf, nv; bool sep; V1LaunchFn launch; V1LaunchFn launch_one; V1LaunchManyFn launch_many; int (*prepare)(const V1Run&, float*); size_t lds_bytes; };
#define V1_VARIANT(NF, NV, SEP) {NF, NV, SEP, &v1_launch<NF, NV, SEP>, &v1_launch_one<NF, NV, SEP>, &v1_launch_many<NF, NV, SEP>, &v1_prepare<NF, NV, SEP>, V1Lds<4, 4, (NV + 3) / 4>::bytes}
#define V1_BOTH(NF, NV) V1_VARIANT(NF, NV, true), V1_VARIANT(NF, NV, false)
const V1Variant kV1Variants[] = {
    V1_BOTH(6, 4),    // BASELINE config 2: 6 fields, emb_dim 16, deep 64-64 (sep = the deep part's own movieId / userId tables, DeepFM.py:106)
    V1_BOTH(4, 3),    // the reference's own DeepFM.py: 4 fields, emb_dim 10 (rows padded to 12)
    V1_BOTH(4, 4),
    V1_BOTH(4, 16),   // BASELINE config 4: emb_dim 64 -- 256-byte rows gathered whole (four pieces per lane)
};

// Recognise the plan models.DeepFM emits (DeepFM.py graph: pair dots + first order + 2-layer deep part) and set up
// k_deepfm_pairs for it.  Leaves v1_variant = -1 (tile interpreter) for any other shape.
int setup_deepfm_pairs(sprk_engine* h) {
    if (!h->tune.v1_chain) return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (p.model_kind != SPRK_MODEL_DEEPFM || p.din.enabled || p.n_ops != 3 || p.n_taps != 3 || p.n_pairs < 1) return SPRK_OK;
    const sprk_op &od = p.ops[0], &o0 = p.ops[1], &o1 = p.ops[2];
    if (od.kind != SPRK_OP_PAIR_DOT || od.src_buf != 0 || od.dst_buf != 0) return SPRK_OK;
    if (o0.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_RELU || o0.src_buf != 0 || o0.dst_buf != 1 || o0.dst_off != 0 || o0.N != 64) return SPRK_OK;
    if (o1.kind != SPRK_OP_DENSE || o1.act != SPRK_ACT_RELU || o1.src_buf != 1 || o1.src_off != 0 || o1.K != o0.N || o1.dst_off != 0 || o1.N != 64) return SPRK_OK;
    V1Run r;
    memset(&r, 0, sizeof(r));
    int row_dst[V1_MAX_FIELDS], nf = 0, Dp = 0, num_dst = -1, scal_dst[V1_MAX_FIELDS], ns = 0, scal_col[V1_MAX_FIELDS], scal_vocab[V1_MAX_FIELDS];
    const float* scal_tab[V1_MAX_FIELDS];
    // the deep part's OWN tables (models.DeepFM without share_deep_tables; DeepFM.py:106): a ROWS segment that lands inside deep0's
    // input slice while ANOTHER ROWS segment of the same ids column lands outside it (the FM part's table of that key)
    const int ds0 = o0.src_off, ds1 = o0.src_off + o0.K;
    int dsep_col[V1_MAX_DEEP], dsep_vocab[V1_MAX_DEEP], dsep_dst[V1_MAX_DEEP], n_dsep = 0;
    const float* dsep_tab[V1_MAX_DEEP];
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& sg = p.segs[i];
        if (sg.kind == SPRK_SEG_ROWS) {
            if (nf == 0 && n_dsep == 0) Dp = sg.row_stride;
            if (sg.row_stride != Dp || sg.count * 4 != Dp || Dp > 64) return SPRK_OK;
            if (h->slot_bytes[sg.slot] < ((size_t)sg.vocab + 1) * Dp * sizeof(float)) return SPRK_OK;   // needs the zero row at index vocab
            bool twin_outside = false;
            for (int j = 0; j < p.n_segs; ++j)
                if (j != i && p.segs[j].kind == SPRK_SEG_ROWS && p.segs[j].field == sg.field &&
                    !(p.segs[j].dst >= ds0 && p.segs[j].dst + p.segs[j].row_stride <= ds1)) twin_outside = true;
            if (twin_outside && sg.dst >= ds0 && sg.dst + Dp <= ds1) {
                if (n_dsep == V1_MAX_DEEP) return SPRK_OK;
                dsep_col[n_dsep] = sg.field; dsep_vocab[n_dsep] = sg.vocab; dsep_tab[n_dsep] = (const float*)h->slot_ptr[sg.slot];
                dsep_dst[n_dsep++] = sg.dst;
                continue;
            }
            if (nf == V1_MAX_FIELDS) return SPRK_OK;
            r.col[nf] = sg.field; r.vocab[nf] = sg.vocab; r.table[nf] = (const float*)h->slot_ptr[sg.slot];
            row_dst[nf++] = sg.dst;
        } else if (sg.kind == SPRK_SEG_SCALAR) {
            if (ns == V1_MAX_FIELDS) return SPRK_OK;
            if (h->slot_bytes[sg.slot] < ((size_t)sg.vocab + 1) * sizeof(float)) return SPRK_OK;
            scal_col[ns] = sg.field; scal_vocab[ns] = sg.vocab; scal_tab[ns] = (const float*)h->slot_ptr[sg.slot]; scal_dst[ns++] = sg.dst;
        } else if (sg.kind == SPRK_SEG_DENSE) {
            if (num_dst >= 0 || sg.field != 0 || sg.count > 8) return SPRK_OK;
            num_dst = sg.dst; r.n_num = sg.count;
        } else if (sg.kind != SPRK_SEG_ZERO) {
            return SPRK_OK;
        }
    }
    if (nf < 2 || ns != nf || num_dst < 0 || r.n_num < 1) return SPRK_OK;
    for (int f = 0; f < nf; ++f) {                            // first-order table of the same ids column
        int hit = -1;
        for (int i = 0; i < ns; ++i) if (scal_col[i] == r.col[f] && scal_vocab[i] == r.vocab[f]) hit = i;
        if (hit < 0) return SPRK_OK;
        r.w1[f] = scal_tab[hit];
    }
    int smin = scal_dst[0];
    for (int i = 1; i < ns; ++i) if (scal_dst[i] < smin) smin = scal_dst[i];
    // taps: first order (all ones), pair dots (weights), deep output (weights)
    const sprk_tap *tf = nullptr, *tpair = nullptr, *tdeep = nullptr;
    for (int t = 0; t < 3; ++t) {
        const sprk_tap& tp = p.taps[t];
        if (tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
        if (tp.buf == 0 && tp.off == smin && tp.len == ns && tp.w_slot < 0) tf = &tp;
        else if (tp.buf == 0 && tp.off == od.dst_off && tp.len == p.n_pairs && tp.w_slot >= 0) tpair = &tp;
        else if (tp.buf == o1.dst_buf && tp.off == 0 && tp.len <= o1.N && tp.w_slot >= 0) tdeep = &tp;
    }
    if (!tf || !tpair || !tdeep) return SPRK_OK;
    for (int i = 0; i < ns; ++i) if (scal_dst[i] < smin || scal_dst[i] >= smin + ns) return SPRK_OK;
    // pairs -> (field a, field b) x head weight
    std::vector<float> pwh(p.n_pairs);
    HIP_TRY(hipMemcpy(pwh.data(), h->slot_ptr[tpair->w_slot], p.n_pairs * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < p.n_pairs; ++i) {
        int a = -1, b = -1;
        for (int f = 0; f < nf; ++f) { if (row_dst[f] == p.pair_a[i]) a = f; if (row_dst[f] == p.pair_b[i]) b = f; }
        if (a < 0 || b < 0 || a == b || od.K != Dp) return SPRK_OK;
        if (a > b) { const int t = a; a = b; b = t; }
        r.pw[a * V1_MAX_FIELDS + b] += pwh[i];
    }
    // deep part: the embedding columns inside deep0's input slice (at most V1_MAX_DEEP).  Tied tables: they are FM fields, which
    // become fields 0.. of the kernel.  Own tables (n_dsep > 0): every deep column must be one of them, and the FM fields with the
    // same ids columns become fields 0.. (the kernel looks deep row d up with field d's id).
    const int s0 = ds0, s1 = ds1;
    if (num_dst < s0 || num_dst + r.n_num > s1) return SPRK_OK;
    int order[V1_MAX_FIELDS], no = 0, deep_off[V1_MAX_DEEP] = {0, 0};
    const bool sep = n_dsep > 0;
    for (int f = 0; f < nf; ++f) {
        if (row_dst[f] >= s0 && row_dst[f] + Dp <= s1) {
            if (sep || r.n_deep == V1_MAX_DEEP) return SPRK_OK;
            deep_off[r.n_deep++] = row_dst[f] - s0;
            order[no++] = f;
        } else if (row_dst[f] < s1 && row_dst[f] + Dp > s0) {
            return SPRK_OK;
        }
    }
    if (sep) {
        for (int d = 0; d < n_dsep; ++d) {
            int twin = -1;
            for (int f = 0; f < nf; ++f) if (r.col[f] == dsep_col[d] && r.vocab[f] == dsep_vocab[d]) twin = f;
            if (twin < 0) return SPRK_OK;
            for (int i = 0; i < no; ++i) if (order[i] == twin) return SPRK_OK;
            deep_off[r.n_deep++] = dsep_dst[d] - s0;
            order[no++] = twin;
        }
    }
    for (int f = 0; f < nf; ++f) {
        bool deep = false;
        for (int i = 0; i < r.n_deep; ++i) deep |= order[i] == f;
        if (!deep) order[no++] = f;
    }
    {
        V1Run t = r;
        int inv[V1_MAX_FIELDS];
        for (int i = 0; i < nf; ++i) {
            const int f = order[i];
            inv[f] = i;
            t.col[i] = r.col[f]; t.vocab[i] = r.vocab[f]; t.table[i] = r.table[f]; t.w1[i] = r.w1[f];
        }
        memset(t.pw, 0, sizeof(t.pw));
        for (int a = 0; a < nf; ++a)
            for (int b = a + 1; b < nf; ++b) {
                const float w = r.pw[a * V1_MAX_FIELDS + b];
                if (w == 0.f) continue;
                int x = inv[a], y = inv[b];
                if (x > y) { const int tt = x; x = y; y = tt; }
                t.pw[x * V1_MAX_FIELDS + y] += w;
            }
        r = t;
    }
    r.sep = sep ? 1 : 0;
    for (int d = 0; d < n_dsep; ++d) r.table[nf + d] = dsep_tab[d];
    const float* const* deep_tables = sep ? &r.table[nf] : &r.table[0];   // tables deep0's embedding block reads
    int variant = -1;
    for (size_t v = 0; v < sizeof(kV1Variants) / sizeof(kV1Variants[0]); ++v)
        if (kV1Variants[v].nf == nf && kV1Variants[v].nv == Dp / 4 && kV1Variants[v].sep == sep) variant = (int)v;
    if (variant < 0) return SPRK_OK;
    const int H0 = o0.N, H1 = o1.N;
    const float* W0 = (const float*)h->slot_ptr[o0.w_slot];
    const int PC = (Dp / 4 + 3) / 4;                          // 16-float chunks per embedding row
    const int KW = 16 * (V1_MAX_DEEP * PC + 1);
    float* w0p = nullptr;
    HIP_TRY(hipMalloc((void**)&w0p, (size_t)H0 * KW * sizeof(float) + 16));
    h->v1_bufs.push_back(w0p);
    hipLaunchKernelGGL(k_v1_pack_w0, dim3(1), dim3(256), 0, 0, W0, o0.ldw, r.n_deep, deep_off[0], deep_off[1], Dp, num_dst - s0, r.n_num,
                       H0, PC, w0p);
    HIP_TRY(hipGetLastError());
    float* hd = nullptr;
    HIP_TRY(hipMalloc((void**)&hd, (size_t)H1 * sizeof(float) + 16));
    h->v1_bufs.push_back(hd);
    HIP_TRY(hipMemset(hd, 0, (size_t)H1 * sizeof(float)));
    HIP_TRY(hipMemcpy(hd, h->slot_ptr[tdeep->w_slot], (size_t)tdeep->len * sizeof(float), hipMemcpyDeviceToDevice));
    HIP_TRY(hipDeviceSynchronize());
    r.F = p.n_id_cols; r.ND = p.n_dense; r.nf = nf; r.row_floats = Dp;
    r.w0 = w0p; r.b0 = (const float*)h->slot_ptr[o0.b_slot];
    r.W1 = (const float*)h->slot_ptr[o1.w_slot]; r.ld1 = o1.ldw; r.b1 = (const float*)h->slot_ptr[o1.b_slot];
    r.hdeep = hd; r.head_bias = p.head_bias;
    r.w1frag = nullptr; r.inv_w1_scale = 0.f; r.w0frag = nullptr; r.inv_w0_scale = 0.f;
    {
        // DYN: deep1's kernel and the embedding columns of deep0's (the first 32 of the packed 48) as split-f16 fragments
        float w_scale = 0.f, w0_scale = 0.f;
        float *frag = nullptr, *frag0 = nullptr;
        int rc2 = make_dyn_fragments(h, r.W1, r.ld1, H1, H0, &frag, &w_scale);
        if (rc2) return rc2;
        if (frag && (rc2 = make_dyn_fragments(h, w0p, KW, H0, 32 * PC, &frag0, &w0_scale))) return rc2;
        if (frag && frag0) { r.w1frag = frag; r.inv_w1_scale = 1.0f / w_scale; r.w0frag = frag0; r.inv_w0_scale = 1.0f / w0_scale; }
    }
    {
        float* img = nullptr;
        HIP_TRY(hipMalloc((void**)&img, kV1Variants[variant].lds_bytes));
        h->v1_bufs.push_back(img);
        { const int rc3 = kV1Variants[variant].prepare(r, img); if (rc3) return rc3; }
        HIP_TRY(hipDeviceSynchronize());
        r.image = img;
    }
    r.e_scale = 0.f; r.e_inv = 0.f;
    {
        // static scale for deep0's embedding block: max |E| over the deep fields' tables, unless a table has outlier rows
        if (r.w0frag) {
            DevProbe d_max_probe;
            unsigned*& d_max = d_max_probe.p;
            HIP_TRY(hipMalloc((void**)&d_max, sizeof(unsigned)));
            HIP_TRY(hipMemset(d_max, 0, sizeof(unsigned)));
            bool wide = false;
            for (int f = 0; f < r.n_deep; ++f) {
                const long long rows = (long long)r.vocab[f] + 1;
                long long blocks = (rows * Dp + 255) / 256;
                if (blocks > 8192) blocks = 8192;
                hipLaunchKernelGGL(k_v2_absmax, dim3((unsigned)blocks), dim3(256), 0, 0, deep_tables[f], rows, Dp, Dp, d_max);
            }
            HIP_TRY(hipGetLastError());
            unsigned bits = 0;
            HIP_TRY(hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
            float mx;
            memcpy(&mx, &bits, sizeof(mx));
            for (int f = 0; f < r.n_deep && !wide && mx > 0.f && mx < 3.0e38f; ++f)
                if (int rcw = wide_dynamic_range(deep_tables[f], (long long)r.vocab[f] + 1, Dp, Dp, mx, &wide)) return rcw;
            if (mx > 0.f && mx < 3.0e38f && !wide) {
                int e = 0;
                (void)frexpf(mx, &e);
                e = 15 - e;
                if (e > 60) e = 60;
                if (e < -60) e = -60;
                r.e_scale = ldexpf(1.f, e);
                r.e_inv = r.inv_w0_scale / r.e_scale;
            }
        }
    }
    r.tab = nullptr;
    if (PC == 1) {
        // own deep tables ride in their field's 128-byte line: rows of <= 12 floats at float 20 (w1 stays at float 16), rows of 16
        // floats at float 16 with the first-order weights of those fields moved to a compact array (V1Run::pack)
        const int pack = !sep ? 0 : (Dp <= 12 ? 80 : (Dp == 16 ? 64 : 0));
        size_t rows = 0;
        for (int f = 0; f < nf; ++f) rows += (size_t)r.vocab[f] + 1;
        if (sep && !pack) for (int d = 0; d < r.n_deep; ++d) rows += (size_t)r.vocab[d] + 1;
        if (rows * 128 < ((size_t)1 << 32)) {                     // 32-bit byte offsets
            float* tab = nullptr;
            { const int rc_ = table_alloc(h, (void**)&tab, rows * 128); if (rc_) return rc_; }
            h->v1_bufs.push_back(tab);
            h->derived_bytes += rows * 128;
            float* w1c = nullptr;
            if (pack == 64) {
                size_t nw = 0;
                for (int d = 0; d < r.n_deep; ++d) { r.w1cbase[d] = (unsigned)nw; nw += (size_t)r.vocab[d] + 1; }
                HIP_TRY(hipMalloc((void**)&w1c, nw * sizeof(float) + 16));
                h->v1_bufs.push_back(w1c);
                for (int d = 0; d < r.n_deep; ++d)
                    HIP_TRY(hipMemcpy(w1c + r.w1cbase[d], r.w1[d], ((size_t)r.vocab[d] + 1) * sizeof(float), hipMemcpyDeviceToDevice));
            }
            size_t base = 0;
            for (int f = 0; f < nf + ((sep && !pack) ? r.n_deep : 0); ++f) {
                const bool deep_row = f >= nf;
                const bool packed_here = pack && f < r.n_deep;
                const long long n = (long long)r.vocab[deep_row ? f - nf : f] + 1;
                long long nb = (n * 32 + 255) / 256;
                if (nb > 65536) nb = 65536;
                hipLaunchKernelGGL(k_v1_build_rows, dim3((unsigned)nb), dim3(256), 0, 0, r.table[f], Dp,
                                   (deep_row || (packed_here && pack == 64)) ? (const float*)nullptr : r.w1[f], n, tab + base * 32,
                                   packed_here ? r.table[nf + f] : (const float*)nullptr, pack / 4);
                r.rowbase[f] = (unsigned)base;
                base += (size_t)n;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            r.tab = tab;
            r.pack = pack;
            r.w1c = w1c;
        }
    }
    {
        h->v1_one = r.tab && r.w1frag && PC == 1 && h->tune.v1_one;
    }
    h->v1_run = r;
    h->v1_variant = variant;
    return SPRK_OK;
}


// host_setup_rows.h -- k_rows_chain (literal DeepFM_v2, NeuralCF): dispatch table and set-up.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- dispatch table for k_rows_chain<KPC, H0C, H1C, G_BIG, NJF, HASNUM> ----
constexpr int RC_WAVES = 8;
typedef void (*RowsLaunchFn)(const RowsRun&, const int*, const float*, float*, int, int*, const float*, int, size_t, hipStream_t);
typedef void (*RowsLaunchManyFn)(const RowsRun&, const RowsMany&, int, int*, const float*, int, size_t, hipStream_t);
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM, bool UNF>
void rows_launch(const RowsRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image, int grid,
                 size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_rows_chain<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES, UNF>), dim3(grid), dim3(RC_WAVES * 64), lds, st, a, ids, dense,
                       out, B, err, image);
}
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM, bool UNF>
void rows_launch_one(const RowsRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image, int grid,
                     size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_rows_chain1<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES, UNF>), dim3(grid), dim3(RC_WAVES * 64), lds, st, a, ids, dense,
                       out, B, err, image);
}
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM, bool UNF>
void rows_launch_many(const RowsRun& a, const RowsMany& m, int B, int* err, const float* image, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_rows_chain_many<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES, UNF>), dim3(grid), dim3(RC_WAVES * 64), lds, st, a, m, B,
                       err, image);
}
struct RowsVariant {
    int kpc, h0c, h1c, g_big, njf;
    bool hasnum;
    bool unf;                             // big fields as raw split-f16 rows, projected on the matrix pipe (k_rows_chain.h, UNF)
    const void* fn;
    const void* fn_many;
    const void* fn_one;
    RowsLaunchFn launch;
    RowsLaunchFn launch_one;
    RowsLaunchManyFn launch_many;
    int image_floats, ss, rb;
};
#define ROWS_VARIANT_X(KPC, H0C, H1C, G_BIG, NJF, HASNUM, UNF)                                                                          \
    {KPC, H0C, H1C, G_BIG, NJF, HASNUM, UNF, reinterpret_cast<const void*>(&k_rows_chain<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES, UNF>),   \
     reinterpret_cast<const void*>(&k_rows_chain_many<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES, UNF>),                               \
     reinterpret_cast<const void*>(&k_rows_chain1<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES, UNF>),                                   \
     &rows_launch<KPC, H0C, H1C, G_BIG, NJF, HASNUM, UNF>, &rows_launch_one<KPC, H0C, H1C, G_BIG, NJF, HASNUM, UNF>,                         \
     &rows_launch_many<KPC, H0C, H1C, G_BIG, NJF, HASNUM, UNF>,                                                                          \
     RowsLds<KPC, H0C, H1C, HASNUM, UNF>::total_pad, RowsLds<KPC, H0C, H1C, HASNUM, UNF>::SS, RowsLds<KPC, H0C, H1C, HASNUM, UNF>::RB}
#define ROWS_VARIANT(KPC, H0C, H1C, G_BIG, NJF, HASNUM) ROWS_VARIANT_X(KPC, H0C, H1C, G_BIG, NJF, HASNUM, false)
const RowsVariant kRowsVariants[] = {
    ROWS_VARIANT_X(4, 2, 1, 2, 2, true, true),   // DeepFM_v2.py as written, big fields unfolded (raw split rows + MFMA projection)
    ROWS_VARIANT_X(4, 2, 1, 2, 1, true, true), ROWS_VARIANT_X(4, 2, 1, 1, 1, true, true), ROWS_VARIANT_X(2, 2, 1, 2, 2, true, true),
    ROWS_VARIANT(4, 2, 1, 2, 2, true),     // DeepFM_v2.py as written: Dense(64) projections, deep 32-16, movieId + userId + two genre fields
    ROWS_VARIANT(4, 2, 1, 2, 1, true), ROWS_VARIANT(4, 2, 1, 1, 1, true), ROWS_VARIANT(4, 2, 1, 3, 3, true), ROWS_VARIANT(4, 2, 1, 3, 1, true),
    ROWS_VARIANT(2, 2, 1, 2, 2, true),     // projection width 32
    ROWS_VARIANT(2, 2, 1, 3, 3, true),
    ROWS_VARIANT(1, 2, 1, 3, 3, true),     // BASELINE config 2's shape on this kernel (A/B against k_deepfm_v2_joint: SPRK_V2_ROWS=1)
    ROWS_VARIANT(0, 1, 1, 2, 0, false),    // NeuralCF.py:45-53: two embedding columns -> Dense(10) -> Dense(10) -> Dense(1)
};
int find_rows_variant(int kpc, int h0c, int h1c, int g_big, int njf, bool hasnum, bool unf = false) {
    for (size_t v = 0; v < sizeof(kRowsVariants) / sizeof(kRowsVariants[0]); ++v) {
        const RowsVariant& r = kRowsVariants[v];
        if (r.kpc == kpc && r.h0c == h0c && r.h1c == h1c && r.g_big == g_big && r.njf == njf && r.hasnum == hasnum && r.unf == unf) return (int)v;
    }
    return -1;
}
// device -> host copy of a small float matrix
int pull(std::vector<float>& dst, const float* src, size_t n) {
    dst.resize(n);
    HIP_TRY(hipMemcpy(dst.data(), src, n * sizeof(float), hipMemcpyDeviceToHost));
    return SPRK_OK;
}
int rows_finish(sprk_engine* h, const RowsVariant& rv, const std::vector<float>& image, size_t small_floats) {
    HIP_TRY(hipMalloc((void**)&h->rows_image, (size_t)rv.image_floats * sizeof(float)));
    HIP_TRY(hipMemcpy(h->rows_image, image.data(), (size_t)rv.image_floats * sizeof(float), hipMemcpyHostToDevice));
    h->rows_lds_bytes = ((size_t)rv.image_floats + RC_WAVES * 256 + small_floats) * sizeof(float);
    if (h->rows_lds_bytes > 160 * 1024) return fail(SPRK_EINVAL, "rows chain needs %zu bytes of LDS", h->rows_lds_bytes);
    HIP_TRY(hipFuncSetAttribute(rv.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->rows_lds_bytes));
    HIP_TRY(hipFuncSetAttribute(rv.fn_many, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->rows_lds_bytes));
    HIP_TRY(hipFuncSetAttribute(rv.fn_one, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->rows_lds_bytes));
    h->rows_one = true;
    HIP_TRY(hipDeviceSynchronize());
    return SPRK_OK;
}

// DeepFM_v2 plans whose projection width has no k_deepfm_v2_joint instantiation (the reference's own Dense(64)): every field
// becomes a table of rows {P | W0^T P} (+ scalars), see k_rows_chain.h.  h->v2 holds the parsed plan (match_v2_chain).
int setup_rows_v2(sprk_engine* h) {
    const V2Args& a = h->v2;
    const sprk_plan& p = h->plan;
    const int G = h->rows_g_emb;
    const sprk_op &d0 = p.ops[G + 2], &d1 = p.ops[G + 3];
    const int KP = p.ops[0].N, H0 = d0.N, H1 = d1.N, Dp = a.ldp_emb;
    int big[V2_MAX_FIELDS], nbig = 0, sm[V2_MAX_FIELDS], nsm = 0;
    for (int g = 0; g < G; ++g) {
        if ((long long)a.emb_vocab[g] + 1 <= 32 && nsm < RC_MAX_SMALL) sm[nsm++] = g;
        else big[nbig++] = g;
    }
    if (nbig < 1 || nbig > RC_MAX_BIG) return SPRK_OK;
    // UNF: the big fields' tables keep the raw embedding (16 padded values, split f16) and the projections run per task on the
    // matrix pipe -- when the fold would multiply the bytes per row (emb_dim 10 -> 96 floats for DeepFM_v2.py as written)
    int variant = -1;
    float p_scale = 1.f;
    if (h->tune.rows_unf && h->tune.dyn_f16 && nbig <= 2 && Dp <= 16 && KP + H0 > 16) {
        const int vu = find_rows_variant(KP / 16, H0 / 16, H1 / 16, nbig, nsm, true, true);
        if (vu >= 0) {
            DevProbe d_max_probe;
            unsigned*& d_max = d_max_probe.p;
            HIP_TRY(hipMalloc((void**)&d_max, sizeof(unsigned)));
            HIP_TRY(hipMemset(d_max, 0, sizeof(unsigned)));
            for (int b = 0; b < nbig; ++b)
                hipLaunchKernelGGL(k_v2_absmax, dim3(1024), dim3(256), 0, 0, a.table[big[b]], (long long)a.emb_vocab[big[b]] + 1, Dp, Dp, d_max);
            unsigned bits = 0;
            HIP_TRY(hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
            float mx;
            memcpy(&mx, &bits, sizeof(mx));
            bool ok = mx < 3.0e38f;
            for (int b = 0; ok && b < nbig; ++b) {
                bool wide = false;
                if (int rcw = wide_dynamic_range(a.table[big[b]], (long long)a.emb_vocab[big[b]] + 1, Dp, Dp, mx, &wide)) return rcw;
                if (wide) ok = false;                              // an outlier row: its neighbours' lo halves would be subnormal
            }
            if (ok) {
                int e = 0;
                if (mx > 0.f) { (void)frexpf(mx, &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; p_scale = ldexpf(1.f, e); }
                variant = vu;
            }
        }
    }
    if (variant < 0) variant = find_rows_variant(KP / 16, H0 / 16, H1 / 16, nbig, nsm, true);
    if (variant < 0) return SPRK_OK;
    const RowsVariant& rv = kRowsVariants[variant];
    const bool unf = rv.unf;
    // first-order weights in embedding-group order (one ids column feeds both)
    const float* w1g[V2_MAX_FIELDS];
    for (int g = 0; g < G; ++g) {
        w1g[g] = nullptr;
        for (int i = 0; i < a.n_fo; ++i)
            if (a.fo_col[i] == a.emb_col[g] && a.fo_vocab[i] == a.emb_vocab[g]) w1g[g] = a.w1[i];
        if (!w1g[g]) return SPRK_OK;
    }
    RowsRun& r = h->rows_run;
    memset(&r, 0, sizeof(r));
    r.F = a.F; r.ND = a.ND; r.n_num = a.n_num;
    size_t big_rows = 0;
    for (int b = 0; b < nbig; ++b) {
        r.big_col[b] = a.emb_col[big[b]]; r.big_vocab[b] = a.emb_vocab[big[b]];
        r.big_rowbase[b] = (unsigned)big_rows; r.big_scal[b] = (unsigned)big_rows;
        big_rows += (size_t)a.emb_vocab[big[b]] + 1;
    }
    if (big_rows >= ((size_t)1 << 31)) return SPRK_OK;
    size_t small_floats = 0;
    for (int f = 0; f < nsm; ++f) {
        r.s_col[f] = a.emb_col[sm[f]]; r.s_vocab[f] = a.emb_vocab[sm[f]]; r.s_off[f] = (int)small_floats;
        small_floats += ((size_t)a.emb_vocab[sm[f]] + 1) * rv.ss;
    }
    small_floats = (small_floats + 255) & ~(size_t)255;
    { const int rc_ = table_alloc(h, (void**)&h->rows_tab, big_rows * rv.rb + 64); if (rc_) return rc_; }
    HIP_TRY(hipMemset(h->rows_tab, 0, big_rows * rv.rb + 64));
    HIP_TRY(hipMalloc((void**)&h->rows_scal, big_rows * sizeof(float) + 16));
    h->derived_bytes += big_rows * rv.rb + big_rows * sizeof(float);
    if (small_floats) {
        HIP_TRY(hipMalloc((void**)&h->rows_small, small_floats * sizeof(float)));
        HIP_TRY(hipMemset(h->rows_small, 0, small_floats * sizeof(float)));
    }
    auto build = [&](int g, float* out, int out_stride, float* scal_out) {
        const long long rows = (long long)a.emb_vocab[g] + 1;
        long long blocks = (rows + 3) / 4;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_rows_build, dim3((unsigned)blocks), dim3(256), 0, 0, a.table[g], Dp, rows, a.Wp[g], a.ldp_emb, a.bp[g], KP,
                           a.W0, d0.ldw, g * KP, H0, KP, (const float*)nullptr, w1g[g], a.hfm, a.n_hfm, a.h0w, out, out_stride, scal_out,
                           scal_out ? 0 : 1);
    };
    // UNF: lin[b][d][n] = column d of {Wp^T | (Wp W0_b)^T} (the builder run on unit vectors without bias), cst[b][n] = {bp | W0_b^T bp}
    std::vector<std::vector<float>> lin(nbig), cst(nbig);
    for (int b = 0; b < nbig; ++b) {
        if (!unf) {
            build(big[b], h->rows_tab + (size_t)r.big_rowbase[b] * (rv.rb / 4), rv.rb / 4, h->rows_scal + r.big_scal[b]);
            continue;
        }
        const int g = big[b];
        const long long rows = (long long)a.emb_vocab[g] + 1;
        build(g, nullptr, 0, h->rows_scal + r.big_scal[b]);                       // the per-id scalars only
        long long nb = (rows * 16 + 255) / 256;
        if (nb > 65536) nb = 65536;
        hipLaunchKernelGGL(k_rows_unf_split, dim3((unsigned)nb), dim3(256), 0, 0, a.table[g], Dp, rows, p_scale,
                           reinterpret_cast<_Float16*>(reinterpret_cast<char*>(h->rows_tab) + (size_t)r.big_rowbase[b] * 64));
        std::vector<float> ident((size_t)(Dp + 1) * Dp, 0.f);
        for (int d = 0; d < Dp; ++d) ident[(size_t)d * Dp + d] = 1.f;
        float *d_id = nullptr, *d_out = nullptr;
        const int W = KP + H0;
        HIP_TRY(hipMalloc((void**)&d_id, ident.size() * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&d_out, (size_t)(Dp + 1) * W * sizeof(float)));
        HIP_TRY(hipMemcpy(d_id, ident.data(), ident.size() * sizeof(float), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_rows_build, dim3(4), dim3(256), 0, 0, (const float*)d_id, Dp, (long long)Dp, a.Wp[g], a.ldp_emb, (const float*)nullptr, KP,
                           a.W0, d0.ldw, g * KP, H0, KP, (const float*)nullptr, (const float*)nullptr, a.hfm, 0, 0.f, d_out, W, (float*)nullptr, 0);
        hipLaunchKernelGGL(k_rows_build, dim3(1), dim3(256), 0, 0, (const float*)(d_id + (size_t)Dp * Dp), Dp, 1ll, a.Wp[g], a.ldp_emb, a.bp[g], KP,
                           a.W0, d0.ldw, g * KP, H0, KP, (const float*)nullptr, (const float*)nullptr, a.hfm, 0, 0.f, d_out + (size_t)Dp * W, W,
                           (float*)nullptr, 0);
        int rcp = hipGetLastError() == hipSuccess ? SPRK_OK : fail(SPRK_EHIP, "rows chain: projection probe launch failed");
        if (!rcp) rcp = pull(lin[b], d_out, (size_t)Dp * W);
        if (!rcp) rcp = pull(cst[b], d_out + (size_t)Dp * W, W);
        (void)hipFree(d_id); (void)hipFree(d_out);
        if (rcp) return rcp;
    }
    for (int f = 0; f < nsm; ++f) build(sm[f], h->rows_small + r.s_off[f], rv.ss, nullptr);
    HIP_TRY(hipGetLastError());
    // weight image (host): Wn, bn, M = W0[:, num block] Wn, c0 = b0 + W0[:, num block] bn, W1, b1, hfm, hd, fn
    std::vector<float> Wn, bn, W0, b0, W1, b1, hfm, hd, fnw;
    int rc;
    if ((rc = pull(Wn, a.Wp[G], (size_t)KP * a.ldp_num)) || (rc = pull(bn, a.bp[G], KP)) || (rc = pull(W0, a.W0, (size_t)H0 * d0.ldw)) ||
        (rc = pull(b0, a.b0, H0)) || (rc = pull(W1, a.W1, (size_t)H1 * d1.ldw)) || (rc = pull(b1, a.b1, H1)) ||
        (rc = pull(hfm, a.hfm, a.n_hfm)) || (rc = pull(hd, a.hdeep, a.n_hdeep)) || (rc = pull(fnw, a.fo_num_w, a.n_num))) return rc;
    std::vector<float> img(rv.image_floats, 0.f);
    const int SN = 12, S1 = H0 + 4;
    int off = 0;
    const int off_wn = off; off += KP * SN;
    const int off_bn = off; off += KP;
    const int off_m = off; off += H0 * SN;
    const int off_c0 = off; off += H0;
    const int off_w1 = off; off += H1 * S1;
    const int off_b1 = off; off += H1;
    const int off_hfm = off; off += KP;
    const int off_hd = off; off += H1;
    const int off_fn = off; off += 8;
    const int off_cp = off; if (unf) off += KP;
    const int off_af = (off + 3) & ~3; if (unf) off = off_af + (KP + H0) / 16 * 2 * 256;
    if (off > rv.image_floats) return fail(SPRK_EINVAL, "rows image layout mismatch");
    for (int n = 0; n < KP; ++n) {
        for (int k = 0; k < a.n_num && k < 8; ++k) img[off_wn + n * SN + k] = Wn[(size_t)n * a.ldp_num + k];
        img[off_bn + n] = bn[n];
    }
    for (int m = 0; m < H0; ++m) {
        const float* w = &W0[(size_t)m * d0.ldw + (size_t)G * KP];
        for (int k = 0; k < a.n_num && k < 8; ++k) {
            double acc = 0.0;
            for (int n = 0; n < KP; ++n) acc += (double)w[n] * (double)Wn[(size_t)n * a.ldp_num + k];
            img[off_m + m * SN + k] = (float)acc;
        }
        double c = b0[m];
        for (int n = 0; n < KP; ++n) c += (double)w[n] * (double)bn[n];
        img[off_c0 + m] = (float)c;
    }
    for (int n = 0; n < H1; ++n) {
        for (int k = 0; k < H0; ++k) img[off_w1 + n * S1 + k] = W1[(size_t)n * d1.ldw + k];
        img[off_b1 + n] = b1[n];
    }
    for (int n = 0; n < a.n_hfm && n < KP; ++n) img[off_hfm + n] = hfm[n];
    for (int n = 0; n < a.n_hdeep && n < H1; ++n) img[off_hd + n] = hd[n];
    for (int k = 0; k < a.n_num && k < 8; ++k) img[off_fn + k] = a.h0w * fnw[k];
    r.unscale = 1.f;
    if (unf) {
        const int W = KP + H0;
        float amax = 0.f;
        for (int b = 0; b < nbig; ++b) {
            for (float v : lin[b]) amax = fmaxf(amax, fabsf(v));
            for (int n = 0; n < KP; ++n) img[off_cp + n] += cst[b][n];
            for (int m = 0; m < H0; ++m) img[off_c0 + m] += cst[b][KP + m];
        }
        if (!(amax < 3.0e38f)) return fail(SPRK_EINVAL, "non-finite projection weights");
        float w_scale = 1.f;
        { int e = 0; if (amax > 0.f) { (void)frexpf(amax, &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; w_scale = ldexpf(1.f, e); } }
        _Float16* fh = reinterpret_cast<_Float16*>(&img[off_af]);
        for (int nb = 0; nb < W / 16; ++nb)
            for (int ln = 0; ln < 64; ++ln)
                for (int e = 0; e < 8; ++e) {
                    const int n = nb * 16 + (ln & 15), k = 8 * (ln >> 4) + e, b = k >> 4, d = k & 15;
                    const float x = (b < nbig && d < Dp) ? lin[b][(size_t)d * W + n] * w_scale : 0.f;
                    const _Float16 hi = (_Float16)x;
                    fh[(size_t)(2 * nb) * 512 + ln * 8 + e] = hi;
                    fh[(size_t)(2 * nb + 1) * 512 + ln * 8 + e] = (_Float16)(x - (float)hi);
                }
        r.unscale = 1.f / (p_scale * w_scale);
    }
    r.rows = h->rows_tab; r.scal = h->rows_scal; r.small = h->rows_small; r.small_floats = (int)small_floats;
    r.bias = a.head_bias + a.h0w * a.fo_bias;
    if ((rc = rows_finish(h, rv, img, small_floats))) return rc;
    h->rows_variant = variant;
    return SPRK_OK;
}

// NeuralCF.py:45-53 (neural_cf_model_1): concat(item row, user row) -> Dense(relu) -> Dense(relu) -> Dense(1, sigmoid).  The first
// Dense is linear in each row, so each field becomes a table of its 16 (padded) pre-activations: Q_f[id] = W0[:, f]^T E_f[id].
int setup_rows_ncf(sprk_engine* h) {
    if (!h->tune.ncf_chain) return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (p.model_kind != SPRK_MODEL_NEURALCF || p.din.enabled || p.n_segs != 2 || p.n_ops != 2 || p.n_taps != 1 || p.n_dense != 0) return SPRK_OK;
    if (p.n_id_cols > 8) return SPRK_OK;
    const sprk_op &o0 = p.ops[0], &o1 = p.ops[1];
    const sprk_tap& tp = p.taps[0];
    if (o0.kind != SPRK_OP_DENSE || o1.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_RELU || o1.act != SPRK_ACT_RELU) return SPRK_OK;
    if (o0.src_buf != 0 || o0.dst_off != 0 || o1.src_buf != o0.dst_buf || o1.src_off != 0 || o1.K != o0.N || o1.dst_off != 0) return SPRK_OK;
    if (tp.buf != o1.dst_buf || tp.off != 0 || tp.len > o1.N || tp.w_slot < 0 || tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
    const int H0 = o0.N, H1 = o1.N;
    const int variant = find_rows_variant(0, H0 / 16, H1 / 16, 2, 0, false);
    if (variant < 0) return SPRK_OK;
    const RowsVariant& rv = kRowsVariants[variant];
    RowsRun& r = h->rows_run;
    memset(&r, 0, sizeof(r));
    r.F = p.n_id_cols; r.ND = 0; r.n_num = 0;
    size_t rows_total = 0;
    for (int b = 0; b < 2; ++b) {
        const sprk_seg& sg = p.segs[b];
        if (sg.kind != SPRK_SEG_ROWS || sg.dst < o0.src_off || sg.dst + 4 * sg.count > o0.src_off + o0.K || 4 * sg.count > 64) return SPRK_OK;
        if (h->slot_bytes[sg.slot] < ((size_t)sg.vocab + 1) * sg.row_stride * sizeof(float)) return SPRK_OK;   // needs the zero row at index vocab
        r.big_col[b] = sg.field; r.big_vocab[b] = sg.vocab; r.big_rowbase[b] = (unsigned)rows_total;
        rows_total += (size_t)sg.vocab + 1;
    }
    if (rows_total >= ((size_t)1 << 31)) return SPRK_OK;
    { const int rc_ = table_alloc(h, (void**)&h->rows_tab, rows_total * rv.rb + 64); if (rc_) return rc_; }
    HIP_TRY(hipMemset(h->rows_tab, 0, rows_total * rv.rb + 64));
    h->derived_bytes += rows_total * rv.rb;
    const float* W0 = (const float*)h->slot_ptr[o0.w_slot];
    for (int b = 0; b < 2; ++b) {
        const sprk_seg& sg = p.segs[b];
        const long long rows = (long long)sg.vocab + 1;
        long long blocks = (rows + 3) / 4;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_rows_build, dim3((unsigned)blocks), dim3(256), 0, 0, (const float*)h->slot_ptr[sg.slot], sg.row_stride, rows,
                           (const float*)nullptr, 0, (const float*)nullptr, 0, W0, o0.ldw, sg.dst - o0.src_off, H0, 4 * sg.count,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0.f,
                           h->rows_tab + (size_t)r.big_rowbase[b] * (rv.rb / 4), rv.rb / 4, (float*)nullptr, 0);
    }
    HIP_TRY(hipGetLastError());
    std::vector<float> b0, W1, b1, hd;
    int rc;
    if ((rc = pull(b0, (const float*)h->slot_ptr[o0.b_slot], H0)) || (rc = pull(W1, (const float*)h->slot_ptr[o1.w_slot], (size_t)H1 * o1.ldw)) ||
        (rc = pull(b1, (const float*)h->slot_ptr[o1.b_slot], H1)) || (rc = pull(hd, (const float*)h->slot_ptr[tp.w_slot], tp.len))) return rc;
    std::vector<float> img(rv.image_floats, 0.f);
    const int S1 = H0 + 4;
    const int off_c0 = 0, off_w1 = off_c0 + H0, off_b1 = off_w1 + H1 * S1, off_hfm = off_b1 + H1, off_hd = off_hfm + 0;
    for (int m = 0; m < H0; ++m) img[off_c0 + m] = b0[m];
    for (int n = 0; n < H1; ++n) {
        for (int k = 0; k < H0; ++k) img[off_w1 + n * S1 + k] = W1[(size_t)n * o1.ldw + k];
        img[off_b1 + n] = b1[n];
    }
    for (int n = 0; n < tp.len; ++n) img[off_hd + n] = hd[n];
    r.rows = h->rows_tab; r.scal = nullptr; r.small = nullptr; r.small_floats = 0;
    r.bias = p.head_bias;
    if ((rc = rows_finish(h, rv, img, 0))) return rc;
    h->rows_variant = variant;
    return SPRK_OK;
}


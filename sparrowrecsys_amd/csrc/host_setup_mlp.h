// host_setup_mlp.h -- EmbeddingMLP / Wide&Deep: k_mlp_rows set-up.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- k_mlp_rows<8, 8, NBIG, NS, WAVES, DYN, WK> ----
constexpr int MR_WAVES = 8;
typedef void (*MlpRowsKernel)(const MlpRowsRun, const int*, const float*, float*, int, int*, const float*);
// NS = 8 (EmbeddingMLP.py / WideNDeep.py as written: eight genre columns) has its own instantiations for the reference's shape
// (two big columns, split-f16 second layer); every other shape runs the generic ones (NS = -1: the count is a run-time value).
template <int NBIG, int NS, bool DYN>
MlpRowsKernel mlp_rows_kernel_wk(int wk) {
    return wk == 1 ? &k_mlp_rows<8, 8, NBIG, NS, MR_WAVES, DYN, 1> : wk == 2 ? &k_mlp_rows<8, 8, NBIG, NS, MR_WAVES, DYN, 2> : &k_mlp_rows<8, 8, NBIG, NS, MR_WAVES, DYN, 0>;
}
typedef void (*MlpRowsManyKernel)(const MlpRowsRun, const MlpRowsMany, int, int*, const float*);
template <int NBIG, int NS, bool DYN>
MlpRowsManyKernel mlp_rows_many_kernel_wk(int wk) {
    return wk == 1 ? &k_mlp_rows_many<8, 8, NBIG, NS, MR_WAVES, DYN, 1> : wk == 2 ? &k_mlp_rows_many<8, 8, NBIG, NS, MR_WAVES, DYN, 2> : &k_mlp_rows_many<8, 8, NBIG, NS, MR_WAVES, DYN, 0>;
}
// [r6] several batches per launch: the split-f16 forms (every model whose weights are finite and bounded: the set-up's choice); the f32-MFMA
// fall-back forms keep going batch by batch
MlpRowsManyKernel mlp_rows_many_kernel(int nbig, int nsmall, bool dyn, int wk) {
    if (!dyn) return nullptr;
    if (nbig == 2 && nsmall == 8) return mlp_rows_many_kernel_wk<2, 8, true>(wk);
    if (nbig == 2) return mlp_rows_many_kernel_wk<2, -1, true>(wk);
    return mlp_rows_many_kernel_wk<1, -1, true>(wk);
}
MlpRowsKernel mlp_rows_kernel(int nbig, int nsmall, bool dyn, int wk) {
    if (nbig == 2 && nsmall == 8 && dyn) return mlp_rows_kernel_wk<2, 8, true>(wk);
    if (nbig == 2) return dyn ? mlp_rows_kernel_wk<2, -1, true>(wk) : mlp_rows_kernel_wk<2, -1, false>(wk);
    return dyn ? mlp_rows_kernel_wk<1, -1, true>(wk) : mlp_rows_kernel_wk<1, -1, false>(wk);
}
// Recognise an EmbeddingMLP / Wide&Deep plan (EmbeddingMLP.py:72-77, WideNDeep.py:99-107) with ReLU layers of 128 and fold
// EVERY embedding column through the first Dense layer (see k_mlp_rows.h).  Leaves mlp_rows_nbig = -1 for any other shape.
int setup_mlp_rows(sprk_engine* h) {
    if (!h->tune.mlp_chain) return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (p.din.enabled || p.n_ops != 2 || p.n_taps < 1 || p.n_taps > 2) return SPRK_OK;
    if (p.model_kind != SPRK_MODEL_EMBEDDING_MLP && p.model_kind != SPRK_MODEL_WIDE_DEEP) return SPRK_OK;
    if (p.n_id_cols > 12 || p.n_dense > 8 || p.n_dense < 1) return SPRK_OK;
    const sprk_op &o0 = p.ops[0], &o1 = p.ops[1];
    if (o0.kind != SPRK_OP_DENSE || o1.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_RELU || o1.act != SPRK_ACT_RELU) return SPRK_OK;
    if (o0.src_buf != 0 || o0.dst_buf == 0 || o0.dst_off != 0 || o1.src_buf != o0.dst_buf || o1.src_off != 0 || o1.K != o0.N ||
        o1.dst_off != 0 || o0.N != 128 || o1.N != 128) return SPRK_OK;
    const int lo = o0.src_off, hi = o0.src_off + o0.K, N0 = 128;
    MlpRowsRun r;
    memset(&r, 0, sizeof(r));
    const sprk_seg* big_seg[MR_MAX_BIG];
    const sprk_seg* small_seg[MR_MAX_SMALL];
    const sprk_seg* cross = nullptr;
    int num_dst = -1;
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& sg = p.segs[i];
        if (sg.kind == SPRK_SEG_ROWS) {
            if (sg.dst < lo || sg.dst + 4 * sg.count > hi) return SPRK_OK;
            if ((long long)sg.vocab <= 31 && r.n_small < MR_MAX_SMALL) small_seg[r.n_small++] = &sg;
            else if (r.n_big < MR_MAX_BIG) big_seg[r.n_big++] = &sg;
            else return SPRK_OK;
            if (((size_t)sg.vocab + 1) * N0 * sizeof(float) >= ((size_t)4 << 30)) return SPRK_OK;      // (32-bit byte offsets into a folded table)
        } else if (sg.kind == SPRK_SEG_DENSE) {
            if (num_dst >= 0 || sg.field != 0 || sg.count > 8 || sg.dst < lo || sg.dst + sg.count > hi) return SPRK_OK;
            num_dst = sg.dst; r.n_num = sg.count;
        } else if (sg.kind == SPRK_SEG_CROSS_ROWS || sg.kind == SPRK_SEG_CROSS_SCALAR) {
            if (cross || (sg.dst < hi && sg.dst + (sg.kind == SPRK_SEG_CROSS_ROWS ? 4 * sg.count : 1) > lo)) return SPRK_OK;
            cross = &sg;
        } else if (sg.kind != SPRK_SEG_ZERO) {
            return SPRK_OK;
        }
    }
    if (r.n_big < 1 || r.n_big > 2 || num_dst < 0 || r.n_num < 1) return SPRK_OK;   // (three big columns spill: 3 x 8 float4 in flight)
    const sprk_tap *tdeep = nullptr, *twide = nullptr;
    for (int t = 0; t < p.n_taps; ++t) {
        const sprk_tap& tp = p.taps[t];
        if (tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
        if (tp.buf == o1.dst_buf && tp.off == 0 && tp.len <= o1.N && tp.w_slot >= 0 && !tdeep) tdeep = &tp;
        else if (cross && tp.buf == 0 && tp.off == cross->dst && !twide) twide = &tp;
        else return SPRK_OK;
    }
    if (!tdeep || (cross != nullptr) != (twide != nullptr)) return SPRK_OK;
    if (cross) {
        if (cross->kind == SPRK_SEG_CROSS_ROWS) {
            if (twide->len != 4 * cross->count || twide->w_slot < 0 || twide->len > 32) return SPRK_OK;
            r.wide_kind = 1; r.wide_dim = twide->len; r.wide_stride = cross->row_stride; r.wide_w = (const float*)h->slot_ptr[twide->w_slot];
        } else {
            if (twide->len != 1 || twide->w_slot >= 0) return SPRK_OK;
            r.wide_kind = 2;
        }
        r.wide_a = cross->field; r.wide_b = cross->field2; r.wide_buckets = cross->vocab;
        r.wide_magic = (cross->vocab > 0 && (long long)cross->vocab < (1ll << 30)) ? ~0ULL / (unsigned long long)cross->vocab : 0ULL;
        r.wide_tab = (const float*)h->slot_ptr[cross->slot];
    }
    // the second layer's form first: the LDS image's size depends on it (split-f16 fragments: 64 KB; W1^T in f32: 66 KB)
    float* w1frag = nullptr;
    {
        float w_scale = 0.f;
        const int rc2 = make_dyn_fragments(h, (const float*)h->slot_ptr[o1.w_slot], o1.ldw, o1.N, o1.K, &w1frag, &w_scale);
        if (rc2) return rc2;
        r.inv_w1_scale = w1frag ? 1.0f / w_scale : 0.f;
    }
    const bool dyn = w1frag != nullptr;
    const int image_lds = dyn ? MlpRowsLds<8, 8, true>::total_pad : MlpRowsLds<8, 8, false>::total_pad;
    const int image_floats = dyn ? MlpRowsLds<8, 8, true>::image_floats : MlpRowsLds<8, 8, false>::image_floats;
    // LDS: fixed image + small tables (rows MR_RS floats apart, + one shared zero row) + a staging slot per wave
    size_t small_floats = 0;
    for (int f = 0; f < r.n_small; ++f) { r.s_off[f] = (int)small_floats; small_floats += (size_t)small_seg[f]->vocab * MR_RS; }
    r.zero_off = (int)small_floats;
    small_floats += MR_RS;
    small_floats = (small_floats + 255) & ~(size_t)255;
    const size_t lds = ((size_t)image_lds + small_floats + (size_t)MR_WAVES * MR_STAGE) * sizeof(float);
    if (lds > 160 * 1024) return SPRK_OK;
    const float* W0 = (const float*)h->slot_ptr[o0.w_slot];
    HIP_TRY(hipMalloc((void**)&h->mlp_rows_small, small_floats * sizeof(float)));
    HIP_TRY(hipMemset(h->mlp_rows_small, 0, small_floats * sizeof(float)));
    auto fold = [&](const sprk_seg& sg, float* F) {
        long long blocks = ((long long)sg.vocab * N0 + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_fold_dense_rows, dim3((unsigned)blocks), dim3(256), 0, 0, (const float*)h->slot_ptr[sg.slot], (long long)sg.vocab,
                           sg.row_stride, 4 * sg.count, W0, o0.ldw, sg.dst - lo, N0, F);
    };
    {
        // small columns: fold into a scratch buffer, then into the padded LDS layout (k_mlp_rows.h)
        float* tmp = nullptr;
        HIP_TRY(hipMalloc((void**)&tmp, (size_t)32 * N0 * sizeof(float)));
        for (int f = 0; f < r.n_small; ++f) {
            r.s_col[f] = small_seg[f]->field; r.s_vocab[f] = small_seg[f]->vocab;
            fold(*small_seg[f], tmp);
            hipLaunchKernelGGL(k_mlp_rows_pad, dim3(4), dim3(256), 0, 0, tmp, h->mlp_rows_small + r.s_off[f], small_seg[f]->vocab);
        }
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(tmp);
    }
    for (int b = 0; b < r.n_big; ++b) {
        const sprk_seg& sg = *big_seg[b];
        float* F = nullptr;
        const size_t bytes = ((size_t)sg.vocab + 1) * N0 * sizeof(float);
        { const int rc_ = table_alloc(h, (void**)&F, bytes); if (rc_) return rc_; }
        h->mlp_rows_bufs.push_back(F);
        h->derived_bytes += bytes;
        HIP_TRY(hipMemset(F + (size_t)sg.vocab * N0, 0, N0 * sizeof(float)));        // the "no id" row
        fold(sg, F);
        r.big_col[b] = sg.field; r.big_vocab[b] = sg.vocab; r.big_tab[b] = F;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMalloc((void**)&h->mlp_rows_image, (size_t)image_floats * sizeof(float)));
    if (dyn)
        hipLaunchKernelGGL((k_mlp_rows_pack<8, 8, true>), dim3(1), dim3(256), 0, 0, W0, o0.ldw, num_dst - lo, r.n_num, (const float*)h->slot_ptr[o0.b_slot],
                           (const float*)h->slot_ptr[o1.w_slot], o1.ldw, (const float*)h->slot_ptr[o1.b_slot],
                           (const float*)h->slot_ptr[tdeep->w_slot], tdeep->len, w1frag, h->mlp_rows_image);
    else
        hipLaunchKernelGGL((k_mlp_rows_pack<8, 8, false>), dim3(1), dim3(256), 0, 0, W0, o0.ldw, num_dst - lo, r.n_num, (const float*)h->slot_ptr[o0.b_slot],
                           (const float*)h->slot_ptr[o1.w_slot], o1.ldw, (const float*)h->slot_ptr[o1.b_slot],
                           (const float*)h->slot_ptr[tdeep->w_slot], tdeep->len, w1frag, h->mlp_rows_image);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    r.F = p.n_id_cols; r.ND = p.n_dense; r.head_bias = p.head_bias;
    r.small = h->mlp_rows_small; r.small_floats = (int)small_floats;
    if (r.n_num < 8) r.flags |= 2;                                       // b0 in the numerics' eighth K slot (k_mlp_rows_pack put it there)
    h->mlp_rows_kernel = mlp_rows_kernel(r.n_big, r.n_small, dyn, r.wide_kind);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(h->mlp_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->mlp_rows_many_kernel = h->tune.mlp_rows_many ? mlp_rows_many_kernel(r.n_big, r.n_small, dyn, r.wide_kind) : nullptr;
    if (h->mlp_rows_many_kernel)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(h->mlp_rows_many_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->mlp_rows_run = r;
    h->mlp_rows_lds = lds;
    h->mlp_rows_nbig = r.n_big;
    return SPRK_OK;
}

// api_engine.h -- C ABI: sprk_last_error .. sprk_create / sprk_upload / sprk_finalize / sprk_workspace_bytes.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
extern "C" {

const char* sprk_last_error(void) { return g_err.c_str(); }

int sprk_runtime_info(int32_t info[4]) {
    if (!info) return fail(SPRK_EINVAL, "info is NULL");
    info[0] = SPRK_ABI_VERSION;
    info[1] = info[2] = info[3] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    info[1] = n;
    if (n > 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            info[2] = prop.multiProcessorCount;
            info[3] = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
        }
    }
    return SPRK_OK;
}

int sprk_create(const sprk_plan* plan, sprk_handle* out) {
    if (!plan || !out) return fail(SPRK_EINVAL, "plan/out is NULL");
    *out = nullptr;
    int rc = validate_plan(*plan);
    if (rc) return rc;
    sprk_engine* h = new (std::nothrow) sprk_engine();
    if (!h) return fail(SPRK_EHIP, "out of host memory");
    h->plan = *plan;
    h->slot_ptr.assign(plan->n_slots, nullptr);
    h->slot_bytes.assign(plan->n_slots, 0);
    h->slot_external.assign(plan->n_slots, 0);
    int off = 0;
    for (int b = 0; b < plan->n_bufs; ++b) {
        h->buf_stride[b] = lds_stride(plan->buf_width[b]);
        h->buf_base[b] = off;
        off += SPRK_TILE_M * h->buf_stride[b];
    }
    // the tile's ids block [64][columns some gather segment reads]
    auto use_col = [&](int c) {
        for (int x : h->idc) if (x == c) return;
        h->idc.push_back(c);
    };
    for (int i = 0; i < plan->n_segs; ++i) {
        const sprk_seg& sg = plan->segs[i];
        if (sg.kind == SPRK_SEG_ROWS || sg.kind == SPRK_SEG_SCALAR || sg.kind == SPRK_SEG_CROSS_ROWS || sg.kind == SPRK_SEG_CROSS_SCALAR) use_col(sg.field);
        if (sg.kind == SPRK_SEG_CROSS_ROWS || sg.kind == SPRK_SEG_CROSS_SCALAR) use_col(sg.field2);
    }
    h->ids_base = off;
    off += (SPRK_TILE_M * (int)h->idc.size() + 3) & ~3;
    h->tile_lds_bytes = (size_t)off * sizeof(float);
    if (h->tile_lds_bytes > 160 * 1024) {
        size_t need = h->tile_lds_bytes;
        delete h;
        return fail(SPRK_EINVAL, "plan needs %zu bytes of LDS per tile (> 160 KiB)", need);
    }
    *out = h;
    return SPRK_OK;
}

int sprk_upload(sprk_handle h, int32_t slot, const void* src, size_t bytes) {
    if (!h || !src || bytes == 0) return fail(SPRK_EINVAL, "bad upload arguments");
    if (slot < 0 || slot >= h->plan.n_slots) return fail(SPRK_EINVAL, "slot %d outside [0,%d)", slot, h->plan.n_slots);
    if (h->finalized) return fail(SPRK_ESTATE, "upload after finalize");
    if (h->slot_ptr[slot] && !h->slot_external[slot]) (void)hipFree(h->slot_ptr[slot]);
    h->slot_ptr[slot] = nullptr;
    h->slot_external[slot] = 0;
    // 16 spare bytes so a float4 tail read of a [len]-float vector never leaves the allocation
    HIP_TRY(hipMalloc(&h->slot_ptr[slot], bytes + 16));
    HIP_TRY(hipMemset(h->slot_ptr[slot], 0, bytes + 16));
    HIP_TRY(hipMemcpy(h->slot_ptr[slot], src, bytes, hipMemcpyDefault));
    h->slot_bytes[slot] = bytes;
    return SPRK_OK;
}

int sprk_finalize(sprk_handle h) {
    RoctxRange roctx_range_("sprk_finalize");
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (h->finalized) return SPRK_OK;
    h->tune = SprkTuning::from_env();                     // the ONE place an engine reads the environment
    struct TuneScope { const SprkTuning*& slot; ~TuneScope() { slot = nullptr; } } tune_scope{g_finalize_tune};
    g_finalize_tune = &h->tune;
    const sprk_plan& p = h->plan;
    DevPlan* dp = new (std::nothrow) DevPlan();
    if (!dp) return fail(SPRK_EHIP, "out of host memory");
    memset(dp, 0, sizeof(DevPlan));
    struct Guard { DevPlan* p; ~Guard() { delete p; } } guard{dp};
    dp->F = p.n_id_cols; dp->ND = p.n_dense; dp->NA = p.n_aux;
    dp->n_segs = p.n_segs; dp->n_ops = p.n_ops; dp->n_taps = p.n_taps; dp->n_pairs = p.n_pairs; dp->n_bufs = p.n_bufs;
    dp->head_bias = p.head_bias;
    for (int b = 0; b < SPRK_MAX_BUFS; ++b) { dp->buf_stride[b] = h->buf_stride[b]; dp->buf_base[b] = h->buf_base[b]; }
    dp->ids_base = h->ids_base;
    dp->n_idc = (int)h->idc.size();
    for (size_t i = 0; i < h->idc.size(); ++i) dp->idc[i] = h->idc[i];
    auto compact = [&](int c) { for (size_t i = 0; i < h->idc.size(); ++i) if (h->idc[i] == c) return (int)i; return 0; };
    for (int i = 0; i < p.n_pairs; ++i) { dp->pair_a[i] = p.pair_a[i]; dp->pair_b[i] = p.pair_b[i]; }
    int rc;
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& s = p.segs[i];
        DevSeg& d = dp->segs[i];
        d.kind = s.kind; d.field = s.field; d.field2 = s.field2; d.row_stride = s.row_stride; d.count = s.count; d.dst = s.dst; d.vocab = s.vocab;
        if (s.kind == SPRK_SEG_ROWS || s.kind == SPRK_SEG_SCALAR || s.kind == SPRK_SEG_CROSS_ROWS || s.kind == SPRK_SEG_CROSS_SCALAR) d.field = compact(s.field);
        if (s.kind == SPRK_SEG_CROSS_ROWS || s.kind == SPRK_SEG_CROSS_SCALAR) d.field2 = compact(s.field2);
        d.table = nullptr;
        if (s.kind == SPRK_SEG_ROWS || s.kind == SPRK_SEG_CROSS_ROWS) {
            if ((rc = need_bytes(h, s.slot, (size_t)s.vocab * s.row_stride * 4, "embedding table"))) return rc;
            d.table = (const float*)h->slot_ptr[s.slot];
        } else if (s.kind == SPRK_SEG_SCALAR || s.kind == SPRK_SEG_CROSS_SCALAR) {
            if ((rc = need_bytes(h, s.slot, (size_t)s.vocab * 4, "first-order table"))) return rc;
            d.table = (const float*)h->slot_ptr[s.slot];
        }
    }
    for (int i = 0; i < p.n_ops; ++i) {
        const sprk_op& o = p.ops[i];
        DevOp& d = dp->ops[i];
        d.kind = o.kind; d.src_buf = o.src_buf; d.src_off = o.src_off; d.K = o.K; d.dst_buf = o.dst_buf; d.dst_off = o.dst_off;
        d.N = o.N; d.ldw = o.ldw; d.act = o.act; d.groups = o.groups; d.group_stride = o.group_stride;
        if (o.kind == SPRK_OP_DENSE) {
            if ((rc = need_bytes(h, o.w_slot, (size_t)o.N * o.ldw * 4, "Dense kernel"))) return rc;
            if ((rc = need_bytes(h, o.b_slot, (size_t)o.N * 4, "Dense bias"))) return rc;
            d.W = (const float*)h->slot_ptr[o.w_slot];
            d.bias = (const float*)h->slot_ptr[o.b_slot];
            if (o.act == SPRK_ACT_PRELU) {
                if ((rc = need_bytes(h, o.alpha_slot, (size_t)o.N * 4, "PReLU alpha"))) return rc;
                d.alpha = (const float*)h->slot_ptr[o.alpha_slot];
            }
        }
    }
    for (int i = 0; i < p.n_taps; ++i) {
        const sprk_tap& t = p.taps[i];
        DevTap& d = dp->taps[i];
        d.buf = t.buf; d.off = t.off; d.len = t.len; d.scale = t.scale; d.bias = t.bias; d.w = nullptr;
        if (t.w_slot >= 0) {
            if ((rc = need_bytes(h, t.w_slot, (size_t)t.len * 4, "tap weights"))) return rc;
            d.w = (const float*)h->slot_ptr[t.w_slot];
        }
    }
    HIP_TRY(hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (p.din.enabled == 2) {
        const sprk_din& s = p.din;
        DevDin& d = dp->din;
        d.enabled = 1; d.T = s.T; d.hist_col = s.hist_col; d.cand_col = s.cand_col; d.row_stride = s.row_stride; d.vocab = s.vocab; d.hidden = s.hidden;
        const size_t img = s.emb_dim == 10 ? DienLayout<10, 32>::total_pad : DienLayout<16, 32>::total_pad;
        if ((rc = need_bytes(h, s.table_slot, (size_t)s.vocab * s.row_stride * 4, "DIEN table"))) return rc;
        if ((rc = need_bytes(h, s.seq_slot, img * 4, "DIEN sequence weights"))) return rc;
        d.table = (const float*)h->slot_ptr[s.table_slot];
        h->dien_run.T = s.T; h->dien_run.F = p.n_id_cols; h->dien_run.hist_col = s.hist_col; h->dien_run.cand_col = s.cand_col;
        h->dien_run.Dp = s.row_stride; h->dien_run.vocab = s.vocab; h->dien_run.NA = p.n_aux;
        h->dien_run.table = d.table;
        h->dien_run.image = (const float*)h->slot_ptr[s.seq_slot];
        if (h->tune.dien_mfma && h->tune.dyn_f16 && s.hidden == 32 && (s.emb_dim == 10 || s.emb_dim == 16)) {
            // sixteen samples per wave on the matrix pipe (k_dien_mfma.h): the host's packed image -> split-f16 MFMA A fragments with
            // static scales from max|E| and the weights' row sums; non-finite weights keep the lane-per-sample kernel
            const size_t fl = s.emb_dim == 10 ? DienFrag<10, 32>::total_pad : DienFrag<16, 32>::total_pad;
            const size_t ok_at = s.emb_dim == 10 ? DienFrag<10, 32>::S_OK : DienFrag<16, 32>::S_OK;
            DevProbe d_max_probe;
            unsigned*& d_max = d_max_probe.p;
            HIP_TRY(hipMalloc((void**)&d_max, sizeof(unsigned)));
            HIP_TRY(hipMemset(d_max, 0, sizeof(unsigned)));
            hipLaunchKernelGGL(k_v2_absmax, dim3(1024), dim3(256), 0, 0, d.table, (long long)s.vocab, s.row_stride, s.row_stride, d_max);
            HIP_TRY(hipMalloc((void**)&h->dien_frag, fl * sizeof(float)));
            if (s.emb_dim == 10) hipLaunchKernelGGL((k_dien_mfma_pack<10, 32>), dim3(1), dim3(256), 0, 0, h->dien_run.image, (const unsigned*)d_max, h->dien_frag);
            else hipLaunchKernelGGL((k_dien_mfma_pack<16, 32>), dim3(1), dim3(256), 0, 0, h->dien_run.image, (const unsigned*)d_max, h->dien_frag);
            HIP_TRY(hipGetLastError());
            float ok = 0.f;
            HIP_TRY(hipMemcpy(&ok, h->dien_frag + ok_at, sizeof(float), hipMemcpyDeviceToHost));
            if (ok != 1.f) { (void)hipFree(h->dien_frag); h->dien_frag = nullptr; }
        }
    } else if (p.din.enabled) {
        const sprk_din& s = p.din;
        DevDin& d = dp->din;
        d.enabled = 1; d.T = s.T; d.hist_col = s.hist_col; d.cand_col = s.cand_col; d.row_stride = s.row_stride; d.vocab = s.vocab; d.hidden = s.hidden; d.b2 = s.b2;
        if ((rc = need_bytes(h, s.table_slot, (size_t)s.vocab * s.row_stride * 4, "DIN table"))) return rc;
        if ((rc = need_bytes(h, s.w_slot, (size_t)s.hidden * 4 * s.row_stride * 4, "DIN att0 kernel"))) return rc;
        if ((rc = need_bytes(h, s.b_slot, (size_t)s.hidden * 4, "DIN att0 bias"))) return rc;
        if ((rc = need_bytes(h, s.alpha_slot, (size_t)s.T * s.hidden * 4, "DIN alpha"))) return rc;
        if ((rc = need_bytes(h, s.w2_slot, (size_t)s.hidden * 4, "DIN att1 kernel"))) return rc;
        d.table = (const float*)h->slot_ptr[s.table_slot];
        d.W = (const float*)h->slot_ptr[s.w_slot];
        d.bias = (const float*)h->slot_ptr[s.b_slot];
        d.alpha = (const float*)h->slot_ptr[s.alpha_slot];
        d.w2 = (const float*)h->slot_ptr[s.w2_slot];
        // samples per workgroup pass: about 256 (sample, slot) rows in LDS
        int ms = 256 / s.T;
        if (ms < 1) ms = 1;
        if (ms > 64) ms = 64;
        h->din_ms = ms;
        const int hs = s.row_stride + 4;
        h->din_lds_bytes = ((size_t)ms * s.T * hs + (size_t)ms * hs + (size_t)ms * s.T) * sizeof(float);
        if (h->din_lds_bytes > 160 * 1024) return fail(SPRK_EINVAL, "DIN stage needs %zu bytes of LDS", h->din_lds_bytes);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_din_pool), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->din_lds_bytes));
        int per_cu = (int)(160 * 1024 / h->din_lds_bytes);
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        h->din_grid_cap = h->num_cus * per_cu;
        // [r6] k_din_attn_cols / k_din_fused when the shape is one of theirs and the operands fit the split-f16 form; anything else (and
        // SPRK_DIN_HALF=0 / SPRK_DIN_COLS=0 / SPRK_DIN_LEGACY=1) stays on the generic k_din_pool -- until round 6 k_din_attn took those
        const int kc = (s.row_stride + 15) / 16, hc = s.hidden / 16;
        const size_t vc_bytes = (size_t)s.vocab * s.hidden * sizeof(float);
        if (!h->tune.din_legacy && h->tune.din_half && h->tune.din_cols && s.T <= 64 && vc_bytes < ((size_t)4 << 30) &&
            (size_t)s.vocab * s.row_stride * sizeof(float) < ((size_t)4 << 30)) {   // 32-bit element offsets
            for (size_t v = 0; v < sizeof(kDinVariants) / sizeof(kDinVariants[0]); ++v) {
                const DinVariant& dv = kDinVariants[v];
                if (dv.kc != kc || dv.hc != hc || s.T > dv.max_t) continue;
                const int KP = kc * 16;
                if (!h->din_w12) {
                    HIP_TRY(hipMalloc((void**)&h->din_w12, (size_t)s.hidden * KP * sizeof(float)));
                    HIP_TRY(hipMalloc((void**)&h->din_w4, (size_t)s.hidden * KP * sizeof(float)));
                    HIP_TRY(hipMalloc((void**)&h->din_vc, vc_bytes));
                    h->derived_bytes += vc_bytes;
                }
                hipLaunchKernelGGL(k_din_prep_w, dim3(8), dim3(256), 0, 0, d.W, s.hidden, s.row_stride, KP, 1.0f, h->din_w12, h->din_w4);
                HIP_TRY(hipGetLastError());
                float h_scale = 1.f, a_scale = 1.f;
                {
                    // power-of-two scales from max|E|, max|W12|, max|W4|: |A_b| <= max|W12| + max|W4| max|E|
                    DevProbe d_max_probe;
                    unsigned*& d_max = d_max_probe.p;
                    HIP_TRY(hipMalloc((void**)&d_max, 3 * sizeof(unsigned)));
                    HIP_TRY(hipMemset(d_max, 0, 3 * sizeof(unsigned)));
                    long long nb_ = ((long long)s.vocab * s.row_stride + 255) / 256;
                    if (nb_ > 8192) nb_ = 8192;
                    hipLaunchKernelGGL(k_v2_absmax, dim3((unsigned)nb_), dim3(256), 0, 0, d.table, (long long)s.vocab, s.row_stride, s.row_stride, d_max);
                    hipLaunchKernelGGL(k_v2_absmax, dim3(4), dim3(256), 0, 0, h->din_w12, (long long)s.hidden, KP, KP, d_max + 1);
                    hipLaunchKernelGGL(k_v2_absmax, dim3(4), dim3(256), 0, 0, h->din_w4, (long long)s.hidden, KP, KP, d_max + 2);
                    HIP_TRY(hipGetLastError());
                    unsigned bits[3];
                    HIP_TRY(hipMemcpy(bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
                    float mx[3];
                    memcpy(mx, bits, sizeof(mx));
                    if (!(mx[0] < 3.0e38f) || !(mx[1] < 3.0e38f) || !(mx[2] < 3.0e38f)) break;   // NaN / Inf weights: the generic stage
                    {
                        bool wide = false;                      // outlier rows: the ordinary rows would lose their lo halves
                        if (int rcw = wide_dynamic_range(d.table, (long long)s.vocab, s.row_stride, s.row_stride, mx[0], &wide)) return rcw;
                        if (wide) break;
                    }
                    const float bound_a = mx[1] + mx[2] * mx[0];
                    int e = 0;
                    if (mx[0] > 0.f) { (void)frexpf(mx[0], &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; h_scale = ldexpf(1.f, e); }
                    if (bound_a > 0.f) { (void)frexpf(bound_a, &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; a_scale = ldexpf(1.f, e); }
                    hipLaunchKernelGGL(k_din_prep_w, dim3(8), dim3(256), 0, 0, d.W, s.hidden, s.row_stride, KP, a_scale, h->din_w12, h->din_w4);
                    HIP_TRY(hipGetLastError());
                    if ((size_t)s.vocab * KP * sizeof(float) >= ((size_t)4 << 30)) break;
                    if (!h->din_tsplit) { HIP_TRY(hipMalloc((void**)&h->din_tsplit, (size_t)s.vocab * KP * sizeof(float) + 16)); h->derived_bytes += (size_t)s.vocab * KP * sizeof(float); }
                    long long sb = ((long long)s.vocab * KP + 255) / 256;
                    if (sb > 65536) sb = 65536;
                    hipLaunchKernelGGL(k_din_split_table, dim3((unsigned)sb), dim3(256), 0, 0, d.table, (long long)s.vocab, s.row_stride, KP,
                                       h_scale, reinterpret_cast<_Float16*>(h->din_tsplit));
                    HIP_TRY(hipGetLastError());
                }
                long long blocks = ((long long)s.vocab * s.hidden + 255) / 256;
                if (blocks > 65536) blocks = 65536;
                hipLaunchKernelGGL(k_din_prep_vc, dim3((unsigned)blocks), dim3(256), 0, 0, d.W, d.bias, d.table, s.hidden,
                                   s.row_stride, (long long)s.vocab, h->din_vc);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipDeviceSynchronize());
                h->din_attn_many = true;
                h->din_variant = (int)v;
                // k_din_attn_cols: the weights as the static MFMA operand, sixteen samples per tile (k_din_cols.h).  Needs the
                // split-f16 tables of this variant and scales that keep W4 * s4 and h * c * sP inside f16's normal range.
                {
                    if (hc == 2 && (kc == 1 || kc == 2) && s.T <= 64) {
                        // U = a_scale h_scale (the accumulators' unit); sP = h_scale^2 2^-15 puts max |h c| sP in [2^13, 2^15);
                        // W4 then carries s4 = U / sP = a_scale 2^15 / h_scale
                        const float rho = 32768.0f / h_scale;        // s4 / a_scale
                        float w4max = 0.f;
                        {
                            DevProbe d_m_probe;
                            unsigned*& d_m = d_m_probe.p;
                            HIP_TRY(hipMalloc((void**)&d_m, sizeof(unsigned)));
                            HIP_TRY(hipMemset(d_m, 0, sizeof(unsigned)));
                            hipLaunchKernelGGL(k_v2_absmax, dim3(4), dim3(256), 0, 0, h->din_w4, (long long)s.hidden, KP, KP, d_m);
                            unsigned bits = 0;
                            HIP_TRY(hipMemcpy(&bits, d_m, sizeof(bits), hipMemcpyDeviceToHost));
                            memcpy(&w4max, &bits, sizeof(w4max));    // max |W4| a_scale
                        }
                        const float w4s = w4max * rho;
                        if (w4max == 0.f || (w4s < 60000.0f && w4s >= 16.0f)) {
                            if (!h->din_frag) HIP_TRY(hipMalloc((void**)&h->din_frag, 2 * 4 * 64 * 16 + 2 * 64 * 36 * sizeof(float)));
                            float* coef = h->din_frag + 2 * 4 * 64 * 4;                  // behind the 8 KB of fragments
                            hipLaunchKernelGGL(k_din_cols_coef, dim3(1), dim3(256), 0, 0, d.alpha, d.w2, s.T, coef);
                            hipLaunchKernelGGL(k_din_cols_pack, dim3(1), dim3(256), 0, 0, h->din_w12, h->din_w4, KP, 1.0f, rho,
                                               reinterpret_cast<_Float16*>(h->din_frag));
                            HIP_TRY(hipGetLastError());
                            HIP_TRY(hipDeviceSynchronize());
                            DinColsRun& c = h->din_cols_run;
                            memset(&c, 0, sizeof(c));
                            c.T = s.T; c.F = p.n_id_cols; c.hist_col = s.hist_col; c.cand_col = s.cand_col; c.Dp = s.row_stride; c.vocab = s.vocab;
                            c.b2 = s.b2; c.acc_scale = a_scale * h_scale; c.unscale = 1.0f / (a_scale * h_scale); c.inv_h_scale = 1.0f / h_scale;
                            c.kappa = 1.0f / 32768.0f;
                            c.tsplit = h->din_tsplit; c.vc = h->din_vc; c.alpha = d.alpha; c.w2 = d.w2; c.frag = h->din_frag;
                            c.coef = coef; c.idp = p.n_id_cols;
                            const int lds_max = (2 * 64 * 36 + DC_WAVES * 16 * p.n_id_cols + DC_WAVES * 2 * 64 * 8) * 4;
                            if (lds_max > 160 * 1024) return fail(SPRK_EINVAL, "DIN attention needs %d bytes of LDS", lds_max);
                            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_attn_cols<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
                            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_attn_cols<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
                            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_attn_cols<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
                            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_attn_cols<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
                            h->din_cols = true;
                            h->din_cols_kc = kc;
                            // k_din_fused (k_din_fused.h) takes the same tables; its tail half is set up by setup_din_tail
                            // (a trip of its slot loop is four slots: for the reference's own hist_len = 5 that is 8 slots of work for 5, and
                            // k_din_attn_cols' three-slot trips measure 10.7 us against 12.5 -- so short histories stay there)
                            if (h->tune.din_fused && s.T >= h->tune.din_fused_min_t) {
                                DinFusedRun& f = h->din_fused_run;
                                memset(&f, 0, sizeof(f));
                                f.T = c.T; f.F = c.F; f.hist_col = c.hist_col; f.cand_col = c.cand_col; f.Dp = c.Dp; f.vocab = c.vocab;
                                f.b2 = c.b2; f.acc_scale = c.acc_scale; f.unscale = c.unscale; f.inv_h_scale = c.inv_h_scale; f.kappa = c.kappa;
                                f.tsplit = c.tsplit; f.vc = c.vc; f.frag = c.frag; f.coef = c.coef; f.idp = c.idp;
                                const int lds_attn = (DF_COEF_FLOATS + DF_WAVES * 16 * p.n_id_cols + DF_WAVES * 2 * 64 * 8) * 4;
                                if (lds_attn <= 160 * 1024) {
                                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<1, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attn));
                                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<1, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attn));
                                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attn));
                                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attn));
                                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<1, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attn));
                                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attn));
#ifdef SPRK_DF_XP
#define DF_XP_ATTR(X) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, false, false, false, X>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_attn));
                                    DF_XP_ATTR(1) DF_XP_ATTR(2) DF_XP_ATTR(4) DF_XP_ATTR(8) DF_XP_ATTR(16) DF_XP_ATTR(32) DF_XP_ATTR(64) DF_XP_ATTR(3) DF_XP_ATTR(56) DF_XP_ATTR(60) DF_XP_ATTR(63) DF_XP_ATTR(127) DF_XP_ATTR(65) DF_XP_ATTR(126)
#undef DF_XP_ATTR
#endif
                                    h->din_fused_attn = true;
                                }
                            }
                        }
                    }
                }
                if (!h->din_cols) h->din_variant = -1;              // (W4's range does not fit the split: the generic stage)
                break;
            }
        }
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_forward), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->tile_lds_bytes));
    {
        int per_cu = (int)(160 * 1024 / (h->tile_lds_bytes ? h->tile_lds_bytes : 1));
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        h->tile_grid_cap = h->num_cus * per_cu;
    }
    {
        if (!h->tune.force_interpreter && match_v2_chain(h)) {
            const V2Variant& vv = kV2Variants[h->v2_variant];
            h->v2_grid_cap = h->num_cus;                             // one 8-wave workgroup per CU (weights in registers: two waves per SIMD)
            // first-order weight blocks back to back
            HIP_TRY(hipMalloc((void**)&h->v2_fo_all, h->v2_fo_floats * sizeof(float)));
            for (int g = 0; g < vv.g_emb; ++g)
                HIP_TRY(hipMemcpy(h->v2_fo_all + h->v2run.fo_off[g], h->v2.w1[g], ((size_t)h->v2run.vocab[g] + 1) * sizeof(float), hipMemcpyDeviceToDevice));
            h->v2run.fo_all = h->v2_fo_all;
            {
                const int KP = vv.kpc * 16;
                size_t rows_total = 0;
                for (int g = 0; g < vv.g_emb; ++g) { h->v2run.rowbase[g] = (unsigned)rows_total; rows_total += (size_t)h->v2run.vocab[g] + 1; }
                { const int rc_ = table_alloc(h, (void**)&h->v2_folded, rows_total * (KP + 16) * sizeof(float)); if (rc_) return rc_; }
                for (int g = 0; g < vv.g_emb; ++g) {
                    const long long rows = (long long)h->v2run.vocab[g] + 1;
                    long long blocks = (rows + 3) / 4;
                    if (blocks > 65536) blocks = 65536;
                    hipLaunchKernelGGL(k_v2_fold, dim3((unsigned)blocks), dim3(256), 0, 0, h->v2.table[g], h->v2.ldp_emb,
                                       h->v2.Wp[g], h->v2.ldp_emb, h->v2.bp[g], h->v2.w1[g], h->v2.hfm, h->v2.n_hfm, h->v2.h0w,
                                       h->v2_folded + (size_t)h->v2run.rowbase[g] * (KP + 16), KP, rows);
                    HIP_TRY(hipGetLastError());
                }
                h->v2run.tab0 = h->v2_folded;
                HIP_TRY(hipDeviceSynchronize());
                if ((rc = setup_v2_joint(h))) return rc;
                if (h->v2j_variant >= 0) h->derived_bytes += rows_total * (KP + 16) * sizeof(float);
            }
            if (h->v2j_variant >= 0) {
                HIP_TRY(hipMalloc((void**)&h->v2_image, vv.lds_bytes));
                HIP_TRY(hipMemset(h->v2_image, 0, vv.lds_bytes));
                vv.pack(h->v2, h->v2_image);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipDeviceSynchronize());
            } else {
                // [r6] no joint form for this model (no small-vocabulary field, more than three large ones, SPRK_V2_JOINT=0): until round 6
                // k_deepfm_v2_chain took it; now the parsed plan goes to k_rows_chain, then to the interpreter
                (void)hipFree(h->v2_fo_all); h->v2_fo_all = nullptr;
                table_free(h, h->v2_folded); h->v2_folded = nullptr;
                h->v2_variant = -1;
                h->rows_from_v2 = h->v2_rows_ok;
            }
        }
    }
    {
        if (!h->tune.force_interpreter && h->v2_variant < 0) {
            if (h->rows_from_v2 && (rc = setup_rows_v2(h))) return rc;
            if (h->rows_variant < 0 && (rc = setup_rows_ncf(h))) return rc;
        }
    }
    const bool rows_on = h->rows_variant >= 0;
    if (!rows_on && h->v2_variant < 0 && (rc = setup_deepfm_pairs(h))) return rc;
    if (!rows_on && h->v2_variant < 0 && h->v1_variant < 0) {
        if (!h->tune.force_interpreter && (rc = setup_mlp_rows(h))) return rc;
    }
    const bool mrows_on = h->mlp_rows_nbig >= 0;
    if (!mrows_on && !rows_on && h->v2_variant < 0 && h->v1_variant < 0 && (rc = fold_first_dense(h, dp))) return rc;
    if (!mrows_on && !rows_on && h->v2_variant < 0 && (rc = setup_din_tail(h, dp))) return rc;
    if (p.din.enabled == 2 && h->dien_frag && h->tune.dien_fused && h->din_tail_variant >= 0) {
        // DIEN in one launch (k_dien_fused.h): the sequence stage on the matrix pipe AND DIN.py's 128 / 64 tail on raw split rows
        const DinTailVariant& tv = kDinTailVariants[h->din_tail_variant];
        if (tv.n0c == 8 && tv.n1c == 4 && tv.kpc == 1 && h->din_tail_run.e_unscale != 0.f) {
            const bool d10 = p.din.emb_dim == 10;
            h->dien_fused_lds = ((d10 ? DienFrag<10, 32>::total_pad : DienFrag<16, 32>::total_pad) + DinTailLds<8, 4, 1>::total_pad) * sizeof(float);
            HIP_TRY(hipFuncSetAttribute(d10 ? reinterpret_cast<const void*>(&k_dien_fused<10, 32, 8, 4>) : reinterpret_cast<const void*>(&k_dien_fused<16, 32, 8, 4>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->dien_fused_lds));
            h->dien_fused = true;
        }
    }
    HIP_TRY(hipMalloc((void**)&h->dev_plan, sizeof(DevPlan)));
    HIP_TRY(hipMemcpy(h->dev_plan, dp, sizeof(DevPlan), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&h->dev_err, sizeof(int)));
    HIP_TRY(hipMemset(h->dev_err, 0, sizeof(int)));
    {
        // helper streams for sprk_forward_many's fan-out (sprk_set_many_streams; SPRK_MANY_STREAMS presets it)
        {
            HIP_TRY(hipEventCreateWithFlags(&h->many_fork, hipEventDisableTiming));
            for (int i = 0; i < 4; ++i) {
                HIP_TRY(hipStreamCreateWithFlags(&h->many_stream[i], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&h->many_join[i], hipEventDisableTiming));
            }
            h->many_streams = h->tune.many_streams;              // 0 = strict stream order (default), 2..4 = fan out
        }
    }
    h->finalized = true;
    return SPRK_OK;
}

size_t sprk_workspace_bytes(sprk_handle h, int32_t B) {
    if (!h || B <= 0 || !h->plan.din.enabled) return 0;
    return (size_t)B * h->plan.n_aux * sizeof(float);
}


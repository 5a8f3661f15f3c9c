// host_setup_din_tail.h -- DIN / DIEN tail: k_din_tail dispatch table and set-up -- closes the host helpers' anonymous namespace.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- dispatch table for k_din_tail<N0C, N1C, KPC, WAVES> ----
constexpr int DT_WAVES = 8;
// UNF: emb_dim <= 16 fits 128 VGPRs -> sixteen waves per CU, ONE task per wave at B = 65 536; emb_dim <= 32 holds four operand pairs
// and would spill 81 dwords at that cap: eight waves
constexpr int dt_waves_unf(int kpc) { return kpc >= 2 ? 8 : 16; }
typedef void (*DinTailLaunchFn)(const DinTailRun&, const int*, const float*, const float*, float*, int, int*, const float*, int, hipStream_t);
typedef void (*DinTailLaunchManyFn)(const DinTailRun&, const DinTailMany&, int, int*, const float*, int, hipStream_t);
typedef void (*DinTailPackFn)(const float*, int, int, int, int, int, const float*, const float*, const float*, int, const float*,
                              const float*, const float*, int, const float*, float*, const float*, const float*);
template <int N0C, int N1C, int KPC>
void din_tail_launch(const DinTailRun& a, const int* ids, const float* dense, const float* aux, float* out, int B, int* err,
                     const float* image, int grid, hipStream_t st) {
    const size_t lds = DinTailLds<N0C, N1C, KPC>::bytes;
    static const DinTailMany none{};
    if (a.e_unscale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, dt_waves_unf(KPC), true, false, true>), dim3(grid), dim3(dt_waves_unf(KPC) * 64), lds, st,
                           a, ids, dense, aux, out, B, err, image, none);
    else if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, true, false>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, ids, dense, aux, out, B, err, image, none);
    else
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, false, false>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, ids, dense, aux, out, B, err, image, none);
}
template <int N0C, int N1C, int KPC>
void din_tail_launch_many(const DinTailRun& a, const DinTailMany& m, int B, int* err, const float* image, int grid, hipStream_t st) {
    const size_t lds = DinTailLds<N0C, N1C, KPC>::bytes;
    if (a.e_unscale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, dt_waves_unf(KPC), true, true, true>), dim3(grid), dim3(dt_waves_unf(KPC) * 64), lds, st,
                           a, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, err, image, m);
    else if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, true, true>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, err, image, m);
    else
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, false, true>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, err, image, m);
}
template <int N0C, int N1C, int KPC>
void din_tail_pack(const float* W0, int ldw0, int p_off, int Dp, int n_off, int n_num, const float* b0, const float* a0,
                   const float* W1, int ldw1, const float* b1, const float* a1, const float* hw, int n_hw, const float* w1frag,
                   float* img, const float* w0pfrag, const float* w0efrag) {
    hipLaunchKernelGGL((k_din_tail_pack<N0C, N1C, KPC>), dim3(1), dim3(256), 0, 0, W0, ldw0, p_off, Dp, n_off, n_num, b0, a0, W1, ldw1,
                       b1, a1, hw, n_hw, w1frag, img, w0pfrag, w0efrag);
}
struct DinTailVariant {
    int n0c, n1c, kpc;
    const void* fn[6];                // [DYN][MB] instantiations, then UNF [MB]
    size_t lds_bytes;
    DinTailLaunchFn launch;
    DinTailLaunchManyFn launch_many;
    DinTailPackFn pack;
};
#define DIN_TAIL_VARIANT(N0C, N1C, KPC) {N0C, N1C, KPC, {reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, false, false>), \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, false, true>),               \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, true, false>),               \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, true, true>),                \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, dt_waves_unf(KPC), true, false, true>),     \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, dt_waves_unf(KPC), true, true, true>)},     \
                                         DinTailLds<N0C, N1C, KPC>::bytes, &din_tail_launch<N0C, N1C, KPC>, &din_tail_launch_many<N0C, N1C, KPC>, \
                                         &din_tail_pack<N0C, N1C, KPC>}
const DinTailVariant kDinTailVariants[] = {
    DIN_TAIL_VARIANT(8, 4, 2),        // DIN.py:161-167 widths 128 / 64, emb_dim 17..32 (BASELINE config 3)
    DIN_TAIL_VARIANT(8, 4, 1),        // ... emb_dim <= 16 (the reference's own emb_dim 10)
    DIN_TAIL_VARIANT(4, 2, 2), DIN_TAIL_VARIANT(4, 2, 1),     // half-width tails (64 / 32)
};

// Recognise the DIN tail the first-Dense fold left behind (every embedding column folded, fc0 reading only the
// pooled history + numerics, two PReLU Dense layers, one weighted tap) and set up k_din_tail for it.
int setup_din_tail(sprk_engine* h, DevPlan* dp) {
    if (!h->tune.din_tail) return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (!p.din.enabled || (p.model_kind != SPRK_MODEL_DIN && p.model_kind != SPRK_MODEL_DIEN) || dp->n_ops != 2 || dp->n_taps != 1) return SPRK_OK;
    if (dp->n_acc < 1 || dp->n_acc > DT_MAX_COLS) return SPRK_OK;
    const DevOp &o0 = dp->ops[0], &o1 = dp->ops[1];
    if (o0.kind != SPRK_OP_DENSE || o1.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_PRELU || o1.act != SPRK_ACT_PRELU) return SPRK_OK;
    if (!o0.acc_init || o0.src_buf != 0 || o0.dst_off != 0 || o1.src_buf != o0.dst_buf || o1.src_off != 0 || o1.K != o0.N ||
        o1.dst_off != 0) return SPRK_OK;
    const DevTap& tp = dp->taps[0];
    if (tp.buf != o1.dst_buf || tp.off != 0 || tp.len > o1.N || !tp.w || tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
    int aux_dst = -1, num_dst = -1, n_num = 0, Dp = 0;
    const int n_plain = dp->n_segs - dp->n_acc;
    for (int i = 0; i < n_plain; ++i) {
        const DevSeg& sg = dp->segs[i];
        if (sg.kind == SPRK_SEG_AUX && aux_dst < 0 && sg.field == 0) { aux_dst = sg.dst; Dp = sg.count; }
        else if (sg.kind == SPRK_SEG_DENSE && num_dst < 0 && sg.field == 0) { num_dst = sg.dst; n_num = sg.count; }
        else if (sg.kind != SPRK_SEG_ZERO) return SPRK_OK;     // an unfolded gather remains: leave it to the interpreter
    }
    if (aux_dst < 0 || num_dst < 0 || Dp != p.n_aux || n_num < 1 || n_num > 8) return SPRK_OK;
    const int p_off = aux_dst - o0.src_off, n_off = num_dst - o0.src_off;
    if (p_off < 0 || p_off + Dp > o0.K || n_off < 0 || n_off + n_num > o0.K) return SPRK_OK;
    const int n0c = o0.N / 16, n1c = o1.N / 16, kpc = (Dp + 15) / 16;
    int variant = -1;
    for (size_t v = 0; v < sizeof(kDinTailVariants) / sizeof(kDinTailVariants[0]); ++v)
        if (kDinTailVariants[v].n0c == n0c && kDinTailVariants[v].n1c == n1c && kDinTailVariants[v].kpc == kpc) variant = (int)v;
    if (variant < 0) return SPRK_OK;
    const DinTailVariant& tv = kDinTailVariants[variant];
    DinTailRun& r = h->din_tail_run;
    memset(&r, 0, sizeof(r));
    r.F = p.n_id_cols; r.ND = p.n_dense; r.NA = p.n_aux; r.n_cols = dp->n_acc; r.n_num = n_num; r.head_bias = dp->head_bias;
    for (int g = 0; g < dp->n_acc; ++g) {
        const DevSeg& sg = dp->segs[n_plain + g];
        r.col[g] = h->idc[sg.field]; r.vocab[g] = sg.vocab; r.Ftab[g] = sg.table;
    }
    HIP_TRY(hipMalloc((void**)&h->din_tail_image, tv.lds_bytes));
    // DYN: fc1's weights split into f16 hi / lo fragments with a static power-of-two scale
    float* w1frag = nullptr;
    {
        float w_scale = 0.f;
        const int rc2 = make_dyn_fragments(h, o1.W, o1.ldw, o1.N, o1.K, &w1frag, &w_scale);
        if (rc2) return rc2;
        if (w1frag) r.inv_w1_scale = 1.0f / w_scale;
    }
    // [r3] DYN: fc0's pooled-history columns W0^T[:, p_off .. p_off + Dp) as split-f16 fragments (K block padded with zeros)
    float* w0pfrag = nullptr;
    if (w1frag && h->tune.tail_pooled_f16) {
        float w_scale = 0.f;
        const int rc3 = make_dyn_fragments(h, o0.W + p_off, o0.ldw, o0.N, 32 * ((kpc + 1) / 2), &w0pfrag, &w_scale, Dp);
        if (rc3) return rc3;
        if (w0pfrag) r.inv_w0p_scale = 1.0f / w_scale;
    }
    // [r3] UNF: raw split rows of the embedding columns + fc0's columns for them as fragments (k_din_tail.h, DinTailRun::Etab)
    float* w0efrag = nullptr;
    if (w1frag && w0pfrag && h->tune.tail_unf && p.ops[0].kind == SPRK_OP_DENSE && p.ops[0].w_slot >= 0) {
        const sprk_op& q0 = p.ops[0];                              // the ORIGINAL first Dense (o0.W has the folded columns zeroed)
        const float* W0full = (const float*)h->slot_ptr[q0.w_slot];
        const sprk_seg* raw[DT_MAX_COLS] = {nullptr, nullptr, nullptr, nullptr};
        const int epb = kpc >= 2 ? 32 : 16, nblk = kpc >= 2 ? 4 : 2;    // DinTailLds::EPB / NBLK
        bool ok = kpc <= 2;
        for (int g = 0; g < dp->n_acc && ok; ++g) {
            const DevSeg& sg = dp->segs[n_plain + g];
            for (int i = 0; i < p.n_segs; ++i)
                if (p.segs[i].kind == SPRK_SEG_ROWS && p.segs[i].field == h->idc[sg.field] && p.segs[i].vocab == sg.vocab &&   // (DevSeg::field is the compact index)
                    p.segs[i].dst >= q0.src_off && p.segs[i].dst + 4 * p.segs[i].count <= q0.src_off + q0.K) { raw[g] = &p.segs[i]; break; }
            ok = raw[g] && 4 * raw[g]->count <= epb && raw[g]->row_stride <= epb;
            for (int g2 = 0; g2 < g && ok; ++g2) ok = raw[g2] != raw[g];
        }
        // one static scale for all columns' rows; an outlier row keeps the folded tables
        float mx = 0.f;
        if (ok) {
            DevProbe d_max_probe;
            unsigned*& d_max = d_max_probe.p;
            HIP_TRY(hipMalloc((void**)&d_max, sizeof(unsigned)));
            HIP_TRY(hipMemset(d_max, 0, sizeof(unsigned)));
            for (int g = 0; g < dp->n_acc; ++g)
                hipLaunchKernelGGL(k_v2_absmax, dim3(256), dim3(256), 0, 0, (const float*)h->slot_ptr[raw[g]->slot], (long long)raw[g]->vocab,
                                   raw[g]->row_stride, 4 * raw[g]->count, d_max);
            HIP_TRY(hipGetLastError());
            unsigned bits = 0;
            HIP_TRY(hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
            memcpy(&mx, &bits, sizeof(mx));
            ok = mx < 3.0e38f;
            for (int g = 0; g < dp->n_acc && ok; ++g) {
                bool wide = false;
                if (int rcw = wide_dynamic_range((const float*)h->slot_ptr[raw[g]->slot], (long long)raw[g]->vocab, raw[g]->row_stride,
                                                 4 * raw[g]->count, mx, &wide)) return rcw;
                ok = !wide;
            }
        }
        if (ok) {
            auto pow2_scale = [](float m, int top) { int e = 0; if (m > 0.f) { (void)frexpf(m, &e); e = top - e; } e = e > 60 ? 60 : (e < -60 ? -60 : e); return ldexpf(1.f, e); };
            const float e_scale = pow2_scale(mx, 15);
            std::vector<float> W0h;
            int rcp = pull(W0h, W0full, (size_t)q0.N * q0.ldw);
            if (rcp) return rcp;
            float amax = 0.f;
            for (int g = 0; g < dp->n_acc; ++g)
                for (int n = 0; n < q0.N; ++n)
                    for (int d = 0; d < 4 * raw[g]->count; ++d) amax = fmaxf(amax, fabsf(W0h[(size_t)n * q0.ldw + raw[g]->dst - q0.src_off + d]));
            ok = amax < 3.0e38f;
            if (ok) {
                const float w_scale = pow2_scale(amax, 15);
                std::vector<float> fr((size_t)n0c * nblk * 512, 0.f);
                _Float16* fh = reinterpret_cast<_Float16*>(fr.data());
                for (int nb = 0; nb < n0c; ++nb)
                    for (int pb = 0; pb < nblk; ++pb)
                        for (int ln = 0; ln < 64; ++ln)
                            for (int e = 0; e < 8; ++e) {
                                const int n = nb * 16 + (ln & 15), k = 8 * (ln >> 4) + e;
                                const int g = epb == 16 ? 2 * pb + (k >> 4) : pb, d = epb == 16 ? (k & 15) : k;
                                float x = 0.f;
                                if (g < dp->n_acc && d < 4 * raw[g]->count) x = W0h[(size_t)n * q0.ldw + raw[g]->dst - q0.src_off + d] * w_scale;
                                const _Float16 hi = (_Float16)x;
                                const size_t base = (size_t)((nb * nblk + pb) * 2) * 512;
                                fh[base + ln * 8 + e] = hi;
                                fh[base + 512 + ln * 8 + e] = (_Float16)(x - (float)hi);
                            }
                HIP_TRY(hipMalloc((void**)&w0efrag, fr.size() * sizeof(float)));
                h->fold_bufs.push_back(w0efrag);
                HIP_TRY(hipMemcpy(w0efrag, fr.data(), fr.size() * sizeof(float), hipMemcpyHostToDevice));
                for (int g = 0; g < DT_MAX_COLS; ++g) r.Etab[g] = nullptr;
                for (int g = 0; g < dp->n_acc; ++g) {
                    const long long rows = (long long)raw[g]->vocab;
                    float* et = nullptr;
                    HIP_TRY(hipMalloc((void**)&et, (size_t)(rows + 1) * 4 * epb));
                    HIP_TRY(hipMemset(et, 0, (size_t)(rows + 1) * 4 * epb));
                    h->fold_bufs.push_back(et);
                    h->derived_bytes += (size_t)(rows + 1) * 4 * epb;
                    long long nbk = (rows * epb + 255) / 256;
                    if (nbk > 65536) nbk = 65536;
                    if (nbk > 0)
                        hipLaunchKernelGGL(k_rows_unf_split, dim3((unsigned)nbk), dim3(256), 0, 0, (const float*)h->slot_ptr[raw[g]->slot], raw[g]->row_stride,
                                           rows, e_scale, reinterpret_cast<_Float16*>(et), epb);
                    HIP_TRY(hipGetLastError());
                    r.Etab[g] = reinterpret_cast<const _Float16*>(et);
                }
                for (int g = dp->n_acc; g < DT_MAX_COLS; ++g) r.Etab[g] = r.Etab[0];
                HIP_TRY(hipGetLastError());
                r.e_unscale = 1.f / (e_scale * w_scale);
            }
        }
    }
    tv.pack(o0.W, o0.ldw, p_off, Dp, n_off, n_num, o0.bias, o0.alpha, o1.W, o1.ldw, o1.bias, o1.alpha, tp.w, tp.len, w1frag, h->din_tail_image, w0pfrag,
            w0efrag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < 6; ++i) HIP_TRY(hipFuncSetAttribute(tv.fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)tv.lds_bytes));
    h->din_tail_variant = variant;
    // k_din_fused: the same tail as the epilogue of the attention kernel (k_din_fused.h).  Needs the cols formulation of the attention
    // (set up before this function), DIN.py's widths 128 / 64, fc1 AND fc0's pooled columns as split-f16 fragments, folded rows for
    // every embedding column, and LDS for both halves.
    if (h->din_fused_attn && p.model_kind == SPRK_MODEL_DIN && p.din.enabled == 1 && n0c == DinFusedImg::N0C && n1c == DinFusedImg::N1C &&
        w1frag && w0pfrag && Dp <= 16 * h->din_cols_kc && tp.len <= DinFusedImg::N1) {
        const int kc = h->din_cols_kc;
        const int lds_full = (DF_COEF_FLOATS + DinFusedImg::dma_floats + DF_WAVES * 16 * p.n_id_cols + DF_WAVES * 64 * 4 * kc) * 4;
        bool rows_ok = true;
        for (int g = 0; g < dp->n_acc; ++g) rows_ok = rows_ok && r.Ftab[g] != nullptr;
        if (rows_ok && lds_full <= 160 * 1024) {
            // UNF (emb_dim 17 .. 32, k_din_tail's raw split rows exist): the (up to) two columns with the LARGEST vocabularies -- DIN.py's
            // userId and candidate movieId, whose folded rows (512 bytes per id) miss every cache -- go to the matrix pipe as raw rows
            // (128 bytes per id); the genre columns' folded tables are a few KB and stay folded.
            int n_unf = 0, unf_g[2] = {0, 0};
            // [r5] emb_dim <= 16 (DIN.py as written): k_din_tail's raw 64-byte rows for ALL the embedding columns, as its two column-pair blocks
            const bool unf_pairs = kc == 1 && w0efrag && r.e_unscale != 0.f && h->tune.din_fused_unf;
            if (unf_pairs) { n_unf = 2; unf_g[0] = 0; unf_g[1] = 1; }
            if (kc == 2 && w0efrag && r.e_unscale != 0.f && h->tune.din_fused_unf) {
                for (int pass = 0; pass < 2; ++pass) {
                    int best = -1;
                    for (int g = 0; g < dp->n_acc; ++g)
                        if (r.vocab[g] > 64 && (long long)(r.vocab[g] + 1) * 128 < (1LL << 32) && !(n_unf == 1 && unf_g[0] == g) && (best < 0 || r.vocab[g] > r.vocab[best])) best = g;
                    if (best >= 0) unf_g[n_unf++] = best;
                }
            }
            HIP_TRY(hipMalloc((void**)&h->din_fused_image, DinFusedImg::dma_floats * sizeof(float)));
            hipLaunchKernelGGL(k_din_fused_pack, dim3(1), dim3(256), 0, 0, o0.W, o0.ldw, p_off, Dp, n_off, n_num, 1.0f / r.inv_w0p_scale, 4 * kc,
                               o0.bias, o0.alpha, w1frag, o1.bias, o1.alpha, tp.w, tp.len, h->din_fused_image, (const float*)w0efrag, n_unf, unf_g[0], unf_g[1], kc == 2 ? 4 : 2);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            DinFusedRun& f = h->din_fused_run;
            f.ND = r.ND; f.n_num = r.n_num;
            // the folded columns (in the tail's order) and, apart from them, the raw-row columns
            f.n_cols = 0;
            for (int g = 0; g < r.n_cols; ++g) {
                if (unf_pairs || (n_unf > 0 && unf_g[0] == g) || (n_unf > 1 && unf_g[1] == g)) continue;
                f.col[f.n_cols] = r.col[g]; f.tvocab[f.n_cols] = r.vocab[g]; f.Ftab[f.n_cols] = r.Ftab[g]; ++f.n_cols;
            }
            for (int g = f.n_cols; g < DT_MAX_COLS; ++g) { f.col[g] = r.col[0]; f.tvocab[g] = r.vocab[0]; f.Ftab[g] = r.Ftab[0]; }
            f.head_bias = r.head_bias; f.inv_w1_scale = r.inv_w1_scale; f.inv_w0p_scale = r.inv_w0p_scale;
            f.b0_slot = n_num < 8 ? n_num : -1;
            f.n_unf = n_unf; f.e_unscale = r.e_unscale;
            for (int u = 0; u < 4; ++u) {
                if (unf_pairs) {                                   // slot u IS column u of the tail's list
                    f.ucol[u] = u < r.n_cols ? r.col[u] : r.col[0]; f.uvocab[u] = u < r.n_cols ? r.vocab[u] : r.vocab[0];
                    f.Etab[u] = u < r.n_cols ? r.Etab[u] : r.Etab[0];
                } else {
                    f.ucol[u] = u < n_unf ? r.col[unf_g[u]] : r.col[0]; f.uvocab[u] = u < n_unf ? r.vocab[unf_g[u]] : 0;
                    f.Etab[u] = u < n_unf ? r.Etab[unf_g[u]] : nullptr;
                }
            }
            f.n_ucols = unf_pairs ? r.n_cols : n_unf;
            f.image = h->din_fused_image;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_full));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_full));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_full));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_full));
#ifdef SPRK_DF_XP
#define DF_XPT_ATTR(X) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, false, true, false, X>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_full));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_din_fused<2, true, true, false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_full));
            DF_XPT_ATTR(128) DF_XPT_ATTR(256) DF_XPT_ATTR(512) DF_XPT_ATTR(896) DF_XPT_ATTR(1024)
#undef DF_XPT_ATTR
#endif
            h->din_fused = true;
        }
    }
    return SPRK_OK;
}

int need_bytes(const sprk_engine* h, int slot, size_t bytes, const char* what) {
    if (!h->slot_ptr[slot]) return fail(SPRK_ESTATE, "%s: slot %d was never uploaded", what, slot);
    if (h->slot_bytes[slot] < bytes) return fail(SPRK_EINVAL, "%s: slot %d holds %zu bytes, needs %zu", what, slot, h->slot_bytes[slot], bytes);
    return SPRK_OK;
}

}  // namespace


// host_setup_din_tail.h -- DIN / DIEN tail: k_din_tail dispatch table and set-up -- closes the host helpers' anonymous namespace.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- dispatch table for k_din_tail<N0C, N1C, KPC, WAVES> ----
constexpr int DT_WAVES = 8;
typedef void (*DinTailLaunchFn)(const DinTailRun&, const int*, const float*, const float*, float*, int, int*, const float*, int, hipStream_t);
typedef void (*DinTailLaunchManyFn)(const DinTailRun&, const DinTailMany&, int, int*, const float*, int, hipStream_t);
typedef void (*DinTailPackFn)(const float*, int, int, int, int, int, const float*, const float*, const float*, int, const float*,
                              const float*, const float*, int, const float*, float*);
template <int N0C, int N1C, int KPC>
void din_tail_launch(const DinTailRun& a, const int* ids, const float* dense, const float* aux, float* out, int B, int* err,
                     const float* image, int grid, hipStream_t st) {
    const size_t lds = DinTailLds<N0C, N1C, KPC>::bytes;
    static const DinTailMany none{};
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, true, false>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, ids, dense, aux, out, B, err, image, none);
    else
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, false, false>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, ids, dense, aux, out, B, err, image, none);
}
template <int N0C, int N1C, int KPC>
void din_tail_launch_many(const DinTailRun& a, const DinTailMany& m, int B, int* err, const float* image, int grid, hipStream_t st) {
    const size_t lds = DinTailLds<N0C, N1C, KPC>::bytes;
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, true, true>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, err, image, m);
    else
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, false, true>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, err, image, m);
}
template <int N0C, int N1C, int KPC>
void din_tail_pack(const float* W0, int ldw0, int p_off, int Dp, int n_off, int n_num, const float* b0, const float* a0,
                   const float* W1, int ldw1, const float* b1, const float* a1, const float* hw, int n_hw, const float* w1frag,
                   float* img) {
    hipLaunchKernelGGL((k_din_tail_pack<N0C, N1C, KPC>), dim3(1), dim3(256), 0, 0, W0, ldw0, p_off, Dp, n_off, n_num, b0, a0, W1, ldw1,
                       b1, a1, hw, n_hw, w1frag, img);
}
struct DinTailVariant {
    int n0c, n1c, kpc;
    const void* fn[4];                // [DYN][MB] instantiations
    size_t lds_bytes;
    DinTailLaunchFn launch;
    DinTailLaunchManyFn launch_many;
    DinTailPackFn pack;
};
#define DIN_TAIL_VARIANT(N0C, N1C, KPC) {N0C, N1C, KPC, {reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, false, false>), \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, false, true>),               \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, true, false>),               \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, true, true>)},               \
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
    tv.pack(o0.W, o0.ldw, p_off, Dp, n_off, n_num, o0.bias, o0.alpha, o1.W, o1.ldw, o1.bias, o1.alpha, tp.w, tp.len, w1frag, h->din_tail_image);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) HIP_TRY(hipFuncSetAttribute(tv.fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)tv.lds_bytes));
    h->din_tail_variant = variant;
    return SPRK_OK;
}

int need_bytes(const sprk_engine* h, int slot, size_t bytes, const char* what) {
    if (!h->slot_ptr[slot]) return fail(SPRK_ESTATE, "%s: slot %d was never uploaded", what, slot);
    if (h->slot_bytes[slot] < bytes) return fail(SPRK_EINVAL, "%s: slot %d holds %zu bytes, needs %zu", what, slot, h->slot_bytes[slot], bytes);
    return SPRK_OK;
}

}  // namespace


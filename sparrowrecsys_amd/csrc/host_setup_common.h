// host_setup_common.h -- the attention stage's shape table, the interpreter's first-Dense fold, the dynamic-range guard, split-f16 fragment packing.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---- the attention shapes k_din_attn_cols / k_din_fused take ([r6] until round 6 the dispatch table of k_din_attn: k_din_attn.h) ----
struct DinVariant { int kc, hc, max_t; };        // emb_dim in (16 (kc - 1), 16 kc], attention hidden 16 hc, history slots <= max_t
const DinVariant kDinVariants[] = {{2, 2, 64}, {1, 2, 64}};

// First-Dense fold for plans the tile interpreter runs.  A Dense layer is linear in its input, so the share of
// an embedding column is a table of its own: F_g[id] = W_g^T E_g[id] (N floats per id).  When a ROWS segment feeds
// nothing but the plan's first Dense op, the fold replaces "gather E_g[id] into the input slice, multiply by W_g
// on the matrix pipe" by "gather F_g[id] and add it to the layer's accumulator": the layer's K shrinks to the
// columns that really are per-sample data (numerics, the DIN pooled vector, crossed columns), at the price of
// N instead of D floats per gathered row.  DIN tail (DIN.py:161-166): K 168 -> 40; EmbeddingMLP: 108 -> 8.
// Same fp32 arithmetic, other association.  SPRK_TILE_FOLD=0 switches it off (A/B, tests).
int fold_first_dense(sprk_engine* h, DevPlan* dp) {
    if (!h->tune.tile_fold) return SPRK_OK;
    if (dp->n_ops < 1) return SPRK_OK;
    DevOp& op = dp->ops[0];
    if (op.kind != SPRK_OP_DENSE || op.src_buf != 0 || op.dst_buf == 0 || op.N > 512) return SPRK_OK;
    const int lo0 = op.src_off, hi0 = op.src_off + op.K;
    auto used_elsewhere = [&](int a, int b) {                 // is the GATHERED content of buffer 0's [a,b) read by anything but ops[0]?
        std::vector<char> live(b - a, 1);                     // columns still holding gathered data (later ops may overwrite buffer 0)
        auto reads = [&](int s0, int s1) {
            for (int c = (s0 > a ? s0 : a); c < (s1 < b ? s1 : b); ++c) if (live[c - a]) return true;
            return false;
        };
        auto writes = [&](int s0, int s1) { for (int c = (s0 > a ? s0 : a); c < (s1 < b ? s1 : b); ++c) live[c - a] = 0; };
        for (int i = 1; i < dp->n_ops; ++i) {
            const DevOp& o = dp->ops[i];
            if (o.src_buf == 0) {
                if (o.kind == SPRK_OP_PAIR_DOT) {
                    for (int p = 0; p < dp->n_pairs; ++p)
                        if (reads(dp->pair_a[p], dp->pair_a[p] + o.K) || reads(dp->pair_b[p], dp->pair_b[p] + o.K)) return true;
                } else if (o.kind == SPRK_OP_FM_SUMSQ) {
                    if (reads(o.src_off, o.src_off + (o.groups - 1) * o.group_stride + o.K)) return true;
                } else if (reads(o.src_off, o.src_off + o.K)) {
                    return true;
                }
            }
            if (o.dst_buf == 0) {
                const int w = o.kind == SPRK_OP_DENSE ? o.N : o.kind == SPRK_OP_PAIR_DOT ? dp->n_pairs : o.K;
                writes(o.dst_off, o.dst_off + w);
            }
        }
        for (int t = 0; t < dp->n_taps; ++t)
            if (dp->taps[t].buf == 0 && reads(dp->taps[t].off, dp->taps[t].off + dp->taps[t].len)) return true;
        return false;
    };
    std::vector<int> fold;
    size_t bytes = 0;
    for (int i = 0; i < dp->n_segs; ++i) {
        const DevSeg& sg = dp->segs[i];
        if (sg.kind != SPRK_SEG_ROWS) continue;
        const int a = sg.dst, b = sg.dst + 4 * sg.count;
        if (a < lo0 || b > hi0 || used_elsewhere(a, b)) continue;
        const size_t need = (size_t)sg.vocab * op.N * sizeof(float);
        if (need > ((size_t)2 << 30) || bytes + need > ((size_t)8 << 30)) continue;   // keep huge tables as plain row gathers
        if (fold.size() == 8) break;                          // the gather keeps at most 8 folded columns in flight per piece
        bytes += need;
        fold.push_back(i);
    }
    if (fold.empty()) return SPRK_OK;
    // new K range: hull of the columns that stay (everything in [lo0,hi0) not covered by a folded segment)
    std::vector<char> keep(hi0 - lo0, 1);
    for (int i : fold)
        for (int c = dp->segs[i].dst; c < dp->segs[i].dst + 4 * dp->segs[i].count; ++c) keep[c - lo0] = 0;
    int lo = hi0, hi = lo0;
    for (int c = lo0; c < hi0; ++c)
        if (keep[c - lo0]) { if (c < lo) lo = c; if (c + 1 > hi) hi = c + 1; }
    if (lo >= hi) { lo = lo0; hi = lo0; }
    lo &= ~3;
    hi = (hi + 3) & ~3;
    if (hi > hi0) hi = hi0;
    // W^T copy with the folded columns inside the hull zeroed; F tables
    const size_t wbytes = (size_t)op.N * op.ldw * sizeof(float);
    float* wcopy = nullptr;
    HIP_TRY(hipMalloc((void**)&wcopy, wbytes + 16));
    h->fold_bufs.push_back(wcopy);
    HIP_TRY(hipMemcpy(wcopy, op.W, wbytes, hipMemcpyDeviceToDevice));
    bool first = true;
    for (int i : fold) {
        DevSeg& sg = dp->segs[i];
        float* F = nullptr;
        HIP_TRY(hipMalloc((void**)&F, (size_t)sg.vocab * op.N * sizeof(float) + 16));
        h->derived_bytes += (size_t)sg.vocab * op.N * sizeof(float);
        h->fold_bufs.push_back(F);
        long long blocks = ((long long)sg.vocab * op.N + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_fold_dense_rows, dim3((unsigned)blocks), dim3(256), 0, 0, sg.table, (long long)sg.vocab, sg.row_stride,
                           4 * sg.count, op.W, op.ldw, sg.dst - lo0, op.N, F);
        HIP_TRY(hipGetLastError());
        const int c0 = sg.dst - lo0, c1 = c0 + 4 * sg.count;
        hipLaunchKernelGGL(k_zero_columns, dim3(16), dim3(256), 0, 0, wcopy, op.N, op.ldw, c0, c1);
        HIP_TRY(hipGetLastError());
        sg.kind = SEG_ROWS_ACC; sg.table = F; sg.row_stride = op.N; sg.count = op.N / 4; sg.dst = op.dst_off;
        sg.buf = op.dst_buf; sg.field2 = first ? 0 : 1;
        first = false;
    }
    HIP_TRY(hipDeviceSynchronize());
    // folded columns go to the end of the segment list (the kernel handles them as one group)
    {
        std::vector<DevSeg> plain, acc;
        for (int i = 0; i < dp->n_segs; ++i) (dp->segs[i].kind == SEG_ROWS_ACC ? acc : plain).push_back(dp->segs[i]);
        int k = 0;
        for (const DevSeg& g : plain) dp->segs[k++] = g;
        for (const DevSeg& g : acc) dp->segs[k++] = g;
        dp->n_acc = (int)acc.size();
        h->n_acc_folded = dp->n_acc;
    }
    op.W = wcopy + (lo - lo0);
    op.src_off = lo;
    op.K = hi - lo;
    op.acc_init = 1;
    return SPRK_OK;
}

// Dynamic-range guard for a STATIC split-f16 scale (one power of two per table from max |x|): true when more than 1 in
// 1024 of the non-zero entries lie over 2^20 below the maximum -- their lo halves would be f16 subnormals and the entries
// would carry fewer than ~20 significand bits (an outlier row next to ordinary ones).  The caller then keeps the f32 MFMA
// variant of the same kernel.  SPRK_HALF_RANGE_GUARD=0 switches the check off (for the test that shows why it is there).
int wide_dynamic_range(const float* rows, long long nrows, int row_floats, int ncols, float mx, bool* wide) {
    *wide = false;
    const bool guard_on = g_finalize_tune ? g_finalize_tune->half_range_guard : SprkTuning::from_env().half_range_guard;
    if (!guard_on || !(mx > 0.f) || nrows <= 0) return SPRK_OK;
    unsigned long long* d_cnt = nullptr;
    HIP_TRY(hipMalloc((void**)&d_cnt, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d_cnt, 0, 2 * sizeof(unsigned long long)));
    long long blocks = (nrows * ncols + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_v2_count_small, dim3((unsigned)blocks), dim3(256), 0, 0, rows, nrows, row_floats, ncols, ldexpf(mx, -20), d_cnt);
    HIP_TRY(hipGetLastError());
    unsigned long long cnt[2] = {0, 0};
    HIP_TRY(hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
    (void)hipFree(d_cnt);
    *wide = cnt[0] * 1024ull > cnt[1];
    return SPRK_OK;
}

// A Dense layer's W^T [N][ld] (K columns) as split-f16 A fragments for the per-sample dynamic-scale path (dyn_split.h):
// static power-of-two scale putting max |W| in [2^14, 2^15).  *frag stays NULL when switched off (SPRK_DYN_F16=0), when
// the shape does not tile (N % 16, K % 32) or the weights are not finite.
// kvalid >= 0: only the first kvalid columns of W^T's rows belong to the matrix (the K block is padded with zeros up to K)
int make_dyn_fragments(sprk_engine* h, const float* W, int ld, int N, int K, float** frag, float* w_scale_out, int kvalid = -1) {
    *frag = nullptr;
    if (kvalid < 0) kvalid = K;
    if (!h->tune.dyn_f16 || (N & 15) || (K & 31) || kvalid < 1 || kvalid > K) return SPRK_OK;
    DevProbe d_max_probe;
    unsigned*& d_max = d_max_probe.p;
    HIP_TRY(hipMalloc((void**)&d_max, sizeof(unsigned)));
    HIP_TRY(hipMemset(d_max, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(k_v2_absmax, dim3(8), dim3(256), 0, 0, W, (long long)N, ld, kvalid, d_max);
    unsigned bits = 0;
    HIP_TRY(hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
    float mx;
    memcpy(&mx, &bits, sizeof(mx));
    if (!(mx < 3.0e38f)) return SPRK_OK;
    bool wide = false;
    if (int rcw = wide_dynamic_range(W, (long long)N, ld, kvalid, mx, &wide)) return rcw;
    if (wide) return SPRK_OK;
    int e = 0;
    float w_scale = 1.f;
    if (mx > 0.f) { (void)frexpf(mx, &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; w_scale = ldexpf(1.f, e); }
    const size_t frag_floats = (size_t)(N / 16) * (K / 32) * 512;
    float* f = nullptr;
    HIP_TRY(hipMalloc((void**)&f, frag_floats * sizeof(float)));
    h->fold_bufs.push_back(f);
    hipLaunchKernelGGL(k_dyn_pack_w, dim3(32), dim3(256), 0, 0, W, ld, N, K, w_scale, reinterpret_cast<_Float16*>(f), kvalid);
    HIP_TRY(hipGetLastError());
    *frag = f;
    *w_scale_out = w_scale;
    return SPRK_OK;
}


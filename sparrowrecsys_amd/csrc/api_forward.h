// api_forward.h -- C ABI: sprk_din_pool, sprk_forward, sprk_forward_many, sprk_describe, sprk_check_ids, sprk_destroy, operators, emb ranker.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// k_din_attn_cols: `n` batches of B rows (n = 1: ids / pooled; n > 1: the per-batch pointers of `many`)
static int launch_din_cols(sprk_handle h, const int32_t* ids, float* pooled, float* att, int32_t B, const DinColsMany* many, hipStream_t st) {
    DinColsRun c = h->din_cols_run;
    const long long ntasks = (long long)(many ? many->n : 1) * ((B + 15) / 16);
    // time slices per task: keep about four waves per SIMD on the chip when one launch has few tasks
    const long long slots = (long long)h->num_cus * 16;
    c.ts = (c.T >= 16 && ntasks * 4 <= slots) ? 4 : ((c.T >= 8 && ntasks * 2 <= slots) ? 2 : 1);
    if (h->tune.din_cols_ts) c.ts = h->tune.din_cols_ts;
    if (many) c.ts = 1;                                        // several batches: the persistent form, one wave per task (k_din_fused.h)
    c.ts_log2 = c.ts == 4 ? 2 : (c.ts == 2 ? 1 : 0);
    c.ql = (c.T + 3) / 4;
    const int EL = 4 * h->din_cols_kc;
    const size_t lds = ((size_t)2 * 64 * 36 + (size_t)DC_WAVES * 16 * c.idp + (size_t)DC_WAVES * 2 * 64 * EL) * sizeof(float);
    const long long grid = (ntasks * c.ts + DC_WAVES - 1) / DC_WAVES;
    if (many) {
        if (h->din_cols_kc == 2)
            hipLaunchKernelGGL((k_din_attn_cols<2, true>), dim3((unsigned)grid), dim3(DC_WAVES * 64), lds, st, c, (const int*)nullptr, (float*)nullptr, (float*)nullptr, B, h->dev_err, *many);
        else
            hipLaunchKernelGGL((k_din_attn_cols<1, true>), dim3((unsigned)grid), dim3(DC_WAVES * 64), lds, st, c, (const int*)nullptr, (float*)nullptr, (float*)nullptr, B, h->dev_err, *many);
    } else {
        if (h->din_cols_kc == 2)
            hipLaunchKernelGGL((k_din_attn_cols<2, false>), dim3((unsigned)grid), dim3(DC_WAVES * 64), lds, st, c, ids, pooled, att, B, h->dev_err, DinColsOne{});
        else
            hipLaunchKernelGGL((k_din_attn_cols<1, false>), dim3((unsigned)grid), dim3(DC_WAVES * 64), lds, st, c, ids, pooled, att, B, h->dev_err, DinColsOne{});
    }
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

// k_din_fused: `n` batches of B rows (n = 1: ids / dense / out; n > 1: the per-batch pointers of `many`).  tail: scores out (the whole
// DIN forward); else pooled vectors out (the attention stage alone: sprk_din_pool, the two-launch path)
static int launch_din_fused(sprk_handle h, const int32_t* ids, const float* dense, float* out, float* att, int32_t B, const DinFusedMany* many,
                            bool tail, hipStream_t st) {
    DinFusedRun c = h->din_fused_run;
    const long long ntasks = (long long)(many ? many->n : 1) * ((B + 15) / 16);
    // time slices per task: one 8-wave workgroup per CU; spread a small launch over the chip
    const long long slots = (long long)h->num_cus * DF_WAVES;
    c.ts = (c.T >= 16 && ntasks * 4 <= slots) ? 4 : ((c.T >= 8 && ntasks * 2 <= slots) ? 2 : 1);
    if (h->tune.din_cols_ts) c.ts = h->tune.din_cols_ts;
    c.ts_log2 = c.ts == 4 ? 2 : (c.ts == 2 ? 1 : 0);
    c.ql = ((c.T + 3) / 4 + 3) & ~3;                          // slots per quarter, a multiple of four: a trip of the slot loop (two pairs) never straddles one
    const int kc = h->din_cols_kc;
    const size_t lds = ((size_t)DF_COEF_FLOATS + (tail ? (size_t)DinFusedImg::dma_floats : 0) + (size_t)DF_WAVES * 16 * c.idp +
                        (size_t)DF_WAVES * (tail ? 1 : 2) * 64 * 4 * kc) * sizeof(float);   // (k_din_fused.h: PREG)
    long long grid = (ntasks * c.ts + DF_WAVES - 1) / DF_WAVES;
    if (many && grid > h->num_cus) grid = h->num_cus;         // persistent: the waves walk the tasks
#define DF_LAUNCH(KC, MB, TAIL, marg)                                                                                                   \
    hipLaunchKernelGGL((k_din_fused<KC, MB, TAIL>), dim3((unsigned)grid), dim3(DF_WAVES * 64), lds, st, c, MB ? (const int*)nullptr : ids, \
                       MB ? (const float*)nullptr : dense, MB ? (float*)nullptr : out, MB ? (float*)nullptr : att, B, h->dev_err, marg)
#ifdef SPRK_DF_XP
    if (h->tune.df_xp == 128 && kc == 2 && many && tail) {        // the persistent form without the folded rows' gathers
        hipLaunchKernelGGL((k_din_fused<2, true, true, false, 128>), dim3((unsigned)grid), dim3(DF_WAVES * 64), lds, st, c, (const int*)nullptr,
                           (const float*)nullptr, (float*)nullptr, (float*)nullptr, B, h->dev_err, *many);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->tune.df_xp >= 128 && kc == 2 && !many && tail) {       // the tail's ablations
#define DF_XPT(X) case X: hipLaunchKernelGGL((k_din_fused<2, false, true, false, X>), dim3((unsigned)grid), dim3(DF_WAVES * 64), lds, st, c, ids, dense, out, att, B, h->dev_err, DinFusedOne{}); break;
        switch (h->tune.df_xp) { DF_XPT(128) DF_XPT(256) DF_XPT(512) DF_XPT(896) DF_XPT(1024) default: return fail(SPRK_EINVAL, "SPRK_DF_XP=%d is not compiled in", h->tune.df_xp); }
#undef DF_XPT
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->tune.df_xp && h->tune.df_xp < 128 && kc == 2 && !many && !tail && !att) {     // ablation builds only (scripts/r04): garbage out, the time is the point
#define DF_XP(X) case X: hipLaunchKernelGGL((k_din_fused<2, false, false, false, X>), dim3((unsigned)grid), dim3(DF_WAVES * 64), lds, st, c, ids, dense, out, att, B, h->dev_err, DinFusedOne{}); break;
        switch (h->tune.df_xp) { DF_XP(1) DF_XP(2) DF_XP(4) DF_XP(8) DF_XP(16) DF_XP(32) DF_XP(64) DF_XP(3) DF_XP(56) DF_XP(60) DF_XP(63) DF_XP(127) DF_XP(65) DF_XP(126) default: return fail(SPRK_EINVAL, "SPRK_DF_XP=%d is not compiled in", h->tune.df_xp); }
#undef DF_XP
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
#endif
    if (many) {
        if (kc == 2) { if (tail) DF_LAUNCH(2, true, true, *many); else DF_LAUNCH(2, true, false, *many); }
        else { if (tail) DF_LAUNCH(1, true, true, *many); else DF_LAUNCH(1, true, false, *many); }
    } else {
        if (att && !tail) {                                     // attention weights out (tests, inspection): its own instantiation
            if (kc == 2) hipLaunchKernelGGL((k_din_fused<2, false, false, true>), dim3((unsigned)grid), dim3(DF_WAVES * 64), lds, st, c, ids, dense, out, att, B, h->dev_err, DinFusedOne{});
            else hipLaunchKernelGGL((k_din_fused<1, false, false, true>), dim3((unsigned)grid), dim3(DF_WAVES * 64), lds, st, c, ids, dense, out, att, B, h->dev_err, DinFusedOne{});
        } else if (kc == 2) { if (tail) DF_LAUNCH(2, false, true, DinFusedOne{}); else DF_LAUNCH(2, false, false, DinFusedOne{}); }
        else { if (tail) DF_LAUNCH(1, false, true, DinFusedOne{}); else DF_LAUNCH(1, false, false, DinFusedOne{}); }
    }
#undef DF_LAUNCH
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

static int launch_din(sprk_handle h, const int32_t* ids, float* pooled, float* att, int32_t B, hipStream_t st) {
    if (h->plan.din.enabled == 2) {                               // DIEN: GRU -> attention gate -> AUGRU, one lane per sample
        if (att) return fail(SPRK_EINVAL, "DIEN stage has no attention output");
        if (h->dien_frag) {
            DienRun run = h->dien_run;
            run.image = h->dien_frag;
            const int ntiles = (B + 15) / 16;
            int gridm = (ntiles + DM_WAVES - 1) / DM_WAVES;
            if (gridm > h->num_cus * 8) gridm = h->num_cus * 8;
            const size_t lds10 = DienFrag<10, 32>::total_pad * sizeof(float), lds16 = DienFrag<16, 32>::total_pad * sizeof(float);
            if (h->plan.din.emb_dim == 10)
                hipLaunchKernelGGL((k_dien_seq_mfma<10, 32>), dim3(gridm), dim3(DM_WAVES * 64), lds10, st, run, ids, pooled, B, h->dev_err);
            else
                hipLaunchKernelGGL((k_dien_seq_mfma<16, 32>), dim3(gridm), dim3(DM_WAVES * 64), lds16, st, run, ids, pooled, B, h->dev_err);
            HIP_TRY(hipGetLastError());
            return SPRK_OK;
        }
        int grid = (B + 63) / 64;
        if (grid > h->num_cus * 8) grid = h->num_cus * 8;
        if (h->plan.din.emb_dim == 10)
            hipLaunchKernelGGL((k_dien_seq<10, 32>), dim3(grid), dim3(64), 0, st, h->dien_run, ids, pooled, B, h->dev_err);
        else
            hipLaunchKernelGGL((k_dien_seq<16, 32>), dim3(grid), dim3(64), 0, st, h->dien_run, ids, pooled, B, h->dev_err);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->din_variant >= 0 && h->din_cols && h->din_fused_attn) return launch_din_fused(h, ids, nullptr, pooled, att, B, nullptr, false, st);
    if (h->din_variant >= 0 && h->din_cols) return launch_din_cols(h, ids, pooled, att, B, nullptr, st);
    const int nchunks = (B + h->din_ms - 1) / h->din_ms;
    const int grid = nchunks < h->din_grid_cap ? nchunks : h->din_grid_cap;
    hipLaunchKernelGGL(k_din_pool, dim3(grid), dim3(256), h->din_lds_bytes, st, h->dev_plan, ids, pooled, att, B, h->din_ms, h->dev_err);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

int sprk_din_pool(sprk_handle h, const int32_t* ids, float* pooled, float* att, int32_t B, void* stream) {
    RoctxRange roctx_range_("sprk_din_pool");
    if (!h || !ids || !pooled) return fail(SPRK_EINVAL, "NULL argument");
    if (!h->finalized) return fail(SPRK_ESTATE, "din_pool before finalize");
    if (!h->plan.din.enabled) return fail(SPRK_EKIND, "handle has no DIN stage");
    if (B <= 0) return B == 0 ? SPRK_OK : fail(SPRK_EINVAL, "negative batch");
    return launch_din(h, ids, pooled, att, B, (hipStream_t)stream);
}

int sprk_forward(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B,
                 void* workspace, size_t workspace_bytes, void* stream) {
    RoctxRange roctx_range_("sprk_forward");
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "forward before finalize");
    if (B < 0) return fail(SPRK_EINVAL, "negative batch");
    if (B == 0) return SPRK_OK;
    if (!out) return fail(SPRK_EINVAL, "out is NULL");
    if (h->plan.n_id_cols > 0 && !ids) return fail(SPRK_EINVAL, "ids is NULL");
    if (h->plan.n_dense > 0 && !dense) return fail(SPRK_EINVAL, "dense is NULL");
    hipStream_t st = (hipStream_t)stream;
    const float* aux = nullptr;
    if (h->plan.din.enabled) {
        const size_t need = sprk_workspace_bytes(h, B);
        if (!workspace || workspace_bytes < need) return fail(SPRK_EINVAL, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
        // DIN in one launch: attention + pooling + tail (k_din_fused); the workspace contract stays, the buffer is not touched.  One
        // 8-wave workgroup per CU carries the tail's weights: up to ONE round of workgroups (16 rows x 8 waves x CUs = 32 768 rows on an
        // MI355X: BASELINE config 3, and every latency-bound request below it) a wave owns one task; a larger batch goes to the
        // persistent form (a "several batches" launch of one batch: every wave walks its tasks, tables staged once)
        // (this branch must stay UNCONDITIONAL on din_fused: forward_many hands it the caller's whole workspace, not a per-stream slice)
        if (h->plan.din.enabled == 1 && h->din_fused) {
            if ((long long)((B + 15) / 16) <= (long long)h->num_cus * DF_WAVES) return launch_din_fused(h, ids, dense, out, nullptr, B, nullptr, true, st);
            DinFusedMany fm;
            memset(&fm, 0, sizeof(fm));
            fm.n = 1; fm.ids[0] = ids; fm.dense[0] = dense; fm.out[0] = out;
            return launch_din_fused(h, nullptr, nullptr, nullptr, nullptr, B, &fm, true, st);
        }
        // DIEN in one launch (k_dien_fused): as above, the workspace is not touched and the branch is unconditional on dien_fused
        if (h->plan.din.enabled == 2 && h->dien_fused) {
            DienRun run = h->dien_run;
            run.image = h->dien_frag;
            const int ntiles = (B + 15) / 16;
            int grid = (ntiles + DNF_WAVES - 1) / DNF_WAVES;
            if (grid > h->num_cus) grid = h->num_cus;              // one 16-wave workgroup per CU (both weight images in its LDS)
            if (h->plan.din.emb_dim == 10)
                hipLaunchKernelGGL((k_dien_fused<10, 32, 8, 4>), dim3(grid), dim3(DNF_WAVES * 64), h->dien_fused_lds, st, run, h->din_tail_run, ids, dense, out, B,
                                   h->dev_err, (const float*)h->din_tail_image, (float*)workspace);
            else
                hipLaunchKernelGGL((k_dien_fused<16, 32, 8, 4>), dim3(grid), dim3(DNF_WAVES * 64), h->dien_fused_lds, st, run, h->din_tail_run, ids, dense, out, B,
                                   h->dev_err, (const float*)h->din_tail_image, (float*)workspace);
            HIP_TRY(hipGetLastError());
            return SPRK_OK;
        }
        int rc = launch_din(h, ids, (float*)workspace, nullptr, B, st);
        if (rc) return rc;
        aux = (const float*)workspace;
    }
    if (h->v2_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + V2_WAVES - 1) / V2_WAVES;
        if (grid > h->v2_grid_cap) grid = h->v2_grid_cap;
        {                                                          // ([r6] a DeepFM_v2 handle always has a joint form: finalize routes the others elsewhere)
            V2JRun jr = h->v2j_run;
            jr.flags = (((uintptr_t)ids | (uintptr_t)dense) & 15) ? 1 : 0;  // unaligned inputs: element-wise staging
            if (h->v2j1_image && ntasks <= V2J1_MAX_TASKS) {
                // one strict launch of one batch: one task per wave, four waves per SIMD (k_chain_v2j1.h)
                for (size_t v = 0; v < sizeof(kV2J1Variants) / sizeof(kV2J1Variants[0]); ++v)
                    if (kV2J1Variants[v].g_big == kV2JVariants[h->v2j_variant].g_big && kV2J1Variants[v].njf == kV2JVariants[h->v2j_variant].njf) {
                        (h->v2j1_hoist ? kV2J1Variants[v].launch_h : kV2J1Variants[v].launch)(jr, ids, dense, out, B, h->dev_err, h->v2j1_image,
                                                                                                (ntasks + h->v2j1_waves - 1) / h->v2j1_waves, h->v2j1_lds_bytes, st);
                        HIP_TRY(hipGetLastError());
                        return SPRK_OK;
                    }
            }
            kV2JVariants[h->v2j_variant].launch(jr, ids, dense, out, B, h->dev_err, h->v2_image, grid, h->v2j_lds_bytes, st);
            HIP_TRY(hipGetLastError());
            return SPRK_OK;
        }
    }
    if (h->rows_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + RC_WAVES - 1) / RC_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (2 waves per SIMD)
        RowsRun rr = h->rows_run;
        rr.flags = (((uintptr_t)ids | (uintptr_t)dense) & 15) ? 1 : 0;
        if (h->rows_one && ntasks <= V2J1_MAX_TASKS)
            kRowsVariants[h->rows_variant].launch_one(rr, ids, dense, out, B, h->dev_err, h->rows_image, (ntasks + RC_WAVES - 1) / RC_WAVES,
                                                      h->rows_lds_bytes, st);
        else
            kRowsVariants[h->rows_variant].launch(rr, ids, dense, out, B, h->dev_err, h->rows_image, grid, h->rows_lds_bytes, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->v1_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + V1_WAVES - 1) / V1_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (2 waves per SIMD; 3 per SIMD measured slower at B = 65 536)
        if (h->v1_one && ntasks <= V1_ONE_MAX_TASKS)
            kV1Variants[h->v1_variant].launch_one(h->v1_run, ids, dense, out, B, h->dev_err, (ntasks + V1_WAVES - 1) / V1_WAVES, st);
        else
            kV1Variants[h->v1_variant].launch(h->v1_run, ids, dense, out, B, h->dev_err, grid, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->mlp_rows_nbig >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + MR_WAVES - 1) / MR_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (the LDS holds weights + genre tables)
        MlpRowsRun rr = h->mlp_rows_run;
        rr.flags = (rr.flags & ~1) | ((((uintptr_t)ids | (uintptr_t)dense) & 15) ? 1 : 0);
        hipLaunchKernelGGL(h->mlp_rows_kernel, dim3(grid), dim3(MR_WAVES * 64), h->mlp_rows_lds, st, rr, ids, dense, out, B, h->dev_err, (const float*)h->mlp_rows_image);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->din_tail_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        const int tw = h->din_tail_run.e_unscale != 0.f ? dt_waves_unf(kDinTailVariants[h->din_tail_variant].kpc) : DT_WAVES;
        int grid = (ntasks + tw - 1) / tw;
        if (grid > h->num_cus) grid = h->num_cus;                  // one workgroup per CU (8 waves, or 16 with raw embedding rows)
        kDinTailVariants[h->din_tail_variant].launch(h->din_tail_run, ids, dense, aux, out, B, h->dev_err, h->din_tail_image, grid, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    const int ntiles = (B + SPRK_TILE_M - 1) / SPRK_TILE_M;
    const int grid = ntiles < h->tile_grid_cap ? ntiles : h->tile_grid_cap;
    hipLaunchKernelGGL(k_tile_forward, dim3(grid), dim3(256), h->tile_lds_bytes, st, h->dev_plan, ids, dense, aux, out, B, h->dev_err);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

// `many_batches` / `many_streams`: batches per launch and helper streams of THIS call (the handle itself is not touched)
static int forward_many_impl(sprk_handle h, int32_t n_batches, const int32_t* const* ids, const float* const* dense,
                             float* const* out, int32_t B, void* workspace, size_t workspace_bytes, void* stream,
                             int many_batches, int many_streams) {
    RoctxRange roctx_range_("sprk_forward_many");
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (n_batches < 0) return fail(SPRK_EINVAL, "negative batch count");
    if (n_batches > 0 && !out) return fail(SPRK_EINVAL, "out is NULL");
    int S = (h->finalized && n_batches > 1) ? many_streams : 0;
    // a model with a workspace (DIN: attention kernel -> pooled vectors -> tail kernel) needs one workspace slice per
    // stream; with a single slice its forwards stay in strict order
    const size_t ws_need = (sprk_workspace_bytes(h, B) + 255) & ~(size_t)255;
    const bool ws_untouched = h->finalized && ((h->plan.din.enabled == 1 && h->din_fused) || (h->plan.din.enabled == 2 && h->dien_fused));   // (sprk_forward's k_din_fused branch: unconditional, see there)
    if (S >= 2 && ws_need > 0 && !ws_untouched) {
        while (S >= 2 && (!workspace || workspace_bytes < (size_t)S * ws_need)) --S;
        if (S < 2) S = 0;
    }
    // several batches per launch (sprk_set_many_batches): the fused DeepFM_v2 kernel takes up to V2J_MB batches' buffers
    // and walks their tasks as one grid; everything else (other models, unaligned buffers, tracing) goes batch by batch
    if (h->finalized && many_batches > 1 && n_batches > 1 && h->v2_variant >= 0 && h->v2j_variant >= 0 &&
        B > 0 && ids && dense) {
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i)
            ok = ids[i] && dense[i] && out[i] && !(((uintptr_t)ids[i] | (uintptr_t)dense[i]) & 15);
        if (ok) {
            const int ntpb = (B + 15) / 16;
            const V2JVariant& jv = kV2JVariants[h->v2j_variant];
            V2JRun jr = h->v2j_run;
            jr.flags = 0;
            for (int32_t i0 = 0; i0 < n_batches; i0 += many_batches) {
                V2JMany m;
                memset(&m, 0, sizeof(m));
                m.n = n_batches - i0 < many_batches ? n_batches - i0 : many_batches;
                m.ntpb = ntpb;
                for (int j = 0; j < m.n; ++j) { m.ids[j] = ids[i0 + j]; m.dense[j] = dense[i0 + j]; m.out[j] = out[i0 + j]; }
                const long long ntasks = (long long)m.n * ntpb;
                long long grid = (ntasks + V2_WAVES - 1) / V2_WAVES;
                if (grid > h->v2_grid_cap) grid = h->v2_grid_cap;
                jv.launch_many(jr, m, B, h->dev_err, h->v2_image, (int)grid, h->v2j_lds_bytes, (hipStream_t)stream);
                HIP_TRY(hipGetLastError());
            }
            return SPRK_OK;
        }
    }
    // k_rows_chain: up to RC_MB batches per launch
    if (h->finalized && many_batches > 1 && n_batches > 1 && h->rows_variant >= 0 && B > 0 && ids && (dense || h->plan.n_dense == 0)) {
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i)
            ok = ids[i] && out[i] && (h->plan.n_dense == 0 || dense[i]) && !(((uintptr_t)ids[i] | (uintptr_t)(h->plan.n_dense ? dense[i] : nullptr)) & 15);
        if (ok) {
            const int per = many_batches < RC_MB ? many_batches : RC_MB;
            const int ntpb = (B + 15) / 16;
            RowsRun rr = h->rows_run;
            rr.flags = 0;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per) {
                RowsMany m;
                memset(&m, 0, sizeof(m));
                m.n = n_batches - i0 < per ? n_batches - i0 : per;
                m.ntpb = ntpb;
                for (int j = 0; j < m.n; ++j) { m.ids[j] = ids[i0 + j]; m.dense[j] = h->plan.n_dense ? dense[i0 + j] : nullptr; m.out[j] = out[i0 + j]; }
                long long grid = ((long long)m.n * ntpb + RC_WAVES - 1) / RC_WAVES;
                if (grid > h->num_cus) grid = h->num_cus;
                kRowsVariants[h->rows_variant].launch_many(rr, m, B, h->dev_err, h->rows_image, (int)grid, h->rows_lds_bytes, (hipStream_t)stream);
                HIP_TRY(hipGetLastError());
            }
            return SPRK_OK;
        }
    }
    // [r6] k_mlp_rows (EmbeddingMLP / Wide&Deep): up to MR_MB batches per launch
    if (h->finalized && many_batches > 1 && n_batches > 1 && h->mlp_rows_nbig >= 0 && h->mlp_rows_many_kernel && B > 0 && ids && dense) {
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i)
            ok = ids[i] && dense[i] && out[i] && !(((uintptr_t)ids[i] | (uintptr_t)dense[i]) & 15);
        if (ok) {
            const int per = many_batches < MR_MB ? many_batches : MR_MB;
            const int ntpb = (B + 15) / 16;
            MlpRowsRun rr = h->mlp_rows_run;
            rr.flags &= ~1;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per) {
                MlpRowsMany m;
                memset(&m, 0, sizeof(m));
                m.n = n_batches - i0 < per ? n_batches - i0 : per;
                for (int j = 0; j < m.n; ++j) { m.ids[j] = ids[i0 + j]; m.dense[j] = dense[i0 + j]; m.out[j] = out[i0 + j]; }
                long long grid = ((long long)m.n * ntpb + MR_WAVES - 1) / MR_WAVES;
                if (grid > h->num_cus) grid = h->num_cus;
                hipLaunchKernelGGL(h->mlp_rows_many_kernel, dim3((int)grid), dim3(MR_WAVES * 64), h->mlp_rows_lds, (hipStream_t)stream, rr, m, B, h->dev_err,
                                   (const float*)h->mlp_rows_image);
                HIP_TRY(hipGetLastError());
            }
            return SPRK_OK;
        }
    }
    // the pairwise-dot DeepFM kernel: up to V1_MB batches per launch
    if (h->finalized && many_batches > 1 && n_batches > 1 && h->v2_variant < 0 && h->v1_variant >= 0 && B > 0 && ids && dense) {
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i) ok = ids[i] && dense[i] && out[i];
        if (ok) {
            const int per = many_batches < V1_MB ? many_batches : V1_MB;
            const int ntpb = (B + 15) / 16;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per) {
                V1Many m;
                memset(&m, 0, sizeof(m));
                m.n = n_batches - i0 < per ? n_batches - i0 : per;
                m.ntpb = ntpb;
                for (int j = 0; j < m.n; ++j) { m.ids[j] = ids[i0 + j]; m.dense[j] = dense[i0 + j]; m.out[j] = out[i0 + j]; }
                long long grid = ((long long)m.n * ntpb + V1_WAVES - 1) / V1_WAVES;
                if (grid > h->num_cus) grid = h->num_cus;
                kV1Variants[h->v1_variant].launch_many(h->v1_run, m, B, h->dev_err, (int)grid, (hipStream_t)stream);
                HIP_TRY(hipGetLastError());
            }
            return SPRK_OK;
        }
    }
    // DIN on k_din_fused: up to DF_MB batches per launch, one PERSISTENT launch does everything (k_din_fused.h: tables staged once, the
    // waves walk the tasks on their own); groups alternate over the helper streams.  32.2 us per batch of BASELINE config 3 against
    // 33.4 for the two-launch pipeline below on the same box (profiles/r04; SPRK_DIN_FUSED_MB=0 brings the pipeline back)
    if (h->finalized && many_batches > 1 && n_batches > 1 && h->plan.din.enabled == 1 && h->din_fused && h->tune.din_fused_mb && B > 0 && ids && dense) {
        const int per = many_batches < DF_MB ? many_batches : DF_MB;
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i) ok = ids[i] && dense[i] && out[i];
        if (ok) {
            const int SG = S >= 2 ? S : 1;
            if (SG >= 2) {
                HIP_TRY(hipEventRecord(h->many_fork, (hipStream_t)stream));
                for (int s = 0; s < SG; ++s) HIP_TRY(hipStreamWaitEvent(h->many_stream[s], h->many_fork, 0));
            }
            int g = 0;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per, ++g) {
                DinFusedMany fm;
                memset(&fm, 0, sizeof(fm));
                fm.n = n_batches - i0 < per ? n_batches - i0 : per;
                for (int j = 0; j < fm.n; ++j) { fm.ids[j] = ids[i0 + j]; fm.dense[j] = dense[i0 + j]; fm.out[j] = out[i0 + j]; }
                const int rcf = launch_din_fused(h, nullptr, nullptr, nullptr, nullptr, B, &fm, true, SG >= 2 ? h->many_stream[g % SG] : (hipStream_t)stream);
                if (rcf) return rcf;
            }
            if (SG >= 2) {
                for (int s = 0; s < SG; ++s) {
                    HIP_TRY(hipEventRecord(h->many_join[s], h->many_stream[s]));
                    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->many_join[s], 0));
                }
            }
            return SPRK_OK;
        }
    }
    // DIN (k_din_attn -> pooled vectors -> k_din_tail): the attention launches of a group of batches, then ONE tail launch for
    // the group; a workspace slice per batch of the group.  Groups alternate over the helper streams when there are slices for that.
    if (h->finalized && many_batches > 1 && n_batches > 1 && h->plan.din.enabled == 1 && h->din_variant >= 0 &&
        h->din_tail_variant >= 0 && B > 0 && ids && dense && workspace && ws_need > 0) {
        int per = many_batches < DIN_MB ? many_batches : DIN_MB;
        if ((size_t)per * ws_need > workspace_bytes) per = (int)(workspace_bytes / ws_need);
        bool ok = per >= 2;
        for (int32_t i = 0; i < n_batches && ok; ++i) ok = ids[i] && dense[i] && out[i];
        if (ok) {
            int SG = S >= 2 ? S : 1;                                   // streams the groups alternate over
            while (SG > 1 && (size_t)SG * per * ws_need > workspace_bytes) --SG;
            if (SG >= 2) {
                HIP_TRY(hipEventRecord(h->many_fork, (hipStream_t)stream));
                for (int s = 0; s < SG; ++s) HIP_TRY(hipStreamWaitEvent(h->many_stream[s], h->many_fork, 0));
            }
            const DinTailVariant& tv = kDinTailVariants[h->din_tail_variant];
            const int ntpb = (B + 15) / 16;
            int g = 0;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per, ++g) {
                const int n = n_batches - i0 < per ? n_batches - i0 : per;
                hipStream_t st = SG >= 2 ? h->many_stream[g % SG] : (hipStream_t)stream;
                char* wbase = (char*)workspace + (size_t)(g % SG) * per * ws_need;
                DinTailMany tm;
                memset(&tm, 0, sizeof(tm));
                tm.n = n; tm.ntpb = ntpb;
                for (int j = 0; j < n; ++j) {
                    float* pooled = (float*)(wbase + (size_t)j * ws_need);
                    tm.ids[j] = ids[i0 + j]; tm.dense[j] = dense[i0 + j]; tm.aux[j] = pooled; tm.out[j] = out[i0 + j];
                }
                // ONE attention launch for the group (k_din_attn<..., MB = true>: no launch boundary and no partial last round of waves
                // between the batches), then one tail launch
                if (h->din_attn_many && n <= DC_MB) {                   // (din_variant >= 0 <=> k_din_attn_cols' tables exist)
                    DinColsMany cm;
                    memset(&cm, 0, sizeof(cm));
                    cm.n = n;
                    for (int j = 0; j < n; ++j) { cm.ids[j] = tm.ids[j]; cm.pooled[j] = const_cast<float*>(tm.aux[j]); }
                    const int rcc = launch_din_cols(h, nullptr, nullptr, nullptr, B, &cm, st);
                    if (rcc) return rcc;
                } else {
                    for (int j = 0; j < n; ++j) {
                        const int rcc = launch_din_cols(h, tm.ids[j], const_cast<float*>(tm.aux[j]), nullptr, B, nullptr, st);
                        if (rcc) return rcc;
                    }
                }
                const int tw = h->din_tail_run.e_unscale != 0.f ? dt_waves_unf(kDinTailVariants[h->din_tail_variant].kpc) : DT_WAVES;
                long long tg = ((long long)n * ntpb + tw - 1) / tw;
                if (tg > h->num_cus) tg = h->num_cus;
                tv.launch_many(h->din_tail_run, tm, B, h->dev_err, h->din_tail_image, (int)tg, st);
                HIP_TRY(hipGetLastError());
            }
            if (SG >= 2) {
                for (int s = 0; s < SG; ++s) {
                    HIP_TRY(hipEventRecord(h->many_join[s], h->many_stream[s]));
                    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->many_join[s], 0));
                }
            }
            return SPRK_OK;
        }
    }
    if (S >= 2) {
        HIP_TRY(hipEventRecord(h->many_fork, (hipStream_t)stream));
        for (int s = 0; s < S; ++s) HIP_TRY(hipStreamWaitEvent(h->many_stream[s], h->many_fork, 0));
    }
    for (int32_t i = 0; i < n_batches; ++i) {
        // (one workspace slice per stream -- except for the one-launch DIN, whose S was NOT reduced to the slices the buffer holds because
        //  its forward never touches the buffer: it gets the caller's buffer and size unchanged, never an offset past its end [ADVICE r04])
        const bool sliced = S >= 2 && ws_need > 0 && !ws_untouched;
        void* wsi = sliced ? (void*)((char*)workspace + (size_t)(i % S) * ws_need) : workspace;
        const int rc = sprk_forward(h, ids ? ids[i] : nullptr, dense ? dense[i] : nullptr, out[i], B, wsi,
                                    sliced ? ws_need : workspace_bytes, S >= 2 ? (void*)h->many_stream[i % S] : stream);
        if (rc) return rc;
    }
    if (S >= 2) {
        for (int s = 0; s < S; ++s) {
            HIP_TRY(hipEventRecord(h->many_join[s], h->many_stream[s]));
            HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->many_join[s], 0));
        }
    }
    return SPRK_OK;
}

int sprk_forward_many(sprk_handle h, int32_t n_batches, const int32_t* const* ids, const float* const* dense,
                      float* const* out, int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    return forward_many_impl(h, n_batches, ids, dense, out, B, workspace, workspace_bytes, stream, h->many_batches, h->many_streams);
}

int sprk_forward_many_opts(sprk_handle h, int32_t n_batches, const int32_t* const* ids, const float* const* dense,
                           float* const* out, int32_t B, void* workspace, size_t workspace_bytes, void* stream,
                           int32_t batches_per_launch, int32_t helper_streams) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (batches_per_launch < 1 || batches_per_launch > V2J_MB) return fail(SPRK_EINVAL, "batches per launch %d outside [1,%d]", batches_per_launch, V2J_MB);
    if (helper_streams < 0 || helper_streams > 4) return fail(SPRK_EINVAL, "stream count %d outside [0,4]", helper_streams);
    return forward_many_impl(h, n_batches, ids, dense, out, B, workspace, workspace_bytes, stream, batches_per_launch,
                             helper_streams < 2 ? 0 : helper_streams);
}

#define SPRK_FORWARD_KIND(name, kind)                                                                      \
    int name(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws,      \
             size_t ws_bytes, void* stream) {                                                              \
        if (!h) return fail(SPRK_EINVAL, "handle is NULL");                                                \
        if (h->plan.model_kind != kind) return fail(SPRK_EKIND, #name ": handle holds model kind %d", h->plan.model_kind); \
        return sprk_forward(h, ids, dense, out, B, ws, ws_bytes, stream);                                  \
    }
SPRK_FORWARD_KIND(sprk_forward_embedding_mlp, SPRK_MODEL_EMBEDDING_MLP)
SPRK_FORWARD_KIND(sprk_forward_widedeep, SPRK_MODEL_WIDE_DEEP)
SPRK_FORWARD_KIND(sprk_forward_neuralcf, SPRK_MODEL_NEURALCF)
SPRK_FORWARD_KIND(sprk_forward_deepfm, SPRK_MODEL_DEEPFM)
SPRK_FORWARD_KIND(sprk_forward_deepfm_v2, SPRK_MODEL_DEEPFM_V2)
SPRK_FORWARD_KIND(sprk_forward_din, SPRK_MODEL_DIN)
SPRK_FORWARD_KIND(sprk_forward_dien, SPRK_MODEL_DIEN)

int sprk_set_many_streams(sprk_handle h, int32_t n) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "set_many_streams before finalize");
    if (n < 0 || n > 4) return fail(SPRK_EINVAL, "stream count %d outside [0,4]", n);
    h->many_streams = n < 2 ? 0 : n;
    return SPRK_OK;
}

int sprk_set_many_batches(sprk_handle h, int32_t n) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "set_many_batches before finalize");
    if (n < 1 || n > V2J_MB) return fail(SPRK_EINVAL, "batches per launch %d outside [1,%d]", n, V2J_MB);   // (DIN caps at DIN_MB)
    h->many_batches = n;
    return SPRK_OK;
}

int sprk_describe(sprk_handle h, char* buf, size_t buf_bytes) {
    if (!h || !buf || buf_bytes == 0) return fail(SPRK_EINVAL, "describe: NULL argument");
    if (!h->finalized) return fail(SPRK_ESTATE, "describe before finalize");
    char kern[160];
    if (h->v2_variant >= 0 && h->v2j_variant >= 0) {
        const V2JVariant& jv = kV2JVariants[h->v2j_variant];
        snprintf(kern, sizeof(kern), "k_deepfm_v2_joint<G_BIG=%d,NJF=%d,KPC=%d,%s>", jv.g_big, jv.njf, jv.kpc, jv.half ? "split-f16" : "f32");
    } else if (h->v1_variant >= 0) {
        snprintf(kern, sizeof(kern), "k_deepfm_pairs<NF=%d,NV=%d>", kV1Variants[h->v1_variant].nf, kV1Variants[h->v1_variant].nv);
    } else if (h->rows_variant >= 0) {
        const RowsVariant& rv = kRowsVariants[h->rows_variant];
        snprintf(kern, sizeof(kern), "k_rows_chain<KPC=%d,H0C=%d,H1C=%d,G_BIG=%d,NJF=%d%s>", rv.kpc, rv.h0c, rv.h1c, rv.g_big, rv.njf,
                 rv.unf ? ",UNF" : "");
    } else if (h->mlp_rows_nbig >= 0) {
        snprintf(kern, sizeof(kern), "k_mlp_rows<8,8,NBIG=%d,NSMALL=%d>", h->mlp_rows_nbig, h->mlp_rows_run.n_small);
    } else if (h->din_tail_variant >= 0) {
        const DinTailVariant& tv = kDinTailVariants[h->din_tail_variant];
        snprintf(kern, sizeof(kern), "k_din_tail<%d,%d,%d%s>", tv.n0c, tv.n1c, tv.kpc, h->din_tail_run.e_unscale != 0.f ? ",UNF" : "");
    } else {
        snprintf(kern, sizeof(kern), "k_tile_forward");
    }
    const char* stage = "";
    if (h->plan.din.enabled == 2) stage = h->dien_frag ? "k_dien_seq_mfma" : "k_dien_seq";
    else if (h->plan.din.enabled == 1) stage = h->din_variant < 0 ? "k_din_pool" : (h->din_fused_attn ? "k_din_fused" : "k_din_attn_cols");
    if (h->plan.din.enabled == 1 && h->din_fused) snprintf(kern, sizeof(kern), "k_din_fused<KC=%d,tail 128/64>", h->din_cols_kc);
    if (h->plan.din.enabled == 2 && h->dien_fused) snprintf(kern, sizeof(kern), "k_dien_fused<D=%d,tail 128/64>", h->plan.din.emb_dim);   // (stage: what sprk_din_pool runs)
    size_t uploaded = 0;
    for (size_t b : h->slot_bytes) uploaded += b;
#ifndef SPRK_BUILD_DEFINES_STR
#define SPRK_BUILD_DEFINES_STR ""                             // (_lib.build_library passes the experiment defines of SPRK_BUILD_DEFINES; the product build has none)
#endif
    const int n = snprintf(buf, buf_bytes, "kernel=%s;stage=%s;stage_waves_per_workgroup=%d;fused=%d;uploaded_bytes=%zu;derived_bytes=%zu;first_dense_fold=%d;build_defines=%s", kern, stage,
                           h->din_variant >= 0 ? DC_WAVES : 0, strcmp(kern, "k_tile_forward") != 0 ? 1 : 0, uploaded, h->derived_bytes, h->n_acc_folded, SPRK_BUILD_DEFINES_STR);
    if (n < 0 || (size_t)n >= buf_bytes) return fail(SPRK_EINVAL, "describe: buffer of %zu bytes is too small", buf_bytes);
    return SPRK_OK;
}

int sprk_check_ids(sprk_handle h, void* stream) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "check_ids before finalize");
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, h->dev_err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        HIP_TRY(hipMemsetAsync(h->dev_err, 0, sizeof(int), (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        return fail(SPRK_ERANGE, "an id was outside its table (TF would raise InvalidArgumentError: assert_less_than_num_buckets) [flag 0x%x]", flag);
    }
    return SPRK_OK;
}

void sprk_destroy(sprk_handle h) {
#ifdef SPRK_DF_XP
    if (h) {
        if (const char* path = getenv("SPRK_V2J1_TS_FILE")) {      // k_deepfm_v2_joint1's timeline: the LAST launch's stamps
            std::vector<unsigned long long> ts((size_t)V2J1_TS_WAVES * 8);
            hipDeviceSynchronize();
            if (hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_v2j1_ts), ts.size() * 8) == hipSuccess) {
                if (FILE* fp = fopen(path, "wb")) { fwrite(ts.data(), 8, ts.size(), fp); fclose(fp); }
            }
        }
    }
    if (h) {
        if (const char* path = getenv("SPRK_MR_TS_FILE")) {        // k_mlp_rows' timeline: the LAST launch's stamps
            std::vector<unsigned long long> ts((size_t)MR_TS_WAVES * MR_TS_SLOTS);
            hipDeviceSynchronize();
            if (hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_mr_ts), ts.size() * 8) == hipSuccess) {
                if (FILE* fp = fopen(path, "wb")) { fwrite(ts.data(), 8, ts.size(), fp); fclose(fp); }
            }
        }
    }
    if (h && h->tune.df_xp == 1024) {                              // the timeline build: the LAST launch's stamps -> $SPRK_DF_TS_FILE
        if (const char* path = getenv("SPRK_DF_TS_FILE")) {
            std::vector<unsigned long long> ts((size_t)DF_TS_WAVES * 8);
            hipDeviceSynchronize();
            if (hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_df_ts), ts.size() * 8) == hipSuccess) {
                if (FILE* fp = fopen(path, "wb")) { fwrite(ts.data(), 8, ts.size(), fp); fclose(fp); }
            }
        }
    }
#endif
    if (!h) return;
    for (size_t i = 0; i < h->slot_ptr.size(); ++i)
        if (h->slot_ptr[i] && !h->slot_external[i]) (void)hipFree(h->slot_ptr[i]);
    if (h->dev_plan) (void)hipFree(h->dev_plan);
    if (h->v2_image) (void)hipFree(h->v2_image);
    if (h->v2_fo_all) (void)hipFree(h->v2_fo_all);
    table_free(h, h->v2_folded);
    if (h->v2j_tab) (void)hipFree(h->v2j_tab);
    for (void* p : h->fold_bufs) if (p) (void)hipFree(p);
    for (int i = 0; i < 4; ++i) { if (h->many_stream[i]) (void)hipStreamDestroy(h->many_stream[i]); if (h->many_join[i]) (void)hipEventDestroy(h->many_join[i]); }
    if (h->many_fork) (void)hipEventDestroy(h->many_fork);
    if (h->din_tail_image) (void)hipFree(h->din_tail_image);
    if (h->din_fused_image) (void)hipFree(h->din_fused_image);
    if (h->mlp_rows_image) (void)hipFree(h->mlp_rows_image);
    if (h->mlp_rows_small) (void)hipFree(h->mlp_rows_small);
    for (void* q : h->mlp_rows_bufs) table_free(h, q);
    for (void* p : h->v1_bufs) table_free(h, p);
    table_free(h, h->v2j_big);
    if (h->v2j1_image) (void)hipFree(h->v2j1_image);
    if (h->din_frag) (void)hipFree(h->din_frag);
    if (h->dien_frag) (void)hipFree(h->dien_frag);
    table_free(h, h->rows_tab);
    if (h->rows_scal) (void)hipFree(h->rows_scal);
    if (h->rows_small) (void)hipFree(h->rows_small);
    if (h->rows_image) (void)hipFree(h->rows_image);
    if (h->din_w12) (void)hipFree(h->din_w12);
    if (h->din_w4) (void)hipFree(h->din_w4);
    if (h->din_vc) (void)hipFree(h->din_vc);
    if (h->din_tsplit) (void)hipFree(h->din_tsplit);
    if (h->dev_err) (void)hipFree(h->dev_err);
    delete h;
}

int sprk_embedding_gather(const float* table, int32_t V, int32_t D, int32_t row_stride, const int32_t* ids,
                          int32_t B, float* out, void* stream) {
    if (!table || !ids || !out) return fail(SPRK_EINVAL, "NULL argument");
    if (V <= 0 || D <= 0 || (D & 3) || row_stride < D || (row_stride & 3)) return fail(SPRK_EINVAL, "bad gather geometry V=%d D=%d row_stride=%d", V, D, row_stride);
    if (B < 0) return fail(SPRK_EINVAL, "negative batch");
    if (B == 0) return SPRK_OK;
    const long long total = (long long)B * (D / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_embedding_gather, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, V, D / 4, row_stride, ids, B, out);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}


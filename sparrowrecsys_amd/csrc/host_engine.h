// host_engine.h -- struct sprk_engine: everything a finalized handle owns.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct sprk_engine {
    sprk_plan plan;
    std::vector<void*> slot_ptr;
    std::vector<size_t> slot_bytes;
    std::vector<char> slot_external;      // [r4] sprk_upload_external: the slot reads caller-owned device memory (never freed here)
    DevPlan* dev_plan = nullptr;
    int* dev_err = nullptr;
    bool finalized = false;
    int device = 0;
    int num_cus = 256;
    int buf_stride[SPRK_MAX_BUFS] = {0, 0, 0};
    int buf_base[SPRK_MAX_BUFS] = {0, 0, 0};
    size_t tile_lds_bytes = 0;
    int ids_base = 0;              // float offset of the tile's ids block inside the tile kernel's LDS
    std::vector<int> idc;          // ids columns read by the gather segments (compact staging order)
    int tile_grid_cap = 0;
    std::vector<void*> fold_bufs;  // first-Dense fold: folded tables + the W^T copy (device)
    // sprk_forward_many fan-out: independent batches alternate over helper streams (hardware queues), so that one
    // kernel's dispatch / drain (3.3 us even for an empty kernel in a dependent launch chain) overlaps its neighbours
    int many_streams = 0;
    hipStream_t many_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t many_fork = nullptr, many_join[4] = {nullptr, nullptr, nullptr, nullptr};
    // register-chained pairwise-dot DeepFM (k_deepfm_pairs); -1 = the tile interpreter
    int v1_variant = -1;
    bool v1_one = false;                  // one-batch launches use k_deepfm_pairs1 (one task per wave, four waves per SIMD)
    V1Run v1_run;
    std::vector<void*> v1_bufs;
    // DenseFeatures -> Dense -> Dense -> Dense(1) graphs with every embedding column folded through the first layer, genre tables
    // in LDS (k_mlp_rows); -1 = the tile interpreter
    int mlp_rows_nbig = -1;
    MlpRowsRun mlp_rows_run;
    void (*mlp_rows_kernel)(const MlpRowsRun, const int*, const float*, float*, int, int*, const float*) = nullptr;
    void (*mlp_rows_many_kernel)(const MlpRowsRun, const MlpRowsMany, int, int*, const float*) = nullptr;   // [r6] several batches per launch (nullptr: batch by batch)
    float* mlp_rows_image = nullptr;
    float* mlp_rows_small = nullptr;
    size_t mlp_rows_lds = 0;
    std::vector<void*> mlp_rows_bufs;
    // register-chained DIN tail (k_din_tail); -1 = the tile interpreter runs the tail
    int din_tail_variant = -1;
    DinTailRun din_tail_run;
    float* din_tail_image = nullptr;
    // DIN launch geometry
    int din_ms = 0;
    size_t din_lds_bytes = 0;
    int din_grid_cap = 0;
    // >= 0: the attention stage's derived tables (W12 / W4 / vc, the pre-split table) exist and k_din_attn_cols / k_din_fused run; -1 = the generic k_din_pool
    int din_variant = -1;
    DienRun dien_run{};
    float* dien_frag = nullptr;          // k_dien_seq_mfma's fragment image (NULL: the lane-per-sample kernel)
    float* din_w12 = nullptr;      // (W1+W2)^T, W4^T fragments and the per-id c-term table (device)
    float* din_w4 = nullptr;
    float* din_vc = nullptr;
    float* din_tsplit = nullptr;   // HALF: the movie table pre-split into f16 hi/lo pairs
    SprkTuning tune;               // the environment's switches as sprk_finalize found them
    bool din_cols = false;         // attention on k_din_attn_cols (16 samples per MFMA tile, static weight operand; k_din_cols.h)
    int din_cols_kc = 0;
    DinColsRun din_cols_run;
    float* din_frag = nullptr;     // its A fragments
    // the whole DIN forward in one launch (k_din_fused: attention + pooling + tail; k_din_fused.h)
    bool din_fused = false;        // TAIL instantiations usable (attention on the cols formulation AND the 128 / 64 tail recognised)
    bool din_fused_attn = false;   // TAIL = false instantiations replace k_din_attn_cols (sprk_din_pool, the unfused two-launch path)
    DinFusedRun din_fused_run;
    float* din_fused_image = nullptr;
    bool dien_fused = false;       // DIEN in one launch (k_dien_fused.h): dien_frag AND the 128 / 64 tail on raw split rows
    size_t dien_fused_lds = 0;
    bool din_attn_many = true;     // forward_many: one attention launch per group of batches 
    // register-chained fast path (k_deepfm_v2_chain); -1 = use the tile interpreter
    int v2_variant = -1;
    bool v2_rows_ok = false;       // [r6] the parsed DeepFM_v2 plan also fits k_rows_chain (where it goes when the joint set-up refuses it)
    V2Args v2;
    V2Run v2run;
    size_t v2_lds_bytes = 0;
    int v2_grid_cap = 0;
    float* v2_image = nullptr;     // pre-packed LDS weight image (device)
    float* v2_fo_all = nullptr;    // concatenated first-order weight blocks (device)
    float* v2_folded = nullptr;    // projected tables of all fields, back to back (device)
    size_t v2_fo_floats = 0;
    // ... with the small-vocabulary fields folded into one joint table (k_deepfm_v2_joint); -1 = not used
    int v2j_variant = -1;
    float* v2j1_image = nullptr;          // k_deepfm_v2_joint1 (one task per wave): its LDS image; NULL = shape not available
    size_t v2j1_lds_bytes = 0;
    bool v2j1_hoist = false;              // k_deepfm_v2_joint1<..., HOIST>: tables larger than the Infinity Cache (k_chain_v2j1.h)
    int v2j1_waves = 8;                   // waves per workgroup of the form chosen (V2J1_WAVES_OF)
    int many_batches = 1;                 // sprk_forward_many: batches scored per launch (sprk_set_many_batches)
    // "one row per id" chain (k_rows_chain): DeepFM_v2 with projections wider than 16 (the reference's Dense(64)) and NeuralCF
    int rows_variant = -1;
    bool rows_one = true;                 // one-batch launches use k_rows_chain1 (one task per wave)
    bool rows_from_v2 = false;            // set by match_v2_chain: h->v2 holds the parsed DeepFM_v2 plan, tables still to build
    int rows_g_emb = 0;
    RowsRun rows_run;
    float* rows_tab = nullptr;            // big fields' rows {P | Q}
    float* rows_scal = nullptr;           // big fields' per-id scalars
    float* rows_small = nullptr;          // small fields' LDS rows (device image)
    float* rows_image = nullptr;          // weight image
    size_t rows_lds_bytes = 0;
    int n_acc_folded = 0;                 // embedding columns folded into the first Dense layer (fold_first_dense)
    size_t derived_bytes = 0;             // device memory of tables DERIVED at finalize (folded rows, split halfs, per-id terms)
    V2JRun v2j_run;
    float* v2j_tab = nullptr;      // small fields' LDS rows (device image)
    size_t v2j_lds_bytes = 0;
    float* v2j_big = nullptr;      // HALF: split-half rows of the big fields (device)
    // [r6] engine-owned gather tables of 256 MB and more (table_alloc below): one physical allocation behind a virtual range
    struct VmmAlloc { void* va; size_t size; hipMemGenericAllocationHandle_t handle; };
    std::vector<VmmAlloc> vmm_allocs;
};

// [r6, VERDICT r05 item 4 (i)] A gather table the engine derives at finalize (folded / split rows).  From 256 MB on -- beyond the Infinity Cache,
// where every row of a batch is its own HBM access AND its own translation -- it is ONE physical allocation (hipMemCreate) mapped at a 1 GB-aligned
// virtual address, not hipMalloc's memory: scripts/ubench/row_gather.hip (`... vmm`) measures the random 128-byte-row gather over a 3.2 GB table
// at 7.15 instead of 7.46 us per 65 536 x 3 rows (8.37 instead of 8.69 with the fused kernel's staging and LDS traffic around it;
// profiles/r06/experiments/r06_15): the one handle lets the driver describe the range with larger page-table fragments.  SPRK_VMM_TABLES=0, any
// failure of the API, or a smaller table: hipMalloc.
static int table_alloc(sprk_engine* h, void** out, size_t bytes) {
    *out = nullptr;
    if (h->tune.vmm_tables && bytes >= ((size_t)256 << 20)) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = h->device;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess) {
            if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;
            const size_t total = (bytes + gran - 1) / gran * gran;
            void* va = nullptr;
            hipMemGenericAllocationHandle_t hnd;
            if (hipMemAddressReserve(&va, total, (size_t)1 << 30, nullptr, 0) == hipSuccess) {
                if (hipMemCreate(&hnd, total, &prop, 0) == hipSuccess) {
                    hipMemAccessDesc acc = {};
                    acc.location = prop.location;
                    acc.flags = hipMemAccessFlagsProtReadWrite;
                    if (hipMemMap(va, total, 0, hnd, 0) == hipSuccess) {
                        if (hipMemSetAccess(va, total, &acc, 1) == hipSuccess) {
                            h->vmm_allocs.push_back(sprk_engine::VmmAlloc{va, total, hnd});
                            *out = va;
                            return SPRK_OK;
                        }
                        (void)hipMemUnmap(va, total);
                    }
                    (void)hipMemRelease(hnd);
                }
                (void)hipMemAddressFree(va, total);
            }
        }
        (void)hipGetLastError();                                  // (the fall-back below is not an error)
    }
    HIP_TRY(hipMalloc(out, bytes));
    return SPRK_OK;
}
static void table_free(sprk_engine* h, void* p) {
    if (!p) return;
    for (size_t i = 0; i < h->vmm_allocs.size(); ++i)
        if (h->vmm_allocs[i].va == p) {
            (void)hipMemUnmap(p, h->vmm_allocs[i].size);
            (void)hipMemRelease(h->vmm_allocs[i].handle);
            (void)hipMemAddressFree(p, h->vmm_allocs[i].size);
            h->vmm_allocs.erase(h->vmm_allocs.begin() + (long)i);
            return;
        }
    (void)hipFree(p);
}


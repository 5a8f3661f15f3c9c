// host_common.h -- error reporting, HIP_TRY, roctx ranges, SprkTuning (the environment's switches, read once per finalize).
// Included first by every translation unit of the library (sparrow_hip.hip and the kernel-family units tu_*.hip).
// [r5] The kernels' namespace is NAMED (sprk_dev, hidden visibility): the heavy kernel templates are instantiated in the family units and
// only DECLARED (`extern template`, tu_instances.h) in sparrow_hip.hip, which needs external linkage for them and for the argument structs
// in their signatures.  Everything that is not a template or inline is `static` / `inline`, so that every unit can include every header.
#pragma GCC visibility push(hidden)
namespace sprk_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

inline thread_local std::string g_err;

inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(SPRK_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- roctx ranges around the C ABI's entry points ("roctx ranges in the C-ABI") ----
// A finalize-time probe buffer (an abs-max word, a flag): freed on EVERY way out of its scope, also when a HIP_TRY in between returns
// (the early returns between hipMalloc and hipFree leaked it).
struct DevProbe {
    unsigned* p = nullptr;
    ~DevProbe() { if (p) (void)hipFree(p); }
    DevProbe() = default;
    DevProbe(const DevProbe&) = delete;
    DevProbe& operator=(const DevProbe&) = delete;
};

// SPRK_ROCTX=1 binds rocprofiler-sdk's roctx at first use (dlopen: no link-time dependency) and every forward / ingest /
// exchange call then shows up as a named range in `rocprofv3 --marker-trace`; otherwise the cost is one predictable branch.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
    Roctx() {
        const char* e = getenv("SPRK_ROCTX");
        if (!e || e[0] != '1') return;
        void* lib = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        on = push && pop;
    }
};
inline const Roctx& roctx() { static const Roctx r; return r; }
struct RoctxRange {
    bool live;
    explicit RoctxRange(const char* name) : live(roctx().on) { if (live) roctx().push(name); }
    ~RoctxRange() { if (live) roctx().pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// ---- every A/B switch and tuning knob of an engine, read from the environment ONCE, at sprk_finalize ----
// (round 2 had 30 getenv() calls spread over the setup functions.)  The defaults are the shipped configuration; the
// switches exist so that tests and profiles can put two code paths side by side on the same inputs (tests/test_gpu_parity.py,
// profiles/rNN/README.md).  "off" = the variable is set and starts with '0', "on" = it starts with '1'.
struct SprkTuning {
    bool force_interpreter = false;   // SPRK_FORCE_INTERPRETER=1   every plan on k_tile_forward
    bool v2_fold = true;              // SPRK_V2_FOLD=0             DeepFM_v2: projections computed per sample instead of folded tables
    bool v2_rows = false;             // SPRK_V2_ROWS=1             DeepFM_v2 on k_rows_chain even where k_deepfm_v2_joint fits
    bool v2_joint = true;             // SPRK_V2_JOINT=0            no LDS-resident small fields (k_deepfm_v2_chain)
    bool v2_half = true;              // SPRK_V2_HALF=0             big fields on f32 MFMA instead of split f16
    bool v2j_one = true;              // SPRK_V2J_ONE=0             one-batch launches on the looped kernel, not k_deepfm_v2_joint1
    int v2j1_hoist = -1;              // SPRK_V2J1_HOIST=0|1        k_deepfm_v2_joint1's weight fragments read behind / in front of the gathers (default: by table size)
    bool tail_unf = true;             // SPRK_TAIL_UNF=0            k_din_tail: folded 512-byte rows for the embedding columns even when emb_dim <= 16
    bool tail_pooled_f16 = true;      // SPRK_TAIL_POOLED_F16=0     k_din_tail: the pooled-history columns of fc0 on f32 MFMA (round 2) instead of split f16
    bool dien_mfma = true;            // SPRK_DIEN_MFMA=0           DIEN sequence stage: one lane per sample (k_dien_seq) instead of 16 samples per MFMA tile
    bool rows_unf = true;             // SPRK_ROWS_UNF=0            k_rows_chain: folded {P | Q} rows for the big fields even when raw rows are 6x smaller
    bool ncf_chain = true;            // SPRK_NCF_CHAIN=0           NeuralCF on the interpreter
    bool tile_fold = true;            // SPRK_TILE_FOLD=0           interpreter without the first-Dense fold
    bool half_range_guard = true;     // SPRK_HALF_RANGE_GUARD=0    skip the dynamic-range guard of static f16 scales (tests only)
    bool dyn_f16 = true;              // SPRK_DYN_F16=0             hidden layers on f32 MFMA instead of dynamic split f16
    bool v1_chain = true;             // SPRK_V1_CHAIN=0            pair-dot DeepFM on the interpreter
    bool v1_one = true;               // SPRK_V1_ONE=0              one-batch launches on the looped kernel, not k_deepfm_pairs1
    bool mlp_chain = true;            // SPRK_MLP_CHAIN=0           EmbeddingMLP / Wide&Deep on the interpreter
    bool vmm_tables = true;           // SPRK_VMM_TABLES=0          derived gather tables of 256 MB and more through hipMalloc, not one hipMemCreate allocation (host_engine.h table_alloc)
    bool din_tail = true;             // SPRK_DIN_TAIL=0            DIN tail on the interpreter
    bool din_legacy = false;          // SPRK_DIN_LEGACY=1          attention on the generic k_din_pool
    bool din_half = true;             // SPRK_DIN_HALF=0            attention on f32 MFMA
    bool din_cols = true;             // SPRK_DIN_COLS=0            attention on the generic k_din_pool, not k_din_attn_cols / k_din_fused
    bool din_fused = true;            // SPRK_DIN_FUSED=0           DIN as two launches (k_din_attn_cols -> pooled vectors -> k_din_tail), not k_din_fused
    bool din_fused_mb = true;         // SPRK_DIN_FUSED_MB=0        forward_many groups on the attention + tail pipeline instead of the persistent k_din_fused<MB>
    bool din_fused_unf = true;        // SPRK_DIN_FUSED_UNF=0       k_din_fused's tail with folded rows for every embedding column (no raw split rows on the matrix pipe)
    bool dien_fused = true;           // SPRK_DIEN_FUSED=0          DIEN as two launches (k_dien_seq_mfma -> final states -> k_din_tail), not k_dien_fused
    int din_fused_min_t = 12;         // SPRK_DIN_FUSED_MIN_T=n     shortest history k_din_fused takes (below: k_din_attn_cols -> k_din_tail)
    bool mlp_rows_many = true;        // SPRK_MLP_ROWS_MANY=0       sprk_forward_many over k_mlp_rows graphs launch by launch (streams), not k_mlp_rows_many
    int df_xp = 0;                    // SPRK_DF_XP=bits            k_din_fused ablation variants (only in a -DSPRK_DF_XP build of the library)
    int din_cols_ts = 0;              // SPRK_DIN_COLS_TS=1|2|4     waves per task of k_din_attn_cols (0 = by the launch's task count)
    int many_streams = 0;             // SPRK_MANY_STREAMS=n        forward_many fans batches over n helper streams (2..4)
    static SprkTuning from_env() {
        auto off = [](const char* n) { const char* e = getenv(n); return e && e[0] == '0'; };
        auto on = [](const char* n) { const char* e = getenv(n); return e && e[0] == '1'; };
        auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
        SprkTuning t;
        t.force_interpreter = on("SPRK_FORCE_INTERPRETER");
        t.v2_fold = !off("SPRK_V2_FOLD"); t.v2_rows = on("SPRK_V2_ROWS"); t.v2_joint = !off("SPRK_V2_JOINT"); t.v2_half = !off("SPRK_V2_HALF");
        t.v2j_one = !off("SPRK_V2J_ONE"); t.v2j1_hoist = num("SPRK_V2J1_HOIST", -1);
        t.rows_unf = !off("SPRK_ROWS_UNF"); t.dien_mfma = !off("SPRK_DIEN_MFMA"); t.tail_pooled_f16 = !off("SPRK_TAIL_POOLED_F16"); t.tail_unf = !off("SPRK_TAIL_UNF"); t.ncf_chain = !off("SPRK_NCF_CHAIN"); t.tile_fold = !off("SPRK_TILE_FOLD");
        t.half_range_guard = !off("SPRK_HALF_RANGE_GUARD"); t.dyn_f16 = !off("SPRK_DYN_F16");
        t.v1_chain = !off("SPRK_V1_CHAIN");
        t.v1_one = !off("SPRK_V1_ONE");
        t.mlp_chain = !off("SPRK_MLP_CHAIN");
        t.vmm_tables = !off("SPRK_VMM_TABLES");
        t.din_tail = !off("SPRK_DIN_TAIL"); t.din_legacy = on("SPRK_DIN_LEGACY"); t.din_half = !off("SPRK_DIN_HALF");
        t.din_cols = !off("SPRK_DIN_COLS"); t.din_fused = !off("SPRK_DIN_FUSED"); t.df_xp = num("SPRK_DF_XP", 0); t.mlp_rows_many = !off("SPRK_MLP_ROWS_MANY"); t.din_fused_mb = !off("SPRK_DIN_FUSED_MB"); t.din_fused_unf = !off("SPRK_DIN_FUSED_UNF"); t.din_fused_min_t = num("SPRK_DIN_FUSED_MIN_T", 12); t.dien_fused = !off("SPRK_DIEN_FUSED");
        { const int n = num("SPRK_DIN_COLS_TS", 0); t.din_cols_ts = (n == 1 || n == 2 || n == 4) ? n : 0; }
        { const int n = num("SPRK_MANY_STREAMS", 0); t.many_streams = n < 2 ? 0 : (n > 4 ? 4 : n); }
        return t;
    }
};
inline thread_local const SprkTuning* g_finalize_tune = nullptr;   // the engine being finalized on this thread (helpers without a handle)


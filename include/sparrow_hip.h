/*
 * sparrow_hip.h -- C ABI of libsparrow_hip.so: the MI355X (gfx950) CTR-ranking forward pass
 * for SparrowRecSys's TFRecModel models.
 *
 * The reference has NO native/FFI interface for this path: every model is a stand-alone Keras
 * script and the boundary a maintainer sees is `model.predict(x)` (reference
 * TFRecModel/src/com/sparrowrecsys/offline/tensorflow/DeepFM.py:131, DIN.py:185,
 * NeuralCF.py:91) and the TF-Serving REST call of the Jetty server
 * (src/main/java/com/sparrowrecsys/online/recprocess/RecForYouProcess.java:113-138).
 * This header is therefore the NEW seam those two call into: each entry
 * point below names the reference construct it replaces.  The Python host
 * (sparrowrecsys_amd/_lib.py) binds it with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * SPRK_E* code and never throws; sprk_last_error() returns a thread-local message.
 * `ids`, `dense`, `aux`, `out`, `workspace` are DEVICE pointers owned by the caller (torch
 * tensors in the Python host); the library owns only the tables/weights it was given through
 * sprk_upload().  A finalized handle is immutable: sprk_forward*() may run concurrently on
 * distinct streams; sprk_upload()/sprk_finalize() must not race with them.
 */
#ifndef SPARROW_HIP_H
#define SPARROW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* [r6] The library is built with -fvisibility=hidden: the entry points declared between this push and its pop are its WHOLE dynamic symbol table
 * (tests/test_cabi.py compares `nm -D` with this header). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define SPRK_ABI_VERSION 2

#define SPRK_OK 0
#define SPRK_EINVAL (-1)    /* bad argument / malformed plan            */
#define SPRK_EHIP (-2)      /* a HIP runtime call failed                */
#define SPRK_ESTATE (-3)    /* wrong call order (e.g. forward before finalize) */
#define SPRK_ERANGE (-4)    /* an id was outside its table (see sprk_check_ids) */
#define SPRK_EKIND (-5)     /* forward_<model> called on a handle of another model */

#define SPRK_TILE_M 64      /* samples per workgroup tile                */
#define SPRK_MAX_SEGS 40
#define SPRK_MAX_OPS 24
#define SPRK_MAX_TAPS 8
#define SPRK_MAX_PAIRS 32
#define SPRK_MAX_BUFS 3

/* model_kind: which reference script the plan restates (checked by sprk_forward_<model>) */
enum sprk_model_kind {
    SPRK_MODEL_GENERIC = 0,
    SPRK_MODEL_EMBEDDING_MLP = 1, /* EmbeddingMLP.py:72-77  */
    SPRK_MODEL_WIDE_DEEP = 2,     /* WideNDeep.py:99-108    */
    SPRK_MODEL_NEURALCF = 3,      /* NeuralCF.py:45-70      */
    SPRK_MODEL_DEEPFM = 4,        /* DeepFM.py:91-115       */
    SPRK_MODEL_DEEPFM_V2 = 5,     /* DeepFM_v2.py:98-157    */
    SPRK_MODEL_DIN = 6,           /* DIN.py:125-169         */
    SPRK_MODEL_DIEN = 7           /* DIEN.py:114-259 (y_pred) */
};

/* Gather segments: how one sample's row of activation buffer 0 is assembled.  Together they
 * replace tf.keras.layers.DenseFeatures + tf.feature_column.{embedding,indicator,numeric,
 * crossed}_column (DeepFM.py:54-97, WideNDeep.py:72-73). */
enum sprk_seg_kind {
    SPRK_SEG_ROWS = 0,         /* embedding_column: copy table[id] (nvec float4); id<0 -> zeros */
    SPRK_SEG_SCALAR = 1,       /* indicator_column x Dense(1): one weight table[id]; id<0 -> 0   */
    SPRK_SEG_DENSE = 2,        /* numeric_column(s): copy `count` floats of the dense row       */
    SPRK_SEG_ZERO = 3,         /* zero-fill `count` floats (K padding)                           */
    SPRK_SEG_CROSS_SCALAR = 4, /* indicator(crossed_column) x Dense(1): table[hash(a,b) % vocab] */
    SPRK_SEG_CROSS_ROWS = 5,   /* embedding_column(crossed_column): row table[hash(a,b) % vocab] */
    SPRK_SEG_AUX = 6           /* copy `count` floats of the aux row (DIN pooled history)        */
};

typedef struct sprk_seg {
    int32_t kind;
    int32_t slot;       /* weight slot holding the table (ROWS/SCALAR/CROSS_*)            */
    int32_t field;      /* ids column | first dense column | first aux column             */
    int32_t field2;     /* second ids column (CROSS_*)                                    */
    int32_t row_stride; /* floats per table row (multiple of 4 for ROWS)                  */
    int32_t count;      /* ROWS/CROSS_ROWS: float4 per row; DENSE/ZERO/AUX: floats        */
    int32_t dst;        /* destination float offset inside buffer 0 (multiple of 4 for ROWS) */
    int32_t vocab;      /* table rows (bounds check) | number of hash buckets             */
} sprk_seg;

enum sprk_op_kind {
    SPRK_OP_DENSE = 0,    /* tf.keras.layers.Dense: dst = act(src @ W + b) on fp32 MFMA        */
    SPRK_OP_FM_SUMSQ = 1, /* DeepFM_v2.py:147-152: (sum_g v_g)^2 - sum_g v_g^2 over `groups`    */
    SPRK_OP_PAIR_DOT = 2  /* tf.keras.layers.Dot(axes=1) for the plan's pair list (DeepFM.py:100-103) */
};
enum sprk_act { SPRK_ACT_NONE = 0, SPRK_ACT_RELU = 1, SPRK_ACT_PRELU = 2 };

typedef struct sprk_op {
    int32_t kind;
    int32_t src_buf, src_off;
    int32_t K;           /* DENSE: input width (multiple of 4); FM_SUMSQ: width per group; PAIR_DOT: vector length */
    int32_t dst_buf, dst_off;
    int32_t N;           /* DENSE: output width (multiple of 16)                           */
    int32_t w_slot;      /* DENSE: W transposed+padded, [N][ldw] floats                    */
    int32_t ldw;         /* DENSE: floats per W^T row (multiple of 4, >= K)                */
    int32_t b_slot;      /* DENSE: bias [N]                                                */
    int32_t alpha_slot;  /* DENSE+PRELU: alpha [N]                                         */
    int32_t act;
    int32_t groups;      /* FM_SUMSQ: number of fields                                     */
    int32_t group_stride;/* FM_SUMSQ: floats between consecutive fields                    */
} sprk_op;

/* Output layer: z = head_bias + sum_t scale_t * (bias_t + sum_j w_t[j] * buf_t[off_t + j]);
 * score = sigmoid(z).  Replaces the final concatenate + Dense(1, sigmoid)
 * (DeepFM.py:111-113, DeepFM_v2.py:154-155, DIN.py:167, ...). */
typedef struct sprk_tap {
    int32_t buf, off, len;
    int32_t w_slot;      /* weights [len]; -1 -> all ones                                  */
    float scale;
    float bias;
} sprk_tap;

/* DIN activation unit + weighted sum pooling (DIN.py:132-158), run by its own kernel before
 * the tile kernel; its [B, out_cols] result is what SPRK_SEG_AUX segments read. */
typedef struct sprk_din {
    int32_t enabled;
    int32_t T;           /* history length                                                 */
    int32_t hist_col;    /* first of T consecutive ids columns                             */
    int32_t cand_col;    /* candidate movieId ids column                                   */
    int32_t table_slot;  /* shared movie Embedding table                                   */
    int32_t row_stride;  /* floats per row (= Dp, multiple of 4)                           */
    int32_t vocab;
    int32_t hidden;      /* attention hidden width (multiple of 16; reference 32)          */
    int32_t w_slot;      /* att0 kernel transposed [hidden][4*Dp] in [h-c | h | c | h*c] blocks */
    int32_t b_slot;      /* att0 bias [hidden]                                             */
    int32_t alpha_slot;  /* PReLU alpha [T][hidden]                                        */
    int32_t w2_slot;     /* att1 kernel [hidden]                                           */
    float b2;            /* att1 bias                                                      */
    /* enabled == 2: DIEN's interest-evolution stage instead (DIEN.py:163-250: Embedding mask -> GRU -> attention gate ->
     * AUGRU; its final state [B, row_stride] is what SPRK_SEG_AUX reads).  Uses T, hist_col, cand_col, table_slot,
     * row_stride, vocab, hidden (= 32) and the two fields below; w_slot .. b2 are ignored. */
    int32_t emb_dim;     /* D: Embedding / GRU / AUGRU width (10 or 16)                    */
    int32_t seq_slot;    /* packed weights, strides padded to 4 floats (Dq = pad4(D), N3 = pad4(3D)):
                          *   GRU kernel [D][N3] | recurrent kernel [D][N3] | bias [2][N3]  (z | r | h columns, reset_after)
                          *   | attention Dense(32) kernel [D][32] | bias [32] | Dense(1) kernel [32] | bias [4]
                          *   | for gate in (R_t, Z_t, H_t_next): input_w kernel [D][Dq] | bias [Dq] | hidden_w kernel [D][Dq]
                          *     | Dense_sigmoid / Dense_tanh kernel [D][Dq] | bias [Dq]
                          *   | AUGRU initial state h0 [Dq]; the whole image zero-padded to a multiple of 64 floats */
} sprk_din;

typedef struct sprk_plan {
    int32_t abi_version;               /* SPRK_ABI_VERSION */
    int32_t model_kind;
    int32_t n_id_cols;                 /* F: int32 columns per ids row   */
    int32_t n_dense;                   /* N: floats per dense row        */
    int32_t n_aux;                     /* floats per aux row (0 if none) */
    int32_t n_slots;                   /* weight slots used              */
    int32_t n_bufs;
    int32_t buf_width[SPRK_MAX_BUFS];  /* floats per sample in each LDS activation buffer */
    int32_t n_segs;
    sprk_seg segs[SPRK_MAX_SEGS];
    int32_t n_ops;
    sprk_op ops[SPRK_MAX_OPS];
    int32_t n_pairs;                   /* pair list of the PAIR_DOT op (offsets in src buffer) */
    int32_t pair_a[SPRK_MAX_PAIRS];
    int32_t pair_b[SPRK_MAX_PAIRS];
    int32_t n_taps;
    sprk_tap taps[SPRK_MAX_TAPS];
    float head_bias;
    sprk_din din;
} sprk_plan;

typedef struct sprk_engine* sprk_handle;

/* Library / device facts.  info[0]=ABI version, [1]=HIP device count (0 when no GPU is
 * visible; never an error), [2]=compute units of the current device, [3]=1 if it is gfx950. */
int sprk_runtime_info(int32_t info[4]);

/* Replaces building the Keras graph (`tf.keras.Model(inputs, output_layer)`, DeepFM.py:115):
 * validates and copies `plan`. */
int sprk_create(const sprk_plan* plan, sprk_handle* out);

/* Replaces Keras weight loading (`model.set_weights` / SavedModel restore, NeuralCF.py:97-105):
 * copies `bytes` from `src` (host OR device pointer) into library-owned device memory for
 * weight slot `slot`.  Layouts are the device layouts documented in DESIGN.md (tables padded
 * to a multiple of 4 floats per row, Dense kernels transposed). */
int sprk_upload(sprk_handle h, int32_t slot, const void* src, size_t bytes);

/* Resolves slots to device pointers; must follow the last sprk_upload. */
int sprk_finalize(sprk_handle h);

/* Bytes of caller-provided device scratch sprk_forward needs for a batch of B (0 for models
 * without a DIN stage). */
size_t sprk_workspace_bytes(sprk_handle h, int32_t B);

/* Replaces `model.predict(x)` for one device-resident batch (DeepFM.py:131):
 *   ids   [B, n_id_cols] int32 row-major, -1 = missing / out-of-vocabulary (-> zero row)
 *   dense [B, n_dense]   float32 row-major
 *   out   [B]            float32 sigmoid scores
 * Asynchronous on `stream` (a hipStream_t; NULL = default stream). */
int sprk_forward(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Replaces `model.predict(dataset)` looping over the dataset's batches (DeepFM.py:131 over the
 * tf.data pipeline of DeepFM.py:14-22): enqueues n_batches forwards of B rows each on `stream`,
 * batch i reading ids[i] / dense[i] and writing out[i] (arrays of DEVICE pointers held in HOST
 * memory).  Exactly equivalent to n_batches calls of sprk_forward; it exists so that a serving or
 * evaluation loop pays one foreign-function call per pass instead of one per batch. */
int sprk_forward_many(sprk_handle h, int32_t n_batches, const int32_t* const* ids, const float* const* dense,
                      float* const* out, int32_t B, void* workspace, size_t workspace_bytes, void* stream);

/* How sprk_forward_many enqueues its batches.  n = 0 (default): all on `stream`, strictly one after the other -- on this
 * stack even an empty kernel costs ~3.3 us per launch in such a dependent chain.  n = 2..4: independent batches alternate
 * over n library-owned helper streams forked from `stream` and joined back into it, so that one kernel's dispatch / drain
 * overlaps its neighbours' execution (config 2: 8.9 -> 6.8 us per 65 536-row batch).  Results are identical; completion is
 * still ordered on `stream`.  Models with a workspace (DIN) fan out only when `workspace_bytes` holds one 256-byte-aligned
 * slice of sprk_workspace_bytes(B) per stream. */
int sprk_set_many_streams(sprk_handle h, int32_t n);

/* How many of sprk_forward_many's batches ONE kernel launch scores (1 = a launch per batch, the default; up to 64).  Each
 * batch keeps its own ids / dense / out buffers of B rows; the launch walks the tasks of all of them, so the fixed cost of
 * a launch is spent once per n batches.  Bit-identical results.  Honoured, with 16-byte aligned buffers, by the fused DeepFM_v2
 * and embedding-rows kernels (up to 64 per launch) and by the pair-dot DeepFM, EmbeddingMLP / Wide&Deep and DIN kernels (up to
 * 16); every other case silently goes batch by batch (and takes sprk_set_many_streams into account). */
int sprk_set_many_batches(sprk_handle h, int32_t n);

/* sprk_forward_many with both knobs as ARGUMENTS of the call: `batches_per_launch` (1..64) and `helper_streams` (0, 2..4) mean
 * what sprk_set_many_batches / sprk_set_many_streams set, but nothing is stored in the handle -- a finalized handle stays
 * immutable, so threads sharing one can each use their own settings (round 2's setters changed state under a handle the
 * header called immutable; they remain as the DEFAULTS sprk_forward_many uses and must not be called while another thread is
 * inside a forward).  With helper_streams >= 2 the call uses the handle's helper streams and fork / join events: that form is
 * not re-entrant on one handle; helper_streams = 0 is.  What `model.predict(dataset)` (DeepFM.py:131) replays per batch. */
int sprk_forward_many_opts(sprk_handle h, int32_t n_batches, const int32_t* const* ids, const float* const* dense,
                           float* const* out, int32_t B, void* workspace, size_t workspace_bytes, void* stream,
                           int32_t batches_per_launch, int32_t helper_streams);

/* Per-model entry points: identical to sprk_forward but fail with
 * SPRK_EKIND unless the handle was created from that model's plan. */
int sprk_forward_embedding_mlp(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws, size_t ws_bytes, void* stream);
int sprk_forward_widedeep(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws, size_t ws_bytes, void* stream);
int sprk_forward_neuralcf(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws, size_t ws_bytes, void* stream);
int sprk_forward_deepfm(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws, size_t ws_bytes, void* stream);
int sprk_forward_deepfm_v2(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws, size_t ws_bytes, void* stream);
int sprk_forward_din(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws, size_t ws_bytes, void* stream);
int sprk_forward_dien(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws, size_t ws_bytes, void* stream);

/* DIN stage alone (DIN.py:132-158): pooled [B, row_stride] and, if att != NULL, the attention
 * weights att [B, T].  For a DIEN handle: the final AUGRU state [B, row_stride]; att must be NULL. */
int sprk_din_pool(sprk_handle h, const int32_t* ids, float* pooled, float* att, int32_t B, void* stream);

/* Replaces TF's assert_greater_or_equal_0 / assert_less_than_num_buckets
 * (categorical_column_with_identity, DeepFM.py:54,59): synchronises `stream` and returns
 * SPRK_ERANGE if any forward since the last call saw an id >= its table size or < -1. */
int sprk_check_ids(sprk_handle h, void* stream);

/* Which kernels a finalized handle dispatches to (no reference counterpart; Keras would answer `model.summary()`).  Writes a
 * NUL-terminated "key=value;..." line into buf: kernel = the fused kernel instantiation that scores a batch, or k_tile_forward
 * when no fused kernel matched the plan and the generic plan interpreter runs it (fused=0: several times slower -- a shape that
 * silently fell off the fast path is visible here); stage = the history stage of DIN / DIEN handles (k_din_attn, k_din_pool,
 * k_dien_seq) or empty; uploaded_bytes / derived_bytes = device memory of the uploaded slots and of the tables derived from them
 * at finalize (folded rows, split halfs, per-id terms); first_dense_fold = embedding columns folded into the first Dense layer.
 * SPRK_EINVAL when the buffer is too small (256 bytes always suffice). */
int sprk_describe(sprk_handle h, char* buf, size_t buf_bytes);

void sprk_destroy(sprk_handle h);

/* ---- stand-alone operators (same kernels' building blocks, for parity tests / reuse) ---- */

/* tf.keras.layers.Embedding / embedding_column row gather (DIN.py:132-136, DeepFM.py:55):
 * out[b, 0:D] = table[ids[b], 0:D]; ids[b] < 0 -> zeros.  table is [V, row_stride] floats,
 * out is [B, D] floats; D and row_stride multiples of 4.  Bit-exact copy. */
int sprk_embedding_gather(const float* table, int32_t V, int32_t D, int32_t row_stride,
                          const int32_t* ids, int32_t B, float* out, void* stream);

/* tf.feature_column.crossed_column([a, b], num_buckets) hashing (WideNDeep.py:72-73):
 * out[b] = FingerprintCat64(FingerprintCat64(0xDECAFCAFFE, a[b]), b[b]) mod num_buckets. */
int sprk_cross_hash(const int32_t* a, const int32_t* b, int32_t B, int64_t num_buckets,
                    int64_t* out, void* stream);

/* The reference's "emb" ranker: RecForYouProcess.java:69-92 (ranker, case "emb") with
 * calculateEmbSimilarScore :100-105, SimilarMovieProcess.java:121-136,167-172, Embedding.calculateSimilarity
 * (Embedding.java:33-47).  For query u (a user's embedding, or a movie's) and its C candidate movies
 *   scores[u][c] = dot / (sqrt(n1) * sqrt(n2)), float products summed into doubles in index order exactly as the Java
 *                  loop does (bit-exact doubles; an all-zero vector gives NaN as in Java);
 *                  -1.0 if the query has no embedding (query_has[u] == 0), cand[u][c] is outside [0, n_items) (a Movie
 *                  unknown to the table) or that item has none (item_has[id] == 0) -- Embedding.java:34-37.
 *   order[u][0..C) (optional, C <= 4096): candidate POSITIONS in the order of
 *                  `sorted(Map.Entry.comparingByValue(Comparator.reverseOrder()))`: descending in Double.compareTo
 *                  order (NaN first, 0.0 before -0.0); equal scores stay in candidate order (Java: unspecified).
 * All pointers are device memory; item_emb [n_items][item_stride], query_emb [n_queries][query_stride] floats (first D
 * used), cand [n_queries][C]; item_has / query_has may be NULL (= all present). */
int sprk_emb_rank(const float* item_emb, const uint8_t* item_has, int32_t n_items, int32_t D, int32_t item_stride,
                  const float* query_emb, const uint8_t* query_has, int32_t n_queries, int32_t query_stride,
                  const int32_t* cand, int32_t C, double* scores, int32_t* order, void* stream);

/* ---- host ingest (no GPU involved): the step before the path ----
 * Replaces `tf.data.experimental.make_csv_dataset(..., na_value="0", ignore_errors=True)` of the reference's
 * get_dataset (DeepFM.py:14-22) plus the feature-column id resolution (DeepFM.py:54-76) for a CSV text held in
 * memory: header line + rows -> the two packed arrays sprk_forward reads.  Identity columns: empty -> 0, values
 * outside [0, vocab) -> SPRK_ERANGE (TF: assert_less_than_num_buckets); genre columns: position in the 19-entry
 * vocabulary of DeepFM.py:64-66, anything else (empty, unknown) -> -1; dense columns: empty -> 0.0.  Rows whose
 * field count differs from the header's are skipped (ignore_errors).  `ids_out` is [max_rows, n_id] int32,
 * `dense_out` [max_rows, n_dense] float32 (HOST memory); *rows_out receives the number of rows packed. */
typedef struct sprk_csv_col {
    const char* name;    /* CSV header name (reference schema key)                  */
    int32_t kind;        /* 0 = categorical_column_with_identity, 1 = genre vocabulary */
    int32_t vocab;       /* buckets / vocabulary size                                */
} sprk_csv_col;
int sprk_pack_csv(const char* text, size_t len, const sprk_csv_col* id_cols, int32_t n_id,
                  const char* const* dense_names, int32_t n_dense, int32_t max_rows,
                  int32_t* ids_out, float* dense_out, int32_t* rows_out);
/* The same on n_threads host threads (clamped to [1, 256]; the body is cut at line boundaries, chunks are parsed
 * concurrently and stitched in file order): identical outputs, identical first error, for any thread count. */
int sprk_pack_csv_mt(const char* text, size_t len, const sprk_csv_col* id_cols, int32_t n_id,
                     const char* const* dense_names, int32_t n_dense, int32_t max_rows, int32_t n_threads,
                     int32_t* ids_out, float* dense_out, int32_t* rows_out);
/* The same packing ON THE DEVICE ("GPU-side tokenizer"): `text_dev` is the CSV text in device
 * memory (16-byte aligned; e.g. the file read straight into a pinned buffer and copied once), `ids_dev` / `dense_dev` are
 * DEVICE arrays [max_rows, n_id] / [max_rows, n_dense].  Same rules and the same bits as sprk_pack_csv for every field that is
 * empty or a plain decimal of at most 15 significant digits with a decimal exponent within +-22 (strtod's exact fast path) --
 * every value the reference's sample files hold (quoted fields are split as the host tokenizer splits them; the files spell
 * an empty string "").  It never guesses: a numeric field outside that shape ("inf", hex, blanks, 16+ digits, text, an escaped
 * quote) fails with SPRK_EKIND and names the first such row; ids outside
 * their bucket range fail with SPRK_ERANGE like the host tokenizer (first bad row in file order).  Synchronises `stream`
 * (the row count comes back to the host).  Device scratch (8 bytes per line + 4 per 4-KB chunk of text) is allocated on first
 * use, grows to the largest text seen and is kept for the calling thread's lifetime. */
int sprk_pack_csv_device(const char* text_dev, size_t len, const sprk_csv_col* id_cols, int32_t n_id,
                         const char* const* dense_names, int32_t n_dense, int32_t max_rows,
                         int32_t* ids_dev, float* dense_dev, int32_t* rows_out, void* stream);
/* Which kernels the calling thread's last sprk_pack_csv_device ran the text through (diagnostics / tests): 1 = the optimistic
 * pass alone (every line of the text is a row: line i is output row i - 1), 2 = the exact keep -> scan -> parse sequence (some
 * line is empty or has another field count than the header: ignore_errors=True drops it), -1 = no call yet / no data line / the
 * call failed early.  The packed arrays are the same bits whichever ran. */
int sprk_csv_last_path(void);

/* ---- multi-GPU: the path's one collective (the reference has no distributed path) ----
 * Batch rows are sharded over one process per GPU, tables and weights replicated; every rank ends with all scores through ONE
 * all-gather of the per-rank score slices over RCCL / xGMI, enqueued on the caller's HIP stream (no host synchronisation).
 * RCCL is loaded at run time on first use.  Rank 0 calls sprk_comm_unique_id and hands the 128 bytes to the other ranks through
 * any host channel (the Python host uses the process group's store); every rank then calls sprk_comm_create (collective).
 * gathered = [world][count] floats, rank r's slice at offset r * count; in-place (local == gathered + rank * count) is allowed. */
#define SPRK_COMM_ID_BYTES 128
typedef struct sprk_comm_s* sprk_comm;
int sprk_comm_unique_id(uint8_t id[SPRK_COMM_ID_BYTES]);
int sprk_comm_create(const uint8_t id[SPRK_COMM_ID_BYTES], int32_t rank, int32_t world, sprk_comm* out);
int sprk_comm_allgather_scores(sprk_comm c, const float* local, float* gathered, size_t count, void* stream);
void sprk_comm_destroy(sprk_comm c);

/* The same exchange as direct peer writes over xGMI: every rank stores its slice straight into all peers'
 * receive buffers (one step on the point-to-point mesh instead of a ring's world-1 dependent hops), then waits for the peers'
 * arrival flags -- two kernels on the caller's stream, no host synchronisation, no RCCL.  Setup: every rank calls
 * sprk_peer_create (allocates its receive buffer [2 parities][world][slot_floats] + flags and returns the 64-byte IPC handle),
 * the handles travel through any host channel, every rank calls sprk_peer_connect with all of them (rank order).
 * sprk_peer_allgather_scores: local [count <= slot_floats] -> *gathered = this rank's receive buffer for this exchange,
 * [world][slot_floats] floats (rank r's slice at r * slot_floats), valid for work enqueued on `stream` until the exchange after
 * the next one on this communicator; one exchange in flight per communicator.  A slice that does not arrive within the deadline
 * (2 s; SPRK_PEER_TIMEOUT_MS) raises a flag that sprk_peer_check (synchronises the stream) reports as SPRK_EHIP. */
#define SPRK_PEER_HANDLE_BYTES 64
typedef struct sprk_peer_s* sprk_peer;
int sprk_peer_create(int32_t rank, int32_t world, size_t slot_floats, uint8_t handle_out[SPRK_PEER_HANDLE_BYTES], sprk_peer* out);
int sprk_peer_connect(sprk_peer c, const uint8_t* handles /* [world][SPRK_PEER_HANDLE_BYTES] */);
int sprk_peer_allgather_scores(sprk_peer c, const float* local, size_t count, const float** gathered, void* stream);
int sprk_peer_check(sprk_peer c, void* stream);
const char* sprk_peer_memory_kind(sprk_peer c);   /* "uncached" | "fine-grained" | "default": how the receive buffer was allocated */
void sprk_peer_destroy(sprk_peer c);

/* [r4] BASELINE.json configs[3]: "DeepFM emb_dim=64, 138 k-movie x 27 M-row synthetic table, ROW-SHARDED across 8 x MI355X" (the
 * reference holds its tables as ordinary tf.Variables of one process -- embedding_column, DeepFM.py:54-60 -- so there is no
 * reference interface to mirror; SURVEY.md section 8(e) names the variant).  A sprk_vtable is an embedding table of rows_total rows
 * whose rows [r S, (r + 1) S) live in rank r's HBM and which EVERY rank sees as one contiguous device array: HIP virtual memory maps
 * the peers' allocations into one reserved range, so the fused kernels gather table[id] unchanged and a row another GPU owns is
 * loaded over the xGMI link between the two -- no collective, no staging pass (the textbook form is an all-to-all of ids and one
 * of rows in front of every forward).  Set-up, once: every rank calls sprk_vtable_create (collective in effect: same geometry
 * everywhere; allocates and zero-fills its own shard), sprk_vtable_export (a POSIX file descriptor of that shard, to be sent to the
 * other ranks of the node with SCM_RIGHTS -- sparrowrecsys_amd/dist.py ShardedTable does it over Unix sockets) and, for each peer's
 * descriptor, sprk_vtable_import.  sprk_vtable_info: the base of the whole table (rank r's rows start at r * shard_rows), rows per
 * rank (ceil(rows_total / world) rounded up so that shards are whole allocation granules), shards mapped so far.  row_bytes must be a
 * multiple of 16.  The table outlives every engine that was given it through sprk_upload_external. */
typedef struct sprk_vtable_s* sprk_vtable;
int sprk_vtable_create(int64_t rows_total, int32_t row_bytes, int32_t world, int32_t rank, sprk_vtable* out);
int sprk_vtable_export(sprk_vtable v, int32_t* fd_out);
int sprk_vtable_import(sprk_vtable v, int32_t peer_rank, int32_t fd);
int sprk_vtable_info(sprk_vtable v, void** base, int64_t* shard_rows, int32_t* ranks_mapped);
void sprk_vtable_destroy(sprk_vtable v);

/* [r4] sprk_upload without the copy: the slot READS `bytes` of caller-owned device memory at dev_ptr (16-byte aligned, in the
 * slot's device layout -- a table as [vocab + 1][Dp] floats with its last row zero) for as long as the handle lives; the engine
 * never frees it.  For tables that are already where they should be: a sprk_vtable, or a 6.9 GB tensor that sprk_upload would
 * duplicate (embedding_column's variable, DeepFM.py:55,60). */
int sprk_upload_external(sprk_handle h, int32_t slot, const void* dev_ptr, size_t bytes);

const char* sprk_last_error(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SPARROW_HIP_H */

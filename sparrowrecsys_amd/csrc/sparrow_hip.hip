// sparrow_hip.hip -- MI355X (gfx950 / CDNA4) CTR-ranking forward engine behind include/sparrow_hip.h.
//
// Two hand-written kernels carry the hot path (DESIGN.md has the full data-layout story):
//
//   k_tile_forward  one 256-thread workgroup per tile of 64 samples.  Phase 1 gathers every sparse
//                   slot's embedding row (16-B lanes, 64-B..256-B rows), the first-order weights and
//                   the numeric columns of the tile into an LDS activation buffer; phase 2 runs the
//                   model's small op list over LDS (fp32 MFMA 16x16x4 Dense layers with the weights
//                   as the A operand so every epilogue store is a 16-B ds_write, FM sum-of-squares,
//                   pairwise dots); phase 3 applies the output layer + sigmoid and stores one score
//                   per sample.  Activations never leave the CU; HBM traffic is the algorithmic
//                   minimum (ids + rows + numerics in, scores out).
//   k_din_pool      DIN's activation unit + weighted sum pooling: the T history rows of a few samples
//                   are gathered ONCE into LDS, the [h-c, h, c, h*c] -> Dense(hidden) contraction runs
//                   on fp32 MFMA with the B operand built on the fly from LDS, PReLU/Dense(1)/sigmoid
//                   finish in registers + two cross-lane adds, and the pooled vector is reduced from
//                   the same LDS rows.
//
// Reference constructs each piece replaces are cited in include/sparrow_hip.h.
//
// The library's MAIN translation unit: the host side (engine, plan validation, per-graph set-up, the C ABI) and the light kernels.  The heavy
// kernel templates are compiled by the kernel-family units tu_1.hip .. tu_6.hip (tu_kernels.h, tu_instances.h); -DSPRK_SINGLE_TU folds
// everything back into this one unit (the ISA scripts).  The pieces share one named namespace for the kernels (sprk_dev) and one
// anonymous namespace for the host helpers, and are meaningful only in this order:
#include "tu_kernels.h"              // system headers, include/sparrow_hip.h, host_common.h .. k_operators.h, tu_instances.h
#include "host_engine.h"             // struct sprk_engine: everything a finalized handle owns
#include "host_plan.h"               // plan validation and small host helpers
#include "host_setup_v2.h"           // DeepFM_v2: k_deepfm_v2_chain / _joint / _joint1 dispatch tables, plan matcher, fold + joint-table set-up
#include "host_setup_rows.h"         // k_rows_chain (literal DeepFM_v2, NeuralCF): dispatch table and set-up
#include "host_setup_common.h"       // k_din_attn dispatch table, the interpreter's first-Dense fold, the dynamic-range guard, split-f16 fragment packing
#include "host_setup_pairs.h"        // pair-dot DeepFM: k_deepfm_pairs / _pairs1 dispatch table, plan matcher and set-up
#include "host_setup_mlp.h"          // EmbeddingMLP / Wide&Deep: k_mlp_rows set-up
#include "host_setup_din_tail.h"     // DIN / DIEN tail: k_din_tail dispatch table and set-up -- closes the host helpers' anonymous namespace
#include "api_engine.h"              // C ABI: sprk_last_error .. sprk_create / sprk_upload / sprk_finalize / sprk_workspace_bytes
#include "api_forward.h"             // C ABI: sprk_din_pool, sprk_forward, sprk_forward_many, sprk_describe, sprk_check_ids, sprk_destroy, operators, emb ranker
#include "api_ingest.h"              // C ABI: CSV ingest on the host (sprk_pack_csv[_mt]) and on the device (sprk_pack_csv_device), sprk_cross_hash
#include "api_comm.h"
#include "api_vtable.h"              // C ABI: a row-sharded table every rank sees as one (sprk_vtable_*: HIP virtual memory over xGMI), sprk_upload_external                // C ABI: the score all-gather over RCCL (sprk_comm_*) and as direct peer writes (sprk_peer_*)

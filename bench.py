#!/usr/bin/env python
"""bench.py -- CTR samples/sec of the HIP forward on synthetic MovieLens-20M-shaped batches.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" = one pass of the hot path over one batch (ids + dense already resident in HBM -> scores in HBM); at N>1
every rank scores its own B-row shard (weak scaling: global batch N*B) and its score slices are all-gathered over RCCL,
--gather-group steps per collective on a second stream (every step's scores have been exchanged on every rank when a
timed region ends).  Rank 0 prints ONE JSON line.

How the K steps are timed.  A 65 536-row forward takes single-digit microseconds, so K = 20 steps are ~100 us of GPU work
-- less than one host synchronisation.  The timed REGION is therefore R back-to-back repetitions of the K-step block,
R chosen (after the warm-up) so that a region lasts >= --min-region-ms (50 ms); a region is bracketed by barrier +
torch.cuda.synchronize() on both sides and timed with the host clock (max over ranks); --regions (5) regions are timed
and `ms_per_step` = median region / (R*K).  `config.repeats` = R.  The warm-up runs W steps AND at least --settle-ms of
work, so clocks have settled whatever W is.

--gpus N without a launcher (no RANK in the environment) starts the N ranks itself, one process per GPU.

Workloads (BASELINE.json configs):
  deepfm_v2_c2 (default) configs[1]: DeepFM (sum-of-squares FM cross, DeepFM_v2 graph), 6 sparse fields, emb_dim 16,
               projection width 16, B = 65 536 per GPU
  deepfm_c2    the pairwise-dot DeepFM graph (DeepFM.py) on the same fields
  din_c3       configs[2]: DIN, hist_len 50, emb_dim 32, B = 32 768 per GPU
  deepfm_v2_c4 configs[3], one GPU's replica: DeepFM_v2 graph, emb_dim 64, tables of 138 493 users x 27 M rows (6.9 GB), B = 65 536
  deepfm_c4    configs[3] for the pairwise-dot graph: real 256-byte row gathers out of the 6.9 GB table
  widedeep_c5  configs[4], one GPU's share: Wide&Deep with the 10 M-bucket x 32 hashed cross table, B = 131 072
  dien_ref     DIEN.py as written (hist_len 5, emb_dim 10): sequence stage k_dien_seq + the DIN tail
  deepfm_v2_ref / neuralcf_ref   the reference's own literal shapes (DeepFM_v2.py: 4 fields, emb_dim 10, Dense(64)
               projections; NeuralCF.py: 2 fields, emb_dim 10, 20->10->10->1) on MovieLens-20M-sized vocabularies
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12   # FLOP/s, fp32-input MFMA (= the fp32 vector rate on gfx950)
MFMA_F16_PEAK = 2.5e15     # FLOP/s, dense f16/bf16 MFMA
CONFIG4_ROWS = 27_000_000  # BASELINE configs[3]: "138 k-movie x 27 M-row synthetic table"
HBM_RESIDENT_BATCHES = 32      # distinct id batches the roofline_hbm_resident loop cycles (row working set 805 MB >> 256 MB MALL)
STRICT_CYCLE = 32              # input batches the strict one-batch-per-launch loop cycles through (see main())


SHARD_TABLES = False
_SHARDED_TABLES = []                       # (kept alive for the life of the process: engines read them in place)


def _device_table(V, D, seed, std, shard=False):
    """[V, D] float32 table drawn ON the device (a 27 M x 64 table is 6.9 GB: numpy would need minutes and as much host
    memory).  Truncated at 2 sigma like tf's truncated_normal initialiser.  shard: the table ROW-SHARDED over the ranks
    (--shard-tables; sparrowrecsys_amd.dist.ShardedTable: every rank draws only its own rows and maps the peers' shards into one
    virtual range) -- BASELINE config 4 read literally; handed to the model without a copy."""
    import torch
    g = torch.Generator(device="cuda")
    if shard:
        import torch.distributed as dist
        from sparrowrecsys_amd.dist import ShardedTable
        st = ShardedTable(V, D)
        g.manual_seed(seed * 1009 + st.rank)

        def draw(lo, hi):
            t = torch.empty((hi - lo, D), dtype=torch.float32, device="cuda")
            return t.normal_(0.0, 1.0, generator=g).clamp_(-2.0, 2.0).mul_(std)
        st.fill_local(draw)
        torch.cuda.synchronize()
        if dist.is_available() and dist.is_initialized():
            dist.barrier()                   # every shard is filled before any rank folds or gathers from it
        _SHARDED_TABLES.append(st)
        return st.table()
    g.manual_seed(seed)
    t = torch.empty((V, D), dtype=torch.float32, device="cuda")
    t.normal_(0.0, 1.0, generator=g).clamp_(-2.0, 2.0).mul_(std)
    return t


def build_workload(name, B, dist_name, seed_offset=0, big_vocab=0, NB=8):
    """NB: distinct input batches (own ids / dense / score buffers each) the steps cycle through.
    Returns (model, feats, desc, roof): roof = the dominant kernel, its bound, its algorithmic bytes / flops per sample."""
    from sparrowrecsys_amd import models as M, synthetic as SY
    from sparrowrecsys_amd.schema import N_GENRES
    env = os.environ.get

    def synth(fields):
        return [SY.synth_fields(B, fields, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name) for i in range(NB)]

    if name in ("deepfm_v2_c2", "deepfm_c2"):
        F, D = 6, 16
        fields = SY.CONFIG2_FIELDS
        if big_vocab:
            # same graph, the three identity fields' tables blown up past the 256 MB Infinity Cache: every row gather is a
            # real HBM access (config 4's "true HBM-resident gather" at config 2's widths)
            fields = [(k, kind, big_vocab if kind == "id" else v) for k, kind, v in fields]
        bytes_per_sample = F * 4 + F * D * 4 + F * 4 + 7 * 4 + 4
        if name == "deepfm_v2_c2":
            model = M.DeepFMv2(seed=101, emb_dim=D, fields=fields, proj_dim=16)
            desc = "DeepFM sum-of-squares FM (DeepFM_v2 graph), F=6 sparse fields, emb_dim=16, proj=16, deep 32-16"
            if big_vocab:
                desc += ", identity tables of %d rows each" % big_vocab
            kernel = "k_rows_chain" if (env("SPRK_V2_JOINT") == "0" or env("SPRK_V2_FOLD") == "0") else "k_deepfm_v2_joint"
        else:
            model = M.DeepFM(seed=101, emb_dim=D, fields=fields, pairs=SY.CONFIG2_PAIRS)
            desc = ("DeepFM pairwise-dot FM (DeepFM graph), F=6 sparse fields, emb_dim=16, 8 pairs, deep 64-64; the deep part reads its OWN "
                    "movieId / userId tables as in the reference (DeepFM.py:106: a second DenseFeatures layer = second variables)")
            kernel = "k_tile_forward" if env("SPRK_V1_CHAIN") == "0" else "k_deepfm_pairs"
            bytes_per_sample += 2 * D * 4                     # the deep part's two rows
        if env("SPRK_FORCE_INTERPRETER") == "1":
            kernel = "k_tile_forward"
        model._bench_fields = fields
        # fused kernel: ids + embedding rows + first-order weights + numerics in, one score out
        roof = {"bound": "hbm", "kernel": kernel, "bytes_per_sample": bytes_per_sample}
        if name == "deepfm_c2":
            roof["mfma_reference_flops"] = 2 * (39 * 64 + 64 * 64 + 64)
        return model, synth(fields), desc, roof
    if name in ("deepfm_v2_c4", "deepfm_c4"):
        # BASELINE configs[3]: emb_dim 64, userId (138 493 users) and a 27 M-row item table (6.9 GB, HBM-resident), plus the two
        # genre fields of the reference's DeepFM.py; tables drawn on the device
        D = 64
        fields = [("movieId", "id", CONFIG4_ROWS), ("userId", "id", SY.ML20M_USER_IDS),
                  ("userGenre1", "genre", N_GENRES), ("movieGenre1", "genre", N_GENRES)]
        cls = M.DeepFMv2 if name == "deepfm_v2_c4" else M.DeepFM
        kw = dict(proj_dim=16) if name == "deepfm_v2_c4" else dict(pairs=None)
        small = cls(seed=107, emb_dim=D, fields=[(k, kind, min(v, 1024)) for k, kind, v in fields], **kw)
        w = dict(small.weights)
        for i, (k, kind, v) in enumerate(fields):
            if kind == "id":
                sh = SHARD_TABLES and v == CONFIG4_ROWS                 # (--shard-tables: the 27 M-row item table; userId's 138 k rows stay replicated)
                w["emb/" + k] = _device_table(v, D, 1000 + i, 1.0 / math.sqrt(D), shard=sh)
                if name == "deepfm_c4":                           # the deep part's own table of the same key (DeepFM.py:106)
                    w["deep_emb/" + k] = _device_table(v, D, 2000 + i, 1.0 / math.sqrt(D), shard=sh)
        rng = np.random.default_rng(7)
        fo_key = "fo_cat/kernel" if name == "deepfm_v2_c4" else "head/kernel"
        model = cls(weights=_resize_first_order(small, w, fields, fo_key, rng), emb_dim=D, fields=fields, **kw)
        model._bench_fields = fields
        F = 4
        if name == "deepfm_v2_c4":
            desc = "DeepFM_v2 graph, emb_dim=64, 4 fields: movieId 27 M rows (6.9 GB table), userId 138 494, two genre fields; proj=16, deep 32-16"
            # folded rows: what the kernel gathers per big field is the 64-B projected row + its scalar, not the 256-B embedding row
            roof = {"bound": "hbm", "kernel": "k_deepfm_v2_joint", "bytes_per_sample": F * 4 + 2 * (64 + 4) + 2 * 4 + 7 * 4 + 4,
                    "reference_bytes_per_sample": F * (4 + D * 4 + 4) + D * 4}
        else:
            desc = "DeepFM pairwise-dot graph, emb_dim=64, 4 fields: movieId 27 M rows (6.9 GB table), userId 138 494, two genre fields; 4 pairs, deep 64-64"
            roof = {"bound": "hbm", "kernel": "k_deepfm_pairs", "bytes_per_sample": F * (4 + D * 4 + 4) + 2 * D * 4 + 7 * 4 + 4}
        return model, synth(fields), desc, roof
    if name == "deepfm_v2_ref":
        # DeepFM_v2.py as written: 4 fields, emb_dim 10, Dense(64) projections, deep 32-16 -- on ML-20M-sized vocabularies
        fields = [("movieId", "id", SY.ML20M_MOVIE_IDS), ("userId", "id", SY.ML20M_USER_IDS),
                  ("userGenre1", "genre", N_GENRES), ("movieGenre1", "genre", N_GENRES)]
        model = M.DeepFMv2(seed=109, emb_dim=10, fields=fields, order=["movieGenre1", "movieId", "userGenre1", "userId"], proj_dim=64)
        model._bench_fields = fields
        desc = "DeepFM_v2.py literal: 4 fields, emb_dim=10, Dense(64) projections, deep 32-16"
        roof = {"bound": "hbm", "kernel": "?", "bytes_per_sample": 4 * 4 + 2 * (64 * 4 + 4) + 2 * 4 + 7 * 4 + 4,
                "reference_bytes_per_sample": 4 * (4 + 40 + 4) + 7 * 4 + 4}
        return model, synth(fields), desc, roof
    if name == "neuralcf_ref":
        model = M.NeuralCF(seed=111, emb_dim=10, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
        desc = "NeuralCF.py literal: movieId/userId emb_dim=10, 20->10->10->1"
        fields = [("movieId", "id", SY.ML20M_MOVIE_IDS), ("userId", "id", SY.ML20M_USER_IDS)]
        model._bench_fields = fields
        roof = {"bound": "hbm", "kernel": "?", "bytes_per_sample": 2 * 4 + 2 * 40 + 4}
        return model, synth(fields), desc, roof
    if name == "deepfm_ref":
        # DeepFM.py as written (4 fields, emb_dim 10, 4 pair dots, deep 64-64, the deep part's own movieId / userId tables) on
        # MovieLens-20M-sized vocabularies
        fields = [("movieId", "id", SY.ML20M_MOVIE_IDS), ("userId", "id", SY.ML20M_USER_IDS),
                  ("userGenre1", "genre", N_GENRES), ("movieGenre1", "genre", N_GENRES)]
        model = M.DeepFM(seed=113, emb_dim=10, fields=fields)
        model._bench_fields = fields
        desc = "DeepFM.py literal: 4 fields, emb_dim=10, 4 pair dots, deep 64-64, own deep tables"
        roof = {"bound": "hbm", "kernel": "?", "bytes_per_sample": 4 * 4 + 4 * 40 + 2 * 40 + 4 * 4 + 7 * 4 + 4,
                "mfma_reference_flops": 2 * (27 * 64 + 64 * 64 + 64)}
        return model, synth(fields), desc, roof
    if name == "embedding_mlp_ref":
        model = M.EmbeddingMLP(seed=115, emb_dim=10, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
        desc = "EmbeddingMLP.py literal: 10 embedding columns emb_dim=10 + 7 numerics, 107->128->128->1"
        feats = [SY.synth_embedding_mlp(B, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name)
                 for i in range(NB)]
        roof = {"bound": "hbm", "kernel": "?", "bytes_per_sample": 10 * 4 + 10 * 40 + 7 * 4 + 4,
                "mfma_reference_flops": 2 * (107 * 128 + 128 * 128 + 128)}
        return model, feats, desc, roof
    if name in ("din_c3", "din_ref"):
        T, D = (50, 32) if name == "din_c3" else (5, 10)         # din_ref: DIN.py as written (RECENT_MOVIES = 5, EMBEDDING_SIZE = 10)
        model = M.DIN(seed=103, emb_dim=D, hist_len=T, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
        desc = ("DIN, hist_len=50, emb_dim=32, attention 128->32->1, tail 167->128->64->1" if name == "din_c3" else
                "DIN.py literal: hist_len=5, emb_dim=10, attention 40->32->1, tail 57->128->64->1")
        feats = [SY.synth_din(B, T, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name) for i in range(NB)]
        flops = T * (2 * 4 * D * 32 + 2 * 32 + 3 * 32)            # the reference's count: K = 4D per (b,t)
        # what k_din_attn issues on the matrix pipe: K = D per (b,t) after folding the c-only and h-only
        # blocks (A_b = W12 + W4 diag(c)), over whole 16-row groups (T=50 -> 64 columns), 3 split-f16 products
        executed = ((T + 15) // 16) * 16 * 2 * D * 32
        legacy = env("SPRK_DIN_LEGACY") == "1"
        att_kernel = "k_din_pool" if (legacy or env("SPRK_DIN_COLS") == "0" or env("SPRK_DIN_HALF") == "0") else "k_din_attn_cols"
        roof = {"bound": "mfma", "kernel": att_kernel, "hist_len": T, "flops_per_sample": flops,
                "executed_flops_per_sample": flops if legacy else executed,
                "bytes_per_sample": (T + 1) * 4 + (T + 1) * D * 4 + D * 4,
                "tail_reference_flops": 2 * ((7 + 5 * D) * 128 + 128 * 64 + 64),
                "tail_bytes_per_sample": 3 * D * 4 + 7 * 4 + 3 * 4 + 4}
        return model, feats, desc, roof
    if name == "dien_ref":
        # DIEN.py as written (RECENT_MOVIES = 5, EMBEDDING_SIZE = 10): GRU -> attention gate -> AUGRU (k_dien_seq, one lane per
        # sample) -> the DIN tail; the sequence stage is priced like DIN's pooling: ids + (T + 1) rows + the pooled vector
        T, D = 5, 10
        model = M.DIEN(seed=117, emb_dim=D, hist_len=T, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
        desc = "DIEN.py literal: hist_len=5, emb_dim=10, GRU(10) -> attention 10->32->1 -> AUGRU, tail 57->128->64->1"
        feats = [SY.synth_din(B, T, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name) for i in range(NB)]
        flops = T * (2 * 2 * D * 3 * D + 2 * D * 32 + 2 * 32 + 3 * (2 * 3 * D * D))      # GRU (x, h) + gate MLP + three AUGRU gates of 3 Dense each
        roof = {"bound": "mfma", "kernel": "k_dien_seq", "hist_len": T, "flops_per_sample": flops, "executed_flops_per_sample": flops,
                "bytes_per_sample": (T + 1) * 4 + (T + 1) * D * 4 + D * 4,
                "tail_reference_flops": 2 * ((7 + 5 * D) * 128 + 128 * 64 + 64),
                "tail_bytes_per_sample": 3 * D * 4 + 7 * 4 + 3 * 4 + 4}
        return model, feats, desc, roof
    if name == "widedeep_c5":
        # BASELINE configs[4], one GPU's share: Wide&Deep, hashed cross (movieId x userRatedMovie1) computed on device into a
        # 10 M-bucket x 32 embedding table (1.28 GB), emb_dim 32, deep 128-128
        D, CB = 32, 10_000_000
        model = M.WideNDeep(seed=105, emb_dim=D, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS,
                            cross_buckets=CB, cross_dim=D)
        desc = "Wide&Deep, 10 embedding columns emb_dim=32 + hashed cross 10M buckets x 32 (on-device FingerprintCat64), deep 128-128"
        feats = [SY.synth_embedding_mlp(B, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name,
                                        rated_vocab=SY.ML20M_MOVIE_IDS) for i in range(NB)]
        # 8 B ids + 128 B cross row per sample for the wide part; + the deep part's 11 ids, 10 rows, numerics, score
        roof = {"bound": "hbm", "kernel": "k_tile_forward" if env("SPRK_MLP_CHAIN") == "0" else "k_mlp_rows",
                "bytes_per_sample": 8 + D * 4 + 9 * 4 + 10 * D * 4 + 7 * 4 + 4,
                "mfma_reference_flops": 2 * (327 * 128 + 128 * 128 + 128)}
        return model, feats, desc, roof
    raise SystemExit("unknown workload %r" % name)


INFINITY_CACHE = 256e6         # bytes of MALL in front of HBM (MI355X_MICROARCH.md)
# [r6, VERDICT r05 item 2] The id columns whose rows come from a table LARGER than the Infinity Cache, per workload, and the bytes a sample
# gathers through them: deepfm_c4 = the 27 M-row item table's 256-byte row twice (FM table + the deep part's own, DeepFM.py:106),
# deepfm_v2_c4 = its folded 128-byte row, widedeep_c5 = the 128-byte row of the 10 M-bucket cross table (bucket = hash(movieId,
# userRatedMovie1): both columns re-drawn).  Everything else these graphs gather (138 k-user tables of 18 - 70 MB, genre rows in LDS) stays in
# the cache whatever the inputs are -- that share is reported, not labelled HBM.
HBM_SIDE = {"deepfm_c4": (("movieId",), 512), "deepfm_v2_c4": (("movieId",), 128), "widedeep_c5": (("movieId", "userRatedMovie1"), 128)}


def hbm_cycle(name, model, batches, B, cap=128):
    """Configs 4 / 5 with enough DISTINCT input batches that the big-table lines touched between two visits of a batch exceed 4 x the Infinity
    Cache: the host-drawn batches first (the oracle checks those), then copies of them with the big-table id columns re-drawn on the device.
    Returns (batches, info)."""
    import torch
    cols, big = HBM_SIDE[name]
    need = int(math.ceil(4 * INFINITY_CACHE / (B * big)))
    n = max(len(batches), min(cap, need))
    keys = [c.key for c in model.id_columns]
    g = torch.Generator(device="cuda")
    g.manual_seed(4242)
    out = list(batches)
    while len(out) < n:
        ids, dense = batches[len(out) % len(batches)]
        ids = ids.clone()
        for cname in cols:
            j = keys.index(cname)
            ids[:, j] = torch.randint(1, int(model.id_columns[j].vocab), (B,), generator=g, device="cuda", dtype=torch.int32)
        out.append((ids, dense))
    info = {"input_batches_cycled": n, "hbm_side_bytes_per_sample": big,
            "working_set_mb": n * B * big / 1e6,
            "working_set": "%d distinct batches x %d rows x %d B of rows from the table(s) beyond the %d MB Infinity Cache = %.0f MB touched between "
                           "two visits of a batch (%.1f x the cache)" % (n, B, big, INFINITY_CACHE / 1e6, n * B * big / 1e6, n * B * big / INFINITY_CACHE)}
    return out, info


def _resize_first_order(small, w, fields, fo_key, rng):
    """Weights of the config-4 models: everything from a small-vocabulary twin except the first-order block (one weight
    per id of every field, 27 M of them) and the big tables (already in `w`, on the device)."""
    from sparrowrecsys_amd.models import first_order_offsets
    fo_small = first_order_offsets(small.fields)
    fo_big = first_order_offsets(fields)
    old = np.asarray(small.weights[fo_key])
    n_small, n_big = fo_small["__total__"], fo_big["__total__"]
    blk = rng.uniform(-0.05, 0.05, size=(n_big, 1)).astype(np.float32)
    w[fo_key] = np.concatenate([blk, old[n_small:]], axis=0) if fo_key == "head/kernel" else blk
    return w


def oracle_forward(name, model, feats, dtype=np.float32):
    """The numpy oracle on `feats` (a few thousand rows).  Models holding device tables (config 4) are compacted first:
    only the table rows the sample references are pulled to the host and the ids are renumbered, so the oracle never sees
    the 6.9 GB table."""
    from oracle import ctr_oracle as O
    from sparrowrecsys_amd import synthetic as SY
    fields = getattr(model, "_bench_fields", SY.CONFIG2_FIELDS)
    weights = model.weights
    if any(hasattr(v, "data_ptr") for v in weights.values()):
        feats, weights, fields = compact_for_oracle(model, feats, fields)
    if name in ("deepfm_v2_c2", "deepfm_v2_c4", "deepfm_v2_ref"):
        return O.deepfm_v2_forward(feats, weights, dtype=dtype, fields=fields, order=model.order)
    if name in ("deepfm_c2", "deepfm_c4", "deepfm_ref"):
        return O.deepfm_forward(feats, weights, dtype=dtype, fields=fields, pairs=model.pairs)
    if name == "embedding_mlp_ref":
        return O.embedding_mlp_forward(feats, weights, dtype=dtype, movie_buckets=model.movie_buckets, user_buckets=model.user_buckets)
    if name == "neuralcf_ref":
        return O.neural_cf_forward(feats, weights, dtype=dtype, movie_buckets=model.movie_buckets, user_buckets=model.user_buckets)
    if name == "widedeep_c5":
        return O.wide_n_deep_forward(feats, weights, dtype=dtype, movie_buckets=model.movie_buckets,
                                     user_buckets=model.user_buckets, cross_buckets=model.cross_buckets,
                                     rated_buckets=model.rated_buckets)
    if name == "dien_ref":
        return O.dien_forward(feats, weights, dtype=dtype, hist_len=model.hist_len,
                              movie_buckets=model.movie_buckets, user_buckets=model.user_buckets)
    return O.din_forward(feats, weights, dtype=dtype, hist_len=model.hist_len,
                         movie_buckets=model.movie_buckets, user_buckets=model.user_buckets)


def compact_for_oracle(model, feats, fields):
    """DeepFM / DeepFM_v2 with device-resident identity tables: renumber each identity field's ids to 0..n_unique-1 and
    keep only those rows of its table and of the first-order block (name-sorted one-hot offsets, models.first_order_offsets)."""
    import torch
    from sparrowrecsys_amd.models import first_order_offsets
    fo_key = "fo_cat/kernel" if "fo_cat/kernel" in model.weights else "head/kernel"
    fo = first_order_offsets(fields)
    fo_w = np.asarray(model.weights[fo_key])
    new_fields, new_feats, uniq = [], dict(feats), {}
    for k, kind, v in fields:
        if kind == "id":
            u, inv = np.unique(np.asarray(feats[k]).astype(np.int64), return_inverse=True)
            uniq[k] = u
            new_feats[k] = inv.astype(np.int64)
            new_fields.append((k, kind, len(u)))
        else:
            new_fields.append((k, kind, v))
    w = {}
    for name, a in model.weights.items():
        key = name.split("/", 1)[1] if (name.startswith("emb/") or name.startswith("deep_emb/")) else None
        if key in uniq:
            a = a[torch.from_numpy(uniq[key]).to(a.device)].cpu().numpy() if hasattr(a, "data_ptr") else np.asarray(a)[uniq[key]]
        elif hasattr(a, "data_ptr"):
            a = a.cpu().numpy()
        w[name] = a
    fo_new = first_order_offsets(new_fields)
    blk = np.zeros((fo_new["__total__"], 1), np.float32)
    for k, kind, v in fields:
        rows = uniq[k] if kind == "id" else np.arange(v)
        blk[fo_new[k]:fo_new[k] + len(rows), 0] = fo_w[fo[k] + rows, 0]
    w[fo_key] = np.concatenate([blk, fo_w[fo["__total__"]:]], axis=0) if fo_key == "head/kernel" else blk
    return new_feats, w, new_fields


def cpu_baseline_c(name, model, feats, budget_s, note):
    """deepfm_v2_c2 / din_c3: the plain-C restatement of the forward (oracle/ctr_c.c, OpenMP over samples, built with
    -march=native on THIS host) on the packed ids / dense of the same synthetic batch -- a fairer stand-in for the
    reference's TF2 CPU forward than the numpy oracle, whose time goes into Python feature handling.  Checked against
    the numpy oracle before it is timed.  Returns None (and says why in `note`) when it cannot be used."""
    if name not in ("deepfm_v2_c2", "din_c3"):
        note.append("no C restatement for this workload")
        return None
    try:
        from oracle import ctr_c
        from sparrowrecsys_amd import synthetic as SY
        ctr_c.load(native=True)
        if name == "deepfm_v2_c2":
            cm = ctr_c.DeepFMv2C(model.weights, getattr(model, "_bench_fields", SY.CONFIG2_FIELDS))
            what = "DeepFM_v2"
        else:
            cm = ctr_c.DinC(model)
            what = "DIN"
        ids, dense = model.pack(feats[0])
        n = ids.shape[0]
        threads = max(1, min(os.cpu_count() or 1, 128))
        got = cm.forward(ids[:2048], dense[:2048], threads=1)
        ref = oracle_forward(name, model, {k: v[:2048] for k, v in feats[0].items()})[:, 0]
        if not (np.abs(got - ref).max() <= 5e-5):
            note.append("C restatement disagreed with the numpy oracle (max |diff| %g): not used" % np.abs(got - ref).max())
            return None
        out = np.empty(n, dtype=np.float32)
        cm.forward(ids, dense, threads=threads, out=out)     # warm-up (threads, page faults)
        t0 = time.perf_counter()
        done = 0
        while True:
            cm.forward(ids, dense, threads=threads, out=out)
            done += n
            el = time.perf_counter() - t0
            if el >= budget_s or done >= 4096 * n:
                break
        return {"value": done / el, "unit": "samples/s", "cores": threads, "kind": "port", "ran": "oracle/ctr_c.c",
                "sample": "plain-C restatement of the %s forward (oracle/ctr_c.c, OpenMP, -march=native; TensorFlow unavailable), "
                          "%d passes over one packed batch of %d rows, %.1f s" % (what, done // n, n, el),
                "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    except Exception as e:                                    # no compiler on this host, ...: say so, fall back to numpy
        note.append("C restatement unavailable (%s: %s)" % (type(e).__name__, e))
        return None


def cpu_model():
    """The host CPU's model name ("always print core count and CPU model")."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def hardware_block(torch, dev):
    """the device's own facts and a MEASURED streaming bandwidth next to the 8.0 TB/s every fraction is quoted
    against.  copy = torch's device-to-device copy of 1 GiB (bytes read + bytes written per second), read = a sum reduction over
    the same 1 GiB (bytes read per second); HIP events, median of 5 after a warm-up.  These are what simple streaming kernels
    reach on this box, not a different denominator: `frac` stays against 8.0e12 B/s."""
    p = torch.cuda.get_device_properties(dev)
    out = {"name": p.name, "arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
           "memory_GiB": round(p.total_memory / 2 ** 30, 1), "hbm_peak_GBps_quoted": HBM_PEAK / 1e9}
    for k in ("clock_rate", "memory_clock_rate", "memory_bus_width", "L2_cache_size"):
        if hasattr(p, k):
            out[k] = getattr(p, k)
    try:
        n = 1 << 28                                                # 2^28 floats = 1 GiB
        src = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
        dst = torch.empty_like(src)
        def timed(fn):
            fn(); torch.cuda.synchronize(dev)
            tt = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize(dev)
                tt.append(e0.elapsed_time(e1) * 1e-3)
            return float(np.median(tt))
        t_copy = timed(lambda: dst.copy_(src))
        t_read = timed(lambda: src.sum())
        out["measured_device_copy_GBps"] = round(2 * n * 4 / t_copy / 1e9, 1)
        out["measured_read_GBps"] = round(n * 4 / t_read / 1e9, 1)
        out["measured_with"] = "torch d2d copy / sum of 1 GiB, HIP events, median of 5"
        del src, dst
        torch.cuda.empty_cache()
    except Exception as e:                                          # (never fail the bench over a probe)
        out["measured_error"] = "%s: %s" % (type(e).__name__, e)
    return out


def cpu_baseline_tf(name, model, feats, budget_s, note):
    """probe ``import tensorflow`` FIRST.  When it works (it does not in the build container or on the
    GPU box: no network), the headline graph is rebuilt with tf.keras layers from the same weights -- Embedding tables with a
    zero row for the missing id, the per-field Dense projections, the FM sum-of-squares cross, the 32-16 MLP (DeepFM_v2.py:98-157
    at config 2's shape; the reference script itself hard-codes 4 fields / emb_dim 10 and is what tests/golden/make_tf_golden.py
    executes) -- checked against the oracle to 1e-4 and ``model.predict(x, batch_size=B)`` is timed on all host cores.  Returns
    None (and says why in `note`) otherwise.  UNTESTED CODE PATH: no TensorFlow has ever been importable where this ran."""
    if name != "deepfm_v2_c2":
        return None
    try:
        import tensorflow as tf
    except Exception as e:
        note.append("TensorFlow not importable (%s): CPU restatement timed instead" % type(e).__name__)
        return None
    try:
        from sparrowrecsys_amd import synthetic as SY
        from sparrowrecsys_amd.models import first_order_offsets
        fields = getattr(model, "_bench_fields", SY.CONFIG2_FIELDS)
        w = model.weights
        K = tf.keras
        ids_in = K.layers.Input(shape=(len(fields),), dtype="int32")
        num_in = K.layers.Input(shape=(7,), dtype="float32")
        fo = first_order_offsets(fields)
        projected, first = [], []
        for i, (k, _, v) in enumerate(fields):
            idc = tf.where(ids_in[:, i] < 0, v, ids_in[:, i])                    # -1 (missing / OOV) -> the zero row at index v
            tab = np.concatenate([np.asarray(w["emb/" + k]), np.zeros((1, w["emb/" + k].shape[1]), np.float32)])
            e = K.layers.Embedding(v + 1, tab.shape[1], weights=[tab], trainable=False)(idc)
            projected.append(K.layers.Dense(w["proj/%s/kernel" % k].shape[1], weights=[w["proj/%s/kernel" % k], w["proj/%s/bias" % k]])(e))
            fw = np.concatenate([np.asarray(w["fo_cat/kernel"])[fo[k]:fo[k] + v], np.zeros((1, 1), np.float32)])
            first.append(K.layers.Embedding(v + 1, 1, weights=[fw], trainable=False)(idc))
        order = [k for k, _, _ in fields]
        projected = [projected[order.index(k)] for k in model.order]
        projected.append(K.layers.Dense(w["proj/num/kernel"].shape[1], weights=[w["proj/num/kernel"], w["proj/num/bias"]])(num_in))
        stack = tf.stack(projected, axis=1)
        fo_num = K.layers.Dense(1, weights=[w["fo_num/kernel"], w["fo_num/bias"]])(num_in)
        first_order = tf.add_n(first) + w["fo_cat/bias"] + fo_num
        s = tf.reduce_sum(stack, axis=1)
        fm = s * s - tf.reduce_sum(stack * stack, axis=1)
        deep = K.layers.Flatten()(stack)
        i = 0
        while "deep%d/kernel" % i in w:
            deep = K.layers.Dense(w["deep%d/kernel" % i].shape[1], activation="relu", weights=[w["deep%d/kernel" % i], w["deep%d/bias" % i]])(deep)
            i += 1
        out = K.layers.Dense(1, activation="sigmoid", weights=[w["head/kernel"], w["head/bias"]])(tf.concat([first_order, fm, deep], axis=1))
        km = K.Model([ids_in, num_in], out)
        ids, dense = model.pack(feats[0])
        n = ids.shape[0]
        got = km.predict([ids[:2048], dense[:2048]], batch_size=2048, verbose=0)[:, 0]
        ref = oracle_forward(name, model, {k: v[:2048] for k, v in feats[0].items()})[:, 0]
        if not (np.abs(got - ref).max() <= 1e-4):
            note.append("TensorFlow graph disagreed with the oracle (max |diff| %g): not used" % np.abs(got - ref).max())
            return None
        km.predict([ids, dense], batch_size=n, verbose=0)
        t0, done = time.perf_counter(), 0
        while True:
            km.predict([ids, dense], batch_size=n, verbose=0)
            done += n
            el = time.perf_counter() - t0
            if el >= budget_s or done >= 64 * n:
                break
        return {"value": done / el, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port", "ran": "tensorflow %s (tf.keras graph of the same weights)" % tf.__version__,
                "sample": "TF2 CPU forward of the DeepFM_v2 graph at config 2's shape, model.predict(batch_size=%d), %d passes, %.1f s" % (n, done // n, el),
                "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    except Exception as e:
        note.append("TensorFlow leg failed (%s: %s): CPU restatement timed instead" % (type(e).__name__, e))
        return None


def cpu_baseline(name, model, feats, budget_s):
    """TensorFlow first when it can be imported (`cpu_baseline_tf`), otherwise -- the expected case --
    the CPU restatement ("port": TensorFlow is not installable) timed on this host's cores over a bounded sample of
    the same workload: the C restatement where there is one (DeepFM_v2, DIN), with the numpy oracle's rate reported next
    to it; the numpy oracle otherwise.  `ran` names the one that produced `value`, `tensorflow` says why TF did not."""
    note = []
    t = cpu_baseline_tf(name, model, feats, 0.5 * budget_s, note)
    if t is not None:
        return t
    tf_note = "; ".join(note)
    note = []
    c = cpu_baseline_c(name, model, feats, 0.5 * budget_s, note)
    if c is not None and tf_note:
        c["tensorflow"] = tf_note
    if c is not None:
        npy = cpu_baseline_numpy(name, model, feats, 0.5 * budget_s)
        c["numpy_oracle_samples_per_sec"] = npy["value"]
        c["numpy_oracle_threads"] = npy["cores"]
        return c
    r = cpu_baseline_numpy(name, model, feats, budget_s)
    r["ran"] = "oracle/ctr_oracle.py (numpy)"
    r["why_not_c"] = "; ".join(note)
    return r


def cpu_baseline_numpy(name, model, feats, budget_s):
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = 1
    f = feats[0]
    n = len(next(iter(f.values())))
    sample = min(n, 16384)
    fs = {k: v[:sample] for k, v in f.items()}
    oracle_forward(name, model, fs)                      # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        oracle_forward(name, model, fs)
        done += sample
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 64 * sample:
            break
    return {"value": done / el, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "numpy oracle (CPU restatement; TensorFlow unavailable), %d passes of %d rows of the same synthetic batch, %.1f s"
                      % (done // sample, sample, el),
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (what
    `python -m torch.distributed.run --nproc-per-node N` would do), rank r on GPU r, rendezvous on 127.0.0.1."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


class Timer:
    """Regions of R*K back-to-back steps, bracketed by fence() (barrier + synchronize) on both sides, host clock,
    max over ranks; HIP events around the same region ride along."""

    def __init__(self, run_steps, fence, dist_on, device_events):
        self.run_steps, self.fence, self.dist_on, self.device_events = run_steps, fence, dist_on, device_events
        self.step = 0

    def region(self, n_steps):
        import torch
        self.fence()
        ev = None
        if self.device_events:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        t0 = time.perf_counter()
        if ev:
            ev[0].record()
        self.run_steps(self.step, n_steps)
        if ev:
            ev[1].record()
        self.fence()
        wall = time.perf_counter() - t0
        self.step += n_steps
        if self.dist_on:
            import torch.distributed as dist
            t = torch.tensor([wall], dtype=torch.float64, device="cuda" if self.device_events else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        return wall, (ev[0].elapsed_time(ev[1]) * 1e-3 if ev else wall)


def strict_loop(eng, batches, outs, ws, n_steps, lb_restore, fan_restore):
    """n_steps forwards, ONE batch per launch, strict stream order, HIP events on the launch stream: the per-launch
    duration rocprofv3's kernel trace reports for the same command.  Returns seconds per launch (median of 3 loops)."""
    import torch
    NB = len(batches)
    eng.set_many_streams(0)
    eng.set_many_batches(1)
    idx = [i % NB for i in range(n_steps)]
    run = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        ev0.record()
        run()
        ev1.record()
        torch.cuda.synchronize()
        times.append(ev0.elapsed_time(ev1) * 1e-3 / n_steps)
    if fan_restore:
        eng.set_many_streams(fan_restore)
    if lb_restore > 1:
        eng.set_many_batches(lb_restore)
    return float(np.median(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="deepfm_v2_c2")
    ap.add_argument("--batch", type=int, default=0, help="rows per GPU (default: the config's batch)")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf", "hot"], help="id distribution")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle spot check of the outputs")
    ap.add_argument("--no-hardware-probe", action="store_true", help="skip the device facts / measured copy bandwidth block")
    ap.add_argument("--min-region-ms", type=float, default=50.0,
                    help="a timed region = R back-to-back repetitions of the K-step block, R chosen so that it lasts at least this long")
    ap.add_argument("--regions", type=int, default=5, help="timed regions; ms_per_step is the median region / (R*K)")
    ap.add_argument("--settle-ms", type=float, default=40.0, help="the warm-up lasts at least this long, whatever --warmup says")
    ap.add_argument("--gather-group", type=int, default=64,
                    help="N>1: batches whose score slices share one RCCL all-gather (overlapped with the next group)")
    ap.add_argument("--overlap-streams", type=int, default=2,
                    help="with --launch-batches 1: fan sprk_forward_many's independent batches over S helper HIP streams (2..4; "
                         "0 = strict stream order) in the TIMED region.  The roofline block is always measured in strict order "
                         "(one kernel at a time), in its own loop after the timed regions.")
    ap.add_argument("--launch-batches", type=int, default=64,
                    help="batches ONE kernel launch scores in the timed region (sprk_set_many_batches, up to 64; 1 = a launch per "
                         "batch).  Each batch keeps its own buffers of --batch rows.  With N > 1 the launches of the timed region run "
                         "in strict order (no stream fan-out).  `roofline` and `value_one_batch_per_launch` stay ONE batch per "
                         "launch; `roofline_timed_region` describes the N-batch launches.")
    ap.add_argument("--input-batches", type=int, default=0,
                    help="distinct synthetic input batches the steps cycle through (default: 64 for the headline workload -- no "
                         "two batches of one 64-batch launch share buffers -- 16 for din_c3, 8 otherwise)")
    ap.add_argument("--big-vocab", type=int, default=0,
                    help="deepfm_v2_c2 / deepfm_c2: rows of each identity table (e.g. 8388608 = 1 GiB of folded rows per table, "
                         "far beyond the Infinity Cache); default 0 = the MovieLens-20M-shaped vocabularies of the config")
    ap.add_argument("--hbm-resident", type=int, default=-1,
                    help="deepfm_v2_c2 at N=1: also measure the same graph with identity tables of this many rows (HBM-resident "
                         "gather, block `roofline_hbm_resident`); default 8388608, 0 = skip")
    ap.add_argument("--collective", default="torch", choices=["torch", "sprk", "peer"],
                    help="N>1: who issues the all-gather of score slices -- torch.distributed (default), the C ABI's "
                         "sprk_comm_allgather_scores (RCCL bound inside libsparrow_hip.so, no torch on the data path), or "
                         "sprk_peer_allgather_scores (direct peer writes into IPC-mapped receive buffers, no RCCL)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (gloo: functional test of the N>1 path with ranks sharing one GPU)")
    ap.add_argument("--side-workloads", default="din_c3,deepfm_c2,deepfm_c4,widedeep_c5,neuralcf_serving,predict_csv",
                    help="default workload at N=1: also measure these (short loops) and put them under `workloads` in the same JSON "
                         "line -- BASELINE's metric names DeepFM and DIN; '' = none")
    ap.add_argument("--variants", type=int, default=1,
                    help="1: also time the headline graph under Zipf(1.05) ids and on all-f32 MFMA, strict order (roofline_variants); 0: skip")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank scores its own batches of --batch rows (per-GPU work fixed).  strong: the batches are GLOBAL ones of "
                         "--batch rows that every rank holds; rank r scores rows [r B/N, (r+1) B/N) and the score slices are all-gathered "
                         "(north_star: 'batches shard row-wise across the 8 GPUs ... all-gather ... for the final score vector'; what one Jetty "
                         "request needs, RecForYouProcess.java:113-138) -- total work fixed as N grows")
    ap.add_argument("--shard-tables", action="store_true",
                    help="deepfm_c4 / deepfm_v2_c4: the 27 M-row item table(s) ROW-SHARDED over the ranks (BASELINE config 4's wording) instead of "
                         "replicated: sparrowrecsys_amd.dist.ShardedTable, peers' rows loaded over xGMI by the unchanged fused kernel")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU, no kernels: the launcher / process-group / grouped all-gather / timing plumbing with a stand-in "
                         "forward on CPU tensors (gloo).  For the CPU test of `--gpus N` self-spawning; the line says dry_run")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    # SPRK_BENCH_FORCE_DIST=1: run the N>1 code path (process group, grouped all-gather on the side stream) with
    # WORLD_SIZE 1 -- a functional check of the RCCL calls on a one-GPU box
    dist_on = world > 1 or os.environ.get("SPRK_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group("gloo")

    B = args.batch or {"din_c3": 32768, "widedeep_c5": 131072}.get(args.workload, 65536)
    nb_in = args.input_batches or {"deepfm_v2_c2": 64, "deepfm_c2": 16, "din_c3": 16, "din_ref": 16, "dien_ref": 16}.get(args.workload, 8)
    if args.batch and args.batch > 262144:
        nb_in = min(nb_in, 8)
    global SHARD_TABLES
    SHARD_TABLES = bool(args.shard_tables)
    strong = args.scaling == "strong"
    B_global, feats_global0 = B, None
    if strong:
        if B % world:
            raise SystemExit("--scaling strong: the global batch of %d rows does not divide over %d ranks" % (B, world))
        # every rank draws the SAME global batches (one seed) and keeps its row shard; batch 0 stays whole for the one-request latency
        model, feats, desc, roof = build_workload(args.workload, B_global, args.dist, seed_offset=0, big_vocab=args.big_vocab, NB=nb_in)
        B = B_global // world
        feats_global0 = feats[0]
        feats = [{k: np.asarray(v)[rank * B:(rank + 1) * B] for k, v in f.items()} for f in feats]
    else:
        model, feats, desc, roof = build_workload(args.workload, B, args.dist, seed_offset=rank, big_vocab=args.big_vocab, NB=nb_in)
    eng = model.engine
    is_din = args.workload in ("din_c3", "din_ref", "dien_ref")
    roof["kernel"] = eng.kernel_name() if not is_din else (eng.describe().get("stage") or roof["kernel"])   # what the handle really dispatches to (from the handle, not from the environment)
    env = os.environ.get
    lb = 1
    if args.launch_batches > 1 and env("SPRK_FORCE_INTERPRETER") != "1":
        if roof["kernel"] in ("k_deepfm_v2_joint", "k_rows_chain"):
            lb = args.launch_batches
        elif roof["kernel"] == "k_deepfm_pairs" or (roof["kernel"] == "k_mlp_rows" and env("SPRK_MLP_ROWS_MANY") != "0"):
            lb = min(args.launch_batches, 16)      # (k_mlp_rows_many [r6]; SPRK_MLP_ROWS_MANY=0: launch by launch over two streams, rounds 2-5)
        elif is_din and args.workload != "dien_ref" and env("SPRK_DIN_LEGACY") != "1" and env("SPRK_DIN_TAIL") != "0" and eng.kernel_name() in ("k_din_tail", "k_din_fused"):
            lb = min(args.launch_batches, 16)   # groups of batches: one attention + one tail launch each, alternating streams
    if lb > 1:
        eng.set_many_batches(lb)
        if not is_din:
            args.overlap_streams = 0           # several batches per launch: strict order measured faster than the fan-out
    fan = args.overlap_streams if (args.overlap_streams >= 2 and eng.set_many_streams(args.overlap_streams)) else 0
    batches = []
    for f in feats:
        ids, dense = model.pack(f)
        batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()))
    NB_host, ws_info = len(batches), None
    if args.workload in HBM_SIDE and not args.input_batches and not strong and not SHARD_TABLES:
        batches, ws_info = hbm_cycle(args.workload, model, batches, B)   # [r6] configs 4 / 5: a working set of 4 x the Infinity Cache
    NB = len(batches)
    outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(NB)]
    ws = torch.empty(max(eng.many_workspace_bytes(B, max(fan, 1) * lb) // 4, 1), dtype=torch.float32, device="cuda")
    gs = None
    if dist_on:
        from sparrowrecsys_amd.dist import GroupedScoreGather, PeerScoreComm, ScoreComm
        comm = ScoreComm() if (args.collective == "sprk" and args.backend == "nccl") else None
        if args.collective == "peer":
            comm = PeerScoreComm(B * max(1, args.gather_group))
        gs = GroupedScoreGather(B, max(1, args.gather_group), torch.device("cuda", local_dev), comm=comm)

    group_run = [None, None]
    prepared = {}

    def run_steps(first, count):
        """`count` steps starting at step index `first`.  One sprk_forward_many call enqueues a run of
        forwards (the same kernel launches, without a Python/ctypes round trip per launch, which at ~7 us
        would out-last the kernel).  N>1: every step's score slice lands in a GroupedScoreGather ring slot;
        each full group of --gather-group steps is exchanged by ONE RCCL all-gather on a second stream
        while the next group is scored (a per-step all-gather of 256 KiB costs more launch latency than
        the forward it follows)."""
        if count <= 0:
            return
        if gs is None:
            # the marshalled (and validated) pointer arrays of a `count`-step run are built once, OUTSIDE any timed region
            # (prime() below), and replayed: a run always starts at input batch 0
            if count not in prepared:
                idx = [i % NB for i in range(count)]
                prepared[count] = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
            prepared[count]()
            return
        i = first
        while i < first + count:
            if gs.fill == 0 and first + count - i >= gs.G:
                # a whole group: prepared pointer arrays (one per ring slot), one foreign call, one collective
                slot = gs.slot
                outs_g = gs.group_outs()
                if group_run[slot] is None:
                    idx = [j % NB for j in range(gs.G)]
                    group_run[slot] = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], outs_g, ws)
                group_run[slot](stream=torch.cuda.current_stream().cuda_stream)
                gs.commit()
                i += gs.G
                continue
            n = min(gs.G - gs.fill, first + count - i)
            idx = [j % NB for j in range(i, i + n)]
            eng.forward_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [gs.out() for _ in idx], ws)
            i += n
            if gs.full():
                gs.commit()

    def fence():
        if gs is not None:
            gs.flush()                     # exchange a partial group, wait for every collective
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    K = args.steps
    timer = Timer(run_steps, fence, dist_on, True)
    # ---- warm-up: W steps as asked, then K-step blocks until --settle-ms have passed (clocks, caches, RCCL channels) ----
    t_w = time.perf_counter()
    fence()
    run_steps(0, args.warmup)
    fence()
    timer.step = args.warmup
    settle = torch.tensor([0.0], dtype=torch.float64, device="cuda")
    while True:
        el = time.perf_counter() - t_w
        if dist_on:                        # every rank must take the same number of blocks
            settle[0] = el
            dist.all_reduce(settle, op=dist.ReduceOp.MIN)
            el = float(settle.item())
        if el * 1e3 >= args.settle_ms:
            break
        timer.region(K)
    eng.check_ids()
    # ---- calibration: one K-step block -> R (second pass: the first builds the block's prepared launch list) ----
    timer.region(K)
    _, blk = timer.region(K)
    if dist_on:
        t = torch.tensor([blk], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blk = float(t.item())
    R = int(max(1, min(math.ceil(args.min_region_ms * 1e-3 / max(blk, 1e-9)), max(1, 2_000_000 // max(K, 1)))))
    if lb > 1 and not dist_on:
        # a region is R*K steps cut into launches of `lb` batches: make it a whole number of launches, so that every launch of the timed
        # regions is a FULL one and rocprofv3's average over the multi-batch kernel's launches reproduces `roofline_timed_region`
        # (the kept trace averaged launches of different batch counts)
        mult = lb // math.gcd(K, lb)
        R = int(math.ceil(R / mult) * mult)
    timer.region(R * K)                        # untimed: builds the region's prepared launch list, one more warm-up pass
    # the K-step calibration block is ONE short launch list (K = 20 < 64 batches per launch) and reads slower per step than the
    # region's full launches: correct R on the region's own time so that it really lasts >= --min-region-ms
    for _ in range(3):
        w_reg, _ = timer.region(R * K)         # (already the max over ranks: every rank takes the same decision)
        if w_reg * 1e3 >= args.min_region_ms or R >= max(1, 2_000_000 // max(K, 1)):
            break
        R = int(min(math.ceil(R * args.min_region_ms * 1.08e-3 / max(w_reg, 1e-9)), max(1, 2_000_000 // max(K, 1))))
        if lb > 1 and not dist_on:
            mult = lb // math.gcd(K, lb)
            R = int(math.ceil(R / mult) * mult)
        timer.region(R * K)                    # (builds the new list)
    # ---- timed regions ----
    walls, evs = [], []
    for _ in range(max(1, args.regions)):
        w, e = timer.region(R * K)
        walls.append(w)
        evs.append(e)
    elapsed = float(np.median(walls))          # seconds per region of R*K steps (max over ranks, median over regions)
    ev_region = float(np.median(evs))
    n_region = R * K

    # kernel time for the roofline: HIP events on the launch stream around forwards in STRICT order, ONE batch per
    # launch (what rocprofv3's per-kernel duration measures), in its own loop after the timed regions
    n_strict = int(max(K, min(20000, math.ceil(0.02 / max(blk / K, 1e-9)))))
    # The strict loop cycles at most 32 input batches (round 2's count).  The headline's timed region needs 64 distinct ones -- no
    # two batches of a 64-batch launch may share buffers -- but 64 x 2.6 MB of ids / numerics / scores next to the 130 MB of tables no
    # longer fit the 256 MB Infinity Cache together, and the strict launch then reads 7.9 instead of 7.5 us: a working-set effect of
    # the BENCH's input count, which `roofline_hbm_resident` reports on purpose and this block should not pick up by accident.
    NBS = NB if ws_info else min(NB, STRICT_CYCLE)               # (configs 4 / 5: the strict loop walks the WHOLE working set, that is its point)
    fwd_s = strict_loop(eng, batches[:NBS], outs[:NBS], ws, n_strict, lb, fan)
    fan2_s = None
    if not dist_on and lb > 1 and not is_din:
        # one batch per launch, independent batches alternating over two helper streams (still a launch per batch)
        eng.set_many_batches(1)
        eng.set_many_streams(2)
        idx = [i % NBS for i in range(n_strict)]
        run2 = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tt = []
        for _ in range(3):
            torch.cuda.synchronize()
            ev0.record(); run2(); ev1.record()
            torch.cuda.synchronize()
            tt.append(ev0.elapsed_time(ev1) * 1e-3 / n_strict)
        fan2_s = float(np.median(tt))
        eng.set_many_streams(fan)
        eng.set_many_batches(lb)

    # output spot check against the oracle (outside the timed region)
    check = None
    if rank == 0 and not args.no_check:
        # through the SAME call the timed region makes (sprk_forward_many with `lb` batches per launch over `fan` streams): the
        # first and the last batch of a group, head and tail rows of each, against the oracle; and the one-batch entry point
        m = min(NB_host, max(lb, 2))                             # (the host-drawn batches: the oracle has their features)
        chk = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(m)]
        eng.set_many_batches(lb)
        eng.set_many_streams(fan)
        eng.forward_many([batches[j][0] for j in range(m)], [batches[j][1] for j in range(m)], chk, ws)
        eng.forward(batches[0][0], batches[0][1], outs[0], ws)
        torch.cuda.synchronize()
        n = min(2048, B)
        check = 0.0
        for j in sorted({0, m - 1}):
            for sl in (slice(0, n), slice(B - n, B)):
                ref = oracle_forward(args.workload, model, {k: v[sl] for k, v in feats[j].items()})[:, 0]
                check = max(check, float(np.abs(chk[j][sl].cpu().numpy() - ref).max()))
        # (DIN on k_din_fused: the same bits from a launch per batch and from the persistent several-batches launch; under
        # SPRK_DIN_FUSED_MB=0 several batches per launch are the attention + tail pipeline -- another instruction sequence for the
        # tail's fc0, equal to 3e-6, tests/test_gpu_parity.py::test_din_tail_paths; every other graph: the same bits)
        same = torch.equal(chk[0], outs[0]) or (eng.kernel_name() == "k_din_fused" and float((chk[0] - outs[0]).abs().max()) <= 3e-6)
        if not same:
            raise SystemExit("bench: sprk_forward_many and sprk_forward disagree on batch 0")
        if not check <= 1e-4:
            raise SystemExit("bench outputs differ from the oracle: max|err| = %g" % check)

    strong_block = None
    if strong:
        # ONE global batch as one request: shard -> forward -> all-gather, nothing overlapped with anything (sparrowrecsys_amd.dist.
        # RowShardedPredictor: what the Jetty ranker's single request per recommendation needs); max over ranks of the median latency
        from sparrowrecsys_amd.dist import PeerScoreComm, RowShardedPredictor, ScoreComm
        gi, gd = model.pack(feats_global0)
        gi, gd = torch.from_numpy(gi).cuda(), torch.from_numpy(gd).cuda()
        wsl = torch.empty(max(eng.workspace_bytes(B) // 4, 1), dtype=torch.float32, device="cuda")
        eng.set_many_batches(1)
        eng.set_many_streams(0)

        def fwd_local(i_, d_):
            return model.predict_device(i_, d_, workspace=wsl)
        if dist_on:
            comm1 = None
            if args.collective == "sprk" and args.backend == "nccl":
                comm1 = ScoreComm()
            elif args.collective == "peer":
                comm1 = PeerScoreComm(B)
            pred = RowShardedPredictor(fwd_local, comm=comm1)
            one = lambda: pred.predict(gi, gd)
        else:
            one = lambda: fwd_local(gi, gd)
        for _ in range(20):
            got_all = one()
        torch.cuda.synchronize()
        lats = []
        for _ in range(200):
            if dist_on:
                dist.barrier()
            t0 = time.perf_counter()
            got_all = one()
            torch.cuda.synchronize()
            lats.append(time.perf_counter() - t0)
        lat = float(np.median(lats))
        if dist_on:
            t = torch.tensor([lat], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            lat = float(t.item())
        chk1 = None
        if not args.no_check and rank == 0:
            nr = min(1024, B_global)
            ref = oracle_forward(args.workload, model, {k: np.asarray(v)[B_global - nr:] for k, v in feats_global0.items()})[:, 0]
            chk1 = float(np.abs(got_all[B_global - nr:].cpu().numpy() - ref).max())      # the LAST rank's rows, as gathered on rank 0
            if not chk1 <= 1e-4:
                raise SystemExit("bench (strong): gathered scores differ from the oracle: max|err| = %g" % chk1)
        strong_block = {"what": "one global batch of %d rows as ONE request: row shard -> forward -> all-gather of the score slices, host clock "
                                "around the call + synchronize, median of 200, max over ranks" % B_global,
                        "latency_us": lat * 1e6, "samples_per_s": B_global / lat, "rows_per_rank": B,
                        "gathered_scores_oracle_check_max_abs_err": chk1}
    if rank == 0:
        value = B * world * n_region / elapsed
        region = "strict-order one-batch-per-launch loop after the timed regions (%d launches over %d input batches, median of 3 loops)" % (n_strict, NBS)
        legacy_din = env("SPRK_DIN_LEGACY") == "1"
        extra = {}
        if roof["bound"] == "hbm":
            achieved = roof["bytes_per_sample"] * B / fwd_s / 1e9
            rl = {"bound": "hbm", "kernel": roof["kernel"] + " (one batch of %d rows per launch)" % B, "achieved": achieved,
                  "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved * 1e9 / HBM_PEAK,
                  "algorithmic_bytes_per_sample": roof["bytes_per_sample"], "avg_launch_us": fwd_s * 1e6,
                  "timed_with": "HIP events, " + region}
            if "reference_bytes_per_sample" in roof:
                rl["reference_bytes_per_sample"] = roof["reference_bytes_per_sample"]
            mi = mfma_issued(roof["kernel"], roof.get("mfma_reference_flops"), wide_rows=args.workload == "deepfm_c4")
            if mi:
                extra["roofline_mfma"] = mfma_block(roof["kernel"], mi, B, fwd_s)
        else:
            # DIN step = k_din_attn + k_din_tail; time the attention kernel alone
            pooled = torch.empty((B, eng.n_aux), dtype=torch.float32, device="cuda")
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_att = max(K, 200)
            tt = []
            for _ in range(3):
                torch.cuda.synchronize()
                ev0.record()
                for i in range(n_att):
                    eng.din_pool(batches[i % NB][0], pooled)
                ev1.record()
                torch.cuda.synchronize()
                tt.append(ev0.elapsed_time(ev1) * 1e-3 / n_att)
            din_s = float(np.median(tt))
            if legacy_din:
                # k_din_pool: the reference's K = 4D contraction on f32 MFMA
                achieved = roof["flops_per_sample"] * B / din_s / 1e12
                rl = {"bound": "mfma", "kernel": roof["kernel"], "achieved": achieved, "peak": MFMA_F32_PEAK / 1e12,
                      "unit": "TFLOP/s", "frac": achieved * 1e12 / MFMA_F32_PEAK,
                      "algorithmic_flops_per_sample": roof["flops_per_sample"]}
            else:
                # k_din_attn: after the K = 4D -> D fold and the move to split-f16 MFMA the matrix pipe is a small
                # share of the kernel; what bounds it is the history gather (+ the VALU work per gathered row),
                # so it is priced against HBM bandwidth on SURVEY 8(d)'s algorithmic bytes -- and, next to it,
                # against the f16 MFMA peak on what it issues (roofline_mfma)
                achieved = roof["bytes_per_sample"] * B / din_s / 1e9
                rl = {"bound": "hbm", "kernel": roof["kernel"], "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                      "frac": achieved * 1e9 / HBM_PEAK,
                      "reference_flops_per_sample": roof["flops_per_sample"],
                      "reference_equivalent_TFLOPs": roof["flops_per_sample"] * B / din_s / 1e12}
                mi_att = mfma_issued(roof["kernel"], roof["flops_per_sample"], hist_len=roof["hist_len"])
                if mi_att:                                        # (k_dien_seq runs on the VALU: nothing to report)
                    extra["roofline_mfma"] = mfma_block(roof["kernel"], mi_att, B, din_s)
                if eng.kernel_name() in ("k_din_tail", "k_din_fused"):
                    tk = "k_din_tail" if eng.kernel_name() == "k_din_tail" else "k_din_fused (tail epilogue)"
                    extra["roofline_mfma_tail"] = mfma_block(tk, mfma_issued(tk, roof["tail_reference_flops"]), B, max(fwd_s - din_s, 1e-9))
            rl.update({"algorithmic_bytes_per_sample": roof["bytes_per_sample"],
                       "avg_launch_us": din_s * 1e6, "step_us_all_kernels": fwd_s * 1e6,
                       "timed_with": "HIP events, %s-only loop after the timed regions" % roof["kernel"]})
            if eng.kernel_name() == "k_dien_fused":
                # [r5] DIEN in ONE launch (k_dien_fused: the recurrence, then the tail as the same wave's epilogue): the step's bytes against the
                # launch's own duration; the sequence stage alone (what sprk_din_pool launches: k_dien_seq_mfma) stays as a sub-field
                fb = roof["bytes_per_sample"] + roof.get("tail_bytes_per_sample", 0)
                seq_only = {k: rl[k] for k in ("kernel", "achieved", "frac", "algorithmic_bytes_per_sample", "avg_launch_us", "timed_with")}
                rl.update({"kernel": "k_dien_fused (sequence stage + tail, one launch of %d rows)" % B, "algorithmic_bytes_per_sample": fb,
                           "avg_launch_us": fwd_s * 1e6, "achieved": fb * B / fwd_s / 1e9, "frac": fb * B / fwd_s / HBM_PEAK,
                           "timed_with": "HIP events, strict order, one batch per launch", "sequence_only": seq_only})
            if eng.kernel_name() == "k_din_fused":
                # the whole DIN step is ONE launch: the attention stage's bytes + the tail's (userId row, two genre rows -- the candidate's
                # row is the attention's --, 7 numerics, 3 ids, the score), against the fused launch's own duration
                fb = roof["bytes_per_sample"] + roof.get("tail_bytes_per_sample", 0)
                att_only = {k: rl[k] for k in ("kernel", "achieved", "frac", "algorithmic_bytes_per_sample", "avg_launch_us", "timed_with")}
                rl.update({"kernel": "k_din_fused<TAIL> (attention + pooling + tail, one launch of %d rows)" % B, "algorithmic_bytes_per_sample": fb,
                           "avg_launch_us": fwd_s * 1e6, "achieved": fb * B / fwd_s / 1e9, "frac": fb * B / fwd_s / HBM_PEAK,
                           "timed_with": "HIP events, strict order, one batch per launch", "attention_only": att_only})
        # memory-side bytes per launch from the committed PMC passes (rocprofv3 --pmc runs are separate from the
        # timed run by design); only quoted for the batch size and kernel they were collected on
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and not args.big_vocab:
            try:
                t = json.load(open(tpath)).get(args.workload)
                if t and t.get("batch") == B and t.get("kernel") == roof["kernel"]:
                    traffic = t["bytes_per_launch"]
                    rl["traffic_source"] = "profiles/traffic.json (PMC FETCH_SIZE x2 + WRITE_SIZE, round %s)" % t.get("round")
            except Exception:
                traffic = None
        rl["traffic"] = traffic
        table_mb = getattr(eng, "table_bytes", lambda: 0)() / 1e6
        line = {
            "metric": "ctr_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": K, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / n_region,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_detail": "fp32 in, fp32 out, fp32 accumulation; contractions of finalize-bounded operands (table rows x weights) run on "
                            "v_mfma_f32_16x16x32_f16 with split operands hi + lo = 22 significand bits (narrower than fp32's 24: measured |err| vs the "
                            "fp64 oracle <= 5e-7 on the checked rows, bar 1e-4); the all-f32-MFMA variant is SPRK_V2_HALF=0 / SPRK_DIN_HALF=0 / SPRK_DYN_F16=0",
            "value_one_batch_per_launch": B * world / fwd_s,
            "config": {"workload": "%s: %s" % (args.workload, desc), "batch_per_gpu": B, "global_batch": B * world,
                       "id_distribution": args.dist, "input_batches_cycled": NB,
                       "repeats": R, "regions": len(walls), "region_ms": [round(w * 1e3, 3) for w in walls],
                       "timed": "each region = %d back-to-back repetitions of the %d-step block (>= %.0f ms), barrier + synchronize on both "
                                "sides, host clock, max over ranks; ms_per_step = median region / %d; value = the same region" % (R, K, args.min_region_ms, n_region),
                       "ms_per_step_hip_events": ev_region * 1e3 / n_region,
                       "collective": None if not dist_on else ("sprk_comm_allgather_scores (RCCL behind the C ABI)" if args.collective == "sprk"
                                                                 else "sprk_peer_allgather_scores (direct peer writes, IPC-mapped receive buffers)" if args.collective == "peer"
                                                                 else "torch.distributed.all_gather_into_tensor (%s)" % ("RCCL" if args.backend == "nccl" else "gloo: functional run only")),
                       "parallelism": ("rows sharded over %d GPU(s), tables replicated, all-gather of scores" % world)
                                      + ("" if world == 1 else " (one collective per %d steps, overlapped; %d collectives per region)"
                                         % (gs.G, (n_region + gs.G - 1) // gs.G)),
                       "oracle_check_max_abs_err": check,
                       "tables_row_sharded": ("%d table(s) of %d rows row-sharded over %d rank(s), one virtual range per rank (sprk_vtable_*), peers' rows "
                                              "loaded by the fused kernel" % (len(_SHARDED_TABLES), CONFIG4_ROWS, world)) if _SHARDED_TABLES else None,
                       "launch_overlap_streams": fan, "batches_per_launch": lb,
                       "tables": "%.0f MB of device tables: %s" % (table_mb, (("%d of the %d algorithmic bytes per sample are rows of the table(s) beyond the 256 MB Infinity Cache "
                                                                                 "(HBM-side rate: roofline.hbm_side_GBps); the other tables fit the cache" % (ws_info["hbm_side_bytes_per_sample"], roof["bytes_per_sample"]))
                                                                                if ws_info else "beyond the 256 MB Infinity Cache (HBM-resident gather)") if table_mb > 512
                                                                      else "resident in the 256 MB Infinity Cache -- `roofline` is a fabric/cache-side rate for this "
                                                                           "config; see roofline_hbm_resident") if table_mb else None,
                       "kernel": eng.kernel_name(),
                       "arithmetic": "fp32 semantics; contractions whose operands are bounded at finalize (table rows x weights) run on "
                                     "v_mfma_f32_16x16x32_f16 with split operands hi + lo (22 significand bits) and f32 accumulation -- "
                                     "fp32-class error, tests/test_gpu_parity.py::test_deepfm_v2_split_f16_is_fp32_class; everything else "
                                     "on f32 MFMA / VALU (SPRK_V2_HALF=0 / SPRK_DIN_HALF=0 force f32 MFMA throughout)"},
            "roofline": rl,
        }
        line["value_one_batch_per_launch_note"] = ("B x n_gpus / the strict-order per-launch time of `roofline` (%.2f us): the north star's "
                                                   "'at batch %d' figure; `value` scores %d batch(es) per launch" % (fwd_s * 1e6, B, lb))
        if fan2_s is not None:
            line["value_one_batch_per_launch_two_streams"] = B * world / fan2_s
        if lb > 1 and roof["bound"] == "hbm":
            # the timed region's own launches: lb batches (own buffers, B rows each) per launch, strict stream order, so the
            # HIP events around the region bracket exactly ceil(R*K / lb) back-to-back launches of the multi-batch instantiation
            n_launch = (n_region + lb - 1) // lb
            ach = roof["bytes_per_sample"] * B * n_region / ev_region / 1e9
            cache_resident = bool(table_mb) and table_mb <= 256
            line["roofline_timed_region"] = {
                # the config's own tables (%.0f MB) sit in the 256 MB Infinity Cache: the rows of this region are served by the cache and
                # the fabric, not by HBM -- `frac` is the algorithmic bytes against the 8 TB/s HBM figure for COMPARISON, and may exceed what
                # HBM itself delivers.  The HBM claim is roofline_hbm_resident (tables of 3.2 GB, every row an HBM access).
                "bound": "infinity_cache" if cache_resident else "hbm",
                "kernel": roof["kernel"] + " (multi-batch instantiation, %d batches of %d rows per launch)" % (lb, B),
                "achieved": ach, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach * 1e9 / HBM_PEAK,
                "frac_is": ("algorithmic bytes / time / 8.0 TB/s; the rows come out of the Infinity Cache (%.0f MB of tables), so this is a cache- and "
                            "fabric-side rate quoted against the HBM peak for comparison, NOT an HBM utilisation -- see roofline_hbm_resident" % table_mb)
                           if cache_resident else "algorithmic bytes / time / 8.0 TB/s, tables beyond the Infinity Cache",
                "launches": n_launch, "avg_launch_us": ev_region * 1e6 / n_launch, "timed_with": "HIP events around the median timed region"
                + (" (includes the grouped all-gathers' share)" if dist_on else "")}
        if lb > 16 and roof["kernel"] in ("k_deepfm_v2_joint", "k_rows_chain") and not dist_on and env("SPRK_BENCH_SKIP_16") != "1":
            # rounds 1-2 quoted `value` at 16 batches per launch: kept beside the 64-batch figure
            eng.set_many_batches(16)
            n16 = int(max(64, min(4096, math.ceil(0.02 / max(elapsed / n_region, 1e-9) / 16) * 16)))
            idx = [i % NB for i in range(n16)]
            run16 = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
            run16(); torch.cuda.synchronize()
            s16 = _event_loop(run16, n16)
            eng.set_many_batches(lb)
            line["value_16_batches_per_launch"] = B * world / s16
            line["us_per_step_16_batches_per_launch"] = s16 * 1e6
        line.update(extra)
        if strong_block is not None:
            line["strong_one_global_batch"] = strong_block
        line["scaling_curve"] = ("not measured: no multi-GPU node has run this bench yet (rounds 1-5: one-GPU boxes); the driver computes scaling "
                                 "efficiency from its own --gpus 1,2,4,8 runs")
        hbm_rows = args.hbm_resident if args.hbm_resident >= 0 else 8388608
        if world == 1 and not dist_on and args.workload == "deepfm_v2_c2" and not args.big_vocab and hbm_rows > 0 and roof["kernel"] == "k_deepfm_v2_joint":
            line["roofline_hbm_resident"] = hbm_resident_block(args, B, hbm_rows, K)
        if world == 1 and not dist_on and args.workload == "deepfm_v2_c2" and not args.big_vocab and args.side_workloads:
            line["workloads"] = {w: side_workload(args, w) for w in args.side_workloads.split(",") if w}
        headline = world == 1 and not dist_on and args.workload == "deepfm_v2_c2" and not args.big_vocab and roof["kernel"] == "k_deepfm_v2_joint"
        if headline and args.variants and args.side_workloads:
            line["roofline_variants"] = {
                "zipf": strict_variant_block(args, B, "zipf", {}, "SURVEY 8(d) config 2 (z): ids ~ Zipf(1.05) over a fixed permutation, 2 % of the small slots missing"),
                "f32_mfma": strict_variant_block(args, B, args.dist, {"SPRK_V2_HALF": "0"}, "every contraction on v_mfma_f32_16x16x4_f32 (SPRK_V2_HALF=0): "
                                                 "the exact-fp32 twin of the split-f16 product kernel")}
        # [r5, VERDICT r04 weak 7 / next-round 4b] the driver's record keeps `roofline`, `config` and `cpu_baseline` whole and only the NAMES of the
        # other blocks: the scalars a reader needs from those blocks are repeated here, as plain numbers, inside the two kept ones
        if ws_info:
            big = ws_info["hbm_side_bytes_per_sample"]
            rl.update({"hbm_side_bytes_per_sample": big, "hbm_side_GBps": big * B / fwd_s / 1e9, "working_set_mb": ws_info["working_set_mb"],
                       "input_batches_cycled": ws_info["input_batches_cycled"], "working_set": ws_info["working_set"]})
            line["config"]["working_set_mb"] = ws_info["working_set_mb"]
        rl["strict_samples_per_s"] = B * world / fwd_s
        if "roofline_hbm_resident" in line:
            rl["hbm_resident_frac"] = line["roofline_hbm_resident"]["frac"]
            rl["hbm_resident_us"] = line["roofline_hbm_resident"]["avg_launch_us"]
            # [r6, VERDICT r05 weak 7] the steady-state HBM figure: 3.2 GB of rows, 16 batches per launch (the driver's record dropped it with its block)
            rl["hbm_resident_frac_16_batches"] = line["roofline_hbm_resident"]["frac_16_batches_per_launch"]
            rl["hbm_resident_us_16_batches"] = line["roofline_hbm_resident"]["us_per_step_16_batches_per_launch"]
            rl["hbm_resident_working_set_mb"] = line["roofline_hbm_resident"]["working_set_mb"]
        for vk, vv in (line.get("roofline_variants") or {}).items():
            rl[vk + "_strict_us"] = vv["avg_launch_us"]
            rl[vk + "_strict_frac"] = vv["frac"]
            rl[vk + "_oracle_err"] = vv["oracle_check_max_abs_err"]
        if "roofline_timed_region" in line:
            rl["timed_region_bound"] = line["roofline_timed_region"]["bound"]
        if args.workload == "deepfm_v2_c2":
            # [r6] what the memory system gives to this config's BARE gather (3 random 128-byte lines per sample, no scoring) in steady state --
            # one launch walking 64 batches with persistent waves, flat over 8..32 waves per CU and 1..4 tasks in flight: the floor under
            # ms_per_step at several batches per launch.  Measured constants, not re-measured per run.
            rl["random_line_floor_us_per_step"] = {"config_2_big_tables_51_MB": [3.37, 3.47], "infinity_cache_200_MB": [3.68, 3.82], "hbm_3200_MB": [4.92, 5.32], "window_26_MB": [3.07, 3.14],
                                                   "source": "scripts/ubench/row_gather_steady.hip, profiles/r06/experiments/r06_32"}
        cfgd = line["config"]
        for wn, wb in (line.get("workloads") or {}).items():
            if "roofline" in wb:
                cfgd[wn + "_strict_us"] = wb["roofline"]["avg_launch_us"]
                cfgd[wn + "_strict_frac"] = wb["roofline"]["frac"]
                cfgd[wn + "_strict_samples_per_s"] = wb["value_one_batch_per_launch"]
                cfgd[wn + "_samples_per_s"] = wb["value"]
                cfgd[wn + "_oracle_err"] = wb.get("oracle_check_max_abs_err")
                if "working_set_mb" in wb:                        # [r6] configs 4 / 5: what the loop touches, not the size of the table
                    cfgd[wn + "_working_set_mb"] = wb["working_set_mb"]
                    cfgd[wn + "_input_batches_cycled"] = wb["input_batches_cycled"]
                    cfgd[wn + "_hbm_side_GBps_strict"] = wb["roofline"].get("hbm_side_GBps")
                if "attention_only" in wb["roofline"]:
                    cfgd[wn + "_attention_only_us"] = wb["roofline"]["attention_only"]["avg_launch_us"]
                    cfgd[wn + "_attention_only_frac"] = wb["roofline"]["attention_only"]["frac"]
            elif "rows_per_sec" in wb:
                cfgd[wn + "_rows_per_s"] = wb["rows_per_sec"]
            elif "latency_ms" in wb:
                cfgd[wn + "_p50_ms"] = wb["latency_ms"]["p50"]
                cfgd[wn + "_requests_per_s_one_client"] = wb["requests_per_sec_one_client"]
                if "requests_per_sec_workers" in wb:
                    cfgd[wn + "_requests_per_s_%d_workers_%d_clients" % (wb["workers"], wb["workers_clients"])] = wb["requests_per_sec_workers"]
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.workload, model, feats, args.cpu_seconds)
        if not args.no_hardware_probe:
            line["hardware"] = hardware_block(torch, torch.device("cuda", local_dev))
        # RCCL prints a version banner through C stdio, which (stdout being a pipe) would otherwise be flushed at exit,
        # AFTER this line: flush it first so the JSON is the last line of output
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


# MFMA instructions each fused kernel ISSUES per 16-sample task (or per sample for the wave-per-sample attention kernel), by
# operand type: v_mfma_f32_16x16x4_f32 (2 048 FLOP) / v_mfma_f32_16x16x32_f16 (16 384 FLOP).  Read off the kernels' source.
MFMA_ISSUED = {
    # numerics K = 8 as two steps x 4 n-blocks; deep0's embedding block 4 n-blocks x 3 split products; deep1 4 x 2 K-blocks x 3
    "k_deepfm_pairs": {"per": 16, "f32": 8, "f16": 12 + 24},
    # numerics two steps x 8 n-blocks; second layer 8 n-blocks x 4 K-blocks x 3 split products (every embedding column is folded)
    "k_mlp_rows": {"per": 16, "f32": 16, "f16": 96},
    # round 1: 5 K-chunks x 4 steps x 8 n-blocks on f32 + the same second layer
    # fc0's per-sample part: pooled history (K = 32) + numerics chunk = 3 K-chunks x 4 steps x 8 n-blocks; fc1 4 n-blocks x 4 K-blocks x 3
    "k_din_tail": {"per": 16, "f32": 96, "f16": 48},
    # per SAMPLE: 4 sixteen-row groups x 2 n-blocks x 3 split products
    "k_din_attn": {"per": 1, "f32": 0, "f16": 24},
    # per (16 samples, history slot): two K blocks ([h], [h * c]) x 2 n-blocks x 3 split products; "f16" is filled in per T below
    "k_din_attn_cols": {"per": 16, "f32": 0, "f16_per_slot": 12},
    # [r4] the same formulation in k_din_fused's slot loop; its tail epilogue: numerics two steps x 8 n-blocks on f32; on split f16 the
    # pooled history 8 n-blocks x 3, userId and the candidate's movieId as raw rows 2 x 8 x 3 (emb_dim 17..32; the genre columns
    # arrive as folded rows: no MFMA) and fc1 4 x 4 K-blocks x 3
    "k_din_fused": {"per": 16, "f32": 0, "f16_per_slot": 12},
    "k_din_fused (tail epilogue)": {"per": 16, "f32": 16, "f16": 24 + 48 + 48},
}


def mfma_issued(kernel, reference_flops_per_sample, wide_rows=False, hist_len=None):
    m = MFMA_ISSUED.get(kernel)
    if not m:
        return None
    if "f16_per_slot" in m:
        m = {"per": m["per"], "f32": m["f32"], "f16": m["f16_per_slot"] * (hist_len or 1)}
    if wide_rows and kernel == "k_deepfm_pairs":
        m = {"per": 16, "f32": 8, "f16": 48 + 24}        # emb_dim 64: deep0's embedding block is four K = 32 blocks
    return {"f32_flops_per_sample": m["f32"] * 2048.0 / m["per"], "f16_flops_per_sample": m["f16"] * 16384.0 / m["per"],
            "reference_flops_per_sample": reference_flops_per_sample}


def mfma_block(kernel, m, B, seconds):
    """north_star's "MFMA utilisation on the MLP against gfx950 peak": the MFMA FLOPs the kernel ISSUES (by operand type,
    whole 16-sample tiles) / its duration / the matching dense peak; `frac` = share of the kernel's duration the matrix
    pipe is busy if it ran at peak = f32 time + f16 time.  The PMC view (SQ_VALU_MFMA_BUSY_CYCLES) is in profiles/."""
    f32, f16 = m.get("f32_flops_per_sample", 0.0) * B, m.get("f16_flops_per_sample", 0.0) * B
    t_peak = f32 / MFMA_F32_PEAK + f16 / MFMA_F16_PEAK
    return {"bound": "mfma", "kernel": kernel, "issued_f32_TFLOPs": f32 / seconds / 1e12, "peak_f32_TFLOPs": MFMA_F32_PEAK / 1e12,
            "issued_f16_TFLOPs": f16 / seconds / 1e12, "peak_f16_TFLOPs": MFMA_F16_PEAK / 1e12,
            "frac": t_peak / seconds, "kernel_us": seconds * 1e6,
            "reference_flops_per_sample": m.get("reference_flops_per_sample"),
            "reference_equivalent_TFLOPs": (m.get("reference_flops_per_sample") or 0.0) * B / seconds / 1e12,
            "note": "frac = (issued f32 FLOPs / 157.3 TF + issued f16 FLOPs / 2.5 PF) / kernel time: the share of the kernel the matrix pipe "
                    "would be busy at peak issue rate; the kernels are gather-latency bound, not MFMA bound"}


def hbm_resident_block(args, B, rows, K):
    """The headline graph with identity tables far beyond the 256 MB Infinity Cache (3 x `rows` x 128 B of folded rows):
    every row gather is a real HBM access.  Strict order, one batch per launch.  The loop cycles HBM_RESIDENT_BATCHES (32)
    distinct id batches: 32 x 65 536 x 3 rows x 128 B = 805 MB of distinct row lines between two visits of the same row --
    three times the Infinity Cache (round 2 cycled 8 = 201 MB, which FIT it and read 3.7 points high)."""
    import torch
    nb = max(1, int(os.environ.get("SPRK_BENCH_HBM_BATCHES", HBM_RESIDENT_BATCHES)))
    model, feats, desc, roof = build_workload("deepfm_v2_c2", B, args.dist, seed_offset=7, big_vocab=rows, NB=nb)
    eng = model.engine
    batches = []
    for f in feats:
        ids, dense = model.pack(f)
        batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()))
    outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in batches]
    n = int(max(K, 1000))
    strict_loop(eng, batches, outs, None, n, 1, 0)           # warm-up
    s = strict_loop(eng, batches, outs, None, n, 1, 0)
    ach = roof["bytes_per_sample"] * B / s / 1e9
    eng.set_many_batches(16)
    idx = [i % len(batches) for i in range(n)]
    run = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], None)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run()
    torch.cuda.synchronize()
    ev0.record(); run(); ev1.record()
    torch.cuda.synchronize()
    s16 = ev0.elapsed_time(ev1) * 1e-3 / n
    check = None
    if not args.no_check:
        nchk = 2048
        ref = oracle_forward("deepfm_v2_c2", model, {k: v[:nchk] for k, v in feats[0].items()})[:, 0]
        eng.set_many_batches(1)
        eng.forward(batches[0][0], batches[0][1], outs[0])
        torch.cuda.synchronize()
        check = float(np.abs(outs[0][:nchk].cpu().numpy() - ref).max())
    table_mb = eng.table_bytes() / 1e6
    eng.close()
    return {"bound": "hbm", "kernel": "k_deepfm_v2_joint (one batch of %d rows per launch)" % B, "achieved": ach, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": ach * 1e9 / HBM_PEAK, "algorithmic_bytes_per_sample": roof["bytes_per_sample"], "avg_launch_us": s * 1e6,
            "frac_16_batches_per_launch": roof["bytes_per_sample"] * B / s16 / HBM_PEAK, "us_per_step_16_batches_per_launch": s16 * 1e6,
            "identity_table_rows": rows, "device_table_mb": table_mb, "oracle_check_max_abs_err": check,
            "input_batches_cycled": len(batches),
            "working_set_mb": len(batches) * B * 3 * 128 / 1e6,
            "working_set_note": "distinct 128-byte row lines touched between two visits of the same batch (3 big fields); "
                                "the Infinity Cache holds 256 MB",
            "timed_with": "HIP events, strict order, %d launches" % n}


def strict_variant_block(args, B, dist_name, env_overrides, what, nb=8, n=1500):
    """The headline graph once more, strict order, one batch per launch, under another id distribution or another arithmetic switch
    (the switches are read from the environment at sprk_finalize): SURVEY 8(d) config 2 "(z) Zipf(1.05)", and the all-f32-MFMA
    variant next to the split-f16 one."""
    import torch
    saved = {k: os.environ.get(k) for k in env_overrides}
    os.environ.update(env_overrides)
    try:
        model, feats, desc, roof = build_workload("deepfm_v2_c2", B, dist_name, seed_offset=23, NB=nb)
        eng = model.engine
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    batches = []
    for f in feats:
        ids, dense = model.pack(f)
        batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()))
    outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in batches]
    strict_loop(eng, batches, outs, None, n, 1, 0)
    s = strict_loop(eng, batches, outs, None, n, 1, 0)
    check = None
    if not args.no_check:
        nchk = 2048
        ref = oracle_forward("deepfm_v2_c2", model, {k: v[:nchk] for k, v in feats[0].items()})[:, 0]
        eng.forward(batches[0][0], batches[0][1], outs[0])
        torch.cuda.synchronize()
        check = float(np.abs(outs[0][:nchk].cpu().numpy() - ref).max())
        if not check <= 1e-4:
            raise SystemExit("bench (%s) outputs differ from the oracle: max|err| = %g" % (what, check))
    kern = eng.describe().get("kernel")
    eng.close()
    return {"what": what, "id_distribution": dist_name, "env": env_overrides, "kernel": kern, "avg_launch_us": s * 1e6,
            "frac": roof["bytes_per_sample"] * B / s / HBM_PEAK, "samples_per_s": B / s, "oracle_check_max_abs_err": check,
            "timed_with": "HIP events, strict order, one batch per launch, %d launches over %d input batches" % (n, nb)}


def _event_loop(run, n, loops=3):
    """seconds per step of `run()` (= n steps), HIP events on the current stream, median of `loops`."""
    import torch
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tt = []
    for _ in range(loops):
        torch.cuda.synchronize()
        ev0.record(); run(); ev1.record()
        torch.cuda.synchronize()
        tt.append(ev0.elapsed_time(ev1) * 1e-3 / n)
    return float(np.median(tt))


def serving_workload(args):
    """What the Jetty server sends per recommendation (RecForYouProcess.java:34,113-138): ONE POST of 800 {"userId", "movieId"}
    instances to /v1/models/recmodel:predict, answered by sparrowrecsys_amd.serving.PredictServer in front of NeuralCF on the GPU.
    Latency, not throughput: p50 / p99 over keep-alive requests from one client in its own PROCESS (in this one it would share the
    server's GIL), and beside it the share that is the model: predict() on the packed 800 rows, and the kernel launch alone."""
    import http.client
    import multiprocessing as mp
    import torch
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from bench_serving import _client
    from sparrowrecsys_amd import models as M
    from sparrowrecsys_amd.serving import PredictServer
    n_inst = 800
    model = M.NeuralCF(seed=7)
    srv = PredictServer(model, port=0)
    srv.start()
    rng = np.random.default_rng(1)
    bodies, feats = [], None
    for _ in range(16):
        u = int(rng.integers(1, 30000))
        mids = rng.integers(1, 1000, n_inst)
        bodies.append(json.dumps({"instances": [{"userId": u, "movieId": int(m)} for m in mids]}).encode())
        feats = {"userId": np.full(n_inst, u), "movieId": mids}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    seconds = 1.5
    p_ = ctx.Process(target=_client, args=(srv.port, bodies, seconds, n_inst, 0, q))
    p_.start()
    lat = sorted(q.get(timeout=seconds + 120))
    p_.join(timeout=30)
    srv.close()
    pct = lambda p: lat[min(len(lat) - 1, int(p * len(lat)))] * 1e3
    # [r5] past one GIL: serving.serve_workers -- N front processes on one port (SO_REUSEPORT) in front of ONE engine process -- under N clients
    workers_blk = {}
    n_workers = int(os.environ.get("SPRK_BENCH_SERVING_WORKERS", "12"))
    if n_workers > 1:
        try:
            from bench_serving import _neuralcf
            from sparrowrecsys_amd.serving import serve_workers
            pool = serve_workers(_neuralcf, (), n_workers=n_workers, port=0)
            try:
                qw = ctx.Queue()
                n_clients = 8
                cl = [ctx.Process(target=_client, args=(pool.port, bodies, seconds, n_inst, k, qw)) for k in range(n_clients)]
                for c_ in cl:
                    c_.start()
                latw = []
                for _ in cl:
                    latw.extend(qw.get(timeout=seconds + 180))
                for c_ in cl:
                    c_.join(timeout=30)
                latw.sort()
                workers_blk = {"workers": n_workers, "workers_clients": n_clients, "requests_per_sec_workers": len(latw) / seconds,
                               "latency_ms_workers": {"p50": latw[len(latw) // 2] * 1e3, "p99": latw[min(len(latw) - 1, int(0.99 * len(latw)))] * 1e3},
                               "workers_note": "serving.serve_workers: %d front processes on one port (SO_REUSEPORT) + ONE engine process with the NeuralCF "
                                               "model, %d keep-alive clients in their own processes" % (n_workers, n_clients)}
            finally:
                pool.close()
        except Exception as e:                                   # (reported, not fatal: the one-process figures above stand on their own)
            workers_blk = {"workers_error": "%s: %s" % (type(e).__name__, e)}
    # the model's share: predict() on host arrays (pack -> copy -> forward -> copy back -> id check), and the forward alone
    model.predict(feats)
    t0 = time.perf_counter()
    n_pred = 300
    for _ in range(n_pred):
        model.predict(feats)
    predict_ms = (time.perf_counter() - t0) * 1e3 / n_pred
    ids, dense = model.pack(feats)
    ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    out = torch.empty(n_inst, dtype=torch.float32, device="cuda")
    eng = model.engine

    def fw():
        for _ in range(200):
            eng.forward(ti, td, out)
    fw(); torch.cuda.synchronize()
    fwd_us = _event_loop(fw, 200) * 1e6
    kernel = eng.kernel_name()
    eng.close()
    return {"workload": "NeuralCF behind the TF-Serving-shaped REST shim, %d candidates per request (RecForYouProcess.java:34,113-138)" % n_inst,
            "unit": "ms per request", "higher_is_better": False, "requests": len(lat), "latency_ms": {"p50": pct(0.5), "p90": pct(0.9), "p99": pct(0.99)},
            "requests_per_sec_one_client": len(lat) / seconds, "candidates_per_sec_one_client": len(lat) * n_inst / seconds,
            "model_predict_ms": predict_ms, "forward_launch_us": fwd_us, "kernel": kernel, **workers_blk,
            "shares": "request = HTTP + JSON parse (800 instances) + micro-batcher hand-off + model.predict (pack, copy in, forward, copy out, id "
                      "check) + response formatting; model_predict_ms and forward_launch_us are measured in this process, outside the server",
            "server": "sparrowrecsys_amd.serving.PredictServer (ThreadingHTTPServer + micro-batcher that only waits while another request is arriving)"}


def predict_csv_workload(args, rows=1048576):
    """[r6, VERDICT r05 item 8] The reference's own workflow end to end -- model.predict(get_dataset(csv)) (DeepFM.py:14-22,131-133) -- through
    CTRModel.predict_csv: raw CSV text (the reference's sample rows, tests/golden/test_samples_512.csv, repeated) -> one copy to the device ->
    tokenised and packed there (sprk_pack_csv_device) -> forward over 65 536-row slices in groups of up to 64 per launch -> scores on the host.
    Wall clock of the whole call, best of three, text already in host memory."""
    import torch
    from sparrowrecsys_amd import models as M
    base = open(os.path.join(ROOT, "tests", "golden", "test_samples_512.csv"), "rb").read()
    head, body = base.split(b"\n", 1)
    reps = max(1, rows // 512)
    text = head + b"\n" + body * reps
    n = 512 * reps
    model = M.DeepFMv2(seed=1)
    small = model.predict_csv(base)                               # engine creation, scratch sizing
    best, res = 1e9, None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = model.predict_csv(text)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = bool(np.array_equal(res[:512], small) and np.array_equal(res[-512:], small))
    kernel = model.engine.kernel_name()
    model.engine.close()
    return {"workload": "predict_csv: DeepFM_v2.py on %d rows of the reference's testSamples.csv format (%.0f MB of text), host bytes -> scores on the host" % (n, len(text) / 1e6),
            "unit": "rows/s", "rows": n, "seconds": best, "rows_per_sec": n / best, "kernel": kernel, "repeats_equal_first_block": ok}


def side_workload(args, name):
    """One more workload inside the default driver line: BASELINE.json's metric names DeepFM AND DIN,
    and SURVEY 8(d) config 2 names the pair-dot graph next to the sum-of-squares one.  Same measurements as the headline,
    shorter: `value` = batches scored back to back the way `sprk_forward_many` is meant to be used (several batches per
    launch; DIN: attention and tail of alternating groups on two streams), `roofline` = strict stream order, ONE batch
    per launch, HIP events (what rocprofv3's kernel trace reports for `--launch-batches 1 --overlap-streams 0`)."""
    import torch
    if name == "neuralcf_serving":
        return serving_workload(args)
    if name == "predict_csv":
        return predict_csv_workload(args)
    B = {"din_c3": 32768, "widedeep_c5": 131072}.get(name, 65536)
    NB = 8 if name in ("deepfm_c4", "widedeep_c5") else 16       # (configs 4 / 5: the largest single-GPU forms; 27 M-row / 10 M-bucket tables)
    model, feats, desc, roof = build_workload(name, B, args.dist, seed_offset=11, NB=NB)
    eng = model.engine
    din = name == "din_c3"
    roof["kernel"] = eng.kernel_name() if not din else (eng.describe().get("stage") or roof["kernel"])
    batches = []
    for f in feats:
        ids, dense = model.pack(f)
        batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()))
    NB_host, ws_info = NB, None
    if name in HBM_SIDE:
        batches, ws_info = hbm_cycle(name, model, batches, B)
        NB = len(batches)
    outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(NB)]
    lb = 16 if (roof["kernel"] in ("k_deepfm_pairs", "k_deepfm_v2_joint", "k_rows_chain") or (roof["kernel"] == "k_mlp_rows" and os.environ.get("SPRK_MLP_ROWS_MANY") != "0")
                or (din and eng.kernel_name() in ("k_din_tail", "k_din_fused"))) else 1
    eng.set_many_batches(lb)
    fan = 2 if (din and lb > 1 and eng.set_many_streams(2)) else 0
    ws = torch.empty(max(eng.many_workspace_bytes(B, max(fan, 1) * lb) // 4, 1), dtype=torch.float32, device="cuda")
    # timed: >= 30 ms of back-to-back steps
    n0 = 64
    idx = [i % NB for i in range(n0)]
    run0 = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
    run0(); torch.cuda.synchronize()
    t0 = _event_loop(run0, n0, loops=1)
    n = int(min(20000, max(n0, math.ceil(0.03 / max(t0, 1e-9) / 16) * 16)))
    idx = [i % NB for i in range(n)]
    run = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
    run(); torch.cuda.synchronize()
    step_s = _event_loop(run, n)
    eng.check_ids()
    # strict: one batch per launch, stream order
    n_strict = int(min(8000, max(200, math.ceil(0.02 / max(step_s, 1e-9)))))
    fwd_s = strict_loop(eng, batches, outs, ws, n_strict, lb, fan)
    # oracle check through the same multi-batch call
    check = None
    if not args.no_check:
        m = min(NB_host, max(lb, 2))                             # (the host-drawn batches: the oracle has their features)
        chk = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(m)]
        eng.forward_many([batches[j][0] for j in range(m)], [batches[j][1] for j in range(m)], chk, ws)
        torch.cuda.synchronize()
        nr = min(1024, B)
        check = 0.0
        for j in sorted({0, m - 1}):
            for sl in (slice(0, nr), slice(B - nr, B)):
                ref = oracle_forward(name, model, {k: v[sl] for k, v in feats[j].items()})[:, 0]
                check = max(check, float(np.abs(chk[j][sl].cpu().numpy() - ref).max()))
        if not check <= 1e-4:
            raise SystemExit("bench (%s) outputs differ from the oracle: max|err| = %g" % (name, check))
    blk = {"workload": "%s: %s" % (name, desc), "batch": B, "value": B / step_s, "unit": "samples/s", "ms_per_step": step_s * 1e3,
           "batches_per_launch": lb, "launch_overlap_streams": fan, "steps_timed": n, "input_batches_cycled": NB,
           "value_one_batch_per_launch": B / fwd_s, "kernel": eng.kernel_name(), "oracle_check_max_abs_err": check,
           "device_table_mb": eng.table_bytes() / 1e6}
    if ws_info:
        blk.update(ws_info)
    if din:
        pooled = torch.empty((B, eng.n_aux), dtype=torch.float32, device="cuda")
        n_att = 400

        def att():
            for i in range(n_att):
                eng.din_pool(batches[i % NB][0], pooled)
        att(); torch.cuda.synchronize()
        din_s = _event_loop(att, n_att)
        ach = roof["bytes_per_sample"] * B / din_s / 1e9
        blk["roofline"] = {"bound": "hbm", "kernel": "%s (one batch of %d rows per launch)" % (roof["kernel"], B), "achieved": ach, "peak": HBM_PEAK / 1e9,
                           "unit": "GB/s", "frac": ach * 1e9 / HBM_PEAK, "algorithmic_bytes_per_sample": roof["bytes_per_sample"],
                           "avg_launch_us": din_s * 1e6, "step_us_all_kernels_strict": fwd_s * 1e6,
                           "reference_flops_per_sample": roof["flops_per_sample"],
                           "reference_equivalent_TFLOPs": roof["flops_per_sample"] * B / din_s / 1e12,
                           "timed_with": "HIP events, attention-only loop, strict order, %d launches" % n_att}
        if eng.kernel_name() == "k_din_fused":
            # [r5, VERDICT r04 weak 5] `roofline` is the PRODUCT kernel -- k_din_fused<TAIL>, the whole forward of a batch in one launch --
            # on the step's own bytes (attention stage + tail); the attention-only loop (k_din_fused<TAIL = false>, what sprk_din_pool
            # runs) is the sub-field.  Both are bookkeeping against the HBM peak: the 16.8 MB table lives in L2 / Infinity Cache.
            fb = roof["bytes_per_sample"] + roof.get("tail_bytes_per_sample", 0)
            att_only = {k: blk["roofline"][k] for k in ("kernel", "achieved", "frac", "algorithmic_bytes_per_sample", "avg_launch_us", "timed_with")}
            blk["roofline"].update({"kernel": "k_din_fused<TAIL> (attention + pooling + tail, one launch of %d rows)" % B,
                                    "achieved": fb * B / fwd_s / 1e9, "frac": fb * B / fwd_s / HBM_PEAK, "algorithmic_bytes_per_sample": fb,
                                    "avg_launch_us": fwd_s * 1e6, "timed_with": "HIP events, strict order, one batch per launch, %d launches" % n_strict,
                                    "attention_only": att_only})
        blk["roofline_mfma"] = mfma_block(roof["kernel"], mfma_issued(roof["kernel"], roof["flops_per_sample"], hist_len=roof["hist_len"]), B, din_s)
        if eng.kernel_name() in ("k_din_tail", "k_din_fused"):
            tk = "k_din_tail" if eng.kernel_name() == "k_din_tail" else "k_din_fused (tail epilogue)"
            blk["roofline_mfma_tail"] = mfma_block(tk, mfma_issued(tk, roof["tail_reference_flops"]), B, max(fwd_s - din_s, 1e-9))
        blk["dispatch"] = ("one launch per batch: k_din_fused (attention + pooling + tail); several batches per launch: ONE persistent k_din_fused<MB> "
                           "launch per group of up to 16 batches (tables staged once, every wave walks its tasks), groups alternating over two streams"
                           if eng.kernel_name() == "k_din_fused" else "attention launch -> pooled vectors -> tail launch")
    else:
        ach = roof["bytes_per_sample"] * B / fwd_s / 1e9
        blk["roofline"] = {"bound": "hbm", "kernel": roof["kernel"] + " (one batch of %d rows per launch)" % B, "achieved": ach,
                           "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach * 1e9 / HBM_PEAK,
                           "algorithmic_bytes_per_sample": roof["bytes_per_sample"], "avg_launch_us": fwd_s * 1e6,
                           "timed_with": "HIP events, strict order, %d launches" % n_strict}
        if ws_info:
            # [r6] what part of the step's bytes HBM itself serves: the rows of the table(s) beyond the Infinity Cache, at the strict rate and at
            # the several-batches-per-launch rate; the rest of the algorithmic bytes (small tables, ids, numerics) is cache / fabric traffic
            big = ws_info["hbm_side_bytes_per_sample"]
            blk["roofline"].update({"hbm_side_bytes_per_sample": big, "hbm_side_GBps": big * B / fwd_s / 1e9,
                                    "hbm_side_GBps_many": big * B / step_s / 1e9, "algorithmic_GBps_many": roof["bytes_per_sample"] * B / step_s / 1e9,
                                    "working_set_mb": ws_info["working_set_mb"], "input_batches_cycled": ws_info["input_batches_cycled"],
                                    "bound_detail": "rows of the table(s) beyond the Infinity Cache: %d of the %d algorithmic bytes per sample; the other tables of "
                                                    "this graph fit the cache and stay there whatever the inputs are" % (big, roof["bytes_per_sample"])})
        mi = mfma_issued(roof["kernel"], roof.get("mfma_reference_flops"))
        if mi:
            blk["roofline_mfma"] = mfma_block(roof["kernel"], mi, B, fwd_s)
    eng.close()
    return blk


def dry_run(args, rank, world):
    """Everything around the kernels -- rank start-up, process group (gloo), GroupedScoreGather's grouped all-gather, the
    region timing with max-over-ranks -- with a stand-in forward on CPU tensors.  No throughput claim: the line says so."""
    import torch
    import torch.distributed as dist
    from sparrowrecsys_amd.dist import GroupedScoreGather
    dist_on = world > 1
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    B, K = args.batch or 1024, args.steps
    strong = args.scaling == "strong"
    strong_ok = None
    if strong:
        # the strong-scaling plumbing without a GPU: every rank holds the same global batch, scores its row shard with a stand-in
        # forward (score = 3 x row index + 1), RowShardedPredictor gathers, every rank checks every row
        from sparrowrecsys_amd.dist import RowShardedPredictor
        if B % world:
            raise SystemExit("--scaling strong: the global batch of %d rows does not divide over %d ranks" % (B, world))
        gids = torch.arange(B, dtype=torch.int32).reshape(B, 1)
        if dist_on:
            pred = RowShardedPredictor(lambda i_, d_: i_[:, 0].float() * 3.0 + 1.0)
            allv = pred.predict(gids, gids.float())
        else:
            allv = gids[:, 0].float() * 3.0 + 1.0
        strong_ok = bool(tuple(allv.shape) == (B,) and (allv == torch.arange(B).float() * 3.0 + 1.0).all())
        B = B // world                                     # rows per rank in the loop below
    got, queued = [], []

    def sink(gi, view, nb):
        # every rank's slice of every batch of the group, in step order: rank r wrote r * 1000 + step % 7 into all B slots
        want = [queued.pop(0) for _ in range(nb)]
        ok = tuple(view.shape) == (world, nb, B)
        for r in range(world):
            for j in range(nb):
                ok = ok and bool((view[r, j] == float(r * 1000 + want[j])).all())
        got.append(ok)

    gs = GroupedScoreGather(B, max(1, args.gather_group), "cpu", sink=sink) if dist_on else None
    scratch = torch.empty(B)

    def run_steps(first, count):
        for i in range(first, first + count):
            out = gs.out() if gs is not None else scratch
            out.fill_(float(rank * 1000 + i % 7))
            queued.append(i % 7)
            if gs is not None and gs.full():
                gs.commit()

    def fence():
        if gs is not None:
            gs.flush()
        if dist_on:
            dist.barrier()

    timer = Timer(run_steps, fence, dist_on, False)
    run_steps(0, args.warmup)
    fence()
    walls = [timer.region(K)[0] for _ in range(max(1, args.regions))]
    elapsed = float(np.median(walls))
    if rank == 0:
        print(json.dumps({"metric": "ctr_samples_per_sec", "value": B * world * K / elapsed, "unit": "samples/s", "n_gpus": world,
                          "steps": K, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True,
                          "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": "DRY RUN (no GPU, stand-in forward): launcher / collective / timing plumbing only",
                                     "batch_per_gpu": B, "global_batch": B * world, "collectives": gs.collectives if gs else 0,
                                     "strong_row_shard_gather_ok": strong_ok,
                                     "groups_seen_by_sink": len(got), "sink_content_ok": bool(all(got)),
                                     "group": gs.G if gs else 0}}), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

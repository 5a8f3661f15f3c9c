"""CPU ORACLE (test infrastructure, NOT the product) for the SparrowRecSys CTR forward pass.

A numpy restatement of the reference's TensorFlow/Keras model scripts
(``/root/reference/TFRecModel/src/com/sparrowrecsys/offline/tensorflow/*.py``),
forward only.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``sparrowrecsys_amd`` never does.

Arithmetic owner: the reference computes everything inside TensorFlow (un-vendored
third-party dependency, version unpinned: reference README.md:11 says "TensorFlow
2.0+", the bundled SavedModels were written by TF 2.0.0 / Keras 2.2.4-tf).
TensorFlow is not installable here, so every ``tf.*`` call is restated from its
published semantics (SURVEY.md §8(c) checklist).

PARITY PIN STATUS: the reference has no tests and no golden vectors
("parity unpinned" by the reference itself).  What pins this oracle instead:
(1) tests/test_farmhash_pins.py -- the cross hash (FingerprintCat64 chain, default key, unsigned modulo) reproduces the
known answers of TensorFlow's own sparse_cross_op_test.py (oracle/farmhash64.py).
(2) tests/test_savedmodel_pins.py -- the seven SavedModels the reference EXPORTED
(``webroot/modeldata/{neuralcf/001,002, MLPRec/001..005}/saved_model.pb``) executed op by op by oracle/tf_graph_exec.py
on rows of the reference's testSamples.csv; this oracle agrees to 1e-6: pins categorical_column_with_identity +
embedding_column, categorical_column_with_vocabulary_list + indicator_column, numeric_column, DenseFeatures' name-sorted
concat, Dense, concatenate and Dot to the wiring TensorFlow itself generated.
(3) tests/test_reference_blocks.py -- the reference's OWN model-building lines (all seven scripts: DIN.py, DeepFM.py, DeepFM_v2.py,
WideNDeep.py, NeuralCF.py, EmbeddingMLP.py, DIEN.py with its own attention / GRU_gate_parameter / AUGRU classes), exec-uted untouched
on oracle/keras_shim.py (and on TensorFlow where importable:
tests/golden/make_tf_golden.py); this oracle agrees to 1e-6 with the shim run -- the WIRING of DIN's attention unit /
PReLU shapes / pooling, DeepFM(_v2)'s FM crosses and Wide&Deep's crossed column is the reference's code, not a reading of
it (that run is what found DeepFM.py's two tables per deep key).
(4) tests/test_oracle_pins.py: the trained checkpoints' known answers (ROC-AUC 0.7514 / 0.7321 / 0.7353 / 0.7320).
(5) tests/test_dien_cpu.py: the GRU recurrence of this file AND of the stand-in against torch.nn.GRU (an independent implementation).
STILL WITHOUT A TENSORFLOW-PRODUCED VECTOR: DIN, DeepFM, DeepFM_v2, Wide&Deep, DIEN end to end (the ``unpinned`` tests XFAIL
until tests/golden/refblock_tf_*.npz exist).

Every function cites the reference file:line it follows.  ``dtype`` selects the
arithmetic type: float32 reproduces the reference's fp32 CPU forward, float64 is
the shadow used to bound fp32 round-off in tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# literals of the reference scripts
# --------------------------------------------------------------------------------------
# DeepFM.py:64-66 (identical literal in every script)
GENRE_VOCAB = ['Film-Noir', 'Action', 'Adventure', 'Horror', 'Romance', 'War', 'Comedy', 'Western',
               'Documentary', 'Sci-Fi', 'Drama', 'Thriller', 'Crime', 'Fantasy', 'Animation', 'IMAX',
               'Mystery', 'Children', 'Musical']
# numeric_column list, e.g. DeepFM.py:81-87 (order as written; DenseFeatures re-sorts by name)
NUMERIC_KEYS = ['releaseYear', 'movieRatingCount', 'movieAvgRating', 'movieRatingStddev',
                'userRatingCount', 'userAvgRating', 'userRatingStddev']
USER_GENRE_KEYS = ['userGenre1', 'userGenre2', 'userGenre3', 'userGenre4', 'userGenre5']
MOVIE_GENRE_KEYS = ['movieGenre1', 'movieGenre2', 'movieGenre3']
MOVIE_BUCKETS = 1001   # DeepFM.py:54
USER_BUCKETS = 30001   # DeepFM.py:59
CROSS_HASH_KEY = 0xDECAFCAFFE  # tf.feature_column.crossed_column default hash_key
_U64 = (1 << 64) - 1


# --------------------------------------------------------------------------------------
# feature-column transforms (tf.feature_column semantics)
# --------------------------------------------------------------------------------------
def identity_ids(values, num_buckets: int, key: str = "") -> np.ndarray:
    """categorical_column_with_identity (DeepFM.py:54,59): ids cast to int64; with
    default_value=None an id outside [0, num_buckets) fails TF's
    assert_greater_or_equal_0 / assert_less_than_num_buckets -> here ValueError."""
    ids = np.asarray(values).astype(np.int64)
    if ids.size and (ids.min() < 0 or ids.max() >= num_buckets):
        bad = ids[(ids < 0) | (ids >= num_buckets)][0]
        raise ValueError("%s id %d outside [0, %d)" % (key or "categorical", int(bad), num_buckets))
    return ids


def vocab_ids(values, vocab: Sequence[str] = GENRE_VOCAB) -> np.ndarray:
    """categorical_column_with_vocabulary_list (DeepFM.py:68-69): index = position in the
    list; out-of-vocabulary -> -1 (default_value=-1, num_oov_buckets=0); the empty string is
    dropped by to_sparse_input(ignore_value='') -> also no id.  Returns -1 for both."""
    arr = np.asarray(values)
    if arr.dtype.kind in "iu":
        # already-resolved indices (synthetic benchmarks skip the string lookup): pass through,
        # anything outside the vocabulary is "no id"
        arr = arr.astype(np.int64)
        return np.where((arr < 0) | (arr >= len(vocab)), -1, arr)
    table = {v: i for i, v in enumerate(vocab)}
    out = np.empty(len(values), dtype=np.int64)
    for i, v in enumerate(values):
        if isinstance(v, bytes):
            v = v.decode("utf-8")
        if v is None or (isinstance(v, float) and np.isnan(v)):
            v = ""
        out[i] = table.get(v, -1)
    return out


def embedding_lookup(table: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """embedding_column(col, dim) -> safe_embedding_lookup_sparse(combiner='mean') with exactly
    one id per row: out[b] = table[id[b]] (exact copy); pruned (negative) id -> all-zero row
    (SparseFillEmptyRows + zeros_like select).  Reference: DeepFM.py:55,60,70,75."""
    out = np.zeros((ids.shape[0], table.shape[1]), dtype=table.dtype)
    ok = ids >= 0
    out[ok] = table[ids[ok]]
    return out


def numeric(features: Dict, key: str, dtype) -> np.ndarray:
    """numeric_column (DeepFM.py:81-87): int inputs are cast to float32 by DenseFeatures;
    NA/empty CSV fields were already replaced by 0 in make_csv_dataset (DeepFM.py:15-21)."""
    v = np.asarray(features[key])
    if v.dtype.kind in "OUS":
        v = np.array([0.0 if (x is None or x == "" or x == b"") else float(x) for x in v])
    v = v.astype(np.float64)
    v = np.where(np.isnan(v), 0.0, v)
    return v.astype(np.float32).astype(dtype)


def int_feature(features: Dict, key: str) -> np.ndarray:
    """make_csv_dataset int column: empty field -> 0 (DeepFM.py:15-21, na_value='0')."""
    v = np.asarray(features[key])
    if v.dtype.kind in "OUS":
        v = np.array([0 if (x is None or x == "" or x == b"") else int(float(x)) for x in v], dtype=np.int64)
    elif v.dtype.kind == "f":
        v = np.where(np.isnan(v), 0, v).astype(np.int64)
    return v.astype(np.int64)


def _shift_mix(x: int) -> int:
    return x ^ (x >> 47)


def fingerprint_cat64(fp1: int, fp2: int) -> int:
    """tensorflow/core/platform/fingerprint.h FingerprintCat64 (restated)."""
    k_mul = 0xC6A4A7935BD1E995
    result = (fp1 ^ k_mul) & _U64
    result ^= (_shift_mix((fp2 * k_mul) & _U64) * k_mul) & _U64
    result = (result * k_mul) & _U64
    result = (_shift_mix(result) * k_mul) & _U64
    result = _shift_mix(result)
    return result & _U64


def crossed_bucket(columns: Sequence[np.ndarray], num_buckets: int,
                   hash_key: int = CROSS_HASH_KEY) -> np.ndarray:
    """crossed_column([movie_col, rated_movie], N) (WideNDeep.py:72-73) ->
    sparse_cross_hashed: h = hash_key; for v in columns (order as passed): h =
    FingerprintCat64(h, uint64(v)); bucket = h mod N (uint64 arithmetic).  int64 inputs are
    used raw."""
    n = len(columns[0])
    out = np.empty(n, dtype=np.int64)
    for i in range(n):
        h = hash_key
        for col in columns:
            h = fingerprint_cat64(h, int(col[i]) & _U64)
        out[i] = h % num_buckets
    return out


def crossed_bucket_np(columns: Sequence[np.ndarray], num_buckets: int,
                      hash_key: int = CROSS_HASH_KEY) -> np.ndarray:
    """Vectorised uint64 numpy form of crossed_bucket (same arithmetic, for big batches)."""
    k_mul = np.uint64(0xC6A4A7935BD1E995)
    s47 = np.uint64(47)
    with np.errstate(over="ignore"):
        h = np.full(len(columns[0]), hash_key, dtype=np.uint64)
        for col in columns:
            v = np.asarray(col).astype(np.int64).astype(np.uint64)
            r = h ^ k_mul
            t = v * k_mul
            t = (t ^ (t >> s47)) * k_mul
            r = r ^ t
            r = r * k_mul
            r = (r ^ (r >> s47)) * k_mul
            r = r ^ (r >> s47)
            h = r
        return (h % np.uint64(num_buckets)).astype(np.int64)


# --------------------------------------------------------------------------------------
# layers (tf.keras.layers semantics)
# --------------------------------------------------------------------------------------
def dense(x, kernel, bias, dtype):
    """tf.keras.layers.Dense: y = x @ W + b, W laid out [in, out]; on rank-3 input the last
    axis is contracted (DIN.py:149)."""
    return x.astype(dtype) @ kernel.astype(dtype) + bias.astype(dtype)


def relu(x):
    return np.maximum(x, 0)


def prelu(x, alpha):
    """tf.keras.layers.PReLU: f(x) = max(x,0) + alpha*min(x,0); alpha has the input's shape
    minus the batch axis (DIN.py:150,164,166)."""
    return np.maximum(x, 0) + alpha.astype(x.dtype) * np.minimum(x, 0)


def sigmoid(x):
    """activation='sigmoid' (DeepFM.py:113): 1/(1+exp(-x)), evaluated stably."""
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def dense_features(named_blocks: Dict[str, np.ndarray]) -> Tuple[np.ndarray, Dict[str, Tuple[int, int]]]:
    """tf.keras.layers.DenseFeatures: per-column outputs concatenated along axis 1 in
    COLUMN-NAME order (ASCII sort of ``<key>``, ``<key>_embedding``, ``<key>_indicator``);
    empirically confirmed via the MLPRec/004 checkpoint (tests/test_oracle_pins.py)."""
    names = sorted(named_blocks)
    offs = {}
    o = 0
    cols = []
    for n in names:
        blk = named_blocks[n]
        if blk.ndim == 1:
            blk = blk[:, None]
        offs[n] = (o, o + blk.shape[1])
        o += blk.shape[1]
        cols.append(blk)
    return np.concatenate(cols, axis=1), offs


def _numeric_blocks(features, dtype, keys=NUMERIC_KEYS):
    return {k: numeric(features, k, dtype) for k in keys}


def indicator(ids: np.ndarray, depth: int, dtype) -> np.ndarray:
    """indicator_column(categorical) (DeepFM.py:56,61,71,76): multi-hot float [B, depth] -- with one id per
    row a one-hot; "no id" (-1: out-of-vocabulary / empty string) -> an all-zero row.  The op chain the
    reference's exported graphs hold for it (modeldata/MLPRec/{001,003}/saved_model.pb, executed by
    oracle/tf_graph_exec.py): SparseToDense(default -1) -> OneHot(depth, 1.0, 0.0) -> Sum(axis=-2)."""
    out = np.zeros((ids.shape[0], depth), dtype=dtype)
    ok = (ids >= 0) & (ids < depth)
    out[np.nonzero(ok)[0], ids[ok]] = 1
    return out


def int_vocab_ids(values, vocab_size: int) -> np.ndarray:
    """categorical_column_with_vocabulary_list(key, list(range(vocab_size))) over an INTEGER feature (the movieId /
    userRatedMovieN columns of the exported MLPRec/{001,003} graphs): index = the value itself when it is in the
    list, else -1 (default_value); the "missing" marker to_sparse_input drops for integers is -1."""
    v = np.asarray(values).astype(np.int64)
    return np.where((v < 0) | (v >= vocab_size), -1, v)


# --------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------
def embedding_mlp_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                          movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS) -> np.ndarray:
    """EmbeddingMLP.py:34-77: DenseFeatures(7 numerics + 8 genre embeddings + movieId +
    userId embeddings) -> Dense(128, relu) -> Dense(128, relu) -> Dense(1, sigmoid)."""
    x = _embedding_mlp_body(features, w, dtype, movie_buckets, user_buckets)
    z = dense(x, w["head/kernel"], w["head/bias"], dtype)
    return sigmoid(z).astype(np.float32)


def _embedding_mlp_body(features, w, dtype, movie_buckets, user_buckets):
    blocks = _numeric_blocks(features, dtype)
    for k in USER_GENRE_KEYS + MOVIE_GENRE_KEYS:                       # EmbeddingMLP.py:36-51
        blocks[k + "_embedding"] = embedding_lookup(w["emb/" + k].astype(dtype), vocab_ids(features[k]))
    blocks["movieId_embedding"] = embedding_lookup(                      # EmbeddingMLP.py:53-55
        w["emb/movieId"].astype(dtype), identity_ids(int_feature(features, "movieId"), movie_buckets, "movieId"))
    blocks["userId_embedding"] = embedding_lookup(                       # EmbeddingMLP.py:58-60
        w["emb/userId"].astype(dtype), identity_ids(int_feature(features, "userId"), user_buckets, "userId"))
    x, _ = dense_features(blocks)                                        # EmbeddingMLP.py:73
    x = relu(dense(x, w["dense0/kernel"], w["dense0/bias"], dtype))      # EmbeddingMLP.py:74
    x = relu(dense(x, w["dense1/kernel"], w["dense1/bias"], dtype))      # EmbeddingMLP.py:75
    return x


def feature_column_mlp_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                               numeric_keys: Sequence[str] = (), genre_keys: Sequence[str] = (),
                               int_vocab_keys: Sequence[str] = (), int_vocab_size: int = MOVIE_BUCKETS) -> np.ndarray:
    """Sequential([DenseFeatures(numeric columns + INDICATOR columns), Dense(128, relu), Dense(128, relu),
    Dense(1, sigmoid)]): EmbeddingMLP.py:72-77 with indicator_column(categorical_column_with_vocabulary_list(...))
    where the script now has embedding columns -- the graph of the reference's exported modeldata/MLPRec/{001..004}
    SavedModels (001: 8 genre + 6 integer-vocabulary indicator columns; 003: the same + 8 numeric columns; 002 / 004:
    numeric columns only).  Exists so that the feature-column restatements above (vocab_ids, indicator, numeric,
    dense_features) can be checked against those graphs executed op by op (tests/test_savedmodel_pins.py)."""
    blocks = _numeric_blocks(features, dtype, numeric_keys)
    for k in genre_keys:
        blocks[k + "_indicator"] = indicator(vocab_ids(features[k]), len(GENRE_VOCAB), dtype)
    for k in int_vocab_keys:
        blocks[k + "_indicator"] = indicator(int_vocab_ids(int_feature(features, k), int_vocab_size), int_vocab_size, dtype)
    x, _ = dense_features(blocks)
    i = 0
    while "dense%d/kernel" % i in w:
        x = relu(dense(x, w["dense%d/kernel" % i], w["dense%d/bias" % i], dtype))
        i += 1
    return sigmoid(dense(x, w["head/kernel"], w["head/bias"], dtype)).astype(np.float32)


def wide_n_deep_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                        movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS,
                        cross_buckets=10000, rated_buckets=MOVIE_BUCKETS) -> np.ndarray:
    """WideNDeep.py:72-107: deep = EmbeddingMLP body; wide = indicator(crossed_column(
    [movieId, userRatedMovie1], 10000)); concatenate([deep(128), wide(N)]) -> Dense(1, sigmoid).
    The one-hot x kernel product reduces to head_kernel[128 + bucket].

    Generalisation used by BASELINE config 5: if ``w`` holds ``emb/cross`` ([N, D]) the crossed
    column is an embedding_column instead of an indicator and concatenate([deep, cross_emb])
    feeds the head."""
    deep = _embedding_mlp_body(features, w, dtype, movie_buckets, user_buckets)   # WideNDeep.py:101-103
    movie = identity_ids(int_feature(features, "movieId"), movie_buckets, "movieId")
    rated = identity_ids(int_feature(features, "userRatedMovie1"), rated_buckets, "userRatedMovie1")
    bucket = crossed_bucket_np([movie, rated], cross_buckets)                      # WideNDeep.py:72-73
    hk = w["head/kernel"].astype(dtype)
    n_deep = deep.shape[1]
    if "emb/cross" in w:
        wide = embedding_lookup(w["emb/cross"].astype(dtype), bucket)
        z = np.concatenate([deep, wide], axis=1) @ hk + w["head/bias"].astype(dtype)
    else:
        z = deep @ hk[:n_deep] + hk[n_deep + bucket] + w["head/bias"].astype(dtype)  # WideNDeep.py:105-107
    return sigmoid(z).astype(np.float32)


def neural_cf_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                      movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS) -> np.ndarray:
    """NeuralCF.py:45-53 (neural_cf_model_1, the one line 74 selects and the Jetty server
    calls): concatenate([item_tower, user_tower]) -> Dense(relu)* -> Dense(1, sigmoid)."""
    item = embedding_lookup(w["emb/movieId"].astype(dtype),
                            identity_ids(int_feature(features, "movieId"), movie_buckets, "movieId"))
    user = embedding_lookup(w["emb/userId"].astype(dtype),
                            identity_ids(int_feature(features, "userId"), user_buckets, "userId"))
    x = np.concatenate([item, user], axis=1)                              # NeuralCF.py:48
    i = 0
    while "dense%d/kernel" % i in w:                                       # NeuralCF.py:49-50
        x = relu(dense(x, w["dense%d/kernel" % i], w["dense%d/bias" % i], dtype))
        i += 1
    return sigmoid(dense(x, w["head/kernel"], w["head/bias"], dtype)).astype(np.float32)  # NeuralCF.py:51


def neural_cf2_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                       movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS,
                       with_head: bool = True) -> np.ndarray:
    """NeuralCF.py:57-70 (neural_cf_model_2): per-tower Dense(relu)*, Dot(axes=1), Dense(1,
    sigmoid).  ``with_head=False`` returns the raw dot (the MLPRec/005 checkpoint has no head)."""
    item = embedding_lookup(w["emb/movieId"].astype(dtype),
                            identity_ids(int_feature(features, "movieId"), movie_buckets, "movieId"))
    user = embedding_lookup(w["emb/userId"].astype(dtype),
                            identity_ids(int_feature(features, "userId"), user_buckets, "userId"))
    i = 0
    while "item%d/kernel" % i in w:                                        # NeuralCF.py:58-60
        item = relu(dense(item, w["item%d/kernel" % i], w["item%d/bias" % i], dtype))
        i += 1
    i = 0
    while "user%d/kernel" % i in w:                                        # NeuralCF.py:62-64
        user = relu(dense(user, w["user%d/kernel" % i], w["user%d/bias" % i], dtype))
        i += 1
    dot = np.sum(item * user, axis=1, keepdims=True)                       # NeuralCF.py:66
    if not with_head:
        return dot.astype(np.float32)
    return sigmoid(dense(dot, w["head/kernel"], w["head/bias"], dtype)).astype(np.float32)  # NeuralCF.py:67


# ---- DeepFM (pairwise dots) -----------------------------------------------------------
# Field description shared by both DeepFM variants: (key, kind, vocab size).
DEEPFM_FIELDS = [("movieId", "id", MOVIE_BUCKETS), ("userId", "id", USER_BUCKETS),
                 ("userGenre1", "genre", len(GENRE_VOCAB)), ("movieGenre1", "genre", len(GENRE_VOCAB))]
# DeepFM.py:100-103, concat order of DeepFM.py:111-112
DEEPFM_PAIRS = [("movieId", "userId"), ("movieGenre1", "userGenre1"),
                ("movieGenre1", "userId"), ("movieId", "userGenre1")]
DEEPFM_DEEP_EMB = ["movieId", "userId"]                                    # DeepFM.py:88-89


def _field_ids(features, fields):
    ids = {}
    for key, kind, vocab in fields:
        if kind == "id":
            ids[key] = identity_ids(int_feature(features, key), vocab, key)
        else:
            ids[key] = vocab_ids(features[key])
    return ids


def first_order_offsets(fields) -> Dict[str, int]:
    """Row offset of each field's one-hot block inside DenseFeatures(indicator columns)
    (DeepFM.py:79,97): blocks are name-sorted (``<key>_indicator``).  Reference fields ->
    movieGenre1 0, movieId 19, userGenre1 1020, userId 1039 (total 31 040)."""
    offs = {}
    o = 0
    for name, key, vocab in sorted((k + "_indicator", k, v) for k, _, v in fields):
        offs[key] = o
        o += vocab
    offs["__total__"] = o
    return offs


def deepfm_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                   fields=DEEPFM_FIELDS, pairs=DEEPFM_PAIRS, deep_emb=DEEPFM_DEEP_EMB,
                   literal_one_hot: bool = False, share_deep_tables: bool = False) -> np.ndarray:
    """DeepFM.py:54-113.  first-order = indicator columns fed straight into the output Dense;
    second-order = hand-picked pairwise Dot(axes=1); deep = DenseFeatures(numerics + movieId/
    userId embeddings) -> Dense(64, relu) x2; concatenate([first_order, dots..., deep]) ->
    Dense(1, sigmoid).  ``literal_one_hot`` materialises the [B, 31040] one-hot exactly as the
    reference does (small B only) instead of gathering the kernel rows.

    Two tables per deep key: ``DenseFeatures([movie_emb_col])`` (DeepFM.py:91, FM dots) and
    ``DenseFeatures(deep_feature_columns)`` (DeepFM.py:106, MLP) are two layers, and every DenseFeatures layer
    creates its columns' variables itself (TF feature_column_v2.py _StateManagerImpl.create_variable), so the deep part
    reads ``deep_emb/<key>``.  Tied tables (``emb/<key>`` for both, rounds 1-2's reading) only on request
    (``share_deep_tables=True``, the same switch as ``sparrowrecsys_amd.models.DeepFM``): a weight dict without
    ``deep_emb/<key>`` is an error here as it is there, never a silent fallback."""
    ids = _field_ids(features, fields)
    emb = {k: embedding_lookup(w["emb/" + k].astype(dtype), ids[k]) for k, _, _ in fields}
    if share_deep_tables:
        demb = {k: emb[k] for k in deep_emb}
    else:
        for k in deep_emb:
            if "deep_emb/" + k not in w:
                raise KeyError("missing weight %r: DeepFM.py's deep part owns its own table of %r (DeepFM.py:106); "
                               "pass share_deep_tables=True for tied tables" % ("deep_emb/" + k, k))
        demb = {k: embedding_lookup(w["deep_emb/" + k].astype(dtype), ids[k]) for k in deep_emb}
    offs = first_order_offsets(fields)
    n_fo = offs["__total__"]
    hk = w["head/kernel"].astype(dtype)
    # deep part                                                            DeepFM.py:106-108
    blocks = _numeric_blocks(features, dtype)
    for k in deep_emb:
        blocks[k + "_embedding"] = demb[k]
    x, _ = dense_features(blocks)
    i = 0
    while "deep%d/kernel" % i in w:
        x = relu(dense(x, w["deep%d/kernel" % i], w["deep%d/bias" % i], dtype))
        i += 1
    dots = np.stack([np.sum(emb[a] * emb[b], axis=1) for a, b in pairs], axis=1)  # DeepFM.py:100-103
    B = x.shape[0]
    if literal_one_hot:
        onehot = np.zeros((B, n_fo), dtype=dtype)                           # DeepFM.py:97
        for k, _, _ in fields:
            ok = ids[k] >= 0
            onehot[np.nonzero(ok)[0], offs[k] + ids[k][ok]] = 1
        concat = np.concatenate([onehot, dots, x], axis=1)                  # DeepFM.py:111-112
        z = concat @ hk + w["head/bias"].astype(dtype)
    else:
        fo = np.zeros((B, 1), dtype=dtype)
        for k, _, _ in sorted(fields):
            ok = ids[k] >= 0
            contrib = np.zeros((B, 1), dtype=dtype)
            contrib[ok] = hk[offs[k] + ids[k][ok]]
            fo = fo + contrib
        z = fo + np.concatenate([dots, x], axis=1) @ hk[n_fo:] + w["head/bias"].astype(dtype)
    return sigmoid(z).astype(np.float32)                                    # DeepFM.py:113


# ---- DeepFM v2 (sum-of-squares FM cross) ------------------------------------------------
# order of second_order_cat_columns_emb, DeepFM_v2.py:106-110
DEEPFM_V2_ORDER = ["movieGenre1", "movieId", "userGenre1", "userId"]


def deepfm_v2_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                      fields=DEEPFM_FIELDS, order=DEEPFM_V2_ORDER,
                      return_parts: bool = False):
    """DeepFM_v2.py:60-155.  first order = Dense(1)(one-hot indicators) + Dense(1)(numerics);
    each embedding and the numeric vector get their own Dense(K, no activation) -> stack
    [B, F+1, K]; FM cross = (sum_f v_f)^2 - sum_f v_f^2 (no 1/2, not reduced over K);
    deep = Flatten -> Dense(32, relu) -> Dense(16, relu); concat([first(1), fm(K), deep]) ->
    Dense(1, sigmoid)."""
    ids = _field_ids(features, fields)
    emb = {k: embedding_lookup(w["emb/" + k].astype(dtype), ids[k]) for k, _, _ in fields}
    offs = first_order_offsets(fields)
    B = len(next(iter(ids.values())))
    # first order                                                          DeepFM_v2.py:98-104
    fk = w["fo_cat/kernel"].astype(dtype)
    fo_cat = np.zeros((B, 1), dtype=dtype)
    for k, _, _ in sorted(fields):
        ok = ids[k] >= 0
        contrib = np.zeros((B, 1), dtype=dtype)
        contrib[ok] = fk[offs[k] + ids[k][ok]]
        fo_cat = fo_cat + contrib
    fo_cat = fo_cat + w["fo_cat/bias"].astype(dtype)
    num, _ = dense_features(_numeric_blocks(features, dtype))               # DeepFM_v2.py:100,118
    fo_num = dense(num, w["fo_num/kernel"], w["fo_num/bias"], dtype)
    first = fo_cat + fo_num                                                 # DeepFM_v2.py:104
    # second-order fields                                                  DeepFM_v2.py:106-121
    projected = [dense(emb[k], w["proj/%s/kernel" % k], w["proj/%s/bias" % k], dtype) for k in order]
    projected.append(dense(num, w["proj/num/kernel"], w["proj/num/bias"], dtype))
    stack = np.stack(projected, axis=1)                                     # [B, F+1, K]
    deep = stack.reshape(B, -1)                                             # DeepFM_v2.py:124
    i = 0
    while "deep%d/kernel" % i in w:                                         # DeepFM_v2.py:125-126
        deep = relu(dense(deep, w["deep%d/kernel" % i], w["deep%d/bias" % i], dtype))
        i += 1
    s = stack.sum(axis=1)                                                   # DeepFM_v2.py:147
    fm = s * s - (stack * stack).sum(axis=1)                                # DeepFM_v2.py:148-152
    concat = np.concatenate([first, fm, deep], axis=1)                      # DeepFM_v2.py:154
    out = sigmoid(dense(concat, w["head/kernel"], w["head/bias"], dtype)).astype(np.float32)
    if return_parts:
        return out, {"first": first, "fm": fm, "deep": deep, "stack": stack}
    return out


# ---- DIN ------------------------------------------------------------------------------
def din_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                hist_len: int = 5, movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS,
                return_parts: bool = False):
    """DIN.py:95-167.  Candidate and history ids arrive as numeric (float32) columns with
    default 0 (DIN.py:95-103) and are cast to int by the shared Embedding(1001, D,
    mask_zero=True) (DIN.py:132-137; the mask has no numeric effect: no mask-aware consumer,
    SURVEY.md §8(c).10, so padded id 0 gathers row 0 like any other id).  Activation unit
    DIN.py:139-151: [h-c, h, c, h*c] -> Dense(32) -> PReLU(alpha[T,32]) -> Dense(1, sigmoid);
    weighted SUM pooling over the T slots DIN.py:152-158; tail DIN.py:161-167.

    History comes either as scalar keys userRatedMovie1..T (reference schema; the
    DenseFeatures name sort equals numeric order for T<=9) or as one [B,T] array under
    ``userRatedMovies`` (configs with T>9, natural order)."""
    if "userRatedMovies" in features:
        hist = np.asarray(features["userRatedMovies"]).astype(np.int64)
    else:
        names = sorted("userRatedMovie%d" % (i + 1) for i in range(hist_len))  # DIN.py:126
        hist = np.stack([int_feature(features, n) for n in names], axis=1)
    cand = int_feature(features, "movieId")
    table = w["emb/movie"].astype(dtype)
    if hist.size and (hist.min() < 0 or hist.max() >= table.shape[0]):
        raise ValueError("history id outside [0, %d)" % table.shape[0])
    if cand.size and (cand.min() < 0 or cand.max() >= table.shape[0]):
        raise ValueError("movieId outside [0, %d)" % table.shape[0])
    h = table[hist]                                                         # [B,T,D] DIN.py:134
    c = table[cand]                                                         # [B,D]   DIN.py:136-137
    cr = np.repeat(c[:, None, :], hist.shape[1], axis=1)                    # DIN.py:139
    a = np.concatenate([h - cr, h, cr, h * cr], axis=-1)                    # DIN.py:141-147
    u = dense(a, w["att0/kernel"], w["att0/bias"], dtype)                   # DIN.py:149
    u = prelu(u, w["att_prelu/alpha"])                                      # DIN.py:150
    wgt = sigmoid(dense(u, w["att1/kernel"], w["att1/bias"], dtype))        # [B,T,1] DIN.py:151
    pooled = (h * wgt).sum(axis=1)                                          # DIN.py:152-158
    # user profile / context                                               DIN.py:108-128
    prof_blocks = {k: numeric(features, k, dtype) for k in ["userRatingCount", "userAvgRating", "userRatingStddev"]}
    prof_blocks["userId_embedding"] = embedding_lookup(
        w["emb/userId"].astype(dtype), identity_ids(int_feature(features, "userId"), user_buckets, "userId"))
    prof_blocks["userGenre1_embedding"] = embedding_lookup(w["emb/userGenre1"].astype(dtype),
                                                           vocab_ids(features["userGenre1"]))
    profile, _ = dense_features(prof_blocks)
    ctx_blocks = {k: numeric(features, k, dtype) for k in
                  ["releaseYear", "movieRatingCount", "movieAvgRating", "movieRatingStddev"]}
    ctx_blocks["movieGenre1_embedding"] = embedding_lookup(w["emb/movieGenre1"].astype(dtype),
                                                           vocab_ids(features["movieGenre1"]))
    context, _ = dense_features(ctx_blocks)
    x = np.concatenate([profile, pooled, c, context], axis=1)               # DIN.py:161-162
    x = prelu(dense(x, w["fc0/kernel"], w["fc0/bias"], dtype), w["fc0_prelu/alpha"])   # DIN.py:163-164
    x = prelu(dense(x, w["fc1/kernel"], w["fc1/bias"], dtype), w["fc1_prelu/alpha"])   # DIN.py:165-166
    out = sigmoid(dense(x, w["head/kernel"], w["head/bias"], dtype)).astype(np.float32)  # DIN.py:167
    if return_parts:
        return out, {"pooled": pooled, "att": wgt[..., 0]}
    return out


def dien_forward(features: Dict, w: Dict[str, np.ndarray], dtype=np.float32,
                 hist_len: int = 5, movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS,
                 return_parts: bool = False):
    """DIEN.py:114-250, the y_pred output (the second model output, the auxiliary loss of DIEN.py:253-292, needs labels
    and sampled negatives and is training-only).  PARITY: the wiring is pinned by the reference's own lines (DIEN.py:51-296
    executed on oracle/keras_shim.py, tests/golden/refblock_shim_dien.npz: 6e-8); the arithmetic of Keras' GRU is restated
    from its source in both places; no TensorFlow-produced vector (tests/test_reference_blocks.py ..._unpinned_...).

    * ids as numeric columns with default 0 -> shared Embedding(1001, D, mask_zero=True) (DIEN.py:111-121,163-169).
    * tf.keras.layers.GRU(D, return_sequences=True) (DIEN.py:173) -- TF2 defaults: reset_after=True, tanh / sigmoid,
      kernel [D, 3D] and recurrent_kernel [D, 3D] in z | r | h order, bias [2, 3D] (input row, recurrent row):
          z = sig(x Wz + bz + h Uz + cz)   r = sig(x Wr + br + h Ur + cr)   hh = tanh(x Wh + bh + r * (h Uh + ch))
          h' = z * h + (1 - z) * hh
      and, unlike DIN, the Embedding's mask IS consumed here ([TF-MEM]: the GRU is the first mask-aware layer down
      stream of mask_zero=True; Keras' masked RNN step keeps the previous state and REPEATS the previous output --
      zeros before the first unmasked step -- at every slot whose id is 0).  The custom layers after it do not
      support masking, so the mask stops there.
    * attention (DIEN.py:175-206): a_t = sigmoid(Dense(1)(sigmoid(Dense(32)(g_t * c)))) -- a plain per-slot gate in
      (0, 1), no softmax; repeated over the D lanes.
    * AUGRU (DIEN.py:210-250), every gate a GRU_gate_parameter: pre = input_w(g_t) + hidden_w(h) (hidden_w without
      bias), r = sigmoid(Dense_r(pre_r)), z likewise, h~ = tanh(Dense_h(input_w_h(g_t) + hidden_w_h(h * z)));
      u = a_t * r;  h <- (1 - u) * h + u * h~.  Note the attention scales the gate named R_t and the candidate
      state uses Z_t -- restated as written, not as in the DIEN paper.
      The initial state is `GlorotUniform()(shape=(1, D))` evaluated INSIDE call() (DIEN.py:239-240): the reference
      draws a fresh random h_0 per forward pass, so its predictions are not reproducible; here h_0 is the explicit
      weight ``augru/h0`` [1, D].
    * tail (DIEN.py:252-259): concat [augru, candidate, user profile, context] -> Dense(128) PReLU Dense(64) PReLU
      Dense(1, sigmoid)."""
    if "userRatedMovies" in features:
        hist = np.asarray(features["userRatedMovies"]).astype(np.int64)
    else:
        names = sorted("userRatedMovie%d" % (i + 1) for i in range(hist_len))
        hist = np.stack([int_feature(features, n) for n in names], axis=1)
    cand = int_feature(features, "movieId")
    table = w["emb/movie"].astype(dtype)
    if hist.size and (hist.min() < 0 or hist.max() >= table.shape[0]):
        raise ValueError("history id outside [0, %d)" % table.shape[0])
    if cand.size and (cand.min() < 0 or cand.max() >= table.shape[0]):
        raise ValueError("movieId outside [0, %d)" % table.shape[0])
    B, T = hist.shape
    D = table.shape[1]
    x = table[hist]                                                         # [B,T,D] DIEN.py:167
    c = table[cand]                                                         # [B,D]   DIEN.py:168-171
    mask = hist != 0                                                        # Embedding.compute_mask
    # ---- GRU (DIEN.py:173) ----
    Wk, Uk, bk = w["gru/kernel"].astype(dtype), w["gru_rec/kernel"].astype(dtype), w["gru/bias"].astype(dtype)
    h = np.zeros((B, D), dtype=dtype)
    prev_out = np.zeros((B, D), dtype=dtype)
    g = np.zeros((B, T, D), dtype=dtype)
    for t in range(T):
        mx = x[:, t, :] @ Wk + bk[0]
        mh = h @ Uk + bk[1]
        z = sigmoid(mx[:, :D] + mh[:, :D])
        r = sigmoid(mx[:, D:2 * D] + mh[:, D:2 * D])
        hh = np.tanh(mx[:, 2 * D:] + r * mh[:, 2 * D:])
        hn = z * h + (1 - z) * hh
        m = mask[:, t][:, None]
        h = np.where(m, hn, h)
        prev_out = np.where(m, hn, prev_out)
        g[:, t, :] = prev_out
    # ---- attention gate (DIEN.py:193-203) ----
    a = sigmoid(dense(sigmoid(dense(g * c[:, None, :], w["att0/kernel"], w["att0/bias"], dtype)),
                      w["att1/kernel"], w["att1/bias"], dtype))[..., 0]      # [B,T]
    # ---- AUGRU (DIEN.py:238-248) ----
    def gate(name, gt, hid, act):
        pre = dense(gt, w["augru_%s_in/kernel" % name], w["augru_%s_in/bias" % name], dtype) \
            + hid @ w["augru_%s_hid/kernel" % name].astype(dtype)
        return act(dense(pre, w["augru_%s_out/kernel" % name], w["augru_%s_out/bias" % name], dtype))
    hs = np.repeat(w["augru/h0"].astype(dtype).reshape(1, D), B, axis=0)
    for t in range(T):
        gt = g[:, t, :]
        r_t = gate("r", gt, hs, sigmoid)
        z_t = gate("z", gt, hs, sigmoid)
        h_next = gate("h", gt, hs * z_t, np.tanh)
        u = a[:, t][:, None] * r_t
        hs = (1 - u) * hs + u * h_next
    # ---- profile / context / tail (DIEN.py:131-150,252-259) ----
    prof_blocks = {k: numeric(features, k, dtype) for k in ["userRatingCount", "userAvgRating", "userRatingStddev"]}
    prof_blocks["userId_embedding"] = embedding_lookup(
        w["emb/userId"].astype(dtype), identity_ids(int_feature(features, "userId"), user_buckets, "userId"))
    prof_blocks["userGenre1_embedding"] = embedding_lookup(w["emb/userGenre1"].astype(dtype),
                                                           vocab_ids(features["userGenre1"]))
    profile, _ = dense_features(prof_blocks)
    ctx_blocks = {k: numeric(features, k, dtype) for k in
                  ["releaseYear", "movieRatingCount", "movieAvgRating", "movieRatingStddev"]}
    ctx_blocks["movieGenre1_embedding"] = embedding_lookup(w["emb/movieGenre1"].astype(dtype),
                                                           vocab_ids(features["movieGenre1"]))
    context, _ = dense_features(ctx_blocks)
    y = np.concatenate([hs, c, profile, context], axis=1)                   # DIEN.py:252
    y = prelu(dense(y, w["fc0/kernel"], w["fc0/bias"], dtype), w["fc0_prelu/alpha"])
    y = prelu(dense(y, w["fc1/kernel"], w["fc1/bias"], dtype), w["fc1_prelu/alpha"])
    out = sigmoid(dense(y, w["head/kernel"], w["head/bias"], dtype)).astype(np.float32)
    if return_parts:
        return out, {"augru": hs, "att": a, "gru": g}
    return out


FORWARDS = {
    "embedding_mlp": embedding_mlp_forward,
    "wide_n_deep": wide_n_deep_forward,
    "neural_cf": neural_cf_forward,
    "neural_cf2": neural_cf2_forward,
    "deepfm": deepfm_forward,
    "deepfm_v2": deepfm_v2_forward,
    "din": din_forward,
    "dien": dien_forward,
}

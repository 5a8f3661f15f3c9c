"""MovieLens feature-column schema of the reference models and the host-side packing that
turns a ``{feature name: column}`` dict into the two device arrays the HIP kernels read:

* ``ids   [B, F] int32`` -- one column per categorical slot, ``-1`` = missing / out-of-vocabulary
* ``dense [B, 7] float32`` -- the numeric columns in name-sorted order

Strings never reach the GPU: the 19-entry genre vocabulary is resolved here.

Reference: the Keras ``inputs`` dicts (TFRecModel/src/com/sparrowrecsys/offline/tensorflow/
DeepFM.py:30-51, DIN.py:34-59, NeuralCF.py:38-41), the CSV reader ``get_dataset``
(DeepFM.py:14-22: ``na_value="0"``, empty int -> 0, empty string -> "") and the feature columns
(DeepFM.py:54-76).  The CSV header is webroot/sampledata/testSamples.csv:1, produced by
src/main/java/com/sparrowrecsys/offline/spark/featureeng/FeatureEngForRecModel.scala:28-124.
"""
from __future__ import annotations

import csv
from typing import Dict, Iterable, List, Mapping, Sequence, Tuple

import numpy as np

# DeepFM.py:64-66 (same literal in every script); index = position, anything else is OOV
GENRE_VOCAB = ['Film-Noir', 'Action', 'Adventure', 'Horror', 'Romance', 'War', 'Comedy', 'Western',
               'Documentary', 'Sci-Fi', 'Drama', 'Thriller', 'Crime', 'Fantasy', 'Animation', 'IMAX',
               'Mystery', 'Children', 'Musical']
_GENRE_INDEX = {g: i for i, g in enumerate(GENRE_VOCAB)}
_GENRE_INDEX.update({g.encode(): i for i, g in enumerate(GENRE_VOCAB)})

FLOAT_KEYS = ['movieAvgRating', 'movieRatingStddev', 'userAvgRating', 'userRatingStddev']
INT_NUMERIC_KEYS = ['movieRatingCount', 'userRatingCount', 'releaseYear']
# DenseFeatures concatenates columns in name order: this is the order of the packed dense matrix
NUMERIC_KEYS = sorted(FLOAT_KEYS + INT_NUMERIC_KEYS)
USER_GENRE_KEYS = ['userGenre1', 'userGenre2', 'userGenre3', 'userGenre4', 'userGenre5']
MOVIE_GENRE_KEYS = ['movieGenre1', 'movieGenre2', 'movieGenre3']
HISTORY_KEYS = ['userRatedMovie1', 'userRatedMovie2', 'userRatedMovie3', 'userRatedMovie4', 'userRatedMovie5']

MOVIE_BUCKETS = 1001   # DeepFM.py:54
USER_BUCKETS = 30001   # DeepFM.py:59
N_GENRES = len(GENRE_VOCAB)


class IdColumn:
    """One column of the packed ids matrix.

    kind 'id'    categorical_column_with_identity: int, missing -> 0 (CSV default), value outside
                 [0, vocab) -> ValueError (TF: InvalidArgumentError from assert_less_than_num_buckets)
    kind 'genre' categorical_column_with_vocabulary_list(GENRE_VOCAB): string -> index, OOV/empty -> -1
    """
    __slots__ = ("key", "kind", "vocab")

    def __init__(self, key: str, kind: str, vocab: int):
        if kind not in ("id", "genre"):
            raise ValueError("unknown id column kind %r" % kind)
        self.key, self.kind, self.vocab = key, kind, int(vocab)

    def __repr__(self):
        return "IdColumn(%r, %r, %d)" % (self.key, self.kind, self.vocab)


def _is_missing(v) -> bool:
    if v is None:
        return True
    if isinstance(v, (str, bytes)):
        return len(v) == 0
    if isinstance(v, float):
        return v != v
    return False


def _as_list(col):
    if hasattr(col, "detach"):          # torch tensor
        col = col.detach().cpu().numpy()
    if isinstance(col, np.ndarray):
        return col
    return np.asarray(list(col), dtype=object) if len(col) and isinstance(col[0], (str, bytes, type(None))) \
        else np.asarray(col)


def to_int_column(col, key: str = "") -> np.ndarray:
    """int32 CSV semantics: missing/empty/NaN -> 0."""
    a = _as_list(col)
    if a.dtype.kind in "iu":
        return a.astype(np.int64)
    if a.dtype.kind == "f":
        return np.where(np.isnan(a), 0, a).astype(np.int64)
    if a.dtype.kind == "b":
        return a.astype(np.int64)
    out = np.empty(len(a), dtype=np.int64)
    for i, v in enumerate(a):
        out[i] = 0 if _is_missing(v) else int(float(v))
    return out


def to_float_column(col, key: str = "") -> np.ndarray:
    """float32 CSV semantics: missing/empty/NaN -> 0.0; ints are cast (DenseFeatures casts to float32)."""
    a = _as_list(col)
    if a.dtype.kind in "iub":
        return a.astype(np.float32)
    if a.dtype.kind == "f":
        return np.where(np.isnan(a), 0.0, a).astype(np.float32)
    out = np.empty(len(a), dtype=np.float32)
    for i, v in enumerate(a):
        out[i] = 0.0 if _is_missing(v) else float(v)
    return out


def _genre_of(v) -> int:
    if isinstance(v, np.bytes_):
        v = bytes(v)
    elif isinstance(v, np.str_):
        v = str(v)
    return -1 if _is_missing(v) else _GENRE_INDEX.get(v, -1)


def to_genre_index(col) -> np.ndarray:
    a = _as_list(col)
    if a.dtype.kind in "iu":           # already indices
        return a.astype(np.int64)
    if len(a) >= (1 << 18):
        # a column of strings holds a handful of distinct values: factorise (hash based, C speed), resolve each distinct
        # value once, look the rest up -- 0.5 s -> 0.1 s per million rows against the per-element loop (only for columns long
        # enough to repay importing pandas)
        try:
            import pandas as pd
            codes, uniques = pd.factorize(a, use_na_sentinel=True)      # None / NaN -> code -1
            lut = np.fromiter((_genre_of(u) for u in uniques), dtype=np.int64, count=len(uniques))
            out = np.full(len(a), -1, dtype=np.int64)
            ok = codes >= 0
            out[ok] = lut[codes[ok]]
            return out
        except Exception:               # (unhashable elements, pandas missing: the loop decides)
            pass
    out = np.empty(len(a), dtype=np.int64)
    for i, v in enumerate(a):
        out[i] = _genre_of(v)
    return out


def batch_size_of(features: Mapping) -> int:
    for v in features.values():
        return len(v)
    return 0


def pack_ids(features: Mapping, columns: Sequence[IdColumn]) -> np.ndarray:
    """-> ids [B, F] int32."""
    B = batch_size_of(features)
    ids = np.empty((B, len(columns)), dtype=np.int32)
    for j, col in enumerate(columns):
        if col.key not in features:
            raise KeyError("missing input feature %r" % col.key)
        if col.kind == "genre":
            v = to_genre_index(features[col.key])
            v = np.where((v < 0) | (v >= col.vocab), -1, v)
        else:
            v = to_int_column(features[col.key], col.key)
            if v.size and (v.min() < 0 or v.max() >= col.vocab):
                bad = v[(v < 0) | (v >= col.vocab)][0]
                raise ValueError("%s id %d outside [0, %d) (reference: assert_less_than_num_buckets)"
                                 % (col.key, int(bad), col.vocab))
        ids[:, j] = v
    return ids


def pack_dense(features: Mapping, keys: Sequence[str] = NUMERIC_KEYS) -> np.ndarray:
    """-> dense [B, len(keys)] float32 in the given (name-sorted) key order."""
    B = batch_size_of(features)
    dense = np.empty((B, len(keys)), dtype=np.float32)
    for j, k in enumerate(keys):
        if k not in features:
            raise KeyError("missing input feature %r" % k)
        dense[:, j] = to_float_column(features[k], k)
    return dense


def read_samples_csv(path: str, limit: int = None) -> Dict[str, np.ndarray]:
    """CSV -> dict of raw string columns (empty fields stay ""), the input of pack_* and of the
    oracle.  Mirrors what make_csv_dataset hands the model (DeepFM.py:14-22) minus batching and
    shuffling."""
    with open(path, newline="") as f:
        reader = csv.reader(f)
        header = next(reader)
        cols: List[List[str]] = [[] for _ in header]
        for n, row in enumerate(reader):
            if limit is not None and n >= limit:
                break
            if len(row) != len(header):
                continue                      # ignore_errors=True
            for c, v in zip(cols, row):
                c.append(v)
    return {h: np.asarray(c, dtype=object) for h, c in zip(header, cols)}


def iter_feature_batches(x, batch_size: int = None) -> Iterable[Tuple[Mapping, int]]:
    """Normalise what ``predict`` accepts -- a dict of columns, or an iterable of dicts /
    ``(dict, label)`` tuples like a tf.data.Dataset -- into feature dicts."""
    if isinstance(x, Mapping):
        B = batch_size_of(x)
        if batch_size is None or batch_size >= B:
            yield x
        else:
            for s in range(0, B, batch_size):
                yield {k: v[s:s + batch_size] for k, v in x.items()}
        return
    for item in x:
        if isinstance(item, tuple):
            item = item[0]
        if not isinstance(item, Mapping):
            raise TypeError("predict() expects a dict of feature columns or an iterable of such dicts")
        yield item

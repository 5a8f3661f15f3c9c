"""Host ingest: CSV text -> the packed ``ids [B,F] int32`` / ``dense [B,N] float32`` arrays in ONE native pass
(``sprk_pack_csv`` in libsparrow_hip.so, plain C++: no GPU involved).

The reference reads its sample files with ``tf.data.experimental.make_csv_dataset(..., na_value="0",
ignore_errors=True)`` (DeepFM.py:14-22) and resolves the feature columns inside the graph (DeepFM.py:54-76);
``schema.read_samples_csv`` + ``pack_ids`` + ``pack_dense`` restate that in Python at ~0.1 M rows/s, which is four
orders of magnitude below what the forward consumes.  Same semantics here: empty int -> 0,
empty float -> 0.0, genre string -> vocabulary position or -1, rows of the wrong width dropped, identity ids outside
their bucket range -> ``ValueError``.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np

from . import _lib as L
from .schema import IdColumn, NUMERIC_KEYS


def pack_csv(text, id_columns: Sequence[IdColumn], numeric_keys: Sequence[str] = NUMERIC_KEYS,
             max_rows: int = None, threads: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """``text``: the CSV file's content (bytes or str, header line first) -> ``(ids, dense)``.
    ``threads`` > 1: the body is cut at line boundaries and parsed on that many host threads (``sprk_pack_csv_mt``);
    outputs and the first error are identical for any thread count."""
    if isinstance(text, str):
        text = text.encode("utf-8")
    lib = L.load_library()
    if max_rows is None:
        max_rows = text.count(b"\n") + 1
    n_id, n_dense = len(id_columns), len(numeric_keys)
    cols = (L.CsvCol * max(n_id, 1))()
    for j, c in enumerate(id_columns):
        cols[j].name, cols[j].kind, cols[j].vocab = c.key.encode(), 1 if c.kind == "genre" else 0, c.vocab
    names = (C.c_char_p * max(n_dense, 1))(*[k.encode() for k in numeric_keys])
    ids = np.empty((max_rows, n_id), dtype=np.int32)
    dense = np.empty((max_rows, n_dense), dtype=np.float32)
    rows = C.c_int32(0)
    L.check(lib.sprk_pack_csv_mt(text, C.c_size_t(len(text)), cols, n_id, names, n_dense, max_rows, int(max(1, threads)),
                                 C.c_void_p(ids.ctypes.data), C.c_void_p(dense.ctypes.data), C.byref(rows)))
    return ids[:rows.value], dense[:rows.value]


def pack_csv_file(path: str, id_columns: Sequence[IdColumn], numeric_keys: Sequence[str] = NUMERIC_KEYS,
                  max_rows: int = None, threads: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "rb") as f:
        return pack_csv(f.read(), id_columns, numeric_keys, max_rows, threads)

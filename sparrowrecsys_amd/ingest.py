"""Ingest: CSV text -> the packed ``ids [B,F] int32`` / ``dense [B,N] float32`` arrays in ONE native pass, on the host
(``sprk_pack_csv`` in libsparrow_hip.so, plain C++: no GPU involved) or on the device (``sprk_pack_csv_device``: HIP kernels
over the raw text in HBM, same bits).

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


def pack_csv_device(text, id_columns: Sequence[IdColumn], numeric_keys: Sequence[str] = NUMERIC_KEYS, max_rows: int = None,
                    stream=None):
    """The same packing on the GPU (``sprk_pack_csv_device``: k_csv_pack.h): ``text`` is the CSV content as bytes / str (copied
    to the device once, as raw text) or a ``torch.uint8`` device tensor already holding it; returns DEVICE tensors
    ``(ids [rows, F] int32, dense [rows, N] float32)`` ready for ``Engine.forward``.  Bit-identical to :func:`pack_csv` for the
    values CSV sample files hold (plain decimals of at most 15 digits); quoted fields or other number spellings raise
    (``SparrowHipError``) and name the row -- use :func:`pack_csv` for such files.  There is no silent host fallback."""
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("pack_csv_device needs a HIP device (use pack_csv for the host tokenizer)")
    if isinstance(text, str):
        text = text.encode("utf-8")
    if isinstance(text, (bytes, bytearray, memoryview)):
        n = len(text)
        if n:
            import warnings
            with warnings.catch_warnings():                       # (a read-only view of the bytes: it is only copied from)
                warnings.simplefilter("ignore")
                host = torch.from_numpy(np.frombuffer(text, dtype=np.uint8))
        else:
            host = torch.empty(0, dtype=torch.uint8)
        buf = torch.empty(n + 16, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))   # (256-byte aligned)
        buf[:n].copy_(host, non_blocking=False)
    else:
        if text.dtype != torch.uint8 or not text.is_cuda or not text.is_contiguous() or text.data_ptr() % 16:
            raise ValueError("pack_csv_device: a contiguous, 16-byte aligned torch.uint8 device tensor is required")
        buf, n = text, text.numel()
    lib = L.load_library()
    if max_rows is None:
        max_rows = int((buf[:n] == 10).sum().item()) + 1 if n else 0
    n_id, n_dense = len(id_columns), len(numeric_keys)
    cols = (L.CsvCol * max(n_id, 1))()
    for j, c in enumerate(id_columns):
        cols[j].name, cols[j].kind, cols[j].vocab = c.key.encode(), 1 if c.kind == "genre" else 0, c.vocab
    names = (C.c_char_p * max(n_dense, 1))(*[k.encode() for k in numeric_keys])
    ids = torch.empty((max(max_rows, 1), n_id), dtype=torch.int32, device=buf.device)
    dense = torch.empty((max(max_rows, 1), n_dense), dtype=torch.float32, device=buf.device)
    rows = C.c_int32(0)
    if stream is None:
        stream = torch.cuda.current_stream(buf.device).cuda_stream
    # the library launches on (and allocates its scratch on) the CURRENT device: make that the text's device
    with torch.cuda.device(buf.device):
        L.check(lib.sprk_pack_csv_device(C.c_void_p(buf.data_ptr()), C.c_size_t(n), cols, n_id, names, n_dense, int(max_rows),
                                         C.c_void_p(ids.data_ptr()), C.c_void_p(dense.data_ptr()), C.byref(rows), C.c_void_p(stream)))
    return ids[:rows.value], dense[:rows.value]


def last_device_path() -> int:
    """Which kernels this thread's last ``pack_csv_device`` ran: 1 = the optimistic pass alone (every line a row), 2 = the
    exact keep -> scan -> parse sequence (``sprk_csv_last_path``)."""
    return int(L.load_library().sprk_csv_last_path())


def read_csv_to_device(path: str):
    """The file's bytes as a ``torch.uint8`` device tensor (16 spare bytes behind the text): read straight into pinned host
    memory (one copy out of the page cache) and sent with one asynchronous host -> device copy -- what ``pack_csv_device``
    wants as its input; ``(buffer, n_bytes)``."""
    import os

    import numpy as np
    import torch
    n = os.path.getsize(path)
    host = torch.empty(n + 16, dtype=torch.uint8).pin_memory()
    view = host.numpy()
    with open(path, "rb", buffering=0) as f:
        got = 0
        while got < n:
            k = f.readinto(memoryview(view)[got:n])
            if not k:
                break
            got += k
    if got != n:
        raise IOError("%s: read %d of %d bytes" % (path, got, n))
    view[n:] = 0
    dev = torch.empty(n + 16, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
    dev.copy_(host, non_blocking=True)
    torch.cuda.current_stream().synchronize()                     # the pinned buffer may be released when this returns
    return dev, n

"""Plan builder: turns a model description + reference-layout weights into the ``sprk_plan`` the
HIP engine executes and the device-layout weight slots it uploads.

Device layouts (DESIGN.md "Data layout in HBM"):

* embedding table ``[V, D]``  ->  ``[V, Dp]`` with ``Dp = ceil(D/4)*4`` zero-padded floats per row, so
  every row is a whole number of 16-byte lanes (gathered values are copied bit-exactly);
* first-order / wide weights (rows of a Dense(1) kernel that multiply a one-hot block)  ->  a
  ``[V]`` float vector per field (the ``[B, 31040]`` one-hot of DeepFM.py:97 is never built);
* Dense kernel ``[in, out]``  ->  ``W^T`` as ``[Np, K]`` (``Np = ceil(out/16)*16`` rows, ``K`` =
  width of the LDS slice the layer reads, columns permuted to where the plan placed each input
  in LDS, zero elsewhere); bias / PReLU alpha padded to ``Np``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib as L
from .schema import IdColumn


def pad4(n: int) -> int:
    return (n + 3) & ~3


def pad16(n: int) -> int:
    return (n + 15) & ~15


class DeviceTable:
    """An embedding table that is ALREADY in the device layout -- ``[V + 1, Dp]`` float32, rows padded to 16-byte lanes, row V all
    zero -- in device memory its owner keeps alive: ``pad_table`` hands it through and the engine takes it with
    ``sprk_upload_external`` (no copy).  Two uses: a table too large to duplicate (BASELINE config 4's 27 M x 64 is 6.9 GB; the
    padded copy plus ``sprk_upload``'s copy made three of them), and a ROW-SHARDED table (``sparrowrecsys_amd.dist.ShardedTable``:
    one virtual range over every rank's shard, include/sparrow_hip.h sprk_vtable_*).  Quacks like the ``[V, D]`` weight it stands
    for where the host code looks at weights: ``shape``, ``data_ptr()``, ``device``, row indexing (what bench.py's oracle check
    pulls)."""

    def __init__(self, tensor, vocab: int, dim: int, keepalive=None):
        if tuple(tensor.shape) != (vocab + 1, pad4(dim)) or not tensor.is_contiguous() or str(tensor.dtype) != "torch.float32":
            raise ValueError("DeviceTable wants a contiguous float32 [vocab + 1, pad4(dim)] device tensor, got %s %s" % (tuple(tensor.shape), tensor.dtype))
        self.tensor, self.vocab, self.dim, self.keepalive = tensor, int(vocab), int(dim), keepalive
        import weakref
        self.engines = weakref.WeakSet()                        # live engines that took this table with sprk_upload_external

    @classmethod
    def from_rows(cls, table):
        """A [V, D] CUDA tensor -> device layout, ONE copy (instead of pad_table's copy plus sprk_upload's)."""
        import torch
        V, D = table.shape
        out = torch.zeros((V + 1, pad4(D)), dtype=torch.float32, device=table.device)
        out[:V, :D] = table
        return cls(out, V, D)

    @property
    def shape(self):
        return (self.vocab, self.dim)

    @property
    def device(self):
        return self.tensor.device

    @property
    def nbytes(self):
        return self.tensor.numel() * 4

    def data_ptr(self):
        return self.tensor.data_ptr()

    def __getitem__(self, idx):
        return self.tensor[:self.vocab, :self.dim][idx]

    def cpu(self):
        return self.tensor[:self.vocab, :self.dim].cpu()


def pad_table(table):
    """[V, D] -> device layout [V+1, Dp]: rows padded to whole 16-byte lanes, plus ONE all-zero row at
    index V.  The fused kernels point a missing / out-of-vocabulary id at that row, so "no id ->
    zero vector" (safe_embedding_lookup_sparse) costs no select; valid rows are copied bit-exactly.
    A torch CUDA tensor (tables too large to build on the host, e.g. BASELINE config 4's 27 M x 64) is padded on its device; a
    ``DeviceTable`` is in that layout already."""
    if isinstance(table, DeviceTable):
        return table
    if hasattr(table, "data_ptr"):
        import torch
        V, D = table.shape
        out = torch.zeros((V + 1, pad4(D)), dtype=torch.float32, device=table.device)
        out[:V, :D] = table
        return out
    table = np.ascontiguousarray(table, dtype=np.float32)
    V, D = table.shape
    Dp = pad4(D)
    out = np.zeros((V + 1, Dp), dtype=np.float32)
    out[:V, :D] = table
    return out


class PlanBuilder:
    def __init__(self, model_kind: int, id_columns: Sequence[IdColumn], n_dense: int):
        self.model_kind = model_kind
        self.id_columns = list(id_columns)
        self.n_dense = n_dense
        self.n_aux = 0
        self.slots: List[np.ndarray] = []
        self._slot_ids: Dict[int, int] = {}
        self.segs: List[L.Seg] = []
        self.ops: List[L.Op] = []
        self.taps: List[L.Tap] = []
        self.pairs: List[tuple] = []
        self.buf_width = [0, 0, 0]
        self.head_bias = 0.0
        self.din: Optional[L.Din] = None
        self._x_end = 0

    # ---- slots -------------------------------------------------------------------------
    def slot(self, arr) -> int:
        """Register a device array; the same ndarray object registered twice shares a slot."""
        key = id(arr)
        if key in self._slot_ids:
            return self._slot_ids[key]
        a = arr if hasattr(arr, "data_ptr") else np.ascontiguousarray(arr, dtype=np.float32)
        self.slots.append(a)
        self._slot_ids[key] = len(self.slots) - 1
        self._keep = getattr(self, "_keep", [])
        self._keep.append(arr)
        return len(self.slots) - 1

    def col(self, key: str) -> int:
        for i, c in enumerate(self.id_columns):
            if c.key == key:
                return i
        raise KeyError("no ids column %r" % key)

    def _touch(self, buf: int, end: int):
        self.buf_width[buf] = max(self.buf_width[buf], pad4(end))

    # ---- buffer-0 allocation -------------------------------------------------------------
    def x_alloc(self, width: int, at: Optional[int] = None) -> int:
        off = pad4(self._x_end) if at is None else at
        self._x_end = max(self._x_end, off + width)
        self._touch(0, off + width)
        return off

    def x_reserve_to(self, end: int):
        """Make later allocations start at or after ``end`` (keeps them clear of regions of
        buffer 0 that later ops overwrite)."""
        self._x_end = max(self._x_end, end)

    # ---- segments ------------------------------------------------------------------------
    def seg_rows(self, key: str, table, vocab: int, dim: int) -> int:
        """embedding_column: gather ``table[id]`` into buffer 0; returns the LDS offset.  ``table``
        is the padded device-layout array (numpy [V, Dp] or a torch CUDA tensor)."""
        Dp = pad4(dim)
        dst = self.x_alloc(Dp)
        self.segs.append(L.Seg(L.SEG_ROWS, self.slot(table), self.col(key), 0, Dp, Dp // 4, dst, vocab))
        return dst

    def seg_cross_rows(self, key_a: str, key_b: str, table, buckets: int, dim: int) -> int:
        Dp = pad4(dim)
        dst = self.x_alloc(Dp)
        self.segs.append(L.Seg(L.SEG_CROSS_ROWS, self.slot(table), self.col(key_a), self.col(key_b),
                               Dp, Dp // 4, dst, buckets))
        return dst

    def seg_scalar(self, key: str, weights, dst: int) -> int:
        w = np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
        n = w.shape[0]
        w = np.concatenate([w, np.zeros(1, np.float32)])       # zero entry at index V for "no id"
        self.segs.append(L.Seg(L.SEG_SCALAR, self.slot(w), self.col(key), 0, 1, 1, dst, n))
        self._touch(0, dst + 1)
        return dst

    def seg_cross_scalar(self, key_a: str, key_b: str, weights, dst: int) -> int:
        w = weights if hasattr(weights, "data_ptr") else np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
        n = int(w.shape[0])
        self.segs.append(L.Seg(L.SEG_CROSS_SCALAR, self.slot(w), self.col(key_a), self.col(key_b), 1, 1, dst, n))
        self._touch(0, dst + 1)
        return dst

    def seg_dense(self, first: int, count: int, dst: int):
        self.segs.append(L.Seg(L.SEG_DENSE, -1, first, 0, 0, count, dst, 0))
        self._touch(0, dst + count)

    def seg_aux(self, first: int, count: int, dst: int):
        self.segs.append(L.Seg(L.SEG_AUX, -1, first, 0, 0, count, dst, 0))
        self._touch(0, dst + count)

    def seg_zero(self, dst: int, count: int):
        if count > 0:
            self.segs.append(L.Seg(L.SEG_ZERO, -1, 0, 0, 0, count, dst, 0))
            self._touch(0, dst + count)

    def seg_numerics(self, n: int) -> int:
        """All ``n`` dense columns + zero fill up to a multiple of 4; returns the LDS offset."""
        w = pad4(n)
        dst = self.x_alloc(w)
        self.seg_dense(0, n, dst)
        self.seg_zero(dst + n, w - n)
        return dst

    # ---- ops -----------------------------------------------------------------------------
    def dense(self, src_buf: int, src_off: int, K: int, xmap: Sequence[int], kernel: np.ndarray,
              bias: np.ndarray, act: int, dst_buf: int, dst_off: int, alpha: Optional[np.ndarray] = None) -> int:
        """Dense layer reading the LDS slice [src_off, src_off+K) of ``src_buf``.  ``xmap[i]`` is
        the position inside that slice of the reference kernel's input row i.  Returns Np."""
        kernel = np.asarray(kernel, dtype=np.float32)
        kin, n = kernel.shape
        if len(xmap) != kin:
            raise ValueError("xmap has %d entries for a kernel with %d input rows" % (len(xmap), kin))
        if K % 4 or src_off % 4 or dst_off % 4:
            raise ValueError("Dense slice must be 4-float aligned")
        Np = pad16(n)
        wt = np.zeros((Np, K), dtype=np.float32)
        xm = np.asarray(xmap, dtype=np.int64)
        if xm.size and (xm.min() < 0 or xm.max() >= K):
            raise ValueError("xmap outside the input slice")
        wt[:n, xm] = kernel.T
        b = np.zeros(Np, dtype=np.float32)
        b[:n] = np.asarray(bias, dtype=np.float32).reshape(-1)
        a_slot = -1
        if act == L.ACT_PRELU:
            a = np.zeros(Np, dtype=np.float32)
            a[:n] = np.asarray(alpha, dtype=np.float32).reshape(-1)
            a_slot = self.slot(a)
        self.ops.append(L.Op(L.OP_DENSE, src_buf, src_off, K, dst_buf, dst_off, Np, self.slot(wt), K,
                             self.slot(b), a_slot, act, 0, 0))
        self._touch(src_buf, src_off + K)
        self._touch(dst_buf, dst_off + Np)
        return Np

    def fm_sumsq(self, src_buf: int, src_off: int, groups: int, group_stride: int, K: int,
                 dst_buf: int, dst_off: int):
        self.ops.append(L.Op(L.OP_FM_SUMSQ, src_buf, src_off, K, dst_buf, dst_off, 0, -1, 0, -1, -1, 0,
                             groups, group_stride))
        self._touch(src_buf, src_off + (groups - 1) * group_stride + K)
        self._touch(dst_buf, dst_off + K)

    def pair_dot(self, src_buf: int, pairs: Sequence[tuple], K: int, dst_buf: int, dst_off: int):
        if self.pairs:
            raise ValueError("only one pair-dot op per plan")
        self.pairs = list(pairs)
        self.ops.append(L.Op(L.OP_PAIR_DOT, src_buf, 0, K, dst_buf, dst_off, 0, -1, 0, -1, -1, 0, 0, 0))
        self._touch(dst_buf, dst_off + len(pairs))

    def tap(self, buf: int, off: int, length: int, weights=None, scale: float = 1.0, bias: float = 0.0):
        w_slot = -1
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
            if w.shape[0] != length:
                raise ValueError("tap weights length %d != %d" % (w.shape[0], length))
            w_slot = self.slot(w)
        self.taps.append(L.Tap(buf, off, length, w_slot, float(scale), float(bias)))
        self._touch(buf, off + length)

    # ---- finish --------------------------------------------------------------------------
    def build(self) -> L.Plan:
        if len(self.segs) > L.MAX_SEGS or len(self.ops) > L.MAX_OPS or len(self.taps) > L.MAX_TAPS \
                or len(self.pairs) > L.MAX_PAIRS:
            raise ValueError("plan too large: %d segs, %d ops, %d taps, %d pairs"
                             % (len(self.segs), len(self.ops), len(self.taps), len(self.pairs)))
        p = L.Plan()
        p.abi_version = L.ABI_VERSION
        p.model_kind = self.model_kind
        p.n_id_cols = len(self.id_columns)
        p.n_dense = self.n_dense
        p.n_aux = self.n_aux
        p.n_slots = len(self.slots)
        n_bufs = max(i + 1 for i in range(3) if self.buf_width[i] > 0)
        p.n_bufs = n_bufs
        for i in range(3):
            p.buf_width[i] = pad4(self.buf_width[i]) if i < n_bufs else 0
            if i < n_bufs and p.buf_width[i] == 0:
                p.buf_width[i] = 4
        p.n_segs = len(self.segs)
        for i, s in enumerate(self.segs):
            p.segs[i] = s
        p.n_ops = len(self.ops)
        for i, o in enumerate(self.ops):
            p.ops[i] = o
        p.n_pairs = len(self.pairs)
        for i, (a, b) in enumerate(self.pairs):
            p.pair_a[i], p.pair_b[i] = a, b
        p.n_taps = len(self.taps)
        for i, t in enumerate(self.taps):
            p.taps[i] = t
        p.head_bias = float(self.head_bias)
        if self.din is not None:
            p.din = self.din
        return p

"""``model.predict()``-shaped Python API over the HIP engine, one class per reference script.

Each class mirrors one Keras model of the reference
(TFRecModel/src/com/sparrowrecsys/offline/tensorflow/{EmbeddingMLP,WideNDeep,NeuralCF,DeepFM,
DeepFM_v2,DIN}.py): same input feature names/dtypes, ``predict(x)`` returns ``ndarray [N,1]
float32`` like ``tf.keras.Model.predict`` (DeepFM.py:131-133), out-of-range ids raise
``ValueError`` (TF raises InvalidArgumentError from assert_less_than_num_buckets).  Weights are
held in the REFERENCE layout (tables ``[V,D]``, Dense kernels ``[in,out]``) under the names listed
by ``weight_shapes()``; ``_compile`` re-lays them out for the device (plan.py).

All arithmetic happens in libsparrow_hip.so on the GPU.  There is no CPU fallback: without the
library or without a HIP device ``predict`` raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from .plan import DeviceTable, PlanBuilder, pad4, pad16, pad_table
from .schema import (GENRE_VOCAB, HISTORY_KEYS, MOVIE_BUCKETS, MOVIE_GENRE_KEYS, N_GENRES, NUMERIC_KEYS,
                     USER_BUCKETS, USER_GENRE_KEYS, IdColumn, iter_feature_batches, pack_dense, pack_ids,
                     to_int_column)

# typical magnitude of each raw numeric column (webroot/sampledata/testSamples.csv ranges); used
# only by init_weights(trained_like=True) so that random weights give non-saturated scores
_NUMERIC_SCALE = {"releaseYear": 2000.0, "movieRatingCount": 10000.0, "userRatingCount": 200.0,
                  "movieAvgRating": 5.0, "userAvgRating": 5.0, "movieRatingStddev": 1.5, "userRatingStddev": 1.5}


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * std).astype(np.float32)


def _glorot(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


class Engine:
    """Owns one ``sprk_handle``: plan + uploaded weights."""

    def __init__(self, plan: L.Plan, slots: Sequence, forward_symbol: str):
        self.lib = L.load_library()
        info = L.runtime_info()
        if info["device_count"] < 1:
            raise RuntimeError("no HIP device visible: the SparrowRecSys HIP forward has no CPU fallback")
        self._plan = plan
        self.handle = C.c_void_p()
        self._external = []
        L.check(self.lib.sprk_create(C.byref(plan), C.byref(self.handle)))
        try:
            for i, a in enumerate(slots):
                if isinstance(a, DeviceTable):                    # already where it should be (6.9 GB tables, row-sharded tables): no copy
                    L.check(self.lib.sprk_upload_external(self.handle, i, C.c_void_p(a.data_ptr()), a.nbytes))
                    self._external.append(a)                      # (the table must outlive the handle)
                    a.engines.add(self)                           # (ShardedTable.close() looks here before it unmaps)
                    continue
                if hasattr(a, "data_ptr"):
                    ptr, nbytes = a.data_ptr(), a.numel() * a.element_size()
                else:
                    ptr, nbytes = a.ctypes.data, a.nbytes
                L.check(self.lib.sprk_upload(self.handle, i, C.c_void_p(ptr), nbytes))
            L.check(self.lib.sprk_finalize(self.handle))
        except Exception:
            self.lib.sprk_destroy(self.handle)
            self.handle = None
            raise
        self._forward = getattr(self.lib, forward_symbol)
        self.n_id_cols = plan.n_id_cols
        self.n_dense = plan.n_dense
        self.n_aux = plan.n_aux
        # forward_many's launch shape: per-call arguments of sprk_forward_many_opts, not handle state.  SPRK_MANY_STREAMS presets the
        # stream count here as it presets the handle's default for C callers of sprk_forward_many (the switch was dead
        # for Python users)
        try:
            env_streams = int(os.environ.get("SPRK_MANY_STREAMS", "0"))
        except ValueError:
            env_streams = 0
        self._many_batches, self._many_streams = 1, (0 if env_streams < 2 else min(env_streams, 4))
        self.has_din = bool(plan.din.enabled)
        self.din_T = plan.din.T
        d = self.describe()
        if d.get("fused") != "1" and os.environ.get("SPRK_FORCE_INTERPRETER") != "1" and os.environ.get("SPRK_QUIET") != "1":
            # a shape without a fused instantiation is served by the plan interpreter, correct but 6-14x slower: say so once,
            # where the model is built, not only to whoever thinks of calling describe()
            import warnings
            warnings.warn("sparrowrecsys_amd: this model shape has no fused kernel and runs on the plan interpreter "
                          "(kernel=%s, stage=%s): expect a tenth of the fused kernels' throughput; the shapes that have one are "
                          "listed in DESIGN.md section 5" % (d.get("kernel"), d.get("stage") or "-"), RuntimeWarning, stacklevel=3)

    def describe(self) -> Dict[str, str]:
        """``sprk_describe``: which kernel instantiation scores a batch (``kernel``; ``k_tile_forward`` = the generic plan
        interpreter, i.e. no fused kernel matched the plan), the history stage, device bytes of tables."""
        buf = C.create_string_buffer(512)
        L.check(self.lib.sprk_describe(self.handle, buf, len(buf)))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if "=" in kv)

    def kernel_name(self) -> str:
        """Name (without template arguments) of the kernel that scores a batch."""
        return self.describe()["kernel"].split("<")[0]

    def table_bytes(self) -> int:
        d = self.describe()
        return int(d["uploaded_bytes"]) + int(d["derived_bytes"])

    def workspace_bytes(self, B: int) -> int:
        return int(self.lib.sprk_workspace_bytes(self.handle, B))

    # ---- argument validation: the C ABI takes raw device pointers, so everything it cannot see is checked here ----
    def _check_tensor(self, t, what, dtype, shape_tail, B=None):
        import torch
        if not isinstance(t, torch.Tensor):
            raise ValueError("%s must be a torch tensor, got %s" % (what, type(t).__name__))
        if not t.is_cuda:
            raise ValueError("%s must live on a HIP device (got %s)" % (what, t.device))
        if t.dtype != dtype:
            raise ValueError("%s must be %s, got %s" % (what, dtype, t.dtype))
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous (row-major); call .contiguous()" % what)
        if t.dim() != 1 + len(shape_tail) or tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError("%s must have shape [B%s], got %s" % (what, "".join(", %d" % d for d in shape_tail), tuple(t.shape)))
        if B is not None and int(t.shape[0]) != B:
            raise ValueError("%s has %d rows, expected %d (rows of `out`)" % (what, int(t.shape[0]), B))

    def _check_batch(self, ids, dense, out, what=""):
        """dtype / device / contiguity / shape of one batch's tensors; returns B.  A torch.long ids tensor, a column
        slice or a short ids tensor would otherwise be read as garbage (or out of bounds) by the kernels."""
        import torch
        self._check_tensor(out, what + "out", torch.float32, ())
        B = int(out.shape[0])
        if self.n_id_cols > 0:
            if ids is None:
                raise ValueError(what + "ids is required (model has %d ids columns)" % self.n_id_cols)
            self._check_tensor(ids, what + "ids", torch.int32, (self.n_id_cols,), B)
            if ids.device != out.device:
                raise ValueError(what + "ids and out live on different devices")
        if self.n_dense > 0:
            if dense is None:
                raise ValueError(what + "dense is required (model has %d numeric columns)" % self.n_dense)
            self._check_tensor(dense, what + "dense", torch.float32, (self.n_dense,), B)
            if dense.device != out.device:
                raise ValueError(what + "dense and out live on different devices")
        return B

    def _check_workspace(self, workspace, B, slices=1):
        if not self.has_din:
            return None, 0
        import torch
        need = self.workspace_bytes(B)
        if workspace is None or not isinstance(workspace, torch.Tensor) or not workspace.is_cuda or not workspace.is_contiguous() \
                or workspace.numel() * workspace.element_size() < need:
            raise ValueError("DIN forward needs a contiguous %d-byte device workspace tensor" % need)
        return workspace.data_ptr(), workspace.numel() * workspace.element_size()

    def _check_many(self, ids_list, dense_list, out_list, what):
        n = len(out_list)
        if self.n_id_cols > 0 and (ids_list is None or len(ids_list) != n):
            raise ValueError("%s: ids_list must hold one tensor per batch (%d)" % (what, n))
        if self.n_dense > 0 and (dense_list is None or len(dense_list) != n):
            raise ValueError("%s: dense_list must hold one tensor per batch (%d)" % (what, n))
        B = None
        for i in range(n):
            b = self._check_batch(ids_list[i] if self.n_id_cols > 0 else None, dense_list[i] if self.n_dense > 0 else None,
                                  out_list[i], "%s: batch %d: " % (what, i))
            if B is None:
                B = b
            elif b != B:
                raise ValueError("%s: every batch must have the same number of rows" % what)
        return B

    def forward(self, ids, dense, out, workspace=None, stream: Optional[int] = None):
        """ids/dense/out/workspace: torch CUDA tensors (int32 [B,F], float32 [B,N], float32 [B])."""
        import torch
        B = self._check_batch(ids, dense, out)
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        ws_ptr, ws_bytes = self._check_workspace(workspace, B)
        L.check(self._forward(self.handle, C.c_void_p(ids.data_ptr() if ids is not None else None),
                              C.c_void_p(dense.data_ptr() if dense is not None else None),
                              C.c_void_p(out.data_ptr()), B, C.c_void_p(ws_ptr), ws_bytes, C.c_void_p(stream)))

    def _many_arrays(self, ids_list, dense_list, out_list):
        n = len(out_list)
        arr = C.c_void_p * n
        ids_a = arr(*[t.data_ptr() for t in ids_list]) if (ids_list is not None and self.n_id_cols > 0) else None
        dense_a = arr(*[t.data_ptr() for t in dense_list]) if (dense_list is not None and self.n_dense > 0) else None
        out_a = arr(*[t.data_ptr() for t in out_list])
        return ids_a, dense_a, out_a

    def forward_many(self, ids_list, dense_list, out_list, workspace=None, stream: Optional[int] = None):
        """``predict`` over a sequence of device-resident batches with ONE foreign call
        (``sprk_forward_many``): batch i reads ids_list[i] / dense_list[i], writes out_list[i]."""
        import torch
        n = len(out_list)
        if n == 0:
            return
        B = self._check_many(ids_list, dense_list, out_list, "forward_many")
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        ws_ptr, ws_bytes = self._check_workspace(workspace, B)
        ids_a, dense_a, out_a = self._many_arrays(ids_list, dense_list, out_list)
        L.check(self.lib.sprk_forward_many_opts(self.handle, n, ids_a, dense_a, out_a, B, C.c_void_p(ws_ptr), ws_bytes,
                                                C.c_void_p(stream), self._many_batches, self._many_streams))

    def prepare_many(self, ids_list, dense_list, out_list, workspace=None):
        """The argument marshalling (and validation) of ``forward_many`` done once: returns ``run(stream=None)`` that
        enqueues the same sequence of batches again (the tensors must stay alive and in place).  For loops that replay a
        fixed set of device buffers -- e.g. one gather group of a multi-GPU predict loop -- the per-call host cost drops
        to one foreign call."""
        import torch
        n = len(out_list)
        if n == 0:
            raise ValueError("prepare_many: no batches")
        B = self._check_many(ids_list, dense_list, out_list, "prepare_many")
        ws_ptr, ws_bytes = self._check_workspace(workspace, B)
        ids_a, dense_a, out_a = self._many_arrays(ids_list, dense_list, out_list)
        keep = (list(ids_list or ()), list(dense_list or ()), list(out_list), workspace)
        fn, handle, ws_p = self.lib.sprk_forward_many_opts, self.handle, C.c_void_p(ws_ptr)
        per_launch, streams = self._many_batches, self._many_streams        # bound NOW: a prepared run keeps its launch shape

        def run(stream: Optional[int] = None, _keep=keep):
            if stream is None:
                stream = torch.cuda.current_stream().cuda_stream
            L.check(fn(handle, n, ids_a, dense_a, out_a, B, ws_p, ws_bytes, C.c_void_p(stream), per_launch, streams))
        return run

    def set_many_streams(self, n: int) -> bool:
        """Fan forward_many's independent batches over ``n`` helper streams (0 = strict order).  Models with a workspace
        (DIN) additionally need ``n`` workspace slices (see ``many_workspace_bytes``).  Kept on the Python object and handed to
        ``sprk_forward_many_opts`` with every call: the C handle is not modified."""
        n = int(n)
        if n < 0 or n > 4:
            raise ValueError("stream count %d outside [0,4]" % n)
        self._many_streams = 0 if n < 2 else n
        return True

    def set_many_batches(self, n: int) -> bool:
        """Let one kernel launch score up to ``n`` of forward_many's batches (1 = a launch per batch).  A per-call argument
        of ``sprk_forward_many_opts`` (see ``set_many_streams``)."""
        n = int(n)
        if n < 1 or n > 64:
            raise ValueError("batches per launch %d outside [1,64]" % n)
        self._many_batches = n
        return True

    def many_workspace_bytes(self, B: int, n: int) -> int:
        """Workspace size that lets forward_many fan a workspace model over ``n`` streams."""
        need = (self.workspace_bytes(B) + 255) & ~255
        return max(1, n) * need

    def din_pool(self, ids, pooled, att=None, stream: Optional[int] = None):
        import torch
        if not self.has_din:
            raise ValueError("din_pool: the model has no DIN / DIEN stage")
        B = int(ids.shape[0]) if hasattr(ids, "shape") and len(ids.shape) else -1
        self._check_tensor(ids, "ids", torch.int32, (self.n_id_cols,))
        self._check_tensor(pooled, "pooled", torch.float32, (self.n_aux,), B)
        if att is not None:
            self._check_tensor(att, "att", torch.float32, (self.din_T,), B)
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        L.check(self.lib.sprk_din_pool(self.handle, C.c_void_p(ids.data_ptr()), C.c_void_p(pooled.data_ptr()),
                                       C.c_void_p(att.data_ptr() if att is not None else None),
                                       int(ids.shape[0]), C.c_void_p(stream)))

    def check_ids(self, stream: Optional[int] = None):
        import torch
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        L.check(self.lib.sprk_check_ids(self.handle, C.c_void_p(stream)))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sprk_destroy(self.handle)
            self.handle = None
            for a in self._external:
                a.engines.discard(self)
            self._external = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CTRModel:
    MODEL_KIND = L.MODEL_GENERIC
    FORWARD_SYMBOL = "sprk_forward"
    numeric_keys = NUMERIC_KEYS

    def __init__(self, weights: Optional[Mapping[str, np.ndarray]] = None, seed: Optional[int] = None):
        self.id_columns: List[IdColumn] = self._id_columns()
        if weights is None:
            weights = self.init_weights(0 if seed is None else seed)
        self.set_weights(weights)

    # ---- to be provided by subclasses ---------------------------------------------------
    def _id_columns(self) -> List[IdColumn]:
        raise NotImplementedError

    def weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        raise NotImplementedError

    def _compile(self, pb: PlanBuilder, w: Mapping[str, np.ndarray]):
        raise NotImplementedError

    # ---- weights ------------------------------------------------------------------------
    def init_weights(self, seed: int = 0, trained_like: bool = True) -> Dict[str, np.ndarray]:
        """Reference initialisers: embedding_column tables truncated-normal(std=1/sqrt(D)), Keras
        Embedding uniform(+-0.05), Dense glorot-uniform with zero bias, PReLU alpha zeros.
        ``trained_like`` (default) additionally scales the Dense rows that multiply RAW numeric
        columns by 1/typical magnitude and draws small non-zero biases / PReLU alphas, so random
        weights yield non-saturated scores and every parameter is exercised by parity tests."""
        rng = np.random.default_rng(seed)
        out = {}
        for name, shape in self.weight_shapes().items():
            if name.startswith("emb/") or name.startswith("deep_emb/"):
                if name == "emb/movie":
                    out[name] = rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)
                    if trained_like:
                        out[name] = _trunc_normal(rng, shape, 1.0 / np.sqrt(shape[1]))
                else:
                    out[name] = _trunc_normal(rng, shape, 1.0 / np.sqrt(shape[1]))
            elif name.endswith("/kernel"):
                out[name] = _glorot(rng, shape)
                if trained_like and shape[0] > 256:
                    # kernels that multiply a wide one-hot block (head of DeepFM / Wide&Deep,
                    # fo_cat): glorot over a 31 040-wide fan-in is ~0, a trained model's is not
                    lim = np.sqrt(6.0 / (128 + shape[1]))
                    out[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
            elif name.endswith("/bias"):
                out[name] = rng.uniform(-0.1, 0.1, size=shape).astype(np.float32) if trained_like \
                    else np.zeros(shape, np.float32)
            elif name.endswith("/alpha"):
                out[name] = rng.uniform(0.0, 0.5, size=shape).astype(np.float32) if trained_like \
                    else np.zeros(shape, np.float32)
            elif name.endswith("/h0"):
                out[name] = _glorot(rng, shape)                    # GlorotUniform()(shape=(1, D)), DIEN.py:239-240
            else:
                raise ValueError("unknown weight kind %r" % name)
        if trained_like:
            for kname, rows in self._numeric_kernel_rows().items():
                for key, row in rows.items():
                    out[kname][row, :] *= 1.0 / _NUMERIC_SCALE[key]
        return out

    def _numeric_kernel_rows(self) -> Dict[str, Dict[str, int]]:
        """{kernel name: {numeric key: input row}} for kernels fed by raw numeric columns."""
        return {}

    def set_weights(self, weights: Mapping[str, np.ndarray]):
        shapes = self.weight_shapes()
        w = {}
        for name, shape in shapes.items():
            if name not in weights:
                raise KeyError("missing weight %r" % name)
            a = weights[name]
            if tuple(a.shape) != tuple(shape):
                raise ValueError("weight %r has shape %s, expected %s" % (name, tuple(a.shape), tuple(shape)))
            w[name] = a if hasattr(a, "data_ptr") else np.ascontiguousarray(a, dtype=np.float32)
        self.weights = w
        self._engine = None

    # ---- plan / engine ------------------------------------------------------------------
    def build_plan(self):
        pb = PlanBuilder(self.MODEL_KIND, self.id_columns, len(self.numeric_keys))
        self._compile(pb, self.weights)
        return pb.build(), pb.slots

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            plan, slots = self.build_plan()
            self._engine = Engine(plan, slots, self.FORWARD_SYMBOL)
        return self._engine

    # ---- host packing -------------------------------------------------------------------
    def pack(self, features: Mapping) -> Tuple[np.ndarray, np.ndarray]:
        return pack_ids(features, self.id_columns), pack_dense(features, self.numeric_keys)

    # ---- predict ------------------------------------------------------------------------
    def predict_device(self, ids, dense, out=None, workspace=None, stream=None):
        """Device-resident forward: torch CUDA tensors in, torch CUDA tensor [B] out (async)."""
        import torch
        if out is None:
            out = torch.empty(ids.shape[0], dtype=torch.float32, device=ids.device)
        eng = self.engine
        if eng.has_din and workspace is None:
            workspace = torch.empty(eng.workspace_bytes(int(out.shape[0])) // 4, dtype=torch.float32, device=ids.device)
        eng.forward(ids, dense, out, workspace, stream)
        return out

    MANY_GROUP = 64                  # batches handed to ONE sprk_forward_many call (and, where the graph has such a kernel, to one launch)

    def predict_device_many(self, ids_list, dense_list, outs=None, workspace=None, stream=None):
        """[r6] Several device-resident batches of ONE row count: one foreign call (``sprk_forward_many``) and -- for the graphs with a
        several-batches kernel (DeepFM_v2 / pair-dot DeepFM / k_rows_chain graphs / EmbeddingMLP, Wide&Deep / DIN) -- one launch per group of up to 64 (16) batches instead of
        a launch per batch; bit-identical to ``predict_device`` batch by batch (tests/test_gpu_parity.py).  The reference's own small
        batches (DeepFM.py:17 batch 12, the 800-candidate request of RecForYouProcess.java:113-130) sit on the launch floor one at a time.
        Returns the list of score tensors (async)."""
        import torch
        n = len(ids_list)
        if n == 0:
            return []
        B = int(ids_list[0].shape[0])
        if outs is None:
            flat = torch.empty(n * B, dtype=torch.float32, device=ids_list[0].device)
            outs = [flat[i * B:(i + 1) * B] for i in range(n)]
        eng = self.engine
        if eng.has_din and workspace is None:
            workspace = torch.empty(max(eng.workspace_bytes(B) // 4, 1), dtype=torch.float32, device=ids_list[0].device)
        saved = eng._many_batches
        eng.set_many_batches(min(max(n, 1), self.MANY_GROUP))
        try:
            eng.forward_many(ids_list, dense_list, outs, workspace, stream)
        finally:
            eng._many_batches = saved
        return outs

    def predict(self, x, batch_size: Optional[int] = None) -> np.ndarray:
        """Like ``tf.keras.Model.predict``: dict of feature columns (or an iterable of dicts /
        ``(dict, label)`` tuples) -> ``ndarray [N, 1] float32``."""
        import torch
        eng = self.engine
        outs = []
        # batches are enqueued back to back (host packing of batch n+1 overlaps the forward of batch n); ONE id check
        # (= one stream synchronisation) and ONE device -> host copy at the end instead of one per batch
        # [r6] consecutive batches of one size go to the device as a GROUP (predict_device_many: one foreign call, one launch where the graph
        # has a several-batches kernel); a group is enqueued while the host packs the next one
        pending = []

        def flush():
            if len(pending) == 1:
                outs.append(self.predict_device(*pending[0]))
            elif pending:
                outs.extend(self.predict_device_many([b[0] for b in pending], [b[1] for b in pending]))
            pending.clear()
        for feats in iter_feature_batches(x, batch_size):
            ids, dense = self.pack(feats)
            if ids.shape[0] == 0:
                continue
            ids_t = torch.from_numpy(ids).cuda(non_blocking=True)
            dense_t = torch.from_numpy(dense).cuda(non_blocking=True)
            if pending and (ids_t.shape[0] != pending[0][0].shape[0] or len(pending) == self.MANY_GROUP):
                flush()
            pending.append((ids_t, dense_t))
        flush()
        if not outs:
            return np.zeros((0, 1), dtype=np.float32)
        eng.check_ids()
        scores = outs[0] if len(outs) == 1 else torch.cat(outs)
        return scores.cpu().numpy().reshape(-1, 1)

    __call__ = predict

    def evaluate(self, x, y=None, batch_size: Optional[int] = None):
        """Like ``tf.keras.Model.evaluate`` under the reference's ``compile()`` (DeepFM.py:117-126): ``[loss, accuracy, roc_auc,
        pr_auc]`` -- binary cross-entropy, accuracy at 0.5 and Keras' 200-threshold ROC / PR AUC (``metrics.py``).  ``x`` is a
        feature dict with the labels in ``y`` (or under the key ``"label"``, as ``make_csv_dataset(label_name='label')`` splits
        them off), or an iterable of ``(features, labels)`` batches like the reference's ``test_dataset``."""
        from .metrics import evaluate_scores
        if isinstance(x, Mapping):
            labels = x["label"] if y is None else y
            return evaluate_scores(np.asarray(labels).astype(np.float64).reshape(-1), self.predict(x, batch_size)[:, 0])
        batches = list(x)
        if not batches or not all(isinstance(b, (tuple, list)) and len(b) == 2 for b in batches):
            raise ValueError("evaluate: an iterable of (features, labels) batches, or a feature dict with labels, is required")
        labels = np.concatenate([np.asarray(b[1]).astype(np.float64).reshape(-1) for b in batches])
        return evaluate_scores(labels, self.predict(batches, batch_size)[:, 0])

    def predict_csv(self, source, batch_size: int = 65536, max_rows: Optional[int] = None) -> np.ndarray:
        """``model.predict(get_dataset(path))`` of the reference (DeepFM.py:14-22,131-133) without the host in the data path:
        ``source`` = a CSV file path, its bytes, or a ``torch.uint8`` device tensor holding the text.  The raw text goes to
        the device once, ``sprk_pack_csv_device`` tokenizes it there into the packed ids / dense arrays, the forward runs over
        ``batch_size``-row slices of them -> ``ndarray [N, 1] float32``.  Same scores as ``predict(read_samples_csv(path))``."""
        import torch

        from .ingest import pack_csv_device, read_csv_to_device
        if isinstance(source, str):
            buf, nbytes = read_csv_to_device(source)              # pinned read + one asynchronous copy of the raw text
            source = buf[:nbytes]
        ids, dense = pack_csv_device(source, self.id_columns, list(self.numeric_keys), max_rows=max_rows)
        n = int(ids.shape[0])
        if n == 0:
            return np.zeros((0, 1), dtype=np.float32)
        out = torch.empty(n, dtype=torch.float32, device=ids.device)
        full = n // batch_size
        for g0 in range(0, full, self.MANY_GROUP):                # [r6] the full slices in groups of up to 64: one launch per group where the graph allows
            sl = [(k * batch_size, (k + 1) * batch_size) for k in range(g0, min(full, g0 + self.MANY_GROUP))]
            if len(sl) == 1:
                self.predict_device(ids[sl[0][0]:sl[0][1]], dense[sl[0][0]:sl[0][1]], out[sl[0][0]:sl[0][1]])
            else:
                self.predict_device_many([ids[a:b] for a, b in sl], [dense[a:b] for a, b in sl], [out[a:b] for a, b in sl])
        if full * batch_size < n:
            lo = full * batch_size
            self.predict_device(ids[lo:n], dense[lo:n], out[lo:n])
        self.engine.check_ids()
        return out.cpu().numpy().reshape(-1, 1)


# =============================================================================================
def _sorted_xmap(blocks: Mapping[str, Tuple[int, int]]) -> List[int]:
    """DenseFeatures column-name order (ASCII sort) -> list of LDS offsets, one per input row."""
    xmap: List[int] = []
    for name in sorted(blocks):
        off, width = blocks[name]
        xmap.extend(range(off, off + width))
    return xmap


def _mlp_stack(pb: PlanBuilder, w, names: Sequence[str], src_buf: int, src_off: int, K: int,
               xmap: Sequence[int], acts: Sequence[int], alphas: Sequence[Optional[str]] = None):
    """Chain of Dense layers ping-ponging between LDS buffers; returns (buf, real width)."""
    buf, off, xm, width = src_buf, src_off, list(xmap), K
    n = None
    for i, name in enumerate(names):
        dst = 1 if buf != 1 else 0
        kernel, bias = w[name + "/kernel"], w[name + "/bias"]
        alpha = w[alphas[i]] if (alphas and alphas[i]) else None
        Np = pb.dense(buf, off, width, xm, kernel, bias, acts[i], dst, 0, alpha)
        n = kernel.shape[1]
        buf, off, width, xm = dst, 0, Np, list(range(n))
    return buf, n


def _max_buf0_write(widths: Sequence[int], first_dst: int) -> int:
    """Largest padded width a ping-pong stack starting with destination ``first_dst`` writes to buffer 0."""
    m = 0
    dst = first_dst
    for wd in widths:
        if dst == 0:
            m = max(m, pad16(wd))
        dst = 1 - dst
    return m


class EmbeddingMLP(CTRModel):
    """EmbeddingMLP.py:34-77."""
    MODEL_KIND = L.MODEL_EMBEDDING_MLP
    FORWARD_SYMBOL = "sprk_forward_embedding_mlp"
    EMB_KEYS = USER_GENRE_KEYS + MOVIE_GENRE_KEYS + ["movieId", "userId"]

    def __init__(self, weights=None, seed=None, emb_dim=10, movie_buckets=MOVIE_BUCKETS,
                 user_buckets=USER_BUCKETS, hidden=(128, 128)):
        self.emb_dim, self.movie_buckets, self.user_buckets, self.hidden = emb_dim, movie_buckets, user_buckets, tuple(hidden)
        super().__init__(weights, seed)

    def _vocab(self, key):
        return {"movieId": self.movie_buckets, "userId": self.user_buckets}.get(key, N_GENRES)

    def _id_columns(self):
        return [IdColumn(k, "id" if k in ("movieId", "userId") else "genre", self._vocab(k)) for k in self.EMB_KEYS]

    def _deep_in(self):
        return len(self.numeric_keys) + len(self.EMB_KEYS) * self.emb_dim

    def weight_shapes(self):
        s = {"emb/" + k: (self._vocab(k), self.emb_dim) for k in self.EMB_KEYS}
        fan = self._deep_in()
        for i, h in enumerate(self.hidden):
            s["dense%d/kernel" % i] = (fan, h)
            s["dense%d/bias" % i] = (h,)
            fan = h
        s["head/kernel"] = (self._head_in(), 1)
        s["head/bias"] = (1,)
        return s

    def _head_in(self):
        return self.hidden[-1]

    def _deep_blocks_order(self):
        names = [k + "_embedding" for k in self.EMB_KEYS] + list(self.numeric_keys)
        widths = {k + "_embedding": self.emb_dim for k in self.EMB_KEYS}
        widths.update({k: 1 for k in self.numeric_keys})
        rows, r = {}, 0
        for n in sorted(names):
            rows[n] = r
            r += widths[n]
        return rows

    def _numeric_kernel_rows(self):
        rows = self._deep_blocks_order()
        return {"dense0/kernel": {k: rows[k] for k in self.numeric_keys}}

    def _compile_body(self, pb, w):
        blocks = {}
        for k in self.EMB_KEYS:
            off = pb.seg_rows(k, pad_table(w["emb/" + k]) if not hasattr(w["emb/" + k], "data_ptr") else w["emb/" + k],
                              self._vocab(k), self.emb_dim)
            blocks[k + "_embedding"] = (off, self.emb_dim)
        num_off = pb.seg_numerics(len(self.numeric_keys))
        for j, k in enumerate(self.numeric_keys):
            blocks[k] = (num_off + j, 1)
        K0 = pad4(pb._x_end)
        pb.x_reserve_to(max(K0, _max_buf0_write(self.hidden, 1)))
        return K0, _sorted_xmap(blocks)

    def _compile(self, pb, w):
        K0, xmap = self._compile_body(pb, w)
        names = ["dense%d" % i for i in range(len(self.hidden))]
        buf, n = _mlp_stack(pb, w, names, 0, 0, K0, xmap, [L.ACT_RELU] * len(names))
        pb.tap(buf, 0, n, w["head/kernel"][:n, 0])
        pb.head_bias = float(w["head/bias"][0])


class WideNDeep(EmbeddingMLP):
    """WideNDeep.py:72-107.  ``cross_dim == 0``: the crossed column is an indicator (reference);
    ``cross_dim > 0``: an embedding_column of that width over the same hashed cross (BASELINE
    config 5's 10 M-bucket x 32 table)."""
    MODEL_KIND = L.MODEL_WIDE_DEEP
    FORWARD_SYMBOL = "sprk_forward_widedeep"

    def __init__(self, weights=None, seed=None, emb_dim=10, movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS,
                 hidden=(128, 128), cross_buckets=10000, cross_dim=0, rated_buckets=None):
        self.cross_buckets, self.cross_dim = cross_buckets, cross_dim
        self.rated_buckets = movie_buckets if rated_buckets is None else rated_buckets
        super().__init__(weights, seed, emb_dim, movie_buckets, user_buckets, hidden)

    def _id_columns(self):
        return super()._id_columns() + [IdColumn("userRatedMovie1", "id", self.rated_buckets)]

    def weight_shapes(self):
        s = super().weight_shapes()
        if self.cross_dim:
            s["emb/cross"] = (self.cross_buckets, self.cross_dim)
        return s

    def _head_in(self):
        return self.hidden[-1] + (self.cross_dim if self.cross_dim else self.cross_buckets)

    def _compile(self, pb, w):
        K0, xmap = self._compile_body(pb, w)
        H = self.hidden[-1]
        hk = w["head/kernel"]
        if self.cross_dim:
            tab = w["emb/cross"]
            tab = tab if hasattr(tab, "data_ptr") else pad_table(tab)
            if pad4(self.cross_dim) != self.cross_dim and hasattr(w["emb/cross"], "data_ptr"):
                raise ValueError("device-resident cross table needs cross_dim % 4 == 0")
            coff = pb.seg_cross_rows("movieId", "userRatedMovie1", tab, self.cross_buckets, self.cross_dim)
        else:
            coff = pb.x_alloc(1)
            pb.seg_cross_scalar("movieId", "userRatedMovie1",
                                hk[H:, 0] if not hasattr(hk, "data_ptr") else hk[H:, 0].contiguous(), coff)
        names = ["dense%d" % i for i in range(len(self.hidden))]
        buf, n = _mlp_stack(pb, w, names, 0, 0, K0, xmap, [L.ACT_RELU] * len(names))
        hk_np = hk if not hasattr(hk, "data_ptr") else hk[:H + (self.cross_dim or 0)].cpu().numpy()
        pb.tap(buf, 0, n, hk_np[:H, 0])
        if self.cross_dim:
            pb.tap(0, coff, self.cross_dim, hk_np[H:H + self.cross_dim, 0])
        else:
            pb.tap(0, coff, 1, None)
        pb.head_bias = float(np.asarray(w["head/bias"]).reshape(-1)[0])


class NeuralCF(CTRModel):
    """NeuralCF.py:45-70; ``arch=1`` is neural_cf_model_1 (the one NeuralCF.py:74 builds and the
    Jetty server queries), ``arch=2`` the two-tower + Dot variant."""
    MODEL_KIND = L.MODEL_NEURALCF
    FORWARD_SYMBOL = "sprk_forward_neuralcf"
    numeric_keys: List[str] = []

    def __init__(self, weights=None, seed=None, emb_dim=10, movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS,
                 hidden=(10, 10), arch=1):
        self.emb_dim, self.movie_buckets, self.user_buckets, self.hidden, self.arch = emb_dim, movie_buckets, user_buckets, tuple(hidden), arch
        super().__init__(weights, seed)

    def _id_columns(self):
        return [IdColumn("movieId", "id", self.movie_buckets), IdColumn("userId", "id", self.user_buckets)]

    def weight_shapes(self):
        s = {"emb/movieId": (self.movie_buckets, self.emb_dim), "emb/userId": (self.user_buckets, self.emb_dim)}
        if self.arch == 1:
            fan = 2 * self.emb_dim
            for i, h in enumerate(self.hidden):
                s["dense%d/kernel" % i] = (fan, h)
                s["dense%d/bias" % i] = (h,)
                fan = h
            s["head/kernel"] = (fan, 1)
        else:
            for tower in ("item", "user"):
                fan = self.emb_dim
                for i, h in enumerate(self.hidden):
                    s["%s%d/kernel" % (tower, i)] = (fan, h)
                    s["%s%d/bias" % (tower, i)] = (h,)
                    fan = h
            s["head/kernel"] = (1, 1)
        s["head/bias"] = (1,)
        return s

    def pack(self, features):
        ids = pack_ids(features, self.id_columns)
        return ids, np.zeros((ids.shape[0], 0), dtype=np.float32)

    def _compile(self, pb, w):
        D, Dp = self.emb_dim, pad4(self.emb_dim)
        ioff = pb.seg_rows("movieId", pad_table(w["emb/movieId"]), self.movie_buckets, D)
        uoff = pb.seg_rows("userId", pad_table(w["emb/userId"]), self.user_buckets, D)
        K0 = pad4(pb._x_end)
        if self.arch == 1:
            pb.x_reserve_to(max(K0, _max_buf0_write(self.hidden, 1)))
            xmap = list(range(ioff, ioff + D)) + list(range(uoff, uoff + D))        # NeuralCF.py:48
            names = ["dense%d" % i for i in range(len(self.hidden))]
            buf, n = _mlp_stack(pb, w, names, 0, 0, K0, xmap, [L.ACT_RELU] * len(names))
            pb.tap(buf, 0, n, w["head/kernel"][:n, 0])
        else:
            # towers run side by side: item in columns [0,W), user in [W,2W) of buffers 1 and 2
            W = max(pad16(h) for h in self.hidden)
            src = {"item": (0, ioff, Dp, list(range(D))), "user": (0, uoff, Dp, list(range(D)))}
            dst_bufs = [1, 2]
            for i, h in enumerate(self.hidden):
                dbuf = dst_bufs[i % 2]
                for t, tower in enumerate(("item", "user")):
                    sbuf, soff, K, xm = src[tower]
                    Np = pb.dense(sbuf, soff, K, xm, w["%s%d/kernel" % (tower, i)], w["%s%d/bias" % (tower, i)],
                                  L.ACT_RELU, dbuf, t * W)
                    src[tower] = (dbuf, t * W, Np, list(range(h)))
            fbuf = src["item"][0]
            Kdot = src["item"][2]
            dot_off = pb.x_alloc(1, at=pad4(max(K0, 4)))
            pb.pair_dot(fbuf, [(0, W)], Kdot, 0, dot_off)
            pb.tap(0, dot_off, 1, None, scale=float(w["head/kernel"][0, 0]))
        pb.head_bias = float(w["head/bias"][0])


# reference field list (DeepFM.py:54-76): key, kind, vocab
def _default_fields(movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS):
    return [("movieId", "id", movie_buckets), ("userId", "id", user_buckets),
            ("userGenre1", "genre", N_GENRES), ("movieGenre1", "genre", N_GENRES)]


def first_order_offsets(fields) -> Dict[str, int]:
    """Row offsets of the one-hot blocks inside DenseFeatures(indicator columns), name-sorted
    (``<key>_indicator``): reference fields -> movieGenre1 0, movieId 19, userGenre1 1020, userId 1039."""
    offs, o = {}, 0
    for _, key, vocab in sorted((k + "_indicator", k, v) for k, _, v in fields):
        offs[key] = o
        o += vocab
    offs["__total__"] = o
    return offs


class DeepFM(CTRModel):
    """DeepFM.py:54-115 (pairwise-dot FM).  ``fields`` / ``pairs`` / ``deep_emb`` default to the
    reference's literals; other values give the generalised shape of BASELINE config 2.

    The deep part owns its OWN embedding tables (weights ``deep_emb/<key>``): the reference hands ``movie_emb_col`` /
    ``user_emb_col`` to two ``DenseFeatures`` layers -- ``DenseFeatures([movie_emb_col])`` for the FM dots (DeepFM.py:91-92)
    and ``DenseFeatures(deep_feature_columns)`` for the MLP (DeepFM.py:106) -- and a DenseFeatures layer creates the
    variables of its columns itself (TF feature_column_v2.py ``_StateManagerImpl.create_variable`` ->
    ``layer.add_weight``), so a trained reference model holds ``dense_features/movieId_embedding`` AND
    ``dense_features_5/movieId_embedding``.  ``share_deep_tables=True`` ties them (one table per key, ``emb/<key>``)."""
    MODEL_KIND = L.MODEL_DEEPFM
    FORWARD_SYMBOL = "sprk_forward_deepfm"
    DEFAULT_PAIRS = [("movieId", "userId"), ("movieGenre1", "userGenre1"),
                     ("movieGenre1", "userId"), ("movieId", "userGenre1")]       # DeepFM.py:100-103,111-112

    def __init__(self, weights=None, seed=None, emb_dim=10, fields=None, pairs=None, deep_emb=("movieId", "userId"),
                 hidden=(64, 64), share_deep_tables=False):
        self.emb_dim = emb_dim
        self.share_deep_tables = bool(share_deep_tables)
        self.fields = list(fields) if fields is not None else _default_fields()
        self.pairs = list(pairs) if pairs is not None else list(self.DEFAULT_PAIRS)
        self.deep_emb = list(deep_emb)
        self.hidden = tuple(hidden)
        self.fo = first_order_offsets(self.fields)
        super().__init__(weights, seed)

    def _id_columns(self):
        return [IdColumn(k, kind, v) for k, kind, v in self.fields]

    def set_weights(self, weights):
        # a round-2 weight dict (or a converter that only knows emb/<key>) has no deep_emb/<key>: say what that means instead of
        # "missing weight" -- tied tables are a choice the caller makes, the oracle follows the same rule
        if not self.share_deep_tables:
            missing = ["deep_emb/" + k for k in self.deep_emb if "deep_emb/" + k not in weights]
            if missing:
                raise KeyError("missing weight %r: DeepFM.py's deep part owns its own movieId / userId tables (DeepFM.py:106); "
                               "construct the model with share_deep_tables=True to tie them to emb/<key>" % missing[0])
        super().set_weights(weights)

    def _deep_rows(self):
        names = [k + "_embedding" for k in self.deep_emb] + list(self.numeric_keys)
        widths = {k + "_embedding": self.emb_dim for k in self.deep_emb}
        widths.update({k: 1 for k in self.numeric_keys})
        rows, r = {}, 0
        for n in sorted(names):
            rows[n] = r
            r += widths[n]
        return rows, r

    def _numeric_kernel_rows(self):
        rows, _ = self._deep_rows()
        return {"deep0/kernel": {k: rows[k] for k in self.numeric_keys}}

    def weight_shapes(self):
        s = {"emb/" + k: (v, self.emb_dim) for k, _, v in self.fields}
        if not self.share_deep_tables:
            vocab = {k: v for k, _, v in self.fields}
            s.update({"deep_emb/" + k: (vocab[k], self.emb_dim) for k in self.deep_emb})
        _, fan = self._deep_rows()
        for i, h in enumerate(self.hidden):
            s["deep%d/kernel" % i] = (fan, h)
            s["deep%d/bias" % i] = (h,)
            fan = h
        s["head/kernel"] = (self.fo["__total__"] + len(self.pairs) + self.hidden[-1], 1)
        s["head/bias"] = (1,)
        return s

    def _compile(self, pb, w):
        D, Dp = self.emb_dim, pad4(self.emb_dim)
        vocab = {k: v for k, _, v in self.fields}
        offs, blocks = {}, {}
        for k in self.deep_emb:                                   # deep inputs first -> one contiguous slice
            if self.share_deep_tables:
                offs[k] = pb.seg_rows(k, pad_table(w["emb/" + k]), vocab[k], D)
                blocks[k + "_embedding"] = (offs[k], D)
            else:                                                 # the deep part's own table of this key (DeepFM.py:106)
                blocks[k + "_embedding"] = (pb.seg_rows(k, pad_table(w["deep_emb/" + k]), vocab[k], D), D)
        num_off = pb.seg_numerics(len(self.numeric_keys))
        for j, k in enumerate(self.numeric_keys):
            blocks[k] = (num_off + j, 1)
        Kdeep = pad4(pb._x_end)
        for k, _, _ in self.fields:
            if k not in offs:
                offs[k] = pb.seg_rows(k, pad_table(w["emb/" + k]), vocab[k], D)
        pb.x_reserve_to(max(pad4(pb._x_end), _max_buf0_write(self.hidden, 1)))
        hk = w["head/kernel"][:, 0]
        n_fo, P = self.fo["__total__"], len(self.pairs)
        scal_off = pb.x_alloc(len(self.fields))
        for i, (k, _, v) in enumerate(self.fields):               # first-order: rows of the head kernel
            pb.seg_scalar(k, hk[self.fo[k]:self.fo[k] + v], scal_off + i)
        dots_off = pb.x_alloc(P)
        pb.pair_dot(0, [(offs[a], offs[b]) for a, b in self.pairs], Dp, 0, dots_off)
        names = ["deep%d" % i for i in range(len(self.hidden))]
        buf, n = _mlp_stack(pb, w, names, 0, 0, Kdeep, _sorted_xmap(blocks), [L.ACT_RELU] * len(names))
        pb.tap(0, scal_off, len(self.fields), None)
        pb.tap(0, dots_off, P, hk[n_fo:n_fo + P])
        pb.tap(buf, 0, n, hk[n_fo + P:n_fo + P + n])
        pb.head_bias = float(w["head/bias"][0])


class DeepFMv2(CTRModel):
    """DeepFM_v2.py:60-157 (sum-of-squares FM cross over per-field Dense projections)."""
    MODEL_KIND = L.MODEL_DEEPFM_V2
    FORWARD_SYMBOL = "sprk_forward_deepfm_v2"
    DEFAULT_ORDER = ["movieGenre1", "movieId", "userGenre1", "userId"]          # DeepFM_v2.py:106-110

    def __init__(self, weights=None, seed=None, emb_dim=10, fields=None, order=None, proj_dim=64, hidden=(32, 16)):
        self.emb_dim = emb_dim
        self.fields = list(fields) if fields is not None else _default_fields()
        self.order = list(order) if order is not None else (list(self.DEFAULT_ORDER) if fields is None else [k for k, _, _ in self.fields])
        self.proj_dim, self.hidden = proj_dim, tuple(hidden)
        self.fo = first_order_offsets(self.fields)
        super().__init__(weights, seed)

    def _id_columns(self):
        return [IdColumn(k, kind, v) for k, kind, v in self.fields]

    def _numeric_kernel_rows(self):
        rows = {k: i for i, k in enumerate(self.numeric_keys)}    # numerics alone, already name-sorted
        return {"fo_num/kernel": rows, "proj/num/kernel": rows}

    def weight_shapes(self):
        s = {"emb/" + k: (v, self.emb_dim) for k, _, v in self.fields}
        s["fo_cat/kernel"] = (self.fo["__total__"], 1)
        s["fo_cat/bias"] = (1,)
        s["fo_num/kernel"] = (len(self.numeric_keys), 1)
        s["fo_num/bias"] = (1,)
        for k in self.order:
            s["proj/%s/kernel" % k] = (self.emb_dim, self.proj_dim)
            s["proj/%s/bias" % k] = (self.proj_dim,)
        s["proj/num/kernel"] = (len(self.numeric_keys), self.proj_dim)
        s["proj/num/bias"] = (self.proj_dim,)
        fan = (len(self.order) + 1) * self.proj_dim
        for i, h in enumerate(self.hidden):
            s["deep%d/kernel" % i] = (fan, h)
            s["deep%d/bias" % i] = (h,)
            fan = h
        s["head/kernel"] = (1 + self.proj_dim + self.hidden[-1], 1)
        s["head/bias"] = (1,)
        return s

    def _compile(self, pb, w):
        D, Dp = self.emb_dim, pad4(self.emb_dim)
        vocab = {k: v for k, _, v in self.fields}
        nnum = len(self.numeric_keys)
        G = len(self.order) + 1
        Kp = pad16(self.proj_dim)
        offs = {k: pb.seg_rows(k, pad_table(w["emb/" + k]), vocab[k], D) for k in self.order}
        # everything the output layer still needs lives past the region deep0 overwrites in buffer 0
        pb.x_reserve_to(max(pad4(pb._x_end), _max_buf0_write(self.hidden, 0)))
        num_off = pb.seg_numerics(nnum)
        scal_off = pb.x_alloc(len(self.fields))
        fk = w["fo_cat/kernel"][:, 0]
        for i, (k, _, v) in enumerate(self.fields):
            pb.seg_scalar(k, fk[self.fo[k]:self.fo[k] + v], scal_off + i)
        fm_off = pb.x_alloc(self.proj_dim)
        for g, k in enumerate(self.order):                        # DeepFM_v2.py:113-116
            pb.dense(0, offs[k], Dp, list(range(D)), w["proj/%s/kernel" % k], w["proj/%s/bias" % k], L.ACT_NONE, 1, g * Kp)
        pb.dense(0, num_off, pad4(nnum), list(range(nnum)), w["proj/num/kernel"], w["proj/num/bias"], L.ACT_NONE,
                 1, (G - 1) * Kp)                                 # DeepFM_v2.py:118-120
        pb.fm_sumsq(1, 0, G, Kp, self.proj_dim, 0, fm_off)        # DeepFM_v2.py:147-152
        flat = [g * Kp + j for g in range(G) for j in range(self.proj_dim)]     # Flatten, DeepFM_v2.py:124
        names = ["deep%d" % i for i in range(len(self.hidden))]
        buf, n = _mlp_stack(pb, w, names, 1, 0, G * Kp, flat, [L.ACT_RELU] * len(names))
        hk = w["head/kernel"][:, 0]
        h0 = float(hk[0])
        pb.tap(0, scal_off, len(self.fields), None, scale=h0, bias=float(w["fo_cat/bias"][0]))
        pb.tap(0, num_off, nnum, w["fo_num/kernel"][:, 0], scale=h0, bias=float(w["fo_num/bias"][0]))
        pb.tap(0, fm_off, self.proj_dim, hk[1:1 + self.proj_dim])
        pb.tap(buf, 0, n, hk[1 + self.proj_dim:1 + self.proj_dim + n])
        pb.head_bias = float(w["head/bias"][0])


class DIN(CTRModel):
    """DIN.py:95-169.  History arrives as scalar keys ``userRatedMovie1..T`` (reference schema) or
    as one ``userRatedMovies`` [B,T] array; missing slots are id 0 exactly as in the reference
    (numeric_column default 0 -> Embedding row 0, no masking of the pooled sum)."""
    MODEL_KIND = L.MODEL_DIN
    FORWARD_SYMBOL = "sprk_forward_din"

    def __init__(self, weights=None, seed=None, emb_dim=10, hist_len=5, att_hidden=32, hidden=(128, 64),
                 movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS):
        self.emb_dim, self.hist_len, self.att_hidden, self.hidden = emb_dim, hist_len, att_hidden, tuple(hidden)
        self.movie_buckets, self.user_buckets = movie_buckets, user_buckets
        super().__init__(weights, seed)

    def _hist_keys(self):
        return ["userRatedMovie%d" % (i + 1) for i in range(self.hist_len)]

    def _id_columns(self):
        return ([IdColumn("movieId", "id", self.movie_buckets)]
                + [IdColumn(k, "id", self.movie_buckets) for k in self._hist_keys()]
                + [IdColumn("userId", "id", self.user_buckets), IdColumn("userGenre1", "genre", N_GENRES),
                   IdColumn("movieGenre1", "genre", N_GENRES)])

    _PROFILE_NUM = ["userRatingCount", "userAvgRating", "userRatingStddev"]       # DIN.py:111-113
    _CONTEXT_NUM = ["releaseYear", "movieRatingCount", "movieAvgRating", "movieRatingStddev"]  # DIN.py:119-122

    def _fc_rows(self):
        """Reference input-row index of every block of the tail's concat (DIN.py:161-162)."""
        D = self.emb_dim
        rows, r = {}, 0
        prof = {k: 1 for k in self._PROFILE_NUM}
        prof.update({"userId_embedding": D, "userGenre1_embedding": D})
        for n in sorted(prof):
            rows[n] = (r, prof[n])
            r += prof[n]
        rows["__pooled__"] = (r, D)
        r += D
        rows["__cand__"] = (r, D)
        r += D
        ctx = {k: 1 for k in self._CONTEXT_NUM}
        ctx["movieGenre1_embedding"] = D
        for n in sorted(ctx):
            rows[n] = (r, ctx[n])
            r += ctx[n]
        return rows, r

    def _numeric_kernel_rows(self):
        rows, _ = self._fc_rows()
        return {"fc0/kernel": {k: rows[k][0] for k in self.numeric_keys}}

    def weight_shapes(self):
        D, T, H = self.emb_dim, self.hist_len, self.att_hidden
        s = {"emb/movie": (self.movie_buckets, D), "emb/userId": (self.user_buckets, D),
             "emb/userGenre1": (N_GENRES, D), "emb/movieGenre1": (N_GENRES, D),
             "att0/kernel": (4 * D, H), "att0/bias": (H,), "att_prelu/alpha": (T, H),
             "att1/kernel": (H, 1), "att1/bias": (1,)}
        _, fan = self._fc_rows()
        for i, h in enumerate(self.hidden):
            s["fc%d/kernel" % i] = (fan, h)
            s["fc%d/bias" % i] = (h,)
            s["fc%d_prelu/alpha" % i] = (h,)
            fan = h
        s["head/kernel"] = (fan, 1)
        s["head/bias"] = (1,)
        return s

    def pack(self, features):
        if "userRatedMovies" in features and self._hist_keys()[0] not in features:
            hist = features["userRatedMovies"]
            hist = hist.detach().cpu().numpy() if hasattr(hist, "detach") else np.asarray(hist)
            if hist.ndim != 2 or hist.shape[1] != self.hist_len:
                raise ValueError("userRatedMovies must be [B, %d]" % self.hist_len)
            features = dict(features)
            for i, k in enumerate(self._hist_keys()):
                features[k] = hist[:, i]
        return super().pack(features)

    def _compile(self, pb, w):
        D, Dp, T = self.emb_dim, pad4(self.emb_dim), self.hist_len
        H, Hp = self.att_hidden, pad16(self.att_hidden)
        movie_tab = pad_table(w["emb/movie"])
        # ---- attention stage (k_din_pool) ----
        a0 = np.zeros((Hp, 4 * Dp), dtype=np.float32)              # [h-c | h | c | h*c] blocks, DIN.py:146-147
        for blk in range(4):
            a0[:H, blk * Dp:blk * Dp + D] = w["att0/kernel"][blk * D:(blk + 1) * D, :].T
        b0 = np.zeros(Hp, np.float32); b0[:H] = w["att0/bias"]
        al = np.zeros((T, Hp), np.float32); al[:, :H] = w["att_prelu/alpha"]
        w2 = np.zeros(Hp, np.float32); w2[:H] = w["att1/kernel"][:, 0]
        pb.n_aux = Dp
        pb.din = L.Din(1, T, pb.col(self._hist_keys()[0]), pb.col("movieId"), pb.slot(movie_tab), Dp,
                       self.movie_buckets, Hp, pb.slot(a0), pb.slot(b0), pb.slot(al), pb.slot(w2),
                       float(w["att1/bias"][0]), 0, 0)
        self._compile_tail(pb, w, movie_tab)

    def _compile_tail(self, pb, w, movie_tab):
        """concat -> Dense PReLU Dense PReLU Dense(1, sigmoid) (DIN.py:161-167, DIEN.py:252-259)."""
        D, Dp = self.emb_dim, pad4(self.emb_dim)
        rows, fan = self._fc_rows()
        x_of = {}
        # LDS layout of the tail's input: the four embedding columns first, then the per-sample data (pooled
        # history + numerics) side by side -- the engine folds embedding columns that feed only the first Dense
        # into per-id tables (fold_first_dense), and fc0's K range then is just this last stretch
        x_of["userGenre1_embedding"] = pb.seg_rows("userGenre1", pad_table(w["emb/userGenre1"]), N_GENRES, D)
        x_of["userId_embedding"] = pb.seg_rows("userId", pad_table(w["emb/userId"]), self.user_buckets, D)
        x_of["__cand__"] = pb.seg_rows("movieId", movie_tab, self.movie_buckets, D)
        x_of["movieGenre1_embedding"] = pb.seg_rows("movieGenre1", pad_table(w["emb/movieGenre1"]), N_GENRES, D)
        poff = pb.x_alloc(Dp)
        pb.seg_aux(0, Dp, poff)
        x_of["__pooled__"] = poff
        num_off = pb.seg_numerics(len(self.numeric_keys))
        for j, k in enumerate(self.numeric_keys):
            x_of[k] = num_off + j
        K0 = pad4(pb._x_end)
        xmap = [0] * fan
        for name, (r0, width) in rows.items():
            for j in range(width):
                xmap[r0 + j] = x_of[name] + j
        names = ["fc%d" % i for i in range(len(self.hidden))]
        alphas = ["fc%d_prelu/alpha" % i for i in range(len(self.hidden))]
        buf, n = _mlp_stack(pb, w, names, 0, 0, K0, xmap, [L.ACT_PRELU] * len(names), alphas)
        pb.tap(buf, 0, n, w["head/kernel"][:n, 0])
        pb.head_bias = float(w["head/bias"][0])


class DIEN(DIN):
    """DIEN.py:114-259, the y_pred output: shared movie Embedding (mask_zero consumed by the GRU) -> GRU ->
    per-slot attention gate -> AUGRU -> concat [augru, candidate, user profile, context] -> Dense(128) PReLU
    Dense(64) PReLU Dense(1, sigmoid).  The auxiliary-loss output (DIEN.py:253-292) is training-only and not built.

    The reference draws the AUGRU's initial state from GlorotUniform INSIDE call() (DIEN.py:239-240), i.e. a new
    random vector per forward pass; here it is the explicit weight ``augru/h0`` [1, D] so predictions are
    reproducible.  Instantiated for emb_dim 10 (the reference) and 16."""
    MODEL_KIND = L.MODEL_DIEN
    FORWARD_SYMBOL = "sprk_forward_dien"
    GATES = ("r", "z", "h")                                       # R_t, Z_t, H_t_next (DIEN.py:233-236)

    def __init__(self, weights=None, seed=None, emb_dim=10, hist_len=5, att_hidden=32, hidden=(128, 64),
                 movie_buckets=MOVIE_BUCKETS, user_buckets=USER_BUCKETS):
        if emb_dim not in (10, 16) or att_hidden != 32:
            raise ValueError("DIEN: emb_dim 10 or 16, attention width 32")
        super().__init__(weights, seed, emb_dim, hist_len, att_hidden, hidden, movie_buckets, user_buckets)

    def _fc_rows(self):
        """Input-row index of every block of the tail's concat (DIEN.py:252): augru, candidate, profile, context."""
        D = self.emb_dim
        rows, r = {"__pooled__": (0, D), "__cand__": (D, D)}, 2 * D
        prof = {k: 1 for k in self._PROFILE_NUM}
        prof.update({"userId_embedding": D, "userGenre1_embedding": D})
        for n in sorted(prof):
            rows[n] = (r, prof[n])
            r += prof[n]
        ctx = {k: 1 for k in self._CONTEXT_NUM}
        ctx["movieGenre1_embedding"] = D
        for n in sorted(ctx):
            rows[n] = (r, ctx[n])
            r += ctx[n]
        return rows, r

    def weight_shapes(self):
        D, H = self.emb_dim, self.att_hidden
        s = {"emb/movie": (self.movie_buckets, D), "emb/userId": (self.user_buckets, D),
             "emb/userGenre1": (N_GENRES, D), "emb/movieGenre1": (N_GENRES, D),
             "gru/kernel": (D, 3 * D), "gru_rec/kernel": (D, 3 * D), "gru/bias": (2, 3 * D),
             "att0/kernel": (D, H), "att0/bias": (H,), "att1/kernel": (H, 1), "att1/bias": (1,)}
        for g in self.GATES:
            s["augru_%s_in/kernel" % g] = (D, D)
            s["augru_%s_in/bias" % g] = (D,)
            s["augru_%s_hid/kernel" % g] = (D, D)
            s["augru_%s_out/kernel" % g] = (D, D)
            s["augru_%s_out/bias" % g] = (D,)
        s["augru/h0"] = (1, D)
        _, fan = self._fc_rows()
        for i, h in enumerate(self.hidden):
            s["fc%d/kernel" % i] = (fan, h)
            s["fc%d/bias" % i] = (h,)
            s["fc%d_prelu/alpha" % i] = (h,)
            fan = h
        s["head/kernel"] = (fan, 1)
        s["head/bias"] = (1,)
        return s

    def _seq_image(self, w):
        """The packed sequence-stage weights, layout of include/sparrow_hip.h sprk_din.seq_slot (k_dien_seq.h)."""
        D, H = self.emb_dim, self.att_hidden
        Dq, N3 = pad4(D), pad4(3 * D)

        def mat(a, stride):
            out = np.zeros((a.shape[0], stride), np.float32)
            out[:, :a.shape[1]] = a
            return out.ravel()

        def vec(a, n):
            out = np.zeros(n, np.float32)
            out[:a.size] = np.asarray(a, np.float32).ravel()
            return out
        parts = [mat(w["gru/kernel"], N3), mat(w["gru_rec/kernel"], N3), mat(w["gru/bias"], N3),
                 mat(w["att0/kernel"], H), vec(w["att0/bias"], H), vec(w["att1/kernel"], H), vec(w["att1/bias"], 4)]
        for g in self.GATES:
            parts += [mat(w["augru_%s_in/kernel" % g], Dq), vec(w["augru_%s_in/bias" % g], Dq),
                      mat(w["augru_%s_hid/kernel" % g], Dq), mat(w["augru_%s_out/kernel" % g], Dq),
                      vec(w["augru_%s_out/bias" % g], Dq)]
        parts.append(vec(w["augru/h0"], Dq))
        img = np.concatenate(parts)
        return np.concatenate([img, np.zeros((-img.size) % 64, np.float32)])

    def _compile(self, pb, w):
        D, Dp, T = self.emb_dim, pad4(self.emb_dim), self.hist_len
        movie_tab = pad_table(w["emb/movie"])
        pb.n_aux = Dp
        pb.din = L.Din(2, T, pb.col(self._hist_keys()[0]), pb.col("movieId"), pb.slot(movie_tab), Dp,
                       self.movie_buckets, self.att_hidden, 0, 0, 0, 0, 0.0, D, pb.slot(self._seq_image(w)))
        self._compile_tail(pb, w, movie_tab)

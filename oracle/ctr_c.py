"""ctypes wrapper of oracle/ctr_c.c (CPU ORACLE, test infrastructure: only tests/ and bench.py's cpu_baseline leg use it).

``deepfm_v2_forward_c`` takes the PACKED arrays the product's host code produces (ids [B,F] int32 in stack order, -1 =
missing; dense [B,N] float32 in name-sorted numeric order) and the reference-layout weights dict of models.DeepFMv2."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libctr_c.so")
_lib = None


def load(native: bool = False):
    """The generic build under oracle/_build (made on demand); ``native=True`` first tries a -march=native build in a
    temporary directory ON THIS HOST (never shipped: a native build from another machine could hit SIGILL here)."""
    global _lib
    if _lib is None:
        if native:
            try:
                import tempfile
                tmp = os.path.join(tempfile.mkdtemp(prefix="sprk_ctr_c_"), "libctr_c_native.so")
                subprocess.run([os.environ.get("CC", "gcc"), "-O3", "-march=native", "-fPIC", "-shared", "-fopenmp",
                                "-ffp-contract=off", os.path.join(HERE, "ctr_c.c"), "-o", tmp, "-lm"],
                               check=True, capture_output=True, timeout=120)
                _lib = C.CDLL(tmp)
                return _lib
            except Exception:
                pass
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
        _lib = C.CDLL(LIB)
    return _lib


def _ptr_array(arrays):
    return (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


class DeepFMv2C:
    """Weights of a models.DeepFMv2 laid out once for the C forward (reference layouts, contiguous fp32)."""

    def __init__(self, w, fields, order=None):
        from oracle.ctr_oracle import first_order_offsets
        order = list(order or [k for k, _, _ in fields])
        vocab = {k: v for k, _, v in fields}
        offs = first_order_offsets(fields)
        c = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.F, self.D = len(order), int(w["emb/" + order[0]].shape[1])
        self.K, self.NN = int(w["proj/num/kernel"].shape[1]), int(w["proj/num/kernel"].shape[0])
        self.H0, self.H1 = int(w["deep0/kernel"].shape[1]), int(w["deep1/kernel"].shape[1])
        fk = c(w["fo_cat/kernel"])[:, 0]
        self.tables = [c(w["emb/" + k]) for k in order]
        self.fo = [np.ascontiguousarray(fk[offs[k]:offs[k] + vocab[k]]) for k in order]
        self.Wp = [c(w["proj/%s/kernel" % k]) for k in order] + [c(w["proj/num/kernel"])]
        self.bp = [c(w["proj/%s/bias" % k]) for k in order] + [c(w["proj/num/bias"])]
        self.fo_bias, self.fo_num_b = float(w["fo_cat/bias"][0]), float(w["fo_num/bias"][0])
        self.fo_num_w = c(w["fo_num/kernel"])[:, 0].copy()
        self.W0, self.b0, self.W1, self.b1 = c(w["deep0/kernel"]), c(w["deep0/bias"]), c(w["deep1/kernel"]), c(w["deep1/bias"])
        self.head_w, self.head_b = c(w["head/kernel"])[:, 0].copy(), float(w["head/bias"][0])
        assert "deep2/kernel" not in w and self.head_w.size == 1 + self.K + self.H1
        self._p = (_ptr_array(self.tables), _ptr_array(self.fo), _ptr_array(self.Wp), _ptr_array(self.bp))

    def forward(self, ids, dense, threads=1, out=None):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        dense = np.ascontiguousarray(dense, dtype=np.float32)
        B = ids.shape[0]
        assert ids.shape[1] == self.F and dense.shape == (B, self.NN)
        if out is None:
            out = np.empty(B, dtype=np.float32)
        vp = C.c_void_p
        load().deepfm_v2_forward_c(
            C.c_int32(B), C.c_int32(self.F), C.c_int32(self.D), C.c_int32(self.K), C.c_int32(self.NN), C.c_int32(self.H0),
            C.c_int32(self.H1), vp(ids.ctypes.data), vp(dense.ctypes.data), self._p[0], self._p[1], self._p[2], self._p[3],
            C.c_float(self.fo_bias), vp(self.fo_num_w.ctypes.data), C.c_float(self.fo_num_b), vp(self.W0.ctypes.data),
            vp(self.b0.ctypes.data), vp(self.W1.ctypes.data), vp(self.b1.ctypes.data), vp(self.head_w.ctypes.data),
            C.c_float(self.head_b), vp(out.ctypes.data), C.c_int32(int(threads)))
        return out


class DinC:
    """Weights of a models.DIN laid out once for the C forward (din_forward_c); ``model`` supplies the concat order of the
    tail (its ``_fc_rows``), the ids column order of ``pack`` and the numeric key order."""

    def __init__(self, model):
        w = model.weights
        c = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.T, self.D, self.H = model.hist_len, model.emb_dim, model.att_hidden
        cols = [col.key for col in model.id_columns]
        self.F, self.ND = len(cols), len(model.numeric_keys)
        self.hist_col, self.cand_col = cols.index(model._hist_keys()[0]), cols.index("movieId")
        self.table = c(w["emb/movie"])
        self.Wa, self.ba, self.alpha = c(w["att0/kernel"]), c(w["att0/bias"]), c(w["att_prelu/alpha"])
        self.w2, self.b2 = c(w["att1/kernel"])[:, 0].copy(), float(w["att1/bias"][0])
        extra_names = ["userId", "userGenre1", "movieGenre1"]
        self.extra = [c(w["emb/" + k]) for k in extra_names]
        rows, fan = model._fc_rows()
        seg = []
        for name, (r0, width) in rows.items():
            if name == "__pooled__":
                seg.append((r0, width, 2, 0, 0))
            elif name == "__cand__":
                seg.append((r0, width, 3, 0, 0))
            elif name.endswith("_embedding"):
                k = name[:-len("_embedding")]
                seg.append((r0, width, 1, extra_names.index(k), cols.index(k)))
            else:
                seg.append((r0, 1, 0, list(model.numeric_keys).index(name), 0))
        self.seg = np.ascontiguousarray(np.array(seg, dtype=np.int32))
        self.X = fan
        self.W0, self.b0, self.a0 = c(w["fc0/kernel"]), c(w["fc0/bias"]), c(w["fc0_prelu/alpha"])
        self.W1, self.b1, self.a1 = c(w["fc1/kernel"]), c(w["fc1/bias"]), c(w["fc1_prelu/alpha"])
        self.N0, self.N1 = self.W0.shape[1], self.W1.shape[1]
        self.hw, self.hb = c(w["head/kernel"])[:, 0].copy(), float(w["head/bias"][0])
        self._extra_p = _ptr_array(self.extra)

    def forward(self, ids, dense, threads=1, out=None):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        dense = np.ascontiguousarray(dense, dtype=np.float32)
        B = ids.shape[0]
        assert ids.shape[1] == self.F and dense.shape == (B, self.ND)
        if out is None:
            out = np.empty(B, dtype=np.float32)
        vp, i32 = C.c_void_p, C.c_int32
        p = lambda a: vp(a.ctypes.data)
        load().din_forward_c(
            i32(B), i32(self.F), i32(self.ND), i32(self.T), i32(self.D), i32(self.H), i32(self.hist_col), i32(self.cand_col),
            p(ids), p(dense), p(self.table), p(self.Wa), p(self.ba), p(self.alpha), p(self.w2), C.c_float(self.b2),
            i32(len(self.seg)), p(self.seg), self._extra_p, i32(self.X), i32(self.N0), i32(self.N1), p(self.W0), p(self.b0),
            p(self.a0), p(self.W1), p(self.b1), p(self.a1), p(self.hw), C.c_float(self.hb), p(out), i32(int(threads)))
        return out

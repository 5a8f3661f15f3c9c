"""TEST INFRASTRUCTURE (never imported by the product path): a small numpy interpreter for the GraphDefs the
REFERENCE ITSELF exported -- `src/main/resources/webroot/modeldata/{neuralcf,MLPRec}/*/saved_model.pb`, written by
`tf.keras.models.save_model` at the end of the reference's training scripts (NeuralCF.py:97-105, EmbeddingMLP.py).

Why it exists: TensorFlow cannot be installed here, so the oracle (oracle/ctr_oracle.py) is a restatement of what the
reference's tf.feature_column / Keras calls MEAN.  The SavedModels pin that meaning to something the reference produced:
they contain the exact op-level wiring TensorFlow generated for `categorical_column_with_identity` +
`embedding_column` (ExpandDims -> Where/GatherNd sparse conversion -> range asserts -> SparseReshape ->
SparseFillEmptyRows -> Unique -> ResourceGather -> SparseSegmentMean -> Select(zeros) ...),
`categorical_column_with_vocabulary_list` + `indicator_column` (LookupTableFindV2 -> SparseToDense -> OneHot -> Sum),
`numeric_column`, `DenseFeatures`' column order (the ConcatV2's input order), `Dense`, `concatenate`, `Dot`
(BatchMatMulV2) -- together with the trained variables.  Executing that wiring op by op and comparing with the oracle
checks every one of those restated semantics against the reference's own artifact, not against a second reading of the
Python scripts.  What it can NOT pin: the arithmetic inside each TF op is still restated here (from the published op
definitions), and graphs the reference never exported (DIN, DeepFM, Wide&Deep's crossed column, DIEN).

Format notes (protobuf wire format, decoded by field number; no TensorFlow / no generated classes):
  SavedModel{2: MetaGraphDef{2: GraphDef{1: NodeDef*, 2: FunctionDefLibrary{1: FunctionDef*}}, 5: SignatureDef map}}
  NodeDef{1 name, 2 op, 3 input*, 5 attr map<string, AttrValue>}
  AttrValue{1 list, 2 s, 3 i, 4 f, 5 b, 6 type, 7 shape, 8 tensor, 10 func{1 name}}
  FunctionDef{1 signature OpDef{1 name, 2 input_arg{1 name}, 3 output_arg{1 name}}, 3 node_def*, 4 ret map}
  inside a function an input is "arg", "node:out_arg:idx" or "^control"; in the main graph "node:idx".
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

# ---------------------------------------------------------------------------------------------
# protobuf wire format
# ---------------------------------------------------------------------------------------------


def _varint(b: bytes, p: int) -> Tuple[int, int]:
    r = s = 0
    while True:
        x = b[p]
        p += 1
        r |= (x & 0x7F) << s
        if not x & 0x80:
            return r, p
        s += 7


def _fields(b: bytes):
    p, n = 0, len(b)
    while p < n:
        key, p = _varint(b, p)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, p = _varint(b, p)
        elif wt == 1:
            v = b[p:p + 8]
            p += 8
        elif wt == 2:
            ln, p = _varint(b, p)
            v = b[p:p + ln]
            p += ln
        elif wt == 5:
            v = b[p:p + 4]
            p += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield f, wt, v


def _sint(v: int) -> int:
    """varint -> signed int64 (protobuf int64 fields are two's complement in 64 bits)."""
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v, wt) -> List[int]:
    if wt == 0:
        return [_sint(v)]
    out, p = [], 0
    while p < len(v):
        x, p = _varint(v, p)
        out.append(_sint(x))
    return out


# tensorflow/core/framework/types.proto
_NP = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
       17: np.uint16, 22: np.uint32, 23: np.uint64}
DT_STRING, DT_RESOURCE = 7, 20


def _shape(b: bytes) -> List[int]:
    dims = []
    for f, wt, v in _fields(b):
        if f == 2:
            size = 0
            for ff, w2, vv in _fields(v):
                if ff == 1:
                    size = _sint(vv)
            dims.append(size)
    return dims


def _tensor(b: bytes):
    dtype, shape, content = 0, [], None
    vals: List = []
    for f, wt, v in _fields(b):
        if f == 1:
            dtype = v
        elif f == 2:
            shape = _shape(v)
        elif f == 4:
            content = v
        elif f == 5:                                  # float_val
            vals += list(struct.unpack("<%df" % (len(v) // 4), v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 6:                                  # double_val
            vals += list(struct.unpack("<%dd" % (len(v) // 8), v)) if wt == 2 else [struct.unpack("<d", v)[0]]
        elif f in (7, 10, 11):                        # int_val / int64_val / bool_val
            vals += _packed_varints(v, wt)
        elif f == 8:                                  # string_val
            vals.append(v)
    n = int(np.prod(shape)) if shape else 1
    if dtype == DT_STRING:
        if not vals:
            vals = [b""]
        if len(vals) == 1 and n > 1:
            vals = vals * n
        return np.array(vals, dtype=object).reshape(shape)
    np_dt = _NP[dtype]
    if content is not None:
        return np.frombuffer(content, dtype=np.dtype(np_dt).newbyteorder("<")).astype(np_dt).reshape(shape)
    if not vals:
        return np.zeros(shape, np_dt)
    a = np.array(vals, dtype=np_dt)
    if a.size == 1 and n != 1:
        a = np.full(n, a[0], dtype=np_dt)            # TensorProto's "one value fills the tensor" encoding
    elif a.size < n:
        a = np.concatenate([a, np.full(n - a.size, a[-1], dtype=np_dt)])
    return a.reshape(shape)


def _attr(b: bytes):
    for f, wt, v in _fields(b):
        if f == 2:
            return v                                  # bytes
        if f == 3:
            return _sint(v)
        if f == 4:
            return struct.unpack("<f", v)[0]
        if f == 5:
            return bool(v)
        if f == 6:
            return ("type", v)
        if f == 7:
            return ("shape", _shape(v))
        if f == 8:
            return _tensor(v)
        if f == 10:
            for ff, w2, vv in _fields(v):
                if ff == 1:
                    return ("func", vv.decode())
        if f == 1:                                    # list
            out = []
            for ff, w2, vv in _fields(v):
                if ff == 2:
                    out.append(vv)
                elif ff == 3:
                    out += _packed_varints(vv, w2)
                elif ff == 6:
                    out += [("type", t) for t in _packed_varints(vv, w2)]
                elif ff == 7:
                    out.append(("shape", _shape(vv)))
            return out
    return None


class Node:
    __slots__ = ("name", "op", "inputs", "attr")

    def __init__(self, b: bytes):
        self.inputs, self.attr = [], {}
        self.name = self.op = ""
        for f, wt, v in _fields(b):
            if f == 1:
                self.name = v.decode()
            elif f == 2:
                self.op = v.decode()
            elif f == 3:
                self.inputs.append(v.decode())
            elif f == 5:
                k = val = None
                for ff, w2, vv in _fields(v):
                    if ff == 1:
                        k = vv.decode()
                    elif ff == 2:
                        val = vv
                self.attr[k] = val                    # decoded lazily (_attr)

    def a(self, key, default=None):
        return _attr(self.attr[key]) if key in self.attr else default


class Function:
    def __init__(self, b: bytes):
        self.nodes: Dict[str, Node] = {}
        self.order: List[Node] = []
        self.args: List[str] = []
        self.outs: List[str] = []
        self.ret: Dict[str, str] = {}
        self.name = ""
        for f, wt, v in _fields(b):
            if f == 1:
                for ff, w2, vv in _fields(v):
                    if ff == 1:
                        self.name = vv.decode()
                    elif ff in (2, 3):
                        nm = [x for g, w3, x in _fields(vv) if g == 1][0].decode()
                        (self.args if ff == 2 else self.outs).append(nm)
            elif f == 3:
                n = Node(v)
                self.nodes[n.name] = n
                self.order.append(n)
            elif f == 4:
                k = val = None
                for ff, w2, vv in _fields(v):
                    if ff == 1:
                        k = vv.decode()
                    elif ff == 2:
                        val = vv.decode()
                self.ret[k] = val


# output-arg names of the multi-output ops these graphs use (tensorflow/core/ops/*.cc), in flat order
_OUT_ARGS = {
    "Unique": ["y", "idx"],
    "SparseFillEmptyRows": ["output_indices", "output_values", "empty_row_indicator", "reverse_index_map"],
    "SparseReshape": ["output_indices", "output_shape"],
    "RestoreV2": ["tensors"],
}


class Resource:
    """A variable or a hash table."""

    def __init__(self, kind, name):
        self.kind, self.name, self.value, self.table, self.default = kind, name, None, None, None


class SavedModel:
    """The serving_default signature of one exported model, executable with numpy."""

    def __init__(self, saved_model_pb: bytes, variables: Dict[str, np.ndarray]):
        """variables: checkpoint key -> array (the `variables/` TensorBundle, read by the caller)."""
        mg = [v for f, wt, v in _fields(saved_model_pb) if f == 2][0]
        gd = [v for f, wt, v in _fields(mg) if f == 2][0]
        self.bundle = variables
        self.main: Dict[str, Node] = {}
        self.main_order: List[Node] = []
        self.funcs: Dict[str, Function] = {}
        for f, wt, v in _fields(gd):
            if f == 1:
                n = Node(v)
                self.main[n.name] = n
                self.main_order.append(n)
            elif f == 2:
                for ff, w2, vv in _fields(v):
                    if ff == 1:
                        fn = Function(vv)
                        self.funcs[fn.name] = fn
        # signature "serving_default": input key -> placeholder tensor name, output key -> tensor name
        self.sig_inputs: Dict[str, str] = {}
        self.sig_outputs: Dict[str, str] = {}
        for f, wt, v in _fields(mg):
            if f != 5:
                continue
            key = sig = None
            for ff, w2, vv in _fields(v):
                if ff == 1:
                    key = vv.decode()
                elif ff == 2:
                    sig = vv
            if key != "serving_default":
                continue
            for ff, w2, vv in _fields(sig):
                if ff in (1, 2):
                    k = info = None
                    for g, w3, x in _fields(vv):
                        if g == 1:
                            k = x.decode()
                        elif g == 2:
                            info = x
                    tname = [x for g, w3, x in _fields(info) if g == 1][0].decode()
                    (self.sig_inputs if ff == 1 else self.sig_outputs)[k] = tname
        self.ops_executed: Dict[str, int] = {}
        self.capture: set = set()                     # node names (inside any function) whose outputs predict() keeps ...
        self.captured: Dict[str, list] = {}           # ... here (e.g. the DenseFeatures ConcatV2, to look at a layer's input)
        self._main_vals: Dict[str, list] = {}
        self._restore_and_init()

    # ---- main graph -------------------------------------------------------------------------
    def _restore_and_init(self):
        """What loading a SavedModel does: run the restore function (RestoreV2 + AssignVariableOp per variable) and the
        table initialisers (LookupTableImportV2)."""
        for n in self.main_order:
            if n.op == "StatefulPartitionedCall":
                fname = n.a("f")[1]
                if "traced_restore" in fname:
                    args = [self._main_eval(i) if not i.startswith("saver_filename") else np.array(b"", dtype=object) for i in n.inputs if not i.startswith("^")]
                    self._call(fname, args)
                elif any(x.op == "LookupTableImportV2" for x in self.funcs[fname].order):
                    self._call(fname, [self._main_eval(i) for i in n.inputs if not i.startswith("^")])

    def _main_eval(self, tensor: str):
        name, _, idx = tensor.partition(":")
        idx = int(idx) if idx else 0
        if name not in self._main_vals:
            n = self.main[name]
            if n.op in ("VarHandleOp", "HashTableV2", "MutableHashTableV2"):
                out = [Resource("var" if n.op == "VarHandleOp" else "table", (n.a("shared_name") or b"").decode() or n.name)]
            elif n.op == "Const":
                out = [n.a("value")]
            elif n.op == "Placeholder":
                raise KeyError("placeholder %s was not fed" % name)
            else:
                out = self._exec(n, [self._main_eval(i) for i in n.inputs if not i.startswith("^")])
            self._main_vals[name] = out
        return self._main_vals[name][idx]

    def predict(self, features: Dict[str, np.ndarray]) -> np.ndarray:
        """features: signature input key -> array [B] (strings as bytes/str objects).  Returns the signature's single output."""
        fed = {}
        for key, tname in self.sig_inputs.items():
            if key not in features:
                raise KeyError("signature input %r missing" % key)
            node = self.main[tname.partition(":")[0]]
            dt = node.a("dtype")[1]
            a = np.asarray(features[key])
            if dt == DT_STRING:
                a = np.array([x if isinstance(x, bytes) else str(x).encode() for x in a.ravel()], dtype=object).reshape(a.shape)
            else:
                a = a.astype(_NP[dt])
            shp = node.a("shape")
            if shp and len(shp[1]) == 2 and a.ndim == 1:          # Keras Input(shape=()) exports as [None, 1] in some versions
                a = a.reshape(-1, 1)
            fed[tname.partition(":")[0]] = [a]
        keep = {k: v for k, v in self._main_vals.items() if self.main[k].op in ("VarHandleOp", "HashTableV2", "MutableHashTableV2", "Const")}
        self._main_vals = dict(keep)
        self._main_vals.update(fed)
        (out_t,) = self.sig_outputs.values()
        res = self._main_eval(out_t)
        self._main_vals = keep
        return res

    # ---- functions --------------------------------------------------------------------------
    def _call(self, fname: str, args: list) -> list:
        fn = self.funcs[fname]
        vals: Dict[str, list] = {a: [x] for a, x in zip(fn.args, args)}
        if len(args) != len(fn.args):
            raise ValueError("%s: %d args for %d parameters" % (fname, len(args), len(fn.args)))

        def ev(ref: str):
            parts = ref.split(":")
            name = parts[0]
            if name not in vals:
                n = fn.nodes[name]
                ins = [ev(i) for i in n.inputs if not i.startswith("^")]
                for i in n.inputs:                               # control dependencies run too (asserts)
                    if i.startswith("^") and i[1:] not in vals and i[1:] in fn.nodes:
                        ev(i[1:])
                vals[name] = self._exec(n, ins)
                if name in self.capture:
                    self.captured[name] = vals[name]
            if len(parts) == 1:
                return vals[name][0]
            if len(parts) == 2:
                return vals[name][int(parts[1])]
            arg, idx = parts[1], int(parts[2])
            n = fn.nodes.get(name)
            names = _OUT_ARGS.get(n.op) if n is not None else None
            if names and len(names) > 1:
                return vals[name][names.index(arg) + idx]
            return vals[name][idx]

        for n in fn.order:                                       # stateful nodes nobody consumes (AssignVariableOp, imports, asserts)
            if n.op in ("AssignVariableOp", "LookupTableImportV2", "Assert") and n.name not in vals:
                ev(n.name)
        return [ev(fn.ret[o]) for o in fn.outs]

    # ---- ops --------------------------------------------------------------------------------
    def _exec(self, n: Node, x: list) -> list:
        op = n.op
        self.ops_executed[op] = self.ops_executed.get(op, 0) + 1
        if op in ("StatefulPartitionedCall", "PartitionedCall"):
            return self._call(n.a("f")[1], x)
        if op in ("If", "StatelessIf"):
            cond = bool(np.asarray(x[0]).reshape(-1)[0]) if np.asarray(x[0]).size else False
            return self._call(n.a("then_branch" if cond else "else_branch")[1], x[1:])
        if op == "Const":
            return [n.a("value")]
        if op in ("Identity", "StopGradient", "PreventGradient", "Snapshot"):
            return [x[0]]
        if op == "IdentityN":
            return list(x)
        if op == "NoOp":
            return [None]
        if op == "Assert":
            if not bool(np.all(x[0])):
                raise AssertionError("tf.Assert failed in %s: %s" % (n.name, [np.asarray(v).tolist() if not isinstance(v, Resource) else v.name for v in x[1:]][:3]))
            return [None]
        if op == "VarHandleOp":
            return [Resource("var", (n.a("shared_name") or b"").decode() or n.name)]
        if op == "ReadVariableOp":
            if x[0].value is None:
                raise ValueError("variable %s was never restored" % x[0].name)
            return [x[0].value]
        if op == "AssignVariableOp":
            x[0].value = np.asarray(x[1])
            return [None]
        if op == "RestoreV2":
            names = [s.decode() for s in np.asarray(x[1]).ravel()]
            return [self.bundle[k] if k in self.bundle else None for k in names]
        if op in ("HashTableV2", "MutableHashTableV2"):
            return [Resource("table", (n.a("shared_name") or b"").decode() or n.name)]
        if op == "LookupTableImportV2":
            keys, vals = np.asarray(x[1]).ravel(), np.asarray(x[2]).ravel()
            x[0].table = {k: v for k, v in zip(keys.tolist(), vals.tolist())}
            return [None]
        if op == "LookupTableFindV2":
            tab, keys, default = x
            if tab.table is None:
                raise ValueError("hash table %s was never initialised" % tab.name)
            k = np.asarray(keys)
            d = np.asarray(default).reshape(-1)[0]
            return [np.array([tab.table.get(v, d) for v in k.ravel().tolist()], dtype=np.asarray(default).dtype).reshape(k.shape)]
        if op == "Cast":
            dst = n.a("DstT")[1]
            a = np.asarray(x[0])
            if dst == DT_STRING:
                raise NotImplementedError("Cast to string")
            if a.dtype.kind == "f" and np.dtype(_NP[dst]).kind in "iu":
                return [np.trunc(a).astype(_NP[dst])]
            return [a.astype(_NP[dst])]
        if op == "Shape":
            return [np.array(np.asarray(x[0]).shape, dtype=_NP[n.a("out_type", ("type", 3))[1]])]
        if op == "ExpandDims":
            return [np.expand_dims(np.asarray(x[0]), int(np.asarray(x[1]).reshape(-1)[0]))]
        if op == "Squeeze":
            dims = n.a("squeeze_dims") or []
            return [np.squeeze(np.asarray(x[0]), axis=tuple(dims) if dims else None)]
        if op == "Reshape":
            return [np.reshape(np.asarray(x[0]), [int(v) for v in np.asarray(x[1]).ravel()])]
        if op == "Pack":
            return [np.stack([np.asarray(v) for v in x], axis=n.a("axis", 0))]
        if op == "ConcatV2":
            return [np.concatenate([np.asarray(v) for v in x[:-1]], axis=int(np.asarray(x[-1]).reshape(-1)[0]))]
        if op == "Tile":
            return [np.tile(np.asarray(x[0]), [int(v) for v in np.asarray(x[1]).ravel()])]
        if op == "ZerosLike":
            return [np.zeros_like(np.asarray(x[0]))]
        if op == "Slice":
            a = np.asarray(x[0])
            begin, size = [int(v) for v in np.asarray(x[1]).ravel()], [int(v) for v in np.asarray(x[2]).ravel()]
            return [a[tuple(slice(b, a.shape[i] if s == -1 else b + s) for i, (b, s) in enumerate(zip(begin, size)))]]
        if op == "StridedSlice":
            return [self._strided_slice(n, x)]
        if op in ("NotEqual", "Equal", "Less", "GreaterEqual", "Greater", "LessEqual"):
            a, b = np.asarray(x[0]), np.asarray(x[1])
            f = {"NotEqual": np.not_equal, "Equal": np.equal, "Less": np.less, "GreaterEqual": np.greater_equal,
                 "Greater": np.greater, "LessEqual": np.less_equal}[op]
            return [np.asarray(f(a, b), dtype=np.bool_)]
        if op in ("All", "Sum", "Prod", "Max"):
            a = np.asarray(x[0])
            axes = tuple(int(v) for v in np.asarray(x[1]).ravel())
            keep = bool(n.a("keep_dims", False))
            f = {"All": np.all, "Sum": np.sum, "Prod": np.prod, "Max": np.max}[op]
            r = f(a, axis=axes, keepdims=keep)
            return [np.asarray(r, dtype=a.dtype if op != "All" else np.bool_)]
        if op == "Where":
            return [np.argwhere(np.asarray(x[0])).astype(np.int64).reshape(-1, np.asarray(x[0]).ndim)]
        if op == "GatherNd":
            a, idx = np.asarray(x[0]), np.asarray(x[1])
            return [a[tuple(idx[..., i] for i in range(idx.shape[-1]))]]
        if op == "GatherV2":
            a, idx, axis = np.asarray(x[0]), np.asarray(x[1]), int(np.asarray(x[2]).reshape(-1)[0])
            return [np.take(a, idx, axis=axis)]
        if op == "ResourceGather":
            table, idx = x[0].value, np.asarray(x[1])
            if idx.size and (idx.min() < 0 or idx.max() >= table.shape[0]):
                raise IndexError("ResourceGather index out of range in %s" % n.name)
            return [table[idx]]
        if op == "Unique":
            a = np.asarray(x[0])
            # tf.unique: y in order of FIRST OCCURRENCE, idx maps each element to its position in y
            _, first, inv = np.unique(a, return_index=True, return_inverse=True)
            order = np.argsort(first, kind="stable")
            rank = np.empty_like(order)
            rank[order] = np.arange(order.size)
            return [a[np.sort(first)], rank[inv].astype(_NP[n.a("out_idx", ("type", 3))[1]])]
        if op == "SparseReshape":
            idx, shape, new = np.asarray(x[0]), np.asarray(x[1]).astype(np.int64), np.asarray(x[2]).astype(np.int64).copy()
            total = int(np.prod(shape))
            if (new == -1).any():
                known = int(np.prod(new[new != -1])) if (new != -1).any() else 1
                new[new == -1] = total // max(known, 1)
            flat = np.ravel_multi_index(tuple(idx[:, i] for i in range(idx.shape[1])), tuple(int(s) for s in shape)) if idx.size else np.zeros(0, np.int64)
            out = np.stack(np.unravel_index(flat, tuple(int(s) for s in new)), axis=1).astype(np.int64) if idx.size else np.zeros((0, new.size), np.int64)
            return [out, new]
        if op == "SparseFillEmptyRows":
            idx, vals, dense_shape, default = np.asarray(x[0]), np.asarray(x[1]), np.asarray(x[2]), np.asarray(x[3]).reshape(-1)[0]
            rows = int(dense_shape[0])
            present = np.zeros(rows, np.bool_)
            present[idx[:, 0]] = True
            empty = ~present
            # output: the original entries plus (row, 0) = default for every empty row, in row-major order
            add = np.nonzero(empty)[0]
            all_idx = np.concatenate([idx, np.stack([add, np.zeros_like(add)], axis=1).astype(idx.dtype)]) if add.size else idx
            all_val = np.concatenate([vals, np.full(add.size, default, dtype=vals.dtype)]) if add.size else vals
            order = np.lexsort((all_idx[:, 1], all_idx[:, 0]))
            pos_of = np.empty(order.size, np.int64)
            pos_of[order] = np.arange(order.size)
            return [all_idx[order], all_val[order], empty, pos_of[:idx.shape[0]]]   # reverse_index_map: input entry -> output position
        if op == "SparseSegmentMean":
            data, idx, seg = np.asarray(x[0]), np.asarray(x[1]).astype(np.int64), np.asarray(x[2]).astype(np.int64)
            nseg = int(seg.max()) + 1 if seg.size else 0
            out = np.zeros((nseg,) + data.shape[1:], dtype=data.dtype)
            cnt = np.zeros(nseg, dtype=np.int64)
            np.add.at(out, seg, data[idx])
            np.add.at(cnt, seg, 1)
            nz = cnt > 0
            out[nz] = (out[nz] / cnt[nz].reshape((-1,) + (1,) * (data.ndim - 1)).astype(data.dtype)).astype(data.dtype)
            return [out]
        if op == "SparseToDense":
            idx, shape, vals, default = np.asarray(x[0]), [int(v) for v in np.asarray(x[1]).ravel()], np.asarray(x[2]), np.asarray(x[3])
            out = np.full(shape, default.reshape(-1)[0], dtype=vals.dtype if vals.dtype != object else object)
            if idx.size:
                out[tuple(idx[:, i] for i in range(idx.shape[1]))] = vals
            return [out]
        if op == "OneHot":
            ind, depth, on, off = np.asarray(x[0]), int(np.asarray(x[1]).reshape(-1)[0]), np.asarray(x[2]).reshape(-1)[0], np.asarray(x[3]).reshape(-1)[0]
            out = np.full(ind.shape + (depth,), off, dtype=np.asarray(x[2]).dtype)
            ok = (ind >= 0) & (ind < depth)
            pos = np.nonzero(ok)
            out[pos + (ind[ok].astype(np.int64),)] = on
            return [out]
        if op in ("Select", "SelectV2"):
            c, a, b = np.asarray(x[0]), np.asarray(x[1]), np.asarray(x[2])
            if op == "Select" and c.ndim == 1 and a.ndim > 1:
                c = c.reshape((-1,) + (1,) * (a.ndim - 1))      # Select broadcasts a vector condition over rows
            return [np.where(c, a, b)]
        if op == "MatMul":
            a, b = np.asarray(x[0]), np.asarray(x[1])
            if n.a("transpose_a", False):
                a = a.T
            if n.a("transpose_b", False):
                b = b.T
            return [np.matmul(a, b).astype(a.dtype)]
        if op == "BatchMatMulV2":
            a, b = np.asarray(x[0]), np.asarray(x[1])
            if n.a("adj_x", False):
                a = np.swapaxes(a, -1, -2)
            if n.a("adj_y", False):
                b = np.swapaxes(b, -1, -2)
            return [np.matmul(a, b).astype(a.dtype)]
        if op == "BiasAdd":
            return [np.asarray(x[0]) + np.asarray(x[1])]
        if op in ("AddV2", "Add"):
            return [np.asarray(x[0]) + np.asarray(x[1])]
        if op == "Sub":
            return [np.asarray(x[0]) - np.asarray(x[1])]
        if op == "Mul":
            return [np.asarray(x[0]) * np.asarray(x[1])]
        if op == "Relu":
            return [np.maximum(np.asarray(x[0]), 0).astype(np.asarray(x[0]).dtype)]
        if op == "Sigmoid":
            a = np.asarray(x[0])
            return [(1.0 / (1.0 + np.exp(-a.astype(np.float64)))).astype(a.dtype)]
        if op == "LogicalAnd":
            return [np.logical_and(x[0], x[1])]
        raise NotImplementedError("op %s (node %s)" % (op, n.name))

    @staticmethod
    def _strided_slice(n: Node, x: list):
        a = np.asarray(x[0])
        begin, end, strides = ([int(v) for v in np.asarray(t).ravel()] for t in x[1:4])
        bm, em, sm, nm, elm = (int(n.a(k, 0) or 0) for k in ("begin_mask", "end_mask", "shrink_axis_mask", "new_axis_mask", "ellipsis_mask"))
        if nm or elm:
            raise NotImplementedError("StridedSlice new_axis / ellipsis masks")
        index = []
        for i in range(len(begin)):
            if sm & (1 << i):
                index.append(begin[i])
                continue
            b = None if bm & (1 << i) else begin[i]
            e = None if em & (1 << i) else end[i]
            index.append(slice(b, e, strides[i]))
        return a[tuple(index)]

"""CPU ORACLE, part 3 (test infrastructure, NOT the product): a numpy stand-in for the slice of the ``tensorflow`` API the
reference's model scripts touch, so that the reference's OWN model-building source -- the untouched lines of
``TFRecModel/src/com/sparrowrecsys/offline/tensorflow/{DIN,DeepFM,DeepFM_v2,WideNDeep,EmbeddingMLP,NeuralCF,DIEN}.py`` between
the dataset definition and ``model = tf.keras.Model(...)`` -- can be ``exec``-uted in a container where TensorFlow cannot be
installed (tests/golden/make_tf_golden.py does that; with a real ``import tensorflow`` the very same harness runs the very
same lines on TensorFlow instead).

What this pins and what it does not.  Running the script text removes one class of oracle error entirely -- the builder's
reading of the SCRIPT (which columns, which order in every concat, which layer feeds which, which activation, which pair
list): the graph is wired by the reference's code, not by a restatement of it.  What remains restated is the arithmetic
INSIDE each ``tf.*`` object, from its published definition; each class below cites the TensorFlow source it follows
(paths relative to tensorflow/python/, TF 2.0-2.15, the versions the reference's ``tf.feature_column`` code runs on).
PARITY PIN STATUS of outputs produced through this module: "wiring pinned by the reference's source, op arithmetic
restated" -- stronger than oracle/ctr_oracle.py alone, weaker than a TensorFlow-produced vector.

Semantics worth spelling out (each is exercised by tests/test_reference_blocks.py):
  * ``DenseFeatures`` sorts its columns by ``column.name`` and concatenates along axis 1
    (feature_column/dense_features.py ``_normalize_feature_columns`` + ``_process_dense_tensor``).
  * ``DenseFeatures`` creates the variables of its columns ITSELF, per layer (feature_column_v2.py ``_StateManagerImpl.
    create_variable`` -> ``self._layer.add_weight``; checkpoint key ``<layer>/<column>/embedding_weights`` -- the
    ``layer_with_weights-0/movieId_embedding.Sembedding_weights`` keys of the reference's exported checkpoints).  An
    ``embedding_column`` object handed to TWO DenseFeatures layers therefore owns TWO tables: in DeepFM.py the FM part
    (``DenseFeatures([movie_emb_col])``, DeepFM.py:91) and the deep part (``DenseFeatures(deep_feature_columns)``,
    DeepFM.py:106) do NOT share their movieId / userId embeddings.
  * ``embedding_column``: combiner "mean" over the single id of a scalar feature = the row itself; id -1 (OOV) and the
    empty string give the zero vector (``safe_embedding_lookup_sparse``).
  * ``indicator_column``: one-hot, OOV -> all-zero row.  ``crossed_column``: see ``CrossedColumn``.
  * ``PReLU``: alpha has the input's shape without the batch axis, zero-initialised (keras/layers/advanced_activations.py).
  * ``Embedding(mask_zero=True)``: casts non-integer input to int32 and gathers; the mask is metadata only.
  * layer names follow Keras' ``unique_object_name(to_snake_case(class))``: dense, dense_1, p_re_lu, dense_features_2 ...
    in CREATION order -- what the harness uses to address weights identically on both backends.
"""
from __future__ import annotations

import re
import types
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import farmhash64 as FH

F32 = np.float32


# ----------------------------------------------------------------------------------------------------------------------
# symbolic tensors: a DAG of closures; every node also carries a 2-row "probe" value computed at build time, which is
# where static shapes come from (Dense needs its input width when it creates its kernel)
# ----------------------------------------------------------------------------------------------------------------------
class Sym:
    def __init__(self, fn: Callable, parents: Sequence["Sym"] = (), probe=None, name: str = ""):
        self.fn, self.parents, self.name = fn, list(parents), name
        self.probe = probe if probe is not None else fn(*[p.probe for p in self.parents])

    @property
    def shape(self):
        return (None,) + tuple(self.probe.shape[1:])


def _eval(node: Sym, memo: Dict[int, np.ndarray]):
    stack = [node]
    while stack:
        n = stack[-1]
        if id(n) in memo:
            stack.pop()
            continue
        todo = [p for p in n.parents if id(p) not in memo]
        if todo:
            stack.extend(todo)
            continue
        memo[id(n)] = n.fn(*[memo[id(p)] for p in n.parents])
        stack.pop()
    return memo[id(node)]


def _op(fn, *args):
    """fn over Sym and/or concrete arguments: symbolic if any argument is."""
    syms = [a for a in args if isinstance(a, Sym)]
    if not syms:
        return fn(*args)
    idx = [i for i, a in enumerate(args) if isinstance(a, Sym)]

    def run(*vals):
        full = list(args)
        for i, v in zip(idx, vals):
            full[i] = v
        return fn(*full)
    return Sym(run, syms)


class Var:
    def __init__(self, name: str, value: np.ndarray):
        self.name, self.value = name, value

    def numpy(self):
        return self.value


_COUNTS: Dict[str, int] = {}


def clear_session():
    _COUNTS.clear()


def _snake(name: str) -> str:                       # keras/utils/generic_utils.py to_snake_case
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    s = re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()
    return "private" + s if s[0] == "_" else s


def _unique(base: str) -> str:                      # keras/backend.py unique_object_name: dense, dense_1, ...
    n = _COUNTS.get(base, 0)
    _COUNTS[base] = n + 1
    return base if n == 0 else "%s_%d" % (base, n)


_ALL_LAYERS: List["Layer"] = []


def _glorot(shape, rng):                            # keras default kernel initialiser (ops/init_ops_v2.py GlorotUniform)
    lim = np.sqrt(6.0 / (shape[0] + shape[-1]))
    return rng.uniform(-lim, lim, size=shape).astype(F32)


_RNG = np.random.default_rng(12345)                 # initial values only matter until set_weights() replaces them


def _activation(a):
    if a is None or a == "linear":
        return lambda x: x
    if a == "relu":
        return lambda x: np.maximum(x, F32(0))
    if a == "sigmoid":
        def sig(x):
            with np.errstate(over="ignore"):           # exp(-x) -> inf for very negative x: 1 / inf = 0, as TensorFlow's logistic
                return (F32(1) / (F32(1) + np.exp(-x))).astype(F32)
        return sig
    if a == "tanh":
        return lambda x: np.tanh(x).astype(F32)
    raise NotImplementedError("activation %r" % (a,))


class _T(np.ndarray):
    """What a custom layer's ``call`` sees: an array that answers ``x == None`` with False, as a TensorFlow tensor does
    (DIEN.py:217 ``if Z_t_inputs==None:``), and otherwise is the ndarray."""

    def __eq__(self, other):
        return False if other is None else np.ndarray.__eq__(self, other)

    def __ne__(self, other):
        return True if other is None else np.ndarray.__ne__(self, other)

    __hash__ = None


def _wrap(v):
    return np.asarray(v).view(_T) if isinstance(v, np.ndarray) and v.dtype != object else v


def _unwrap(v):
    return np.asarray(v) if isinstance(v, _T) else v


class Layer:
    """keras/engine/base_layer.py: __call__ builds once from the input shape, then calls ``call``.  A layer that holds other
    layers as attributes (a subclass written in the reference script: DIEN.py's attention / GRU_gate_parameter / AUGRU) reports
    their variables behind its own, in attribute-creation order, names prefixed with its own -- Keras' layer tracking."""

    def __init__(self, name: Optional[str] = None, **kwargs):
        self.name = name or _unique(_snake(type(self).__name__))
        self._w: List[Var] = []
        self.built = False
        _ALL_LAYERS.append(self)

    def _ensure_base(self):                         # a subclass that forgot nothing: ReduceLayer calls super().__init__()
        if not hasattr(self, "_w"):
            Layer.__init__(self)

    def add_weight(self, name, value):
        v = Var("%s/%s:0" % (self.name, name), value)
        self._w.append(v)
        return v

    def _sublayers(self):
        return [v for v in vars(self).values() if isinstance(v, Layer)]

    @property
    def weights(self):
        out = list(self._w)
        for sub in self._sublayers():
            for v in sub.weights:
                out.append(types.SimpleNamespace(name="%s/%s" % (self.name, v.name), var=getattr(v, "var", v)))
        return out

    def add_loss(self, *a, **kw):                   # training-only bookkeeping (DIEN.py:288-290)
        pass

    def add_metric(self, *a, **kw):
        pass

    def build(self, input_shape):
        pass

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    def get_weights(self):
        return [getattr(v, "var", v).value for v in self.weights]

    def set_weights(self, values):
        ws = self.weights
        if len(values) != len(ws):
            raise ValueError("%s: %d values for %d weights" % (self.name, len(values), len(ws)))
        for v, a in zip(ws, values):
            a = np.asarray(a, dtype=F32)
            var = getattr(v, "var", v)
            if a.shape != var.value.shape:
                raise ValueError("%s: shape %s for weight %s of shape %s" % (self.name, a.shape, v.name, var.value.shape))
            var.value = a

    def __call__(self, inputs, *args, **kwargs):
        self._ensure_base()
        custom = type(self).__module__ != __name__                # a Layer subclass defined by the executed script
        if custom:
            base_call = self.call

            def call(x, *a, **kw):
                x = [_wrap(v) for v in x] if isinstance(x, (list, tuple)) else _wrap(x)
                return _unwrap(base_call(x, *[_wrap(v) for v in a], **{k: _wrap(v) for k, v in kw.items()}))
        else:
            call = self.call
        syms = inputs if isinstance(inputs, (list, tuple)) else [inputs]
        if isinstance(inputs, dict):
            syms = list(inputs.values())
        if not any(isinstance(v, Sym) for v in syms):
            # eager call on concrete arrays (tf.keras.Sequential builds its layers at the first batch it sees)
            if not self.built:
                shapes = [(None,) + tuple(np.asarray(v).shape[1:]) for v in syms]
                self.build(shapes[0] if not isinstance(inputs, (list, tuple, dict)) else shapes)
                self.built = True
            return call(inputs, *args, **kwargs)
        if not self.built:
            shapes = [s.shape for s in syms if isinstance(s, Sym)]
            self.build(shapes[0] if not isinstance(inputs, (list, tuple, dict)) else shapes)
            self.built = True
        if isinstance(inputs, dict):
            keys = list(inputs.keys())
            return Sym(lambda *vals: call(dict(zip(keys, vals))), [inputs[k] for k in keys], name=self.name)
        if isinstance(inputs, (list, tuple)):
            return Sym(lambda *vals: call(list(vals), *args, **kwargs), list(inputs), name=self.name)
        return Sym(lambda v: call(v, *args, **kwargs), [inputs], name=self.name)


# ----------------------------------------------------------------------------------------------------------------------
# tf.keras.layers
# ----------------------------------------------------------------------------------------------------------------------
def Input(name=None, shape=(), dtype="float32", **kwargs):
    """keras/engine/input_layer.py: a placeholder [batch] + shape.  Probe: two rows of zeros / empty strings."""
    if dtype == "string":
        probe = np.array([""] * 2, dtype=object).reshape((2,) + tuple(shape))
    else:
        probe = np.zeros((2,) + tuple(shape), dtype=np.dtype(dtype))
    s = Sym(None, [], probe=probe, name=name)
    s.is_input, s.dtype = True, dtype
    return s


class Dense(Layer):
    """keras/layers/core.py Dense: ``activation(dot(input, kernel) + bias)``, kernel [in, units], contraction over the LAST
    axis of inputs of any rank (``tensordot`` for rank > 2)."""

    def __init__(self, units, activation=None, use_bias=True, **kw):
        super().__init__(**kw)
        self.units, self.act, self.use_bias = int(units), _activation(activation), use_bias

    def build(self, input_shape):
        self.add_weight("kernel", _glorot((int(input_shape[-1]), self.units), _RNG))
        if self.use_bias:
            self.add_weight("bias", np.zeros((self.units,), F32))

    def call(self, x, **kw):
        y = np.matmul(x.astype(F32), self._w[0].value)
        if self.use_bias:
            y = y + self._w[1].value
        return self.act(y.astype(F32))


class Embedding(Layer):
    """keras/layers/embeddings.py: ``if dtype != int32/int64: inputs = cast(inputs, 'int32')``; ``embedding_lookup``;
    ``mask_zero`` only computes a mask (``compute_mask``), the lookup itself is unchanged.  Out-of-range ids fail (the CPU
    gather kernel raises InvalidArgument)."""

    def __init__(self, input_dim, output_dim, mask_zero=False, **kw):
        super().__init__(**kw)
        self.input_dim, self.output_dim, self.mask_zero = int(input_dim), int(output_dim), mask_zero

    def build(self, input_shape):
        self.add_weight("embeddings", _RNG.uniform(-0.05, 0.05, size=(self.input_dim, self.output_dim)).astype(F32))

    def call(self, x, **kw):
        ids = x if np.issubdtype(np.asarray(x).dtype, np.integer) else np.asarray(x).astype(np.int32)   # C-style truncation
        if ids.size and (ids.min() < 0 or ids.max() >= self.input_dim):
            raise ValueError("InvalidArgumentError: indices out of range [0, %d)" % self.input_dim)
        return self._w[0].value[ids]

    def __call__(self, inputs, *args, **kwargs):
        out = super().__call__(inputs, *args, **kwargs)
        if self.mask_zero and isinstance(out, Sym):
            out.mask = _op(lambda v: np.asarray(v) != 0, inputs)      # compute_mask: not_equal(inputs, 0)
        return out


class GRU(Layer):
    """keras/layers/recurrent_v2.py GRU(units, return_sequences=True), TF2 defaults: activation tanh, recurrent_activation
    sigmoid, reset_after=True, zero initial state.  Variables (gru/gru_cell/...): kernel [in, 3u], recurrent_kernel [u, 3u],
    bias [2, 3u] (input row, recurrent row), gate order z | r | h (recurrent.py GRUCell.call, reset_after branch):
        z = sig(x Wz + bz + h Uz + cz)   r = sig(x Wr + br + h Ur + cr)   hh = tanh(x Wh + bh + r * (h Uh + ch))
        h' = z * h + (1 - z) * hh
    An incoming mask (Embedding(mask_zero=True)) is consumed (keras/backend.py rnn, mask branch): at a masked step the state
    is kept and the OUTPUT repeats the previous output -- zeros before the first unmasked step."""

    def __init__(self, units, return_sequences=False, **kw):
        super().__init__(**kw)
        assert return_sequences, "only the form DIEN.py:169 uses"
        self.units = int(units)

    def build(self, input_shape):
        u, d = self.units, int(input_shape[-1])
        self.add_weight("gru_cell/kernel", _glorot((d, 3 * u), _RNG))
        self.add_weight("gru_cell/recurrent_kernel", _glorot((u, 3 * u), _RNG))
        self.add_weight("gru_cell/bias", np.zeros((2, 3 * u), F32))

    def run(self, x, mask):
        K, U, b = (v.value for v in self._w)
        u = self.units
        B, T, _ = x.shape
        sig = _activation("sigmoid")
        h = np.zeros((B, u), F32)
        prev = np.zeros((B, u), F32)
        out = np.zeros((B, T, u), F32)
        for t in range(T):
            mx = (np.matmul(x[:, t, :].astype(F32), K) + b[0]).astype(F32)
            mh = (np.matmul(h, U) + b[1]).astype(F32)
            z = sig(mx[:, :u] + mh[:, :u])
            r = sig(mx[:, u:2 * u] + mh[:, u:2 * u])
            hh = np.tanh(mx[:, 2 * u:] + r * mh[:, 2 * u:]).astype(F32)
            hn = (z * h + (F32(1) - z) * hh).astype(F32)
            m = np.ones((B, 1), bool) if mask is None else mask[:, t].reshape(B, 1)
            h = np.where(m, hn, h)
            prev = np.where(m, hn, prev)
            out[:, t, :] = prev
        return out

    def __call__(self, inputs, *args, **kwargs):
        self._ensure_base()
        if not self.built:
            self.build(inputs.shape)
            self.built = True
        mask = getattr(inputs, "mask", None)
        if mask is None:
            return Sym(lambda v: self.run(v, None), [inputs], name=self.name)
        return Sym(lambda v, m: self.run(v, m), [inputs, mask], name=self.name)


class PReLU(Layer):
    """keras/layers/advanced_activations.py PReLU: ``alpha`` of shape input_shape[1:] (no shared axes given), zeros;
    ``relu(x) + -alpha * relu(-x)``."""

    def build(self, input_shape):
        self.add_weight("alpha", np.zeros(tuple(int(d) for d in input_shape[1:]), F32))

    def call(self, x, **kw):
        a = self._w[0].value
        return (np.maximum(x, F32(0)) + (-a) * np.maximum(-x, F32(0))).astype(F32)


class Dot(Layer):
    """keras/layers/merge.py Dot(axes=1) on two [B, D] inputs: ``batch_dot`` -> sum over axis 1 -> [B, 1]."""

    def __init__(self, axes, normalize=False, **kw):
        super().__init__(**kw)
        assert axes in (1, (1, 1)) and not normalize
        self.axes = axes

    def call(self, xs, **kw):
        a, b = xs
        assert a.ndim == 2 and b.ndim == 2
        return np.sum(a * b, axis=1, keepdims=True, dtype=F32)


class _Merge(Layer):
    pass


class Add(_Merge):
    def call(self, xs, **kw):
        out = xs[0]
        for x in xs[1:]:
            out = out + x
        return out


class Subtract(_Merge):
    def call(self, xs, **kw):
        assert len(xs) == 2
        return xs[0] - xs[1]


class Multiply(_Merge):
    def call(self, xs, **kw):
        out = xs[0]
        for x in xs[1:]:
            out = out * x
        return out


class Concatenate(_Merge):
    def __init__(self, axis=-1, **kw):
        super().__init__(**kw)
        self.axis = axis

    def call(self, xs, **kw):
        return np.concatenate(list(xs), axis=self.axis)


def concatenate(inputs, axis=-1, **kw):
    return Concatenate(axis=axis, **kw)(inputs)


def multiply(inputs, **kw):
    return Multiply(**kw)(inputs)


def subtract(inputs, **kw):
    return Subtract(**kw)(inputs)


def add(inputs, **kw):
    return Add(**kw)(inputs)


class Flatten(Layer):
    def call(self, x, **kw):
        return x.reshape(x.shape[0], -1)


class RepeatVector(Layer):
    """keras/layers/core.py: [B, D] -> [B, n, D] (``K.repeat``)."""

    def __init__(self, n, **kw):
        super().__init__(**kw)
        self.n = int(n)

    def call(self, x, **kw):
        assert x.ndim == 2
        return np.repeat(x[:, None, :], self.n, axis=1)


class Permute(Layer):
    """keras/layers/core.py: ``dims`` are 1-based positions of the non-batch axes."""

    def __init__(self, dims, **kw):
        super().__init__(**kw)
        self.dims = tuple(dims)

    def call(self, x, **kw):
        return np.transpose(x, (0,) + self.dims)


class Reshape(Layer):
    def __init__(self, target_shape, **kw):
        super().__init__(**kw)
        self.target = tuple(target_shape)

    def call(self, x, **kw):
        return x.reshape((x.shape[0],) + self.target)


class Lambda(Layer):
    def __init__(self, function, **kw):
        super().__init__(**kw)
        self.function = function

    def call(self, x, **kw):
        return self.function(x)


# ----------------------------------------------------------------------------------------------------------------------
# tf.feature_column  (feature_column/feature_column_v2.py)
# ----------------------------------------------------------------------------------------------------------------------
class NumericColumn:
    """numeric_column(key, shape=(1,), default_value=None, dtype=float32): ``to_float`` of the input, reshaped [B, 1]."""

    def __init__(self, key, default_value=None):
        self.key, self.name, self.default_value = key, key, default_value
        self.width = 1

    def dense(self, feats, _vars):
        return np.asarray(feats[self.key]).astype(F32).reshape(-1, 1)


class IdentityCategoricalColumn:
    """categorical_column_with_identity(key, num_buckets, default_value=None): ids as int64; out-of-range fails
    (``assert_less_than_num_buckets`` / ``assert_greater_or_equal_0``)."""

    def __init__(self, key, num_buckets, default_value=None):
        self.key, self.name, self.num_buckets, self.default_value = key, key, int(num_buckets), default_value

    def ids(self, feats):
        v = np.asarray(feats[self.key]).astype(np.int64)
        bad = (v < 0) | (v >= self.num_buckets)
        if bad.any():
            if self.default_value is None:
                raise ValueError("InvalidArgumentError: %s id outside [0, %d)" % (self.key, self.num_buckets))
            v = np.where(bad, self.default_value, v)
        return v


class VocabularyListCategoricalColumn:
    """categorical_column_with_vocabulary_list(key, vocabulary_list, default_value=-1, num_oov_buckets=0): position in the
    list, -1 when absent; the empty string is dropped by ``_to_sparse_input_and_drop_ignore_values`` (no id at all),
    which downstream is the same as -1 (zero embedding / zero indicator row)."""

    def __init__(self, key, vocabulary_list, default_value=-1, num_oov_buckets=0):
        assert num_oov_buckets == 0
        self.key, self.name, self.vocab, self.default_value = key, key, list(vocabulary_list), default_value
        self.num_buckets = len(self.vocab)
        self._index = {s: i for i, s in enumerate(self.vocab)}

    def ids(self, feats):
        out = np.empty(len(feats[self.key]), np.int64)
        for i, s in enumerate(feats[self.key]):
            if isinstance(s, bytes):
                s = s.decode("utf-8")
            s = "" if s is None else str(s)
            out[i] = -1 if s == "" else self._index.get(s, self.default_value)
        return out


class CrossedColumn:
    """crossed_column(keys, hash_bucket_size, hash_key=None): ``sparse_cross_hashed(inputs, num_buckets, hash_key)``
    (core/kernels/sparse_cross_op.cc, HashCrosser): ``h = hash_key (default 0xDECAFCAFFE); for each column in the order
    GIVEN: h = FingerprintCat64(h, feature)``, where an int64 feature is its value and a string feature is
    ``Fingerprint64(string)``; result ``h mod num_buckets`` (uint64).  A categorical-column key contributes its id tensor.
    name = '_X_'.join(sorted(leaf key names)).  FingerprintCat64 / Fingerprint64 (oracle/farmhash64.py) reproduce the
    known answers of TensorFlow's own sparse_cross_op_test.py (tests/test_farmhash_pins.py)."""

    def __init__(self, keys, hash_bucket_size, hash_key=None):
        self.keys, self.num_buckets = list(keys), int(hash_bucket_size)
        self.hash_key = FH.DEFAULT_HASH_KEY if hash_key is None else int(hash_key)
        leaves = [k if isinstance(k, str) else k.key for k in self.keys]
        self.name = "_X_".join(sorted(leaves))

    def ids(self, feats):
        cols = []
        for k in self.keys:
            if isinstance(k, str):
                v = np.asarray(feats[k])
                cols.append(v.astype(np.int64) if v.dtype.kind in "iu" else v)
            else:
                cols.append(k.ids(feats))
        n = len(cols[0])
        out = np.empty(n, np.int64)
        for i in range(n):
            h = self.hash_key
            for c in cols:
                x = c[i]
                fp = FH.fingerprint64(x if isinstance(x, bytes) else str(x).encode()) if c.dtype == object else int(x) & FH.M64
                h = FH.fingerprint_cat64(h, fp)
            out[i] = h % self.num_buckets
        return out


class EmbeddingColumn:
    """embedding_column(categorical_column, dimension, combiner='mean'): variable ``embedding_weights`` [num_buckets, dim]
    (truncated normal, stddev 1/sqrt(dim)); ``safe_embedding_lookup_sparse``: ids < 0 pruned, an example left without
    ids gets the zero vector; one id per example -> mean == the row."""

    def __init__(self, categorical_column, dimension, combiner="mean"):
        assert combiner == "mean"
        self.cat, self.dim = categorical_column, int(dimension)
        self.name = categorical_column.name + "_embedding"

    def create(self, layer):
        std = 1.0 / np.sqrt(self.dim)
        init = np.clip(_RNG.normal(0, std, size=(self.cat.num_buckets, self.dim)), -2 * std, 2 * std).astype(F32)
        return layer.add_weight(self.name + "/embedding_weights", init)

    def dense(self, feats, var):
        ids = self.cat.ids(feats)
        out = np.zeros((len(ids), self.dim), F32)
        ok = ids >= 0
        out[ok] = var.value[ids[ok]]
        return out


class IndicatorColumn:
    """indicator_column(categorical_column): multi-hot float32 [B, num_buckets]; id -1 -> zero row."""

    def __init__(self, categorical_column):
        self.cat = categorical_column
        self.name = categorical_column.name + "_indicator"

    def dense(self, feats, _var):
        ids = self.cat.ids(feats)
        out = np.zeros((len(ids), self.cat.num_buckets), F32)
        ok = ids >= 0
        out[np.nonzero(ok)[0], ids[ok]] = 1
        return out


class DenseFeatures(Layer):
    """feature_column/dense_features.py: columns sorted by name; each column's variables created by THIS layer; outputs
    concatenated along axis 1."""

    def __init__(self, feature_columns, **kw):
        super().__init__(**kw)
        cols = list(feature_columns) if isinstance(feature_columns, (list, tuple)) else [feature_columns]
        self.columns = sorted(cols, key=lambda c: c.name)
        names = [c.name for c in self.columns]
        if len(set(names)) != len(names):
            raise ValueError("Duplicate feature column name found for columns")
        self._vars = {}

    def build(self, _shapes):
        for c in self.columns:
            self._vars[c.name] = c.create(self) if isinstance(c, EmbeddingColumn) else None

    def call(self, feats, **kw):
        return np.concatenate([c.dense(feats, self._vars[c.name]) for c in self.columns], axis=1)


# ----------------------------------------------------------------------------------------------------------------------
# tf.keras.Model, tf.* ops, module tree
# ----------------------------------------------------------------------------------------------------------------------
class Model:
    def __init__(self, inputs, outputs, **kw):
        self.inputs, self.output = inputs, outputs
        seen, order = set(), []

        def walk(n):
            if id(n) in seen:
                return
            seen.add(id(n))
            for p in n.parents:
                walk(p)
            order.append(n)
        import sys
        sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
        for o in (outputs if isinstance(outputs, (list, tuple)) else [outputs]):
            walk(o)
        names = {n.name for n in order}
        self.layers = [l for l in _ALL_LAYERS if l.name in names]

    def compile(self, **kw):
        pass

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: %s" % name)

    def predict(self, x, batch_size=None, **kw):
        """x: dict {input name: array[N]}; unknown keys ignored (functional models ignore unused dict entries)."""
        memo = {}
        for key, node in self.inputs.items():
            v = np.asarray(x[key])
            if node.dtype == "string":
                v = v.astype(object)
            else:
                v = v.astype(np.dtype(node.dtype))
            memo[id(node)] = v
        if isinstance(self.output, (list, tuple)):
            return [np.asarray(_eval(o, memo), dtype=F32) for o in self.output]
        return np.asarray(_eval(self.output, memo), dtype=F32)


class Sequential:
    """keras/engine/sequential.py: a stack of layers; without an Input layer the variables are created at the first batch."""

    def __init__(self, layers=None, **kw):
        self.layers = list(layers or [])

    def compile(self, **kw):
        pass

    def predict(self, x, batch_size=None, **kw):
        v = x
        for layer in self.layers:
            v = layer(v)
        return np.asarray(v, dtype=F32)


def _squeeze(x, axis=None):
    return _op(lambda v: np.squeeze(v, axis=axis), x)


def _reduce_sum(x, axis=None, keepdims=False):
    return _op(lambda v: np.sum(v, axis=axis, keepdims=keepdims, dtype=v.dtype), x)


def _reduce_mean(x, axis=None, keepdims=False):
    return _op(lambda v: np.mean(v, axis=axis, keepdims=keepdims, dtype=v.dtype), x)


class _AUC:
    def __init__(self, curve="ROC", **kw):
        self.curve = curve

    def update_state(self, *a, **kw):               # training-only bookkeeping (DIEN.py:289-290)
        pass

    def result(self):
        return F32(0)


def _reshape(x, shape):
    return _op(lambda v: np.reshape(v, tuple(shape)), x)


def _binary_crossentropy(y_true, y_pred):
    """keras/losses.py binary_crossentropy (from_logits=False): mean over the last axis of the clipped log loss."""
    def f(t, p):
        eps = F32(1e-7)
        p = np.clip(np.asarray(p, F32), eps, F32(1) - eps)
        t = np.asarray(t, F32)
        return np.mean(-(t * np.log(p) + (F32(1) - t) * np.log(F32(1) - p)), axis=-1).astype(F32)
    return _op(f, y_true, y_pred)


# tf.keras.initializers.GlorotUniform()(shape): random per call in TensorFlow.  DIEN.py:238-239 draws the AUGRU's initial
# state that way INSIDE call() -- the reference's own forward pass is not reproducible -- so the harness installs the value it
# wants the run to use (make_tf_golden.py does the same to tensorflow when that backend runs).
INITIALIZER_OVERRIDE: Optional[Callable] = None


class _GlorotUniform:
    def __init__(self, seed=None):
        pass

    def __call__(self, shape, dtype=None):
        if INITIALIZER_OVERRIDE is not None:
            return np.asarray(INITIALIZER_OVERRIDE(tuple(shape)), F32)
        return _glorot(tuple(shape), _RNG)


def build_module():
    """A fresh ``tf`` namespace (and a fresh Keras name scope: ``clear_session``)."""
    clear_session()
    del _ALL_LAYERS[:]
    tf = types.SimpleNamespace()
    tf.__version__ = "0.0-keras-shim"
    layers = types.SimpleNamespace(
        Input=Input, Dense=Dense, Embedding=Embedding, PReLU=PReLU, Dot=Dot, Add=Add, Subtract=Subtract, Multiply=Multiply,
        Concatenate=Concatenate, concatenate=concatenate, multiply=multiply, subtract=subtract, add=add, Flatten=Flatten,
        RepeatVector=RepeatVector, Permute=Permute, Reshape=Reshape, Lambda=Lambda, Layer=Layer, DenseFeatures=DenseFeatures, GRU=GRU)
    tf.keras = types.SimpleNamespace(
        layers=layers, Model=Model, Sequential=Sequential,
        backend=types.SimpleNamespace(sum=_reduce_sum, clear_session=clear_session),
        metrics=types.SimpleNamespace(AUC=_AUC), losses=types.SimpleNamespace(binary_crossentropy=_binary_crossentropy),
        initializers=types.SimpleNamespace(GlorotUniform=_GlorotUniform))
    tf.feature_column = types.SimpleNamespace(
        numeric_column=NumericColumn, categorical_column_with_identity=IdentityCategoricalColumn,
        categorical_column_with_vocabulary_list=VocabularyListCategoricalColumn, embedding_column=EmbeddingColumn,
        indicator_column=IndicatorColumn, crossed_column=CrossedColumn)
    tf.squeeze, tf.reduce_sum, tf.reduce_mean, tf.reshape = _squeeze, _reduce_sum, _reduce_mean, _reshape
    return tf

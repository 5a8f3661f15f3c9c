"""A THIRD implementation of the op arithmetic inside the reference's Keras layers, independent of the two the builder wrote
(oracle/ctr_oracle.py and oracle/keras_shim.py): ``torch.nn.functional`` on CPU, float64.  TensorFlow cannot run in this image, so
no TensorFlow-produced vector pins ``Dense`` on rank-3 input / ``Embedding`` / ``PReLU`` with a per-(time step, unit) alpha /
``Dot(axes=1)`` for DIN, DeepFM and DeepFM_v2 (VERDICT r03 "what's weak" 1); what CAN be done is to hold both restatements to a
library whose authors never saw them -- as tests/test_dien_cpu.py does for the GRU.  Two layers of checks:

  * layer by layer: the shim's ``Dense`` (rank 2 and rank 3), ``Embedding`` (float ids, as DIN.py:95-103 feeds them), ``PReLU``
    (alpha of the input's shape without the batch axis, DIN.py:150), ``Dot(axes=1)`` (DeepFM.py:100-103), ``RepeatVector`` /
    ``Permute`` (DIN.py:153-156) called eagerly on random arrays against F.linear / F.embedding / F.prelu / torch.bmm;
  * graph by graph: DIN.py:132-167, DeepFM.py:91-113 (the [B, 31 040] one-hot block materialised with F.one_hot, as the script
    does) and DeepFM_v2.py:98-155 written out with torch ops only, fed the oracle's weight dict, against the oracle's forward.

The parity status stays "partial" until a ``refblock_tf_*.npz`` exists (tests/test_reference_blocks.py); this narrows what a
TensorFlow run could still find to TensorFlow-specific behaviour, not arithmetic slips shared by two same-author restatements."""
import numpy as np
import pytest

from oracle import ctr_oracle as O
from oracle import keras_shim as KS
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY

torch = pytest.importorskip("torch")
F = torch.nn.functional


def _t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------------
# layer by layer: the shim's objects called eagerly
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(7, 40), (5, 6, 40), (3, 4, 5, 40)])
def test_shim_dense_is_a_contraction_over_the_last_axis(shape):
    rng = np.random.default_rng(1)
    KS.clear_session()
    layer = KS.Dense(32, activation="sigmoid")
    x = _rand(rng, *shape)
    layer(x)                                                       # builds
    k, b = _rand(rng, 40, 32), _rand(rng, 32)
    layer.set_weights([k, b])
    want = torch.sigmoid(F.linear(_t(x), _t(k).T, _t(b))).numpy()  # F.linear contracts the last axis with weight [out, in]
    np.testing.assert_allclose(layer(x), want, atol=2e-6)


def test_shim_embedding_casts_float_ids_and_ignores_its_mask():
    """DIN.py:95-103 feeds the ids as float32 numeric columns; Embedding(mask_zero=True) casts and gathers, row 0 is a row."""
    rng = np.random.default_rng(2)
    KS.clear_session()
    layer = KS.Embedding(50, 10, mask_zero=True)
    ids = rng.integers(0, 50, size=(9, 5)).astype(np.float32)
    ids[0, :] = 0
    layer(ids)
    table = _rand(rng, 50, 10)
    layer.set_weights([table])
    want = F.embedding(torch.from_numpy(ids).long(), _t(table)).numpy()
    np.testing.assert_array_equal(layer(ids), want.astype(np.float32))


def test_shim_prelu_alpha_has_the_inputs_shape_without_the_batch_axis():
    """DIN.py:150: PReLU() on [B, T, 32] owns alpha [T, 32] -- per time step AND unit.  torch's F.prelu is per channel (dim 1):
    flatten (T, 32) into the channel axis."""
    rng = np.random.default_rng(3)
    KS.clear_session()
    layer = KS.PReLU()
    x = _rand(rng, 11, 5, 32)
    layer(x)
    assert layer.get_weights()[0].shape == (5, 32) and not layer.get_weights()[0].any()      # zero-initialised
    alpha = _rand(rng, 5, 32)
    layer.set_weights([alpha])
    want = F.prelu(_t(x).reshape(11, 160), _t(alpha).reshape(160)).reshape(11, 5, 32).numpy()
    np.testing.assert_allclose(layer(x), want, atol=1e-6)
    np.testing.assert_allclose(O.prelu(x.astype(np.float64), alpha), want, atol=1e-12)


def test_shim_dot_axes1_is_a_batched_inner_product():
    rng = np.random.default_rng(4)
    KS.clear_session()
    a, b = _rand(rng, 13, 10), _rand(rng, 13, 10)
    got = KS.Dot(axes=1)([a, b])
    want = torch.bmm(_t(a).unsqueeze(1), _t(b).unsqueeze(2)).reshape(13, 1).numpy()
    assert got.shape == (13, 1)
    np.testing.assert_allclose(got, want, atol=1e-5)


def test_shim_repeat_permute_broadcast_the_attention_weight():
    """DIN.py:152-156: Flatten -> RepeatVector(D) -> Permute((2, 1)) turns [B, T, 1] into [B, T, D] (every lane the weight)."""
    rng = np.random.default_rng(5)
    KS.clear_session()
    w = _rand(rng, 6, 5, 1)
    got = KS.Permute((2, 1))(KS.RepeatVector(10)(KS.Flatten()(w)))
    want = _t(w).reshape(6, 5).unsqueeze(1).repeat(1, 10, 1).permute(0, 2, 1).numpy()
    np.testing.assert_array_equal(got, want.astype(np.float32))


# ---------------------------------------------------------------------------------------------------------------------------
# graph by graph: torch ops only, the oracle's weight dict
# ---------------------------------------------------------------------------------------------------------------------------
def _dense_features(blocks):
    """DenseFeatures: columns sorted by name, concatenated along axis 1 (pinned by the reference's exported graphs)."""
    return torch.cat([blocks[k] for k in sorted(blocks)], dim=1)


def _emb_col(w, key, ids):
    """embedding_column over one id per sample: the row; id -1 (OOV / empty) -> the zero vector."""
    table = _t(w[key])
    rows = F.embedding(torch.from_numpy(np.maximum(ids, 0)), table)
    return rows * torch.from_numpy((ids >= 0).astype(np.float64)).unsqueeze(1)


def _num(f, k):
    return _t(O.numeric(f, k, np.float64)).reshape(-1, 1)


def torch_din(f, w, T, user_buckets):
    hist = torch.from_numpy(np.asarray(f["userRatedMovies"]).astype(np.int64))
    cand = torch.from_numpy(O.int_feature(f, "movieId"))
    table = _t(w["emb/movie"])
    h = F.embedding(hist, table)                                                     # DIN.py:134
    c = F.embedding(cand, table)                                                     # DIN.py:136-137
    cr = c.unsqueeze(1).repeat(1, T, 1)                                              # RepeatVector, DIN.py:139
    a = torch.cat([h - cr, h, cr, h * cr], dim=-1)                                   # DIN.py:141-147
    u = F.linear(a, _t(w["att0/kernel"]).T, _t(w["att0/bias"]))                      # Dense(32) on rank 3, DIN.py:149
    B = u.shape[0]
    u = F.prelu(u.reshape(B, T * 32), _t(w["att_prelu/alpha"]).reshape(T * 32)).reshape(B, T, 32)   # DIN.py:150
    wgt = torch.sigmoid(F.linear(u, _t(w["att1/kernel"]).T, _t(w["att1/bias"])))     # [B, T, 1], DIN.py:151
    rep = wgt.reshape(B, T).unsqueeze(1).repeat(1, h.shape[2], 1).permute(0, 2, 1)   # DIN.py:152-156
    pooled = (rep * h).sum(dim=1)                                                    # Multiply + Lambda(K.sum), DIN.py:157-158
    prof = {k: _num(f, k) for k in ("userRatingCount", "userAvgRating", "userRatingStddev")}
    prof["userId_embedding"] = _emb_col(w, "emb/userId", O.identity_ids(O.int_feature(f, "userId"), user_buckets, "userId"))
    prof["userGenre1_embedding"] = _emb_col(w, "emb/userGenre1", O.vocab_ids(f["userGenre1"]))
    ctx = {k: _num(f, k) for k in ("releaseYear", "movieRatingCount", "movieAvgRating", "movieRatingStddev")}
    ctx["movieGenre1_embedding"] = _emb_col(w, "emb/movieGenre1", O.vocab_ids(f["movieGenre1"]))
    x = torch.cat([_dense_features(prof), pooled, c, _dense_features(ctx)], dim=1)    # DIN.py:161-162
    x = F.prelu(F.linear(x, _t(w["fc0/kernel"]).T, _t(w["fc0/bias"])), _t(w["fc0_prelu/alpha"]))   # DIN.py:163-164
    x = F.prelu(F.linear(x, _t(w["fc1/kernel"]).T, _t(w["fc1/bias"])), _t(w["fc1_prelu/alpha"]))   # DIN.py:165-166
    out = torch.sigmoid(F.linear(x, _t(w["head/kernel"]).T, _t(w["head/bias"])))      # DIN.py:167
    return out.numpy(), wgt.reshape(B, T).numpy(), pooled.numpy()


@pytest.mark.parametrize("T,D", [(5, 10), (50, 32), (17, 24)])
def test_din_graph_in_torch_equals_the_oracle(T, D):
    B, V, U = 257, 900, 300
    f = SY.synth_din(B, T, V, U, seed=50 + T)
    f["userGenre1"][::7] = -1
    model = M.DIN(seed=9 + T, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    w = dict(model.weights)
    rng = np.random.default_rng(T)
    for k in ("att_prelu/alpha", "fc0_prelu/alpha", "fc1_prelu/alpha"):            # Keras initialises alpha to 0: make it matter
        w[k] = (rng.standard_normal(w[k].shape) * 0.5).astype(np.float32)
    ref, parts = O.din_forward(f, w, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U, return_parts=True)
    got, att, pooled = torch_din(f, w, T, U)
    np.testing.assert_allclose(att, parts["att"], atol=1e-12)
    np.testing.assert_allclose(pooled, parts["pooled"], atol=1e-12)
    np.testing.assert_allclose(got, ref.astype(np.float64), atol=1e-7)               # (the oracle returns float32)
    assert ref.std() > 0.01


def torch_deepfm(f, w, fields, pairs, deep_emb):
    ids = O._field_ids(f, fields)
    emb = {k: _emb_col(w, "emb/" + k, ids[k]) for k, _, _ in fields}
    demb = {k: _emb_col(w, "deep_emb/" + k, ids[k]) for k in deep_emb}                # DeepFM.py:106: the deep part's own tables
    # first order: indicator columns, name-sorted, as ONE dense [B, sum vocab] block           DeepFM.py:97
    onehot = torch.cat([F.one_hot(torch.from_numpy(np.maximum(ids[k], 0)), v).double() * torch.from_numpy((ids[k] >= 0).astype(np.float64)).unsqueeze(1)
                        for _, k, v in sorted((k + "_indicator", k, v) for k, _, v in fields)], dim=1)
    dots = [torch.bmm(emb[a].unsqueeze(1), emb[b].unsqueeze(2)).reshape(-1, 1) for a, b in pairs]       # Dot(axes=1), DeepFM.py:100-103
    blocks = {k: _num(f, k) for k in O.NUMERIC_KEYS}
    blocks.update({k + "_embedding": demb[k] for k in deep_emb})
    x = _dense_features(blocks)                                                       # DeepFM.py:106
    i = 0
    while "deep%d/kernel" % i in w:                                                   # DeepFM.py:107-108
        x = F.relu(F.linear(x, _t(w["deep%d/kernel" % i]).T, _t(w["deep%d/bias" % i])))
        i += 1
    concat = torch.cat([onehot] + dots + [x], dim=1)                                  # DeepFM.py:111-112
    return torch.sigmoid(F.linear(concat, _t(w["head/kernel"]).T, _t(w["head/bias"]))).numpy()          # DeepFM.py:113


def test_deepfm_graph_in_torch_equals_the_oracle(samples):
    from tests.golden.make_golden import make_model
    model = make_model("deepfm")
    ref = O.deepfm_forward(samples, model.weights, dtype=np.float64)
    got = torch_deepfm(samples, model.weights, O.DEEPFM_FIELDS, O.DEEPFM_PAIRS, O.DEEPFM_DEEP_EMB)
    np.testing.assert_allclose(got, ref.astype(np.float64), atol=1e-7)
    assert ref.std() > 0.01


def test_deepfm_v2_graph_in_torch_equals_the_oracle(samples):
    from tests.golden.make_golden import make_model
    model = make_model("deepfm_v2")
    w = model.weights
    fields, order = O.DEEPFM_FIELDS, O.DEEPFM_V2_ORDER
    ids = O._field_ids(samples, fields)
    emb = {k: _emb_col(w, "emb/" + k, ids[k]) for k, _, _ in fields}
    onehot = torch.cat([F.one_hot(torch.from_numpy(np.maximum(ids[k], 0)), v).double() * torch.from_numpy((ids[k] >= 0).astype(np.float64)).unsqueeze(1)
                        for _, k, v in sorted((k + "_indicator", k, v) for k, _, v in fields)], dim=1)
    num = _dense_features({k: _num(samples, k) for k in O.NUMERIC_KEYS})             # DeepFM_v2.py:100,118
    first = F.linear(onehot, _t(w["fo_cat/kernel"]).T, _t(w["fo_cat/bias"])) + F.linear(num, _t(w["fo_num/kernel"]).T, _t(w["fo_num/bias"]))   # :98-104
    proj = [F.linear(emb[k], _t(w["proj/%s/kernel" % k]).T, _t(w["proj/%s/bias" % k])) for k in order]
    proj.append(F.linear(num, _t(w["proj/num/kernel"]).T, _t(w["proj/num/bias"])))   # :106-120
    stack = torch.stack(proj, dim=1)                                                  # [B, 5, K], :121
    deep = stack.reshape(stack.shape[0], -1)                                          # Flatten, :124
    i = 0
    while "deep%d/kernel" % i in w:
        deep = F.relu(F.linear(deep, _t(w["deep%d/kernel" % i]).T, _t(w["deep%d/bias" % i])))
        i += 1
    s = stack.sum(dim=1)                                                              # ReduceLayer, :129-147
    fm = s * s - (stack * stack).sum(dim=1)                                           # no 1/2, not reduced over K, :148-152
    out = torch.sigmoid(F.linear(torch.cat([first, fm, deep], dim=1), _t(w["head/kernel"]).T, _t(w["head/bias"]))).numpy()   # :154-155
    ref, parts = O.deepfm_v2_forward(samples, w, dtype=np.float64, return_parts=True)
    np.testing.assert_allclose(fm.numpy(), parts["fm"], atol=1e-10)
    np.testing.assert_allclose(out, ref.astype(np.float64), atol=1e-7)
    assert ref.std() > 0.01


# ---------------------------------------------------------------------------------------------------------------------------
# [r5] Wide&Deep and DIEN end to end (VERDICT r04 "missing" 6): the two graphs the third implementation did not cover yet
# ---------------------------------------------------------------------------------------------------------------------------
def torch_wide_n_deep(f, w, movie_buckets, user_buckets, cross_buckets, rated_buckets):
    blocks = {k: _num(f, k) for k in O.NUMERIC_KEYS}                                  # WideNDeep.py:63-69
    for k in O.USER_GENRE_KEYS + O.MOVIE_GENRE_KEYS:                                  # :33-48, embedding columns over the genre vocabulary
        blocks[k + "_embedding"] = _emb_col(w, "emb/" + k, O.vocab_ids(f[k]))
    movie = O.identity_ids(O.int_feature(f, "movieId"), movie_buckets, "movieId")
    blocks["movieId_embedding"] = _emb_col(w, "emb/movieId", movie)                   # :51-54
    blocks["userId_embedding"] = _emb_col(w, "emb/userId", O.identity_ids(O.int_feature(f, "userId"), user_buckets, "userId"))   # :57-60
    deep = _dense_features(blocks)                                                    # :101
    deep = F.relu(F.linear(deep, _t(w["dense0/kernel"]).T, _t(w["dense0/bias"])))     # :102
    deep = F.relu(F.linear(deep, _t(w["dense1/kernel"]).T, _t(w["dense1/bias"])))     # :103
    rated = O.identity_ids(O.int_feature(f, "userRatedMovie1"), rated_buckets, "userRatedMovie1")
    bucket = torch.from_numpy(O.crossed_bucket_np([movie, rated], cross_buckets).astype(np.int64))   # :72-73 (the hash itself: tests/test_farmhash_pins.py)
    if "emb/cross" in w:                                                              # BASELINE config 5's form: the cross as an embedding column
        wide = F.embedding(bucket, _t(w["emb/cross"]))
    else:
        wide = F.one_hot(bucket, cross_buckets).double()                              # indicator_column: the [B, 10 000] block, materialised as the script does (:73,105)
    both = torch.cat([deep, wide], dim=1)                                             # :106
    return torch.sigmoid(F.linear(both, _t(w["head/kernel"]).T, _t(w["head/bias"]))).numpy()   # :107


@pytest.mark.parametrize("kind", ["indicator", "cross_rows"])
def test_wide_n_deep_graph_in_torch_equals_the_oracle(kind):
    B, V, U = 301, 1001, 3001
    f = SY.synth_embedding_mlp(B, V, U, seed=71, rated_vocab=V)
    kw = dict(cross_buckets=10000, cross_dim=0) if kind == "indicator" else dict(cross_buckets=5000, cross_dim=32)
    model = M.WideNDeep(seed=31, emb_dim=10 if kind == "indicator" else 32, movie_buckets=V, user_buckets=U, **kw)
    w = model.weights
    ref = O.wide_n_deep_forward(f, w, dtype=np.float64, movie_buckets=V, user_buckets=U, cross_buckets=model.cross_buckets, rated_buckets=V)
    got = torch_wide_n_deep(f, w, V, U, model.cross_buckets, V)
    np.testing.assert_allclose(got, ref.astype(np.float64), atol=1e-7)
    assert ref.std() > 0.01


def torch_dien(f, w, T, user_buckets):
    """DIEN.py:163-259 in torch ops: the GRU as torch.nn.GRUCell steps (gates permuted z|r|h -> r|z|n) under Keras' mask rule, the
    script's own attention / GRU_gate_parameter / AUGRU classes written out, the tail."""
    hist = torch.from_numpy(np.asarray(f["userRatedMovies"]).astype(np.int64))
    cand = torch.from_numpy(O.int_feature(f, "movieId"))
    table = _t(w["emb/movie"])
    D = table.shape[1]
    x = F.embedding(hist, table)                                                      # DIEN.py:167
    c = F.embedding(cand, table)                                                      # :168-171
    B = x.shape[0]
    cell = torch.nn.GRUCell(D, D).double()
    perm = np.r_[D:2 * D, 0:D, 2 * D:3 * D]
    with torch.no_grad():
        cell.weight_ih.copy_(torch.from_numpy(w["gru/kernel"].astype(np.float64).T[perm]))
        cell.weight_hh.copy_(torch.from_numpy(w["gru_rec/kernel"].astype(np.float64).T[perm]))
        cell.bias_ih.copy_(torch.from_numpy(w["gru/bias"][0].astype(np.float64)[perm]))
        cell.bias_hh.copy_(torch.from_numpy(w["gru/bias"][1].astype(np.float64)[perm]))
        h = torch.zeros(B, D, dtype=torch.float64)
        prev = torch.zeros(B, D, dtype=torch.float64)
        g = []
        for t in range(T):                                                            # :173, mask_zero's mask consumed by the GRU
            m = (hist[:, t] != 0).unsqueeze(1)
            hn = cell(x[:, t, :], h)
            h = torch.where(m, hn, h)                                                 # a masked step keeps the state ...
            prev = torch.where(m, hn, prev)                                           # ... and repeats the previous output (zeros before the first live slot)
            g.append(prev)
        g = torch.stack(g, dim=1)                                                     # [B, T, D]
        # class attention (:175-204): RepeatVector(c) * g -> Dense(32, sigmoid) -> Dense(1, sigmoid), squeezed, repeated over D, permuted
        prod = g * c.unsqueeze(1).repeat(1, T, 1)
        a = torch.sigmoid(F.linear(torch.sigmoid(F.linear(prod, _t(w["att0/kernel"]).T, _t(w["att0/bias"]))), _t(w["att1/kernel"]).T, _t(w["att1/bias"])))
        att = a.squeeze(2).unsqueeze(1).repeat(1, D, 1).permute(0, 2, 1)              # [B, T, D]

        def gate(name, gt, hid, act, z_t=None):                                       # class GRU_gate_parameter (:210-227)
            hin = hid if z_t is None else hid * z_t
            pre = F.linear(gt, _t(w["augru_%s_in/kernel" % name]).T, _t(w["augru_%s_in/bias" % name])) + F.linear(hin, _t(w["augru_%s_hid/kernel" % name]).T)
            return act(F.linear(pre, _t(w["augru_%s_out/kernel" % name]).T, _t(w["augru_%s_out/bias" % name])))
        hs = _t(w["augru/h0"]).reshape(1, D).repeat(B, 1)                             # (:239-240 draws it per call; pinned to a weight on every backend)
        for t in range(T):                                                            # class AUGRU (:241-248)
            r_t = gate("r", g[:, t, :], hs, torch.sigmoid)
            z_t = gate("z", g[:, t, :], hs, torch.sigmoid)
            h_next = gate("h", g[:, t, :], hs, torch.tanh, z_t)
            ra = att[:, t, :] * r_t
            hs = (1 - ra) * hs + ra * h_next
        prof = {k: _num(f, k) for k in ("userRatingCount", "userAvgRating", "userRatingStddev")}
        prof["userId_embedding"] = _emb_col(w, "emb/userId", O.identity_ids(O.int_feature(f, "userId"), user_buckets, "userId"))
        prof["userGenre1_embedding"] = _emb_col(w, "emb/userGenre1", O.vocab_ids(f["userGenre1"]))
        ctx = {k: _num(f, k) for k in ("releaseYear", "movieRatingCount", "movieAvgRating", "movieRatingStddev")}
        ctx["movieGenre1_embedding"] = _emb_col(w, "emb/movieGenre1", O.vocab_ids(f["movieGenre1"]))
        y = torch.cat([hs, c, _dense_features(prof), _dense_features(ctx)], dim=1)    # :252
        y = F.prelu(F.linear(y, _t(w["fc0/kernel"]).T, _t(w["fc0/bias"])), _t(w["fc0_prelu/alpha"]))   # :254-255
        y = F.prelu(F.linear(y, _t(w["fc1/kernel"]).T, _t(w["fc1/bias"])), _t(w["fc1_prelu/alpha"]))   # :256-257
        out = torch.sigmoid(F.linear(y, _t(w["head/kernel"]).T, _t(w["head/bias"])))  # :259
    return out.numpy(), a.squeeze(2).numpy(), hs.numpy(), g.numpy()


@pytest.mark.parametrize("T,D,holes", [(5, 10, "tail"), (20, 16, "anywhere"), (9, 10, "anywhere")])
def test_dien_graph_in_torch_equals_the_oracle(T, D, holes):
    B, V, U = 129, 400, 150
    f = SY.synth_din(B, T, V, U, seed=60 + T)
    h = f["userRatedMovies"]
    rng = np.random.default_rng(T)
    if holes == "anywhere":
        h[rng.random(h.shape) < 0.3] = 0                                              # masked slots in the middle of a history
    h[0] = 0                                                                          # a user without history
    h[2, :T // 2] = 0                                                                 # leading holes
    model = M.DIEN(seed=11 + T, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    w = dict(model.weights)
    for k in ("fc0_prelu/alpha", "fc1_prelu/alpha"):
        w[k] = (rng.standard_normal(w[k].shape) * 0.5).astype(np.float32)
    ref, parts = O.dien_forward(f, w, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U, return_parts=True)
    got, att, hs, g = torch_dien(f, w, T, U)
    np.testing.assert_allclose(g, parts["gru"], atol=1e-12)
    np.testing.assert_allclose(att, parts["att"], atol=1e-12)
    np.testing.assert_allclose(hs, parts["augru"], atol=1e-12)
    np.testing.assert_allclose(got, ref.astype(np.float64), atol=1e-7)
    assert ref.std() > 0.005

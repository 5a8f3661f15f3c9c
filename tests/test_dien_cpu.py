"""DIEN on CPU: the oracle's GRU against an independent implementation (torch.nn.GRU),
the Keras masking semantics restated, and the compiled plan (packed sequence weights + tail layout) against the oracle."""
import numpy as np
import pytest

from oracle import ctr_oracle as O
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
from tests.plan_interp import dien_seq, run_plan


def _feats(B, T, V, U, seed, holes="tail"):
    f = SY.synth_din(B, T, V, U, seed=seed)
    h = f["userRatedMovies"]
    rng = np.random.default_rng(seed + 1)
    if holes == "anywhere":
        h[rng.random(h.shape) < 0.3] = 0
    h[0] = 0                                              # a user without history: every slot masked
    if B > 2:
        h[1] = np.maximum(h[1], 1)                        # a full history
        h[2, :T // 2] = 0                                 # leading holes (Keras: zero outputs until the first live slot)
    return f


def test_oracle_gru_matches_torch_gru():
    """Keras GRU(reset_after=True) and torch.nn.GRU are the same recurrence with gates stored z|r|h vs r|z|n."""
    import torch
    D, T, B = 10, 7, 33
    model = M.DIEN(seed=3, emb_dim=D, hist_len=T, movie_buckets=50, user_buckets=20)
    w = model.weights
    f = SY.synth_din(B, T, 50, 20, seed=9)
    f["userRatedMovies"] = np.maximum(f["userRatedMovies"], 1)       # no masked slot: plain GRU
    _, parts = O.dien_forward(f, w, dtype=np.float64, hist_len=T, movie_buckets=50, user_buckets=20, return_parts=True)
    gru = torch.nn.GRU(D, D, batch_first=True).double()
    perm = np.r_[D:2 * D, 0:D, 2 * D:3 * D]                          # keras z|r|h -> torch r|z|n
    with torch.no_grad():
        gru.weight_ih_l0.copy_(torch.from_numpy(w["gru/kernel"].astype(np.float64).T[perm]))
        gru.weight_hh_l0.copy_(torch.from_numpy(w["gru_rec/kernel"].astype(np.float64).T[perm]))
        gru.bias_ih_l0.copy_(torch.from_numpy(w["gru/bias"][0].astype(np.float64)[perm]))
        gru.bias_hh_l0.copy_(torch.from_numpy(w["gru/bias"][1].astype(np.float64)[perm]))
        x = torch.from_numpy(w["emb/movie"].astype(np.float64)[f["userRatedMovies"]])
        out, _ = gru(x)
    np.testing.assert_allclose(parts["gru"], out.numpy(), atol=1e-12)


def test_oracle_mask_semantics():
    """mask_zero: a masked slot repeats the previous GRU output (zeros before the first live slot) and keeps the state."""
    D, T = 10, 5
    model = M.DIEN(seed=4, emb_dim=D, hist_len=T, movie_buckets=50, user_buckets=20)
    f = SY.synth_din(4, T, 50, 20, seed=2)
    f["userRatedMovies"] = np.array([[0, 0, 0, 0, 0], [7, 0, 0, 9, 0], [0, 0, 7, 9, 0], [7, 9, 0, 0, 0]])
    _, p = O.dien_forward(f, model.weights, dtype=np.float64, hist_len=T, movie_buckets=50, user_buckets=20, return_parts=True)
    g = p["gru"]
    assert (g[0] == 0).all()
    assert (g[1, 1] == g[1, 0]).all() and (g[1, 2] == g[1, 0]).all() and (g[1, 4] == g[1, 3]).all()
    assert (g[2, :2] == 0).all()
    np.testing.assert_allclose(g[2, 2], g[1, 0], atol=0)              # same first live id, same zero state
    np.testing.assert_allclose(g[2, 3], g[1, 3], atol=0)              # holes do not advance the state
    np.testing.assert_allclose(g[3, 1], g[1, 3], atol=0)


@pytest.mark.parametrize("D,T,holes", [(10, 5, "tail"), (16, 20, "anywhere"), (10, 1, "tail")])
def test_dien_plan_matches_oracle(D, T, holes):
    B, V, U = 192, 300, 500
    f = _feats(B, T, V, U, seed=31 + T, holes=holes)
    model = M.DIEN(seed=5, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    plan, slots = model.build_plan()
    ids, dense = model.pack(f)
    ref, parts = O.dien_forward(f, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U, return_parts=True)
    np.testing.assert_allclose(dien_seq(plan, slots, ids, np.float64)[:, :D], parts["augru"], atol=1e-12)
    got = run_plan(plan, slots, ids, dense, np.float64)
    np.testing.assert_allclose(got, ref[:, 0], atol=2e-7)
    assert ref.std() > 0.01


def test_dien_reference_schema(samples):
    """The reference's own CSV columns (userRatedMovie1..5 with empty fields, DIEN.py:111-121)."""
    model = M.DIEN(seed=6)
    plan, slots = model.build_plan()
    ids, dense = model.pack(samples)
    got = run_plan(plan, slots, ids, dense, np.float64)
    ref = O.dien_forward(samples, model.weights, dtype=np.float64)[:, 0]
    np.testing.assert_allclose(got, ref, atol=2e-7)
    f32 = O.dien_forward(samples, model.weights, dtype=np.float32)[:, 0]
    assert np.abs(f32 - ref).max() < 1e-5


def test_keras_stand_in_gru_matches_torch_gru_and_the_mask_rule():
    """oracle/keras_shim.py's GRU layer -- the one DIEN.py:169 runs on when the reference's lines are executed without TensorFlow --
    against torch.nn.GRU (an independent implementation of the same recurrence) on an unmasked batch, and against the oracle's
    restatement of Keras' mask-consuming step (state kept, previous output repeated) on a batch with holes."""
    import torch
    from oracle import keras_shim as KS
    D, T, B, V = 10, 6, 29, 40
    rng = np.random.default_rng(5)
    tf = KS.build_module()
    ids = tf.keras.layers.Input(name="h", shape=(T,), dtype="float32")
    emb_layer = tf.keras.layers.Embedding(input_dim=V, output_dim=D, mask_zero=True)
    emb = emb_layer(ids)
    gru_layer = tf.keras.layers.GRU(D, return_sequences=True)
    out = gru_layer(emb)
    model = tf.keras.Model(inputs={"h": ids}, outputs=out)
    table = rng.normal(0, 0.5, (V, D)).astype(np.float32)
    K, U, b = (rng.normal(0, 0.4, s).astype(np.float32) for s in ((D, 3 * D), (D, 3 * D), (2, 3 * D)))
    emb_layer.set_weights([table])
    gru_layer.set_weights([K, U, b])
    assert [v.name for v in gru_layer.weights] == ["gru/gru_cell/kernel:0", "gru/gru_cell/recurrent_kernel:0", "gru/gru_cell/bias:0"]
    # (1) no masked slot: torch.nn.GRU with the gates permuted z|r|h -> r|z|n
    h = rng.integers(1, V, (B, T))
    got = model.predict({"h": h.astype(np.float32)})
    gru = torch.nn.GRU(D, D, batch_first=True).double()
    perm = np.r_[D:2 * D, 0:D, 2 * D:3 * D]
    with torch.no_grad():
        gru.weight_ih_l0.copy_(torch.from_numpy(K.astype(np.float64).T[perm]))
        gru.weight_hh_l0.copy_(torch.from_numpy(U.astype(np.float64).T[perm]))
        gru.bias_ih_l0.copy_(torch.from_numpy(b[0].astype(np.float64)[perm]))
        gru.bias_hh_l0.copy_(torch.from_numpy(b[1].astype(np.float64)[perm]))
        want, _ = gru(torch.from_numpy(table.astype(np.float64)[h]))
    assert np.abs(got - want.numpy()).max() <= 2e-6
    # (2) holes: a masked slot repeats the previous output (zeros before the first live one) and does not advance the state
    h2 = h.copy()
    h2[rng.random(h2.shape) < 0.35] = 0
    h2[0] = 0
    g2 = model.predict({"h": h2.astype(np.float32)})
    assert (g2[0] == 0).all()
    for r in range(B):
        live = [t for t in range(T) if h2[r, t] != 0]
        packed = np.zeros((1, T), np.int64)
        packed[0, :len(live)] = h2[r, live]                       # the same ids without the holes: same states in order
        ref = model.predict({"h": np.maximum(packed, 1).astype(np.float32)})[0]
        k = -1
        for t in range(T):
            if h2[r, t] != 0:
                k += 1
            want_t = ref[k] if k >= 0 else np.zeros(D, np.float32)
            np.testing.assert_array_equal(g2[r, t], want_t)

"""Parity pinned to what the REFERENCE produced: the seven SavedModels it exported (webroot/modeldata/{neuralcf/001,002,
MLPRec/001..005}/saved_model.pb) executed op by op by oracle/tf_graph_exec.py -- the wiring TensorFlow generated for the
reference's feature columns and Keras layers, with the reference's trained variables where it ships them.

* CPU (everywhere): the oracle agrees with the committed executed-graph outputs (tests/golden/savedmodel_exec.npz) to 1e-6.
* CPU (build container, /root/reference mounted): the graphs are re-executed live -- all 22 440 test rows for NeuralCF --
  and must reproduce the fixture and the oracle.
* GPU (-m gpu): the HIP path agrees with the executed graphs to 1e-4 (north_star's tolerance).

What this pins: A3 identity + embedding_column, A4 vocabulary-list genre columns (index = list
position, OOV / empty -> no id), A5 indicator columns x Dense (one-hot product == kernel-row gather), A6 DenseFeatures'
name-sorted concat and numeric casts, A9 Dense / concatenate, A15 predict's output shape, A16 sigmoid, NeuralCF.py's Dot.
What stays unpinned: DIN (A10-A13), DeepFM(_v2)'s FM ops (A7, A8), the crossed-column hash (A14), DIEN."""
import os

import numpy as np
import pytest

from oracle import ctr_oracle as O
from tests.conftest import GOLDEN, REFERENCE, needs_reference
from tests.golden.make_golden import ncf_weights
from tests.golden.make_savedmodel_golden import K0, savedmodel_standin_variables

EXEC_TOL = 1e-6       # oracle vs executed graph (both fp32 numpy; only the summation order of the matmuls differs)
HIP_TOL = 1e-4        # north_star: "outputs within 1e-4 of TF2 CPU"

MLP_COLUMNS = {
    # DenseFeatures columns of each exported MLPRec graph (read off the graphs' placeholders / lookup tables)
    "001": dict(genre_keys=O.USER_GENRE_KEYS + O.MOVIE_GENRE_KEYS, int_vocab_keys=["movieId"] + ["userRatedMovie%d" % i for i in range(1, 6)]),
    "002": dict(numeric_keys=O.NUMERIC_KEYS + ["userReleaseYearStddev"]),
    "003": dict(numeric_keys=O.NUMERIC_KEYS + ["userReleaseYearStddev"], genre_keys=O.USER_GENRE_KEYS + O.MOVIE_GENRE_KEYS,
                int_vocab_keys=["movieId"] + ["userRatedMovie%d" % i for i in range(1, 6)]),
    "004": dict(numeric_keys=O.NUMERIC_KEYS),
}


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "savedmodel_exec.npz"))


@pytest.fixture(scope="module")
def ckpt():
    return np.load(os.path.join(GOLDEN, "neuralcf_ckpt.npz"))


def _ncf_w(ckpt, ver):
    w = {k[len(ver) + 1:]: ckpt[k] for k in ckpt.files if k.startswith(ver + "/")}
    table = np.zeros((30001, 10), np.float32)
    table[ckpt["users"]] = ckpt["user_rows_" + ver]
    w["emb/userId"] = table
    return w


def _tower_w(g):
    table = np.zeros((30001, 10), np.float32)
    table[g["w005/users"]] = g["w005/user_rows"]
    return {"emb/movieId": g["w005/emb/movieId"], "emb/userId": table,
            "item0/kernel": g["w005/item0/kernel"], "item0/bias": g["w005/item0/bias"],
            "user0/kernel": g["w005/user0/kernel"], "user0/bias": g["w005/user0/bias"]}


def _mlp_w(g, ver):
    if "w%s/l0/kernel" % ver in g.files:
        return {"dense0/kernel": g["w%s/l0/kernel" % ver], "dense0/bias": g["w%s/l0/bias" % ver],
                "dense1/kernel": g["w%s/l1/kernel" % ver], "dense1/bias": g["w%s/l1/bias" % ver],
                "head/kernel": g["w%s/l2/kernel" % ver], "head/bias": g["w%s/l2/bias" % ver]}
    # the reference ships no data shard for this model: the seeded stand-ins the generator executed the graph with
    shapes = {}
    for line in g["shapes_" + ver]:
        k, shp = str(line).split("=")
        shapes[k] = tuple(int(x) for x in shp.split("x"))
    v = savedmodel_standin_variables(shapes)
    v[K0] = (v[K0] / g["rowscale_" + ver][:, None]).astype(np.float32)
    sfx = "/.ATTRIBUTES/VARIABLE_VALUE"
    return {"dense0/kernel": v[K0], "dense0/bias": v["layer_with_weights-0/bias" + sfx],
            "dense1/kernel": v["layer_with_weights-1/kernel" + sfx], "dense1/bias": v["layer_with_weights-1/bias" + sfx],
            "head/kernel": v["layer_with_weights-2/kernel" + sfx], "head/bias": v["layer_with_weights-2/bias" + sfx]}


# ---------------------------------------------------------------------------------------------------------------------
# CPU: oracle == executed graph (fixture)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ver", ["001", "002"])
def test_oracle_neuralcf_matches_the_exported_graph(g, ckpt, ver):
    """A3 (categorical_column_with_identity + embedding_column), concatenate([item, user]), Dense x3, sigmoid."""
    feats = {"movieId": ckpt["movieId"], "userId": ckpt["userId"]}
    p = O.neural_cf_forward(feats, _ncf_w(ckpt, ver))[:, 0]
    assert np.abs(p - g["ncf_" + ver]).max() <= EXEC_TOL
    ops = {s.split(":")[0] for s in g["ops_ncf_" + ver]}
    assert {"SparseFillEmptyRows", "SparseSegmentMean", "ResourceGather", "Unique", "Select", "MatMul", "Sigmoid"} <= ops


def test_oracle_two_tower_matches_the_exported_graph(g, ckpt):
    """NeuralCF.py:57-66: per-tower Dense(relu), Dot(axes=1) -- exported without a head (MLPRec/005)."""
    feats = {"movieId": ckpt["movieId"], "userId": ckpt["userId"]}
    p = O.neural_cf2_forward(feats, _tower_w(g), with_head=False)[:, 0]
    assert np.abs(p - g["tower_005"]).max() <= EXEC_TOL


@pytest.mark.parametrize("ver", ["001", "002", "003", "004"])
def test_oracle_feature_columns_match_the_exported_mlp_graphs(g, samples, ver):
    """A4 / A5 / A6: vocabulary-list indicator columns, numeric columns and the DenseFeatures order, against the graphs
    TensorFlow generated for them (002 / 004 with the reference's trained variables, 001 / 003 with seeded stand-ins)."""
    p = O.feature_column_mlp_forward(samples, _mlp_w(g, ver), **MLP_COLUMNS[ver])[:, 0]
    ref = g["mlp_" + ver]
    assert ref.shape == (256,) and 0.02 < ref.std() and ref.min() > 1e-3 and ref.max() < 1 - 1e-3     # not saturated: a sensitive comparison
    assert np.abs(p - ref).max() <= EXEC_TOL
    if ver in ("001", "003"):
        ops = {s.split(":")[0] for s in g["ops_mlp_" + ver]}
        assert {"LookupTableFindV2", "SparseToDense", "OneHot", "Sum"} <= ops


def test_the_pin_is_sensitive_to_the_semantics_it_pins(g, samples):
    """A wrong genre vocabulary order, an unsorted DenseFeatures order or OOV -> index 0 must each break the agreement."""
    w = _mlp_w(g, "003")
    ref = g["mlp_003"]
    good = O.feature_column_mlp_forward(samples, w, **MLP_COLUMNS["003"])[:, 0]
    assert np.abs(good - ref).max() <= EXEC_TOL
    # (1) rows of the first kernel permuted as if the numeric columns came in script order instead of name order
    cols = MLP_COLUMNS["003"]
    _, offs = O.dense_features({**{k: np.zeros((1, 1)) for k in cols["numeric_keys"]},
                                **{k + "_indicator": np.zeros((1, 19)) for k in cols["genre_keys"]},
                                **{k + "_indicator": np.zeros((1, 1001)) for k in cols["int_vocab_keys"]}})
    k0 = w["dense0/kernel"].copy()
    a, b = offs["movieAvgRating"][0], offs["releaseYear"][0]
    k0[[a, b]] = k0[[b, a]]
    assert np.abs(O.feature_column_mlp_forward(samples, {**w, "dense0/kernel": k0}, **cols)[:, 0] - ref).max() > 1e-3
    # (2) genre vocabulary in another order
    saved = list(O.GENRE_VOCAB)
    try:
        O.GENRE_VOCAB[:] = sorted(saved)
        bad = O.feature_column_mlp_forward(samples, w, **cols)[:, 0]
    finally:
        O.GENRE_VOCAB[:] = saved
    assert np.abs(bad - ref).max() > 1e-4
    # (3) a missing history id (empty CSV field -> 0 -> vocabulary entry 0) is a real one-hot, not a zero row
    empties = np.nonzero(samples["userRatedMovie5"] == "")[0]
    assert len(empties) > 0


# ---------------------------------------------------------------------------------------------------------------------
# CPU, build container only: re-execute the reference's graphs live
# ---------------------------------------------------------------------------------------------------------------------
@needs_reference
def test_live_execution_reproduces_fixture_and_oracle_on_the_full_test_file(g):
    from sparrowrecsys_amd.schema import read_samples_csv
    from sparrowrecsys_amd.tensorbundle import model_variables
    from tests.golden.make_savedmodel_golden import REF, feed, load_model
    samples = read_samples_csv(os.path.join(REFERENCE, "src/main/resources/webroot/sampledata/testSamples.csv"))
    n = len(samples["movieId"])
    assert n == 22440
    for ver, first3 in (("001", [0.6695241, 0.5660429, 0.08600407]), ("002", [0.8525178, 0.51808727, 0.35965464])):
        m, _, standin = load_model("neuralcf/" + ver)
        assert not standin
        p = m.predict(feed(m, samples, n))[:, 0]
        np.testing.assert_allclose(p[:2048], g["ncf_" + ver], atol=1e-7)
        np.testing.assert_allclose(p[:3], first3, atol=2e-6)                        # SURVEY.md section 4's known answers
        w = ncf_weights(model_variables(REF + "modeldata/neuralcf/%s/variables" % ver))
        assert np.abs(O.neural_cf_forward(samples, w)[:, 0] - p).max() <= EXEC_TOL
    # TF's range assert is part of the exported graph: an id outside the table must fail there too
    bad = feed(m, samples, 8)
    bad["movieId"] = bad["movieId"].copy()
    bad["movieId"][3] = 1001
    with pytest.raises(AssertionError):
        m.predict(bad)


# ---------------------------------------------------------------------------------------------------------------------
# GPU: HIP == executed graph
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ver", ["001", "002"])
def test_hip_neuralcf_matches_the_exported_graph(g, ckpt, ver):
    from sparrowrecsys_amd import models as M
    feats = {"movieId": ckpt["movieId"], "userId": ckpt["userId"]}
    p = M.NeuralCF(weights=_ncf_w(ckpt, ver)).predict(feats)[:, 0]
    assert np.abs(p - g["ncf_" + ver]).max() <= HIP_TOL


@pytest.mark.gpu
def test_hip_two_tower_matches_the_exported_graph(g, ckpt):
    """The exported two-tower graph ends at the Dot; the HIP model's head Dense(1, sigmoid) is given kernel 1 / bias 0 and
    the executed graph's dot goes through the same sigmoid."""
    from sparrowrecsys_amd import models as M
    feats = {"movieId": ckpt["movieId"], "userId": ckpt["userId"]}
    w = _tower_w(g)
    w["head/kernel"] = np.ones((1, 1), np.float32)
    w["head/bias"] = np.zeros(1, np.float32)
    p = M.NeuralCF(weights=w, arch=2, hidden=(10,)).predict(feats)[:, 0]
    ref = O.sigmoid(g["tower_005"].astype(np.float64))
    assert np.abs(p - ref).max() <= HIP_TOL

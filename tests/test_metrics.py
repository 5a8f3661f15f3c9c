"""``model.evaluate``'s numbers (sparrowrecsys_amd/metrics.py) -- Keras' bucketed AUCs against the exact rank statistics and
sklearn, the cross-entropy / accuracy definitions, and (``-m gpu``) ``evaluate`` on the reference's own test rows."""
import numpy as np
import pytest

from sparrowrecsys_amd import metrics as MT


def _data(n=20000, seed=3):
    rng = np.random.default_rng(seed)
    y = (rng.random(n) < 0.56).astype(np.int64)
    p = 1.0 / (1.0 + np.exp(-(rng.normal(0, 1, n) + 1.2 * (y - 0.5))))
    return y, p.astype(np.float32)


def test_exact_auc_matches_sklearn():
    from sklearn.metrics import roc_auc_score
    y, p = _data()
    p = np.round(p, 2)                                                   # plenty of ties
    assert abs(MT.exact_roc_auc(y, p) - roc_auc_score(y, p)) < 1e-12


def test_keras_bucketed_aucs_approximate_the_exact_curves():
    from sklearn.metrics import average_precision_score, roc_auc_score
    y, p = _data()
    assert abs(MT.keras_auc(y, p, "ROC") - roc_auc_score(y, p)) < 2e-3     # 200 thresholds: an approximation by design
    assert abs(MT.keras_auc(y, p, "PR") - average_precision_score(y, p)) < 5e-3
    # perfect and inverted rankers
    yy = np.array([0, 0, 1, 1])
    assert MT.keras_auc(yy, np.array([0.1, 0.2, 0.8, 0.9]), "ROC") == pytest.approx(1.0)
    assert MT.keras_auc(yy, np.array([0.9, 0.8, 0.2, 0.1]), "ROC") == pytest.approx(0.0)
    assert MT.keras_auc(yy, np.array([0.1, 0.2, 0.8, 0.9]), "PR") == pytest.approx(1.0, abs=1e-6)


def test_threshold_semantics():
    # a prediction is positive when it is GREATER than the threshold: all-equal predictions give the chance diagonal
    y = np.array([0, 1, 0, 1, 1, 0])
    assert MT.keras_auc(y, np.full(6, 0.5), "ROC") == pytest.approx(0.5)
    assert MT.binary_accuracy(y, np.full(6, 0.5)) == pytest.approx(0.5)   # 0.5 is NOT > 0.5: everything predicted negative
    assert MT.binary_crossentropy(np.array([1.0, 0.0]), np.array([1.0, 0.0])) == pytest.approx(-np.log(1 - 1e-7), rel=1e-6)
    assert MT.binary_crossentropy(np.array([1.0]), np.array([0.25])) == pytest.approx(-np.log(0.25))


@pytest.mark.gpu
def test_evaluate_on_the_reference_test_rows_with_its_trained_neuralcf(samples):
    """NeuralCF/001's trained weights on the first 2 048 rows of the reference's testSamples.csv: evaluate() = the metrics of
    the oracle's predictions (ROC-AUC ~0.75 on the full file)."""
    import os

    from sparrowrecsys_amd import models as M
    from tests.conftest import GOLDEN
    from tests.test_savedmodel_pins import _ncf_w
    ckpt = np.load(os.path.join(GOLDEN, "neuralcf_ckpt.npz"))
    feats = {"movieId": ckpt["movieId"], "userId": ckpt["userId"], "label": ckpt["label"]}
    model = M.NeuralCF(weights=_ncf_w(ckpt, "001"))
    loss, acc, roc, pr = model.evaluate(feats)
    want = MT.evaluate_scores(ckpt["label"], ckpt["pred_001"])
    assert [loss, acc, roc, pr] == pytest.approx(want, abs=2e-5)
    assert 0.70 < roc < 0.80 and 0.6 < acc < 0.75
    halves = [({k: v[:1024] for k, v in feats.items() if k != "label"}, feats["label"][:1024]),
              ({k: v[1024:] for k, v in feats.items() if k != "label"}, feats["label"][1024:])]
    assert model.evaluate(halves) == pytest.approx([loss, acc, roc, pr], abs=1e-9)

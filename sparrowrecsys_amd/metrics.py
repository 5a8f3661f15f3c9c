"""``model.evaluate``'s numbers (DeepFM.py:117-126: ``loss='binary_crossentropy'``, ``metrics=['accuracy',
tf.keras.metrics.AUC(curve='ROC'), tf.keras.metrics.AUC(curve='PR')]`` -> ``[loss, accuracy, roc_auc, pr_auc]``), computed the
way Keras computes them so that a reference user reads the same figures:

* loss      mean binary cross-entropy with predictions clipped to [1e-7, 1 - 1e-7] (keras/backend.py binary_crossentropy)
* accuracy  ``binary_accuracy``: mean(label == (prediction > 0.5))
* AUC       ``tf.keras.metrics.AUC(num_thresholds=200, summation_method='interpolation')``: confusion counts at 200
            thresholds {-1e-7, 1/199 ... 198/199, 1 + 1e-7} (a prediction is positive when it is GREATER than the threshold);
            ROC: trapezoids over (FPR, TPR); PR: the Davis & Goadrich interpolation of keras/metrics.py ``interpolate_pr_auc``.
            These are approximations of the exact rank statistics by design (the exact ROC-AUC is ``exact_roc_auc``).
Host-side numpy: scoring is the hot path, these are reductions over its [N] output."""
from __future__ import annotations

import numpy as np

EPS = 1e-7


def _confusion(labels: np.ndarray, preds: np.ndarray, num_thresholds: int = 200):
    th = np.array([0.0 - EPS] + [(i + 1) / (num_thresholds - 1) for i in range(num_thresholds - 2)] + [1.0 + EPS])
    y = labels.astype(bool)
    order = np.sort(preds[y]), np.sort(preds[~y])
    # predictions > threshold, per class
    tp = len(order[0]) - np.searchsorted(order[0], th, side="right")
    fp = len(order[1]) - np.searchsorted(order[1], th, side="right")
    fn = len(order[0]) - tp
    tn = len(order[1]) - fp
    return tp.astype(np.float64), fp.astype(np.float64), tn.astype(np.float64), fn.astype(np.float64)


def _div_no_nan(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(b != 0, a / np.where(b != 0, b, 1), 0.0)


def keras_auc(labels, preds, curve: str = "ROC", num_thresholds: int = 200) -> float:
    labels, preds = np.asarray(labels).reshape(-1), np.asarray(preds, dtype=np.float64).reshape(-1)
    tp, fp, tn, fn = _confusion(labels, preds, num_thresholds)
    n = num_thresholds
    if curve == "ROC":
        x = _div_no_nan(fp, fp + tn)
        y = _div_no_nan(tp, tp + fn)
        return float(np.sum((x[:n - 1] - x[1:]) * (y[:n - 1] + y[1:]) / 2.0))
    if curve != "PR":
        raise ValueError("curve must be 'ROC' or 'PR'")
    dtp = tp[:n - 1] - tp[1:]
    p = tp + fp
    dp = p[:n - 1] - p[1:]
    slope = _div_no_nan(dtp, np.maximum(dp, 0))
    intercept = tp[1:] - slope * p[1:]
    ratio = np.where((p[:n - 1] > 0) & (p[1:] > 0), _div_no_nan(p[:n - 1], np.maximum(p[1:], 0)), 1.0)
    inc = _div_no_nan(slope * (dtp + intercept * np.log(ratio)), np.maximum(tp[1:] + fn[1:], 0))
    return float(np.sum(inc))


def exact_roc_auc(labels, preds) -> float:
    """The rank statistic itself (ties count one half) -- what sklearn.metrics.roc_auc_score returns."""
    labels, preds = np.asarray(labels).reshape(-1).astype(bool), np.asarray(preds, dtype=np.float64).reshape(-1)
    order = np.argsort(preds, kind="mergesort")
    ranks = np.empty(len(preds), dtype=np.float64)
    sp = preds[order]
    i = 0
    while i < len(sp):
        j = i
        while j + 1 < len(sp) and sp[j + 1] == sp[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    n_pos, n_neg = int(labels.sum()), int((~labels).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    return float((ranks[labels].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def binary_crossentropy(labels, preds) -> float:
    y = np.asarray(labels, dtype=np.float64).reshape(-1)
    p = np.clip(np.asarray(preds, dtype=np.float64).reshape(-1), EPS, 1.0 - EPS)
    return float(np.mean(-(y * np.log(p) + (1.0 - y) * np.log(1.0 - p))))


def binary_accuracy(labels, preds, threshold: float = 0.5) -> float:
    y = np.asarray(labels).reshape(-1).astype(np.float64)
    return float(np.mean((np.asarray(preds, dtype=np.float64).reshape(-1) > threshold).astype(np.float64) == y))


def evaluate_scores(labels, preds):
    """``[loss, accuracy, roc_auc, pr_auc]`` -- the list ``model.evaluate`` returns for the reference's compile() call."""
    return [binary_crossentropy(labels, preds), binary_accuracy(labels, preds), keras_auc(labels, preds, "ROC"), keras_auc(labels, preds, "PR")]

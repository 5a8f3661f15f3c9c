"""CPU ORACLE (test infrastructure, NOT the product) for the reference's "emb" ranker.

Restates, in plain Python / numpy, the Java of the online server:
``Embedding.calculateSimilarity`` (online/model/Embedding.java:33-47),
``RecForYouProcess.calculateEmbSimilarScore`` (:100-105) / ``SimilarMovieProcess.calculateEmbSimilarScore``
(:167-172), the ``"emb"`` case of ``ranker`` (RecForYouProcess.java:69-92, SimilarMovieProcess.java:121-136)
and ``Utility.parseEmbStr`` (online/util/Utility.java:6-13).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s cpu_baseline leg may import this module.

PARITY PIN STATUS: "parity unpinned" -- there is no JVM in this image (`javac` / `java` not found), and the
reference holds no test or golden vector for these functions.  What the tests do pin: IEEE known answers of
the restated arithmetic (identical / opposite / orthogonal vectors, a 3-4-5 case, zero vector -> NaN,
missing embedding -> -1) and Java's documented ordering (Double.compareTo: NaN greatest, 0.0 > -0.0).

Arithmetic (Java language spec): ``embVector.get(i) * other.get(i)`` is a float * float multiplication
(binary numeric promotion keeps float), rounded to float, then widened and added to a double accumulator;
the sums run in index order; ``Math.sqrt`` is correctly rounded; the quotient is one double division.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np


def parse_emb_str(s: str) -> np.ndarray:
    """Utility.parseEmbStr (Utility.java:6-13): split on single whitespace characters, Float.parseFloat each.
    (np.float32(str) rounds through double; for the <= 9 significant digits the reference's files carry this is
    the same float as Float.parseFloat.)"""
    return np.array([np.float32(tok) for tok in s.split(" ")], dtype=np.float32)


def calculate_similarity(a: Optional[np.ndarray], b: Optional[np.ndarray]) -> float:
    """Embedding.calculateSimilarity (Embedding.java:33-47), scalar loop -- the definition."""
    if a is None or b is None or len(a) != len(b):
        return -1.0
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    dot = n1 = n2 = 0.0                                   # Python floats are IEEE doubles
    for i in range(len(a)):
        dot += float(np.float32(a[i] * b[i]))             # float product, then widened
        n1 += float(np.float32(a[i] * a[i]))
        n2 += float(np.float32(b[i] * b[i]))
    den = math.sqrt(n1) * math.sqrt(n2)
    if den == 0.0:
        return float("nan") if dot == 0.0 or dot != dot else math.copysign(float("inf"), dot)   # Java: x / 0.0
    return dot / den


def scores(item_emb: np.ndarray, item_has: Optional[np.ndarray], query_emb: np.ndarray,
           query_has: Optional[np.ndarray], cand: np.ndarray) -> np.ndarray:
    """Vectorised calculateEmbSimilarScore over [Q, C] candidate lists (ids into item_emb, < 0 or >= N = movie
    without embedding); the same sums in the same order as calculate_similarity."""
    item_emb = np.asarray(item_emb, dtype=np.float32)
    query_emb = np.asarray(query_emb, dtype=np.float32)
    cand = np.asarray(cand)
    Q, C = cand.shape
    N, D = item_emb.shape
    ok = (cand >= 0) & (cand < N)
    safe = np.where(ok, cand, 0)
    if item_has is not None:
        ok &= np.asarray(item_has).astype(bool)[safe]
    if query_has is not None:
        ok &= np.asarray(query_has).astype(bool)[:, None]
    rows = item_emb[safe]                                            # [Q, C, D]
    q = query_emb[:, None, :]
    dot = np.zeros((Q, C)); n1 = np.zeros((Q, 1)); n2 = np.zeros((Q, C))
    for i in range(D):                                               # index order, double accumulators
        dot += (q[:, :, i] * rows[:, :, i]).astype(np.float32).astype(np.float64)
        n1 += (q[:, :, i] * q[:, :, i]).astype(np.float32).astype(np.float64)
        n2 += (rows[:, :, i] * rows[:, :, i]).astype(np.float32).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = dot / (np.sqrt(n1) * np.sqrt(n2))
    return np.where(ok, s, -1.0)


def _compare_key(s: np.ndarray) -> np.ndarray:
    """Order-preserving uint64 image of Double.compareTo: -inf < ... < -0.0 < 0.0 < ... < inf < NaN."""
    b = np.asarray(s, dtype=np.float64).view(np.uint64)
    neg = (b >> np.uint64(63)).astype(bool)
    k = np.where(neg, ~b, b | np.uint64(1 << 63))
    return np.where(np.isnan(s), np.uint64(0xFFFFFFFFFFFFFFFF), k)


def rank(s: np.ndarray) -> np.ndarray:
    """`sorted(Map.Entry.comparingByValue(Comparator.reverseOrder()))` (RecForYouProcess.java:90): candidate positions
    by descending score in Double.compareTo order; ties stay in candidate order (the reference's HashMap leaves ties
    unspecified -- this is the deterministic choice the HIP path makes too)."""
    s = np.atleast_2d(np.asarray(s, dtype=np.float64))
    k = _compare_key(s)
    return np.argsort(~k, axis=1, kind="stable").astype(np.int32)


def ranker_emb(query: Optional[np.ndarray], candidates: Sequence[Optional[np.ndarray]]) -> list:
    """ranker(user, candidates, "emb") for one query, object style (None = no embedding): positions ranked."""
    sc = np.array([calculate_similarity(query, c) if query is not None else -1.0 for c in candidates])
    return rank(sc)[0].tolist()

"""Synthetic MovieLens-20M-shaped inputs and the model shapes of BASELINE.json's configs
(SURVEY.md section 8(d)).  There is no network for datasets or checkpoints, so benchmarks and the
large-size parity tests draw ids and numerics of the right shape and ranges from a seeded RNG.

Feature dicts produced here carry INTEGER columns (genres already as vocabulary indices, -1 =
out-of-vocabulary): that is what ``CTRModel.pack`` and the oracle both accept.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .schema import N_GENRES, NUMERIC_KEYS

ML20M_MOVIE_IDS = 131263     # movieId <= 131 262 in MovieLens-20M
ML20M_USER_IDS = 138494      # 138 493 users
SEED = 20260921

# BASELINE config 2: "DeepFM emb_dim=16, 6 sparse fields"
CONFIG2_FIELDS = [("movieId", "id", ML20M_MOVIE_IDS), ("userId", "id", ML20M_USER_IDS),
                  ("userRatedMovie1", "id", ML20M_MOVIE_IDS), ("userGenre1", "genre", N_GENRES),
                  ("userGenre2", "genre", N_GENRES), ("movieGenre1", "genre", N_GENRES)]
# pair list for the pairwise-dot variant at 6 fields: every item-side x user-side pair
CONFIG2_PAIRS = [(a, b) for a in ("movieId", "movieGenre1")
                 for b in ("userId", "userRatedMovie1", "userGenre1", "userGenre2")]


def _draw_ids(rng, n, vocab, dist: str, low: int = 0):
    if dist == "uniform":
        return rng.integers(low, vocab, size=n, dtype=np.int64)
    if dist == "hot":
        # ablation only: every id inside a 1 k-row window, so all gathers hit L1/L2
        return rng.integers(low, min(vocab, low + 1024), size=n, dtype=np.int64)
    if dist == "zipf":
        # Zipf(s=1.05) popularity over a fixed random permutation of the ids
        ranks = np.arange(1, vocab - low + 1, dtype=np.float64)
        p = ranks ** -1.05
        p /= p.sum()
        perm = np.random.default_rng(SEED + vocab).permutation(vocab - low) + low
        return perm[rng.choice(vocab - low, size=n, p=p)].astype(np.int64)
    raise ValueError("unknown id distribution %r" % dist)


def synth_numerics(rng, B: int) -> Dict[str, np.ndarray]:
    """The 7 numeric columns in the empirical ranges of the reference's samples."""
    def log_uniform(lo, hi):
        return np.floor(np.exp(rng.uniform(np.log(lo), np.log(hi), size=B))).astype(np.int32)
    return {
        "releaseYear": rng.integers(1926, 2016, size=B).astype(np.int32),
        "movieRatingCount": log_uniform(2, 67000),
        "userRatingCount": log_uniform(2, 9000),
        "movieAvgRating": np.round(rng.uniform(0.5, 5.0, size=B), 2).astype(np.float32),
        "userAvgRating": np.round(rng.uniform(0.5, 5.0, size=B), 2).astype(np.float32),
        "movieRatingStddev": np.round(rng.uniform(0.0, 3.2, size=B), 2).astype(np.float32),
        "userRatingStddev": np.round(rng.uniform(0.0, 3.2, size=B), 2).astype(np.float32),
    }


def synth_fields(B: int, fields: Sequence[Tuple[str, str, int]], seed: int = SEED, dist: str = "uniform",
                 missing: float = 0.02) -> Dict[str, np.ndarray]:
    """Features for a DeepFM-style field list: identity ids in [1, vocab), genre indices with
    ``missing`` of the slots out-of-vocabulary (-1); a missing history id is 0 as in the CSV."""
    rng = np.random.default_rng(seed)
    feats = synth_numerics(rng, B)
    for key, kind, vocab in fields:
        if kind == "id":
            v = _draw_ids(rng, B, vocab, dist, low=1)
            if key.startswith("userRatedMovie"):
                v[rng.random(B) < missing] = 0
        else:
            v = rng.integers(0, vocab, size=B, dtype=np.int64)
            v[rng.random(B) < missing] = -1
        feats[key] = v
    return feats


def synth_din(B: int, hist_len: int, movie_vocab: int, user_vocab: int, seed: int = SEED,
              dist: str = "uniform") -> Dict[str, np.ndarray]:
    """DIN inputs: slot t of the history is missing (id 0) with probability rising 1% -> 60%
    (mirrors 1.3% -> 16.4% over the reference's 5 slots)."""
    rng = np.random.default_rng(seed)
    feats = synth_numerics(rng, B)
    feats["movieId"] = _draw_ids(rng, B, movie_vocab, dist, low=1)
    feats["userId"] = _draw_ids(rng, B, user_vocab, dist, low=1)
    hist = _draw_ids(rng, B * hist_len, movie_vocab, dist, low=1).reshape(B, hist_len)
    p_missing = np.linspace(0.01, 0.60, hist_len)[None, :]
    hist[rng.random((B, hist_len)) < p_missing] = 0
    feats["userRatedMovies"] = hist
    for key in ("userGenre1", "movieGenre1"):
        v = rng.integers(0, N_GENRES, size=B, dtype=np.int64)
        v[rng.random(B) < 0.02] = -1
        feats[key] = v
    return feats


def synth_embedding_mlp(B: int, movie_vocab: int, user_vocab: int, seed: int = SEED, dist: str = "uniform",
                        rated_vocab: Optional[int] = None) -> Dict[str, np.ndarray]:
    """EmbeddingMLP / Wide&Deep inputs (8 genre slots + movieId + userId [+ userRatedMovie1])."""
    rng = np.random.default_rng(seed)
    feats = synth_numerics(rng, B)
    feats["movieId"] = _draw_ids(rng, B, movie_vocab, dist, low=1)
    feats["userId"] = _draw_ids(rng, B, user_vocab, dist, low=1)
    for key in ["userGenre%d" % i for i in range(1, 6)] + ["movieGenre%d" % i for i in range(1, 4)]:
        v = rng.integers(0, N_GENRES, size=B, dtype=np.int64)
        v[rng.random(B) < 0.02] = -1
        feats[key] = v
    if rated_vocab:
        v = _draw_ids(rng, B, rated_vocab, dist, low=1)
        v[rng.random(B) < 0.02] = 0
        feats["userRatedMovie1"] = v
    return feats

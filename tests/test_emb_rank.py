"""The "emb" ranker: oracle known answers on CPU, HIP-vs-oracle bit-exact on the GPU."""
import math
import os

import numpy as np
import pytest

from oracle import emb_rank_oracle as EO
from tests.conftest import GOLDEN


# ---------------------------------------------------------------- CPU: the oracle's known answers
def test_parse_emb_str_reference_format():
    # first line of the reference's item2vecEmb.csv (webroot/modeldata/item2vecEmb.csv:1), value part
    v = EO.parse_emb_str("-1.1897237 0.48152843 -0.6113423 -0.40622446 -0.5651619 1.6490159 -0.5101197 1.1188633 -1.3080181 1.0282396")
    assert v.dtype == np.float32 and v.shape == (10,)
    assert v[0] == np.float32(-1.1897237) and v[9] == np.float32(1.0282396)


def test_calculate_similarity_known_answers():
    a = np.array([3, 4, 0], dtype=np.float32)
    assert EO.calculate_similarity(a, a) == 1.0                          # 25 / (5 * 5)
    assert EO.calculate_similarity(a, -a) == -1.0
    assert EO.calculate_similarity(a, np.array([0, 0, 2], dtype=np.float32)) == 0.0
    assert EO.calculate_similarity(a, np.array([4, 3, 0], dtype=np.float32)) == 24.0 / 25.0
    assert EO.calculate_similarity(a, None) == -1.0                      # Embedding.java:34
    assert EO.calculate_similarity(a, np.zeros(2, dtype=np.float32)) == -1.0     # size mismatch, :35
    assert math.isnan(EO.calculate_similarity(a, np.zeros(3, dtype=np.float32)))  # 0.0 / 0.0 in Java
    # float products: 0.1f * 0.1f is rounded to float BEFORE the double sum
    x = np.array([0.1], dtype=np.float32)
    p = float(np.float32(x[0] * x[0]))
    assert EO.calculate_similarity(x, x) == p / (math.sqrt(p) * math.sqrt(p))


def test_vectorised_scores_equal_scalar_definition():
    rng = np.random.default_rng(3)
    items = rng.normal(size=(50, 10)).astype(np.float32)
    items[7] = 0
    has = np.ones(50, dtype=np.uint8); has[11] = 0
    q = rng.normal(size=(4, 10)).astype(np.float32)
    qh = np.array([1, 1, 0, 1], dtype=np.uint8)
    cand = rng.integers(-1, 52, size=(4, 33))
    s = EO.scores(items, has, q, qh, cand)
    for u in range(4):
        for c in range(33):
            i = cand[u, c]
            b = items[i] if 0 <= i < 50 and has[i] else None
            want = EO.calculate_similarity(q[u], b) if qh[u] else -1.0
            assert (math.isnan(want) and math.isnan(s[u, c])) or want == s[u, c]


def test_rank_follows_double_compare_order():
    s = np.array([0.5, float("nan"), -1.0, 0.0, -0.0, float("inf"), 0.5, -1.0])
    assert EO.rank(s)[0].tolist() == [1, 5, 0, 6, 3, 4, 2, 7]            # NaN first, 0.0 before -0.0, ties by position
    assert EO.ranker_emb(None, [np.ones(2, dtype=np.float32)] * 3) == [0, 1, 2]


# ---------------------------------------------------------------- GPU: sprk_emb_rank against the oracle, bit for bit
def _case(Q, C, N, D, seed):
    rng = np.random.default_rng(seed)
    items = rng.normal(size=(N, D)).astype(np.float32)
    items[rng.integers(0, N, size=max(1, N // 50))] = 0                   # zero vectors -> NaN scores
    dup = rng.integers(0, N, size=(max(1, N // 10), 2))
    items[dup[:, 0]] = items[dup[:, 1]]                                   # exact ties
    has = (rng.random(N) > 0.05).astype(np.uint8)
    q = rng.normal(size=(Q, D)).astype(np.float32)
    qh = (rng.random(Q) > 0.1).astype(np.uint8)
    cand = rng.integers(-1, N + 2, size=(Q, C)).astype(np.int32)
    return items, has, q, qh, cand


@pytest.mark.gpu
@pytest.mark.parametrize("Q,C,N,D", [(1, 800, 881, 10), (37, 800, 881, 10), (5, 1, 3, 10), (3, 4096, 20000, 32), (16, 1000, 1001, 7), (9, 256, 300, 10), (6, 257, 300, 16), (3, 1024, 5000, 10), (2, 1025, 5000, 10)])
def test_emb_rank_bit_exact(Q, C, N, D):
    import torch
    from sparrowrecsys_amd.ranker import EmbRanker
    items, has, q, qh, cand = _case(Q, C, N, D, seed=Q * 131 + C)
    r = EmbRanker({i: items[i] for i in range(N)})
    r.has = torch.from_numpy(has).to(r.device)
    scores, order = r.score_many(q, cand, qh)
    want = EO.scores(items, has, q, qh, cand)
    got = scores.cpu().numpy()
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint64), want[~nan].view(np.uint64))      # bit-exact doubles
    assert np.array_equal(order.cpu().numpy(), EO.rank(want))
    os.environ["SPRK_EMB_RANK_GENERIC"] = "1"                           # the one-workgroup-per-query LDS kernel
    try:
        s3, o3 = r.score_many(q, cand, qh)
    finally:
        os.environ.pop("SPRK_EMB_RANK_GENERIC", None)
    assert np.array_equal(s3.cpu().numpy().view(np.uint64), got.view(np.uint64)) and np.array_equal(o3.cpu().numpy(), order.cpu().numpy())
    # scores only (no ranking) takes the same path
    s2, o2 = r.score_many(q, cand, qh, want_order=False)
    assert o2 is None and np.array_equal(s2.cpu().numpy().view(np.uint64), got.view(np.uint64))


@pytest.mark.gpu
def test_emb_rank_near_ties_take_the_exact_path():
    """Scores that agree in the upper 52 bits of their keys but differ below (cos of (1, k 1e-7) against (1, 0):
    1 - 5e-15 k^2): the one-word register sort would order them by position; the kernel must detect it and rank exactly.
    Also the generic LDS kernel (SPRK_EMB_RANK_GENERIC) must agree."""
    import torch
    from sparrowrecsys_amd.ranker import EmbRanker
    N = 300
    items = np.zeros((N, 2), dtype=np.float32)
    items[:, 0] = 1.0
    items[:, 1] = (1e-7 * np.arange(N, 0, -1)).astype(np.float32)      # candidate order = ASCENDING score
    q = np.array([[1.0, 0.0], [0.5, 0.0]], dtype=np.float32)
    cand = np.tile(np.arange(N, dtype=np.int32), (2, 1))
    want = EO.scores(items, None, q, None, cand)
    assert len(np.unique(want[0])) > N // 2 and np.ptp(want[0]) < 1e-9
    r = EmbRanker({i: items[i] for i in range(N)})
    for generic in (False, True):
        if generic:
            os.environ["SPRK_EMB_RANK_GENERIC"] = "1"
        try:
            scores, order = r.score_many(q, cand)
        finally:
            os.environ.pop("SPRK_EMB_RANK_GENERIC", None)
        assert np.array_equal(scores.cpu().numpy().view(np.uint64), want.view(np.uint64))
        assert np.array_equal(order.cpu().numpy(), EO.rank(want))
    assert EO.rank(want)[0, 0] != 0                                      # the order really is not the position order


@pytest.mark.gpu
def test_emb_ranker_object_api_and_errors():
    from sparrowrecsys_amd import _lib as L
    from sparrowrecsys_amd.ranker import EmbRanker
    emb = {10: np.array([1, 0], dtype=np.float32), 20: np.array([0, 1], dtype=np.float32), 30: np.array([1, 1], dtype=np.float32)}
    r = EmbRanker(emb)
    assert r.rank(np.array([1, 0.1], dtype=np.float32), [20, 30, 10, 99]) == [10, 30, 20, 99]     # 99: no embedding -> -1
    assert r.rank(None, [20, 30, 10]) == [20, 30, 10]                                                # user without embedding
    assert r.rank(np.ones(3, dtype=np.float32), [20, 30, 10]) == [20, 30, 10]                        # size mismatch -> all -1
    with pytest.raises(L.SparrowHipError):
        r.score_many(np.zeros((1, 2), dtype=np.float32), np.zeros((1, 5000), dtype=np.int32))        # ranking capped at 4096
    s, _ = r.score_many(np.zeros((1, 2), dtype=np.float32), np.zeros((1, 5000), dtype=np.int32), want_order=False)
    assert bool(np.isnan(s.cpu().numpy()).all())


@pytest.mark.gpu
def test_emb_rank_reference_embeddings_fixture():
    """Real embeddings from the reference's modeldata (first users / movies; tests/golden/make_golden.py writes the
    fixture with the oracle's scores): the HIP path reproduces them bit for bit and ranks the same way."""
    from sparrowrecsys_amd.ranker import EmbRanker
    z = np.load(os.path.join(GOLDEN, "emb_rank.npz"))
    r = EmbRanker({int(m): z["item_emb"][i] for i, m in enumerate(z["item_ids"])})
    cand = np.tile(r.rows(z["item_ids"]), (len(z["user_emb"]), 1))          # fixture order (the table is sorted by id)
    scores, order = r.score_many(z["user_emb"], cand)
    assert np.array_equal(scores.cpu().numpy().view(np.uint64), z["scores"].view(np.uint64))
    assert np.array_equal(order.cpu().numpy(), z["order"])


# ---------------------------------------------------------------- CPU: two independent restatements agree bit for bit
def test_c_restatement_equals_numpy_oracle():
    """oracle/emb_rank_c.c (plain C, float products / double sums written as the Java loop) against the numpy oracle:
    identical doubles, NaN and -1 cases included."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "_build", "libemb_rank_c.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True, capture_output=True)
    lib = C.CDLL(so)
    for (Q, Cn, N, D, seed) in [(7, 300, 500, 10, 1), (3, 64, 40, 32, 2), (2, 5, 3, 1, 3)]:
        items, has, q, qh, cand = _case(Q, Cn, N, D, seed)
        out = np.empty((Q, Cn), dtype=np.float64)
        lib.emb_rank_scores(items.ctypes.data_as(C.c_void_p), has.ctypes.data_as(C.c_void_p), C.c_int32(N), C.c_int32(D),
                            q.ctypes.data_as(C.c_void_p), qh.ctypes.data_as(C.c_void_p), C.c_int32(Q),
                            np.ascontiguousarray(cand).ctypes.data_as(C.c_void_p), C.c_int32(Cn), out.ctypes.data_as(C.c_void_p))
        want = EO.scores(items, has, q, qh, cand)
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(out), nan)
        assert np.array_equal(out[~nan].view(np.uint64), want[~nan].view(np.uint64))

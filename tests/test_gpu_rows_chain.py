"""k_rows_chain (``-m gpu``): the reference's LITERAL DeepFM_v2 (Dense(64) projections, DeepFM_v2.py:114,119) and NeuralCF
(NeuralCF.py:45-53, the model the Jetty server calls) on the fused "one table row per id" kernel instead of the plan
interpreter (VERDICT r01 missing #3 / #4) -- against the fp64 oracle, against the interpreter, through sprk_forward_many
with several batches per launch, with missing ids, ragged sizes and unaligned buffers."""
import numpy as np
import pytest

from oracle import ctr_oracle as O
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
from sparrowrecsys_amd.schema import N_GENRES

pytestmark = pytest.mark.gpu
TIGHT = 3e-5

REF_FIELDS = [("movieId", "id", SY.ML20M_MOVIE_IDS), ("userId", "id", SY.ML20M_USER_IDS),
              ("userGenre1", "genre", N_GENRES), ("movieGenre1", "genre", N_GENRES)]
REF_ORDER = ["movieGenre1", "movieId", "userGenre1", "userId"]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available()
    return t


def _v2(fields=REF_FIELDS, order=REF_ORDER, proj=64, D=10, seed=21):
    return M.DeepFMv2(seed=seed, emb_dim=D, fields=fields, order=order, proj_dim=proj)


def _oracle_v2(model, feats, dtype=np.float64):
    return O.deepfm_v2_forward(feats, model.weights, dtype=dtype, fields=model.fields, order=model.order)[:, 0]


@pytest.mark.parametrize("B", [65536, 4099, 17, 1])
def test_reference_deepfm_v2_runs_on_the_rows_chain(torch, B):
    model = _v2()
    assert model.engine.describe()["kernel"].startswith("k_rows_chain<KPC=4,H0C=2,H1C=1,G_BIG=2,NJF=2")
    feats = SY.synth_fields(B, REF_FIELDS, seed=5)
    p = model.predict(feats)[:, 0]
    n = min(B, 8192)
    ref = _oracle_v2(model, {k: v[:n] for k, v in feats.items()})
    assert np.abs(p[:n] - ref).max() <= TIGHT
    assert B < 100 or 0.05 < p.std()                                       # not a saturated comparison


@pytest.mark.parametrize("proj,fields,kernel", [
    (64, SY.CONFIG2_FIELDS, "k_rows_chain<KPC=4,H0C=2,H1C=1,G_BIG=3,NJF=3"),
    (32, REF_FIELDS, "k_rows_chain<KPC=2,H0C=2,H1C=1,G_BIG=2,NJF=2"),
    (40, REF_FIELDS, "k_rows_chain<KPC=3"),                                  # no instantiation: must still be right (interpreter)
    (64, [("movieId", "id", 5000), ("userGenre1", "genre", N_GENRES)], "k_rows_chain<KPC=4,H0C=2,H1C=1,G_BIG=1,NJF=1")])
def test_other_projection_widths_and_field_splits(torch, proj, fields, kernel):
    model = M.DeepFMv2(seed=23, emb_dim=16, fields=fields, proj_dim=proj)
    d = model.engine.describe()
    if proj == 40:
        assert d["kernel"] == "k_tile_forward" and d["fused"] == "0"      # the fallback is visible
    else:
        assert d["kernel"].startswith(kernel), d
    B = 3001
    feats = SY.synth_fields(B, fields, seed=7, missing=0.2)
    p = model.predict(feats)[:, 0]
    assert np.abs(p - _oracle_v2(model, feats)).max() <= TIGHT


def test_rows_chain_equals_interpreter_and_handles_missing_and_bad_ids(torch, monkeypatch):
    model = _v2()
    B = 2051
    feats = SY.synth_fields(B, REF_FIELDS, seed=9, missing=0.3)
    p = model.predict(feats)[:, 0]
    monkeypatch.setenv("SPRK_FORCE_INTERPRETER", "1")
    slow = M.DeepFMv2(weights=model.weights, emb_dim=10, fields=REF_FIELDS, order=REF_ORDER, proj_dim=64)
    assert slow.engine.describe()["kernel"] == "k_tile_forward"
    q = slow.predict(feats)[:, 0]
    monkeypatch.delenv("SPRK_FORCE_INTERPRETER")
    assert np.abs(p - q).max() <= 2e-6
    bad = dict(feats)
    bad["userId"] = feats["userId"].copy()
    bad["userId"][1000] = SY.ML20M_USER_IDS                                 # one past the table
    with pytest.raises(ValueError):
        model.predict(bad)
    np.testing.assert_array_equal(model.predict(feats)[:, 0], p)           # flag cleared, engine intact


def test_config2_shape_on_the_rows_chain_matches_the_joint_kernel(torch, monkeypatch):
    """A/B switch SPRK_V2_ROWS=1: BASELINE config 2 on k_rows_chain<KPC=1> (exact fp32) against k_deepfm_v2_joint (split f16)."""
    B = 8192
    feats = SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=11)
    joint = M.DeepFMv2(seed=101, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
    pj = joint.predict(feats)[:, 0]
    monkeypatch.setenv("SPRK_V2_ROWS", "1")
    rows = M.DeepFMv2(weights=joint.weights, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
    assert rows.engine.describe()["kernel"].startswith("k_rows_chain<KPC=1")
    pr = rows.predict(feats)[:, 0]
    ref = _oracle_v2(joint, feats)
    assert np.abs(pr - ref).max() <= TIGHT and np.abs(pj - ref).max() <= TIGHT


@pytest.mark.parametrize("B", [65536, 1000, 16, 3])
def test_neuralcf_runs_on_the_rows_chain(torch, monkeypatch, B):
    model = M.NeuralCF(seed=31, emb_dim=10, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
    assert model.engine.describe()["kernel"].startswith("k_rows_chain<KPC=0,H0C=1,H1C=1,G_BIG=2,NJF=0")
    rng = np.random.default_rng(3)
    feats = {"movieId": rng.integers(0, SY.ML20M_MOVIE_IDS, B), "userId": rng.integers(0, SY.ML20M_USER_IDS, B)}
    p = model.predict(feats)[:, 0]
    ref = O.neural_cf_forward(feats, model.weights, dtype=np.float64, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)[:, 0]
    assert np.abs(p - ref).max() <= TIGHT and (B < 100 or p.std() > 0.01)
    monkeypatch.setenv("SPRK_NCF_CHAIN", "0")
    slow = M.NeuralCF(weights=model.weights, emb_dim=10, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
    assert slow.engine.describe()["kernel"] == "k_tile_forward"
    assert np.abs(slow.predict(feats)[:, 0] - p).max() <= 2e-6


@pytest.mark.parametrize("which,B,n,k", [("v2", 4099, 7, 4), ("v2", 65536, 3, 16), ("ncf", 1000, 19, 16), ("ncf", 16, 3, 2), ("v2", 33, 70, 64)])
def test_rows_chain_several_batches_per_launch_is_bit_identical(torch, which, B, n, k):
    if which == "v2":
        model = _v2()
        packs = [model.pack(SY.synth_fields(B, REF_FIELDS, seed=40 + i, missing=0.1)) for i in range(n)]
    else:
        model = M.NeuralCF(seed=31, emb_dim=10, movie_buckets=5000, user_buckets=7000)
        rng = np.random.default_rng(5)
        packs = [model.pack({"movieId": rng.integers(0, 5000, B), "userId": rng.integers(0, 7000, B)}) for i in range(n)]
    eng = model.engine
    ids = [torch.from_numpy(a).cuda() for a, _ in packs]
    dense = [torch.from_numpy(b).cuda() for _, b in packs]
    one = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(n)]
    many = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    for i in range(n):
        eng.forward(ids[i], dense[i], one[i])
    eng.set_many_batches(k)
    eng.forward_many(ids, dense, many)
    eng.set_many_batches(1)
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(one[i], many[i]), (i, float((one[i] - many[i]).abs().max()), int((one[i] != many[i]).sum()))
    eng.check_ids()


def test_rows_chain_unaligned_views_and_every_tail_length(torch):
    model = _v2()
    eng = model.engine
    B = 200
    ids, dense = model.pack(SY.synth_fields(B + 8, REF_FIELDS, seed=51))
    ids_t, dense_t = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    full = torch.empty(B + 8, dtype=torch.float32, device="cuda")
    eng.forward(ids_t, dense_t, full)
    for start in (1, 3):
        for n in list(range(1, 34)) + [B]:
            out = torch.empty(n, dtype=torch.float32, device="cuda")
            eng.forward(ids_t[start:start + n], dense_t[start:start + n], out)     # row views: contiguous, not 16-byte aligned
            torch.cuda.synchronize()
            assert torch.equal(out, full[start:start + n]), (start, n)


@pytest.mark.parametrize("fields,proj,emb", [(REF_FIELDS, 64, 10), (REF_FIELDS, 32, 16),
                                             ([("movieId", "id", 5000), ("userGenre1", "genre", N_GENRES)], 64, 16)])
def test_unfolded_big_rows_equal_the_folded_rows(torch, monkeypatch, fields, proj, emb):
    """UNF (raw split-f16 rows of the big fields, projections on the matrix pipe: 64 bytes per id instead of 384) against the folded
    {P | Q} rows (SPRK_ROWS_UNF=0, exact fp32) and the fp64 oracle: missing ids (the all-zero row), one and two big fields, every
    launch form (one task per wave, the task loop, several batches per launch)."""
    B = 40000
    feats = SY.synth_fields(B, fields, seed=11, missing=0.15)
    new = M.DeepFMv2(seed=29, emb_dim=emb, fields=fields, proj_dim=proj)
    assert ",UNF>" in new.engine.describe()["kernel"], new.engine.describe()
    p = new.predict(feats)[:, 0]
    monkeypatch.setenv("SPRK_ROWS_UNF", "0")
    old = M.DeepFMv2(seed=29, emb_dim=emb, fields=fields, proj_dim=proj)
    assert "UNF" not in old.engine.describe()["kernel"]
    q = old.predict(feats)[:, 0]
    monkeypatch.delenv("SPRK_ROWS_UNF")
    n = 8192
    ref = _oracle_v2(new, {k: v[:n] for k, v in feats.items()})
    assert np.abs(p - q).max() <= 3e-6 and np.abs(p[:n] - ref).max() <= TIGHT and p.std() > 0.02
    assert new.engine.table_bytes() < old.engine.table_bytes()
    # the looped kernel (a launch of more tasks than one-per-wave allows) and small launches give the same bits per row
    ids, dense = new.pack(feats)
    ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    full = new.predict_device(ti, td)
    for lo, hi in ((0, 17), (5, 4101), (B - 33, B)):
        assert torch.equal(new.predict_device(ti[lo:hi], td[lo:hi]), full[lo:hi]), (lo, hi)
    big = {k: np.concatenate([v] * 8) for k, v in feats.items()}              # 320 000 rows: beyond V2J1_MAX_TASKS * 16
    pb = new.predict(big)[:, 0]
    np.testing.assert_array_equal(pb[:B], p)
    np.testing.assert_array_equal(pb[-B:], p)

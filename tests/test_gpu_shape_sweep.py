"""Randomised shape sweep (seeded): models of shapes NO kernel instantiation was written for -- odd field counts, vocabularies
on both sides of the LDS threshold, embedding / projection / hidden widths that do or do not tile -- through whatever the
engine dispatches them to (fused kernels where a variant matches, the rows chain, the interpreter otherwise), every time
against the fp64 oracle at the path's tolerance, at ragged batch sizes.  The dispatch is printed: a shape that silently falls
to the interpreter is visible (`-m gpu -s`)."""
import numpy as np
import pytest

from oracle import ctr_oracle as O
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY

pytestmark = pytest.mark.gpu
TOL = 1e-4

GENRES = ["userGenre1", "userGenre2", "userGenre3", "movieGenre1", "movieGenre2", "movieGenre3"]
IDS = ["movieId", "userId", "userRatedMovie1"]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "gpu tests need a HIP device"
    return t


def _fields(rng):
    n_id = int(rng.integers(1, 4))
    n_g = int(rng.integers(1, 5))
    fields = [(k, "id", int(rng.choice([25, 40, 1001, 30001, 131071]))) for k in rng.permutation(IDS)[:n_id]]
    fields += [(k, "genre", 19) for k in rng.permutation(GENRES)[:n_g]]
    return [fields[i] for i in rng.permutation(len(fields))]


@pytest.mark.parametrize("case", range(14))
def test_deepfm_v2_random_shapes(torch, case):
    rng = np.random.default_rng(1000 + case)
    fields = _fields(rng)
    order = [k for k, _, _ in fields]
    emb_dim = int(rng.choice([4, 6, 10, 16, 32]))
    proj_dim = int(rng.choice([8, 16, 24, 64]))
    hidden = [(32, 16), (64, 32), (16,), (48, 16), (32, 32, 16)][int(rng.integers(0, 5))]
    B = int(rng.choice([1, 17, 1000, 4099]))
    model = M.DeepFMv2(seed=2000 + case, emb_dim=emb_dim, fields=fields, proj_dim=proj_dim, hidden=hidden)
    feats = SY.synth_fields(B, fields, seed=3000 + case)
    got = model.predict(feats)[:, 0]
    ref = O.deepfm_v2_forward(feats, model.weights, dtype=np.float64, fields=fields, order=order)[:, 0]
    err = float(np.abs(got - ref).max())
    print("deepfm_v2 #%d: %d fields %s, emb %d, proj %d, hidden %s, B %d -> %s, max|err| %.2e"
          % (case, len(fields), [v for _, _, v in fields], emb_dim, proj_dim, hidden, B, model.engine.describe()["kernel"], err))
    assert err <= TOL


@pytest.mark.parametrize("case", range(8))
def test_deepfm_pair_dot_random_shapes(torch, case):
    rng = np.random.default_rng(4000 + case)
    fields = _fields(rng)
    names = [k for k, _, _ in fields]
    all_pairs = [(a, b) for i, a in enumerate(names) for b in names[i + 1:]]
    pairs = [all_pairs[i] for i in rng.permutation(len(all_pairs))[:max(1, int(rng.integers(1, len(all_pairs) + 1)))]]
    id_names = [k for k, kind, _ in fields if kind == "id"]
    deep_emb = id_names[:int(rng.integers(1, min(2, len(id_names)) + 1))]
    emb_dim = int(rng.choice([4, 10, 16, 64]))
    hidden = [(64, 64), (32, 16), (128, 64)][int(rng.integers(0, 3))]
    B = int(rng.choice([1, 33, 2051]))
    model = M.DeepFM(seed=5000 + case, emb_dim=emb_dim, fields=fields, pairs=pairs, deep_emb=deep_emb, hidden=hidden)
    feats = SY.synth_fields(B, fields, seed=6000 + case)
    try:
        got = model.predict(feats)[:, 0]
    except RuntimeError as e:
        # a shape with no fused kernel runs on the plan interpreter, whose 64-sample tile must fit the CU's 160 KiB of LDS: 7 FM
        # fields + the deep part's own table at emb_dim 64 = 8 x 64 floats per sample do not.  The engine must say so, loudly.
        if "bytes of LDS per tile" in str(e) and (len(fields) + len(deep_emb)) * emb_dim * 64 * 4 > 120 * 1024:
            pytest.skip("shape beyond the interpreter's LDS tile, refused explicitly: %s" % e)
        raise
    ref = O.deepfm_forward(feats, model.weights, dtype=np.float64, fields=fields, pairs=pairs, deep_emb=deep_emb)[:, 0]
    err = float(np.abs(got - ref).max())
    print("deepfm #%d: %d fields, %d pairs, deep %s, emb %d, hidden %s, B %d -> %s, max|err| %.2e"
          % (case, len(fields), len(pairs), deep_emb, emb_dim, hidden, B, model.engine.describe()["kernel"], err))
    assert err <= TOL


@pytest.mark.parametrize("case", range(8))
def test_din_random_shapes(torch, case):
    rng = np.random.default_rng(7000 + case)
    D = int(rng.choice([6, 10, 16, 32]))
    T = int(rng.choice([1, 5, 17, 50, 64]))
    V, U = int(rng.choice([101, 1001, 20000])), int(rng.choice([50, 30001]))
    hidden = [(128, 64), (64, 32)][int(rng.integers(0, 2))]
    B = int(rng.choice([1, 13, 777, 3001]))
    model = M.DIN(seed=8000 + case, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U, hidden=hidden)
    feats = SY.synth_din(B, T, V, U, seed=9000 + case)
    got = model.predict(feats)[:, 0]
    ref = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    err = float(np.abs(got - ref).max())
    d = model.engine.describe()
    print("din #%d: D %d, T %d, vocab %d / %d, hidden %s, B %d -> %s + %s, max|err| %.2e" % (case, D, T, V, U, hidden, B, d["stage"], d["kernel"], err))
    assert err <= TOL


@pytest.mark.parametrize("case", range(6))
def test_neuralcf_random_shapes(torch, case):
    rng = np.random.default_rng(10000 + case)
    D = int(rng.choice([4, 10, 16, 24]))
    hidden = [(10, 10), (32, 16, 8), (16,), (64, 32)][int(rng.integers(0, 4))]
    V, U = int(rng.choice([101, 1001, 50000])), int(rng.choice([77, 30001]))
    arch = 1 if case % 3 else 2
    B = int(rng.choice([1, 19, 5000]))
    model = M.NeuralCF(seed=11000 + case, emb_dim=D, movie_buckets=V, user_buckets=U, hidden=hidden, arch=arch)
    r2 = np.random.default_rng(12000 + case)
    feats = {"movieId": r2.integers(0, V, B).astype(np.int32), "userId": r2.integers(0, U, B).astype(np.int32)}
    got = model.predict(feats)[:, 0]
    fwd = O.neural_cf_forward if arch == 1 else O.neural_cf2_forward
    ref = fwd(feats, model.weights, dtype=np.float64, movie_buckets=V, user_buckets=U)[:, 0]
    err = float(np.abs(got - ref).max())
    print("neuralcf #%d: arch %d, D %d, hidden %s, vocab %d / %d, B %d -> %s, max|err| %.2e"
          % (case, arch, D, hidden, V, U, B, model.engine.describe()["kernel"], err))
    assert err <= TOL


_FULL = [("din", 50, 32, {}), ("din", 20, 16, {}), ("din", 12, 10, {}), ("din", 5, 10, {}), ("din", 64, 32, {}), ("din", 30, 20, {}),
         ("din", 50, 32, {"SPRK_DIN_FUSED": "0"}), ("din", 20, 16, {"SPRK_DIN_FUSED": "0"}), ("din", 5, 10, {"SPRK_DIN_FUSED_MIN_T": "1"}),
         ("dien", 5, 10, {}), ("dien", 7, 16, {}), ("dien", 20, 16, {}), ("dien", 50, 10, {}),
         ("dien", 5, 10, {"SPRK_DIEN_FUSED": "0"}), ("dien", 7, 16, {"SPRK_DIEN_FUSED": "0"}), ("dien", 20, 16, {"SPRK_DIEN_FUSED": "0"}),
         ("dien", 7, 16, {"SPRK_DIEN_MFMA": "0"})]


@pytest.mark.parametrize("case", range(len(_FULL)), ids=lambda i: "%s-T%d-D%d%s" % (_FULL[i][0], _FULL[i][1], _FULL[i][2], "".join("-%s=%s" % kv for kv in _FULL[i][3].items())))
def test_din_and_dien_dispatch_shapes_at_full_occupancy(torch, monkeypatch, case):
    """[r5] Every DIN / DIEN dispatch (one launch, two launches, raw-row and folded tails, the lane-per-sample DIEN stage) at B = 65 536 -- every CU
    full --, 20 launches each bit for bit the first, and the first within 2e-6 of the fp64 oracle on every 16th tile.  The sweeps above run a few
    thousand rows; k_dien_seq_mfma<16, 32>'s flaky tiles (k_dien_fused.h) only showed with four waves per SIMD.  scripts/r05/dbg/
    full_occupancy_sweep.py is the same loop as a script (profiles/r05/experiments/r05_37: all 17 clean, max 6.6e-7)."""
    kind, T, D, env = _FULL[case]
    for k in ("SPRK_DIN_FUSED", "SPRK_DIN_FUSED_MIN_T", "SPRK_DIEN_FUSED", "SPRK_DIEN_MFMA"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B, V, U = 65536, 5000, 7000
    feats = SY.synth_din(B, T, V, U, seed=7 + T + D)
    h = feats["userRatedMovies"]
    h[np.random.default_rng(T).random(h.shape) < 0.2] = 0
    m = (M.DIN if kind == "din" else M.DIEN)(seed=5, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    ids, dense = m.pack(feats)
    ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    first = m.predict_device(ti, td).clone()
    bad = sum(int((m.predict_device(ti, td) != first).sum().item()) for _ in range(20))
    rows = np.concatenate([np.arange(t * 16, t * 16 + 16) for t in range(0, B // 16, 16)])
    fwd = O.din_forward if kind == "din" else O.dien_forward
    ref = fwd({k: v[rows] for k, v in feats.items()}, m.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    err = float(np.abs(first.cpu().numpy()[rows] - ref).max())
    m.engine.close()
    assert bad == 0, "%d scores of 20 launches differ from the first launch" % bad
    assert err <= 2e-6, err

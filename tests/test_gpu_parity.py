"""Parity tests proper (``-m gpu``): the HIP path, called through the C ABI, against the oracle and
the committed golden vectors.  Bar: bit-exact for index/integer work (row gather, cross hash),
<= 1e-4 absolute on the post-sigmoid score for floating point (BASELINE.json north_star); the
asserts below use a tighter 3e-5 because both sides are fp32-class on these inputs."""
import os

import numpy as np
import pytest

from oracle import ctr_oracle as O
from sparrowrecsys_amd import _lib as L
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
from tests.conftest import GOLDEN
from tests.golden.make_golden import SEEDS, make_model

pytestmark = pytest.mark.gpu

TOL = 1e-4        # the stated bar
TIGHT = 3e-5      # what we actually hold on non-saturating inputs (fp32 summation-order noise is ~1e-5)


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "gpu tests need a HIP device"
    assert os.path.exists(L.LIB_PATH), "libsparrow_hip.so must be built in-tree"
    return t


def _cuda(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# --------------------------------------------------------------------------------------------
# golden vectors, all seven reference graphs (BASELINE config 1 = embedding_mlp on the bundled batch)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(SEEDS))
def test_reference_models_match_golden(torch, samples, name):
    g = np.load(os.path.join(GOLDEN, "oracle_%s.npz" % name))
    model = make_model(name)
    p = model.predict(samples)
    assert p.shape == (256, 1) and p.dtype == np.float32
    assert np.abs(p[:, 0] - g["pred32"]).max() <= TOL
    assert np.abs(p[:, 0] - g["pred64"]).max() <= TIGHT
    # predict(batch_size=12) (the reference's batch, DeepFM.py:17) gives the same rows
    p12 = model.predict(samples, batch_size=12)
    np.testing.assert_array_equal(p12, p)


def test_trained_neuralcf_checkpoint_subset(torch):
    """Weights TRAINED by the reference (modeldata/neuralcf/{001,002}), 2048 real test rows."""
    g = np.load(os.path.join(GOLDEN, "neuralcf_ckpt.npz"))
    feats = {"movieId": g["movieId"], "userId": g["userId"]}
    for ver in ("001", "002"):
        w = {k[len(ver) + 1:]: g[k] for k in g.files if k.startswith(ver + "/")}
        table = np.zeros((30001, 10), np.float32)
        table[g["users"]] = g["user_rows_" + ver]
        w["emb/userId"] = table
        p = M.NeuralCF(weights=w).predict(feats)[:, 0]
        assert np.abs(p - g["pred_" + ver]).max() <= TIGHT
    np.testing.assert_allclose(p[:3], [0.8525178, 0.51808727, 0.35965464], atol=TIGHT)


# --------------------------------------------------------------------------------------------
# ragged / edge sizes
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [1, 2, 63, 64, 65, 127, 129, 1000])
def test_ragged_batch_sizes(torch, samples, B):
    reps = (B + 255) // 256
    feats = {k: np.concatenate([v] * reps)[:B] for k, v in samples.items()}
    for name in ("deepfm_v2", "din"):
        model = make_model(name)
        p = model.predict(feats)
        ref = O.FORWARDS[name](feats, model.weights, dtype=np.float64)
        assert p.shape == (B, 1)
        assert np.abs(p - ref).max() <= TIGHT


def test_empty_batch(torch, samples):
    model = make_model("deepfm")
    empty = {k: v[:0] for k, v in samples.items()}
    p = model.predict(empty)
    assert p.shape == (0, 1) and p.dtype == np.float32


def test_out_of_range_ids_raise(torch, samples):
    model = make_model("deepfm")
    bad = dict(samples)
    bad["userId"] = np.array(["30001"] * 256, dtype=object)
    with pytest.raises(ValueError):
        model.predict(bad)
    # device-resident ids skip host packing: the kernel flags them (TF: assert_less_than_num_buckets)
    ids, dense = model.pack(samples)
    ids[5, 0] = 5000
    model.predict_device(_cuda(torch, ids), _cuda(torch, dense))
    with pytest.raises(ValueError):
        model.engine.check_ids()
    model.predict_device(_cuda(torch, model.pack(samples)[0]), _cuda(torch, dense))
    model.engine.check_ids()        # flag was cleared; clean batch passes


def test_missing_id_is_a_zero_row(torch, samples):
    """-1 (OOV / missing) must behave exactly like an all-zero embedding row and zero first-order weight."""
    model = make_model("deepfm_v2")
    ids, dense = model.pack(samples)
    ids[:, 2] = -1                                   # userGenre1 column
    p = model.predict_device(_cuda(torch, ids), _cuda(torch, dense)).cpu().numpy()
    w = dict(model.weights)
    w["emb/userGenre1"] = np.zeros_like(w["emb/userGenre1"])
    fk = w["fo_cat/kernel"].copy()
    fk[model.fo["userGenre1"]:model.fo["userGenre1"] + 19] = 0
    w["fo_cat/kernel"] = fk
    ids2, _ = model.pack(samples)
    ids2[:, 2] = np.where(ids2[:, 2] < 0, 0, ids2[:, 2])
    q = M.DeepFMv2(weights=w).predict_device(_cuda(torch, ids2), _cuda(torch, dense)).cpu().numpy()
    np.testing.assert_array_equal(p, q)


# --------------------------------------------------------------------------------------------
# integer / index work: bit-exact
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("V,D", [(1001, 10), (30001, 16), (5000, 32), (2000, 64)])
def test_embedding_gather_bit_exact(torch, V, D):
    from sparrowrecsys_amd.plan import pad_table
    import ctypes as C
    rng = np.random.default_rng(V + D)
    table = rng.standard_normal((V, D)).astype(np.float32)
    padded = pad_table(table)
    Dp = padded.shape[1]
    ids = rng.integers(0, V, size=4099).astype(np.int32)
    ids[::7] = -1
    ids[1], ids[2] = 0, V - 1
    t, i = _cuda(torch, padded), _cuda(torch, ids)
    out = torch.empty((len(ids), Dp), dtype=torch.float32, device="cuda")
    lib = L.load_library()
    L.check(lib.sprk_embedding_gather(C.c_void_p(t.data_ptr()), V, Dp, Dp, C.c_void_p(i.data_ptr()), len(ids),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    got = out.cpu().numpy()
    want = O.embedding_lookup(table, ids.astype(np.int64))
    assert got[:, :D].tobytes() == want.tobytes()          # bit-exact copy
    assert not got[:, D:].any()


def test_cross_hash_bit_exact(torch):
    import ctypes as C
    g = np.load(os.path.join(GOLDEN, "cross_hash.npz"))
    rng = np.random.default_rng(9)
    a = np.concatenate([g["a"], rng.integers(0, 2 ** 31, 100000)]).astype(np.int32)
    b = np.concatenate([g["b"], rng.integers(0, 2 ** 31, 100000)]).astype(np.int32)
    lib = L.load_library()
    for buckets, key in ((10000, "b10000"), (10_000_000, "b10m")):
        ta, tb = _cuda(torch, a), _cuda(torch, b)
        out = torch.empty(len(a), dtype=torch.int64, device="cuda")
        L.check(lib.sprk_cross_hash(C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), len(a), buckets,
                                    C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        got = out.cpu().numpy()
        np.testing.assert_array_equal(got[:64], g[key])                                   # python-int golden
        np.testing.assert_array_equal(got, O.crossed_bucket_np([a, b], buckets))           # uint64 numpy


# --------------------------------------------------------------------------------------------
# DIN stage
# --------------------------------------------------------------------------------------------
def test_din_attention_and_pooling_parts(torch, samples):
    model = make_model("din")
    ids, _ = model.pack(samples)
    eng = model.engine
    Dp = eng.n_aux
    pooled = torch.empty((256, Dp), dtype=torch.float32, device="cuda")
    att = torch.empty((256, 5), dtype=torch.float32, device="cuda")
    eng.din_pool(_cuda(torch, ids), pooled, att)
    _, parts = O.din_forward(samples, model.weights, dtype=np.float64, return_parts=True)
    assert np.abs(att.cpu().numpy() - parts["att"]).max() <= TIGHT
    assert np.abs(pooled.cpu().numpy()[:, :10] - parts["pooled"]).max() <= TIGHT
    assert not pooled.cpu().numpy()[:, 10:].any()


@pytest.mark.parametrize("T,D,B", [(50, 32, 4099), (64, 32, 515), (17, 32, 1000), (1, 32, 130), (33, 24, 700),
                                   (50, 16, 1025), (20, 10, 900), (30, 10, 901), (5, 10, 64), (50, 8, 333)])
def test_din_attention_kernel_shapes(torch, monkeypatch, T, D, B):
    """The attention stage (A_b = W12 + W4 diag(c), per-id c-term table) on k_din_fused, on k_din_attn_cols + k_din_tail and on the generic
    k_din_pool (SPRK_DIN_LEGACY=1; [r6] also where SPRK_DIN_COLS=0 / SPRK_DIN_HALF=0 lead since k_din_attn is retired) against the fp64
    oracle's attention weights and pooled vector, over history lengths / row widths that hit every instantiation, partial 16-row groups
    and ragged batches."""
    V, U = 5000, 700
    feats = SY.synth_din(B, T, V, U, seed=100 + T + D)
    got = {}
    modes = ("0", "cols", "wave", "1")                              # k_din_fused / k_din_attn_cols + k_din_tail / SPRK_DIN_COLS=0 (the generic stage) / k_din_pool
    for legacy in modes:
        monkeypatch.setenv("SPRK_DIN_LEGACY", "1" if legacy == "1" else "0")
        monkeypatch.setenv("SPRK_DIN_COLS", "0" if legacy == "wave" else "1")
        monkeypatch.setenv("SPRK_DIN_FUSED", "1" if legacy == "0" else "0")
        model = M.DIN(seed=50 + T, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        ids, dense = model.pack(feats)
        eng = model.engine
        Dp = eng.n_aux
        pooled = torch.full((B, Dp), float("nan"), dtype=torch.float32, device="cuda")
        att = torch.full((B, T), float("nan"), dtype=torch.float32, device="cuda")
        eng.din_pool(_cuda(torch, ids), pooled, att)
        eng.check_ids()
        score = model.predict_device(_cuda(torch, ids), _cuda(torch, dense)).cpu().numpy()
        got[legacy] = (att.cpu().numpy(), pooled.cpu().numpy(), score)
        eng.close()
    ref, parts = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U,
                               return_parts=True)
    for legacy in modes:
        a, p, sc = got[legacy]
        assert np.isfinite(a).all() and np.isfinite(p).all()
        assert np.abs(a - parts["att"]).max() <= TIGHT
        assert np.abs(p[:, :D] - parts["pooled"]).max() <= TIGHT
        assert not p[:, D:].any()
        assert np.abs(sc - ref[:, 0]).max() <= TOL
    assert np.abs(got["0"][0] - got["1"][0]).max() <= 2e-6      # two summation orders of the same fp32 math
    assert np.abs(got["0"][0] - got["wave"][0]).max() <= 2e-6
    assert np.abs(got["0"][0] - got["cols"][0]).max() <= 2e-6
    assert np.abs(got["0"][2] - got["cols"][2]).max() <= 3e-6     # the whole forward: one launch against attention -> pooled -> tail


@pytest.mark.parametrize("T,D", [(50, 32), (5, 10), (23, 16)])
def test_din_cols_result_does_not_depend_on_the_launch_shape(torch, T, D):
    """k_din_attn_cols gives a task to 1, 2 or 4 waves depending on how many tasks a launch has (time slices of the history);
    the pooled sum is formed as (q0 + q1) + (q2 + q3) over the four quarters' in-order partial sums whichever it is, so a
    row's pooled vector must be the SAME BITS in a 20 000-row launch (one wave per task), a 9 000-row launch (two) and a
    600-row launch (four) -- and in the several-batches-per-launch form."""
    V, U, B = 5000, 700, 20000
    feats = SY.synth_din(B, T, V, U, seed=300 + T)
    model = M.DIN(seed=60 + T, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    ids, dense = model.pack(feats)
    eng = model.engine
    ti = _cuda(torch, ids)
    full = torch.empty((B, eng.n_aux), dtype=torch.float32, device="cuda")
    eng.din_pool(ti, full, None)
    for n in (9000, 600, 37):
        part = torch.empty((n, eng.n_aux), dtype=torch.float32, device="cuda")
        eng.din_pool(ti[:n].contiguous(), part, None)
        assert torch.equal(part, full[:n]), "pooled vectors of a %d-row launch differ from the %d-row launch" % (n, B)
    eng.check_ids()
    _, parts = O.din_forward({k: v[:2048] for k, v in feats.items()}, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V,
                             user_buckets=U, return_parts=True)
    assert np.abs(full[:2048, :D].cpu().numpy() - parts["pooled"]).max() <= TIGHT


@pytest.mark.parametrize("T,D,B", [(50, 32, 4099), (12, 10, 1000), (23, 16, 37), (64, 32, 600)])
def test_din_fused_scores_do_not_depend_on_the_launch_shape(torch, monkeypatch, T, D, B):
    """k_din_fused (attention + pooling + tail in ONE launch, DIN.py:132-167): the same bits whether a task's history is walked by one,
    two or four waves (SPRK_DIN_COLS_TS), for a slice of the batch, and in the several-batches-per-launch form; <= 3e-6 from the
    two-launch path (k_din_fused<TAIL = false> -> pooled vectors -> k_din_tail) and within the bar of the fp64 oracle; an id outside
    a tail column's table (userId) is flagged like one outside the history's."""
    V, U = 5000, 700
    feats = SY.synth_din(B, T, V, U, seed=400 + T)
    feats["movieGenre1"][::5] = -1                                  # no id: the all-zero row of the folded table
    got = {}
    for tag, env in (("ts1", {"SPRK_DIN_COLS_TS": "1"}), ("ts2", {"SPRK_DIN_COLS_TS": "2"}), ("ts4", {"SPRK_DIN_COLS_TS": "4"}),
                     ("auto", {"SPRK_DIN_COLS_TS": "0"}), ("two", {"SPRK_DIN_COLS_TS": "0", "SPRK_DIN_FUSED": "0"})):
        monkeypatch.setenv("SPRK_DIN_FUSED", "1")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = M.DIN(seed=90 + T, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        assert model.engine.describe()["kernel"].startswith("k_din_fused" if tag != "two" else "k_din_tail")
        ids, dense = model.pack(feats)
        ti, td = _cuda(torch, ids), _cuda(torch, dense)
        full = model.predict_device(ti, td)
        model.engine.check_ids()
        got[tag] = full.cpu().numpy()
        if tag == "auto":
            lo, hi = B // 3, B // 3 + min(B // 2, 513)
            assert torch.equal(model.predict_device(ti[lo:hi].contiguous(), td[lo:hi].contiguous()), full[lo:hi])
            bad = ids.copy()
            bad[B // 2, 1 + T] = U + 3                                               # userId (columns: candidate, T history slots, userId, two genres): one of the tail's folded columns
            eng = model.engine
            eng.forward(_cuda(torch, bad), td, torch.empty(B, dtype=torch.float32, device="cuda"),
                        torch.empty(eng.workspace_bytes(B) // 4, dtype=torch.float32, device="cuda"))
            with pytest.raises(ValueError):
                eng.check_ids()
        model.engine.close()
    for tag in ("ts2", "ts4", "auto"):
        np.testing.assert_array_equal(got[tag], got["ts1"])
    assert np.abs(got["ts1"] - got["two"]).max() <= 3e-6
    ref = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    assert np.abs(got["ts1"] - ref).max() <= TIGHT


@pytest.mark.parametrize("T,D,B,min_t", [(5, 10, 4099, "1"), (20, 16, 1000, "12"), (12, 10, 333, "12")])
def test_din_fused_tail_on_raw_rows_for_narrow_embeddings(torch, monkeypatch, T, D, B, min_t):
    """[r5] k_din_fused with emb_dim <= 16: the tail's four embedding columns as raw 64-byte rows (the two column-pair K = 32 blocks of
    k_din_tail's form) instead of 4 x 512-byte folded rows.  Against the folded form of the same kernel (SPRK_DIN_FUSED_UNF=0) and the
    two-launch path <= 3e-6, the fp64 oracle within the bar; missing genre ids and a user without history; DIN.py's own shape (5 slots,
    emb 10) reaches the kernel through SPRK_DIN_FUSED_MIN_T=1 (it is not dispatched there by default: two launches measure the same)."""
    V, U = 1001, 3001
    feats = SY.synth_din(B, T, V, U, seed=500 + T)
    feats["movieGenre1"][::5] = -1
    feats["userGenre1"][1::7] = -1
    feats["userRatedMovies"][3] = 0
    got = {}
    for tag, env in (("raw", {}), ("folded", {"SPRK_DIN_FUSED_UNF": "0"}), ("two", {"SPRK_DIN_FUSED": "0"})):
        for k in ("SPRK_DIN_FUSED_UNF", "SPRK_DIN_FUSED"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("SPRK_DIN_FUSED_MIN_T", min_t)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = M.DIN(seed=70 + T, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        assert model.engine.describe()["kernel"].startswith("k_din_fused<KC=1" if tag != "two" else "k_din_tail")
        got[tag] = model.predict(feats)[:, 0]
        model.engine.close()
    assert np.abs(got["raw"] - got["folded"]).max() <= 3e-6
    assert np.abs(got["raw"] - got["two"]).max() <= 3e-6
    ref = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    assert np.abs(got["raw"] - ref).max() <= TIGHT and ref.std() > 0.01


def test_din_fused_batch_beyond_one_round_of_workgroups(torch, monkeypatch):
    """sprk_forward on a DIN batch of more than 16 rows x 8 waves x CUs (32 768 on an MI355X): the PERSISTENT form of k_din_fused (every
    wave walks several tasks, tables staged once) -- the same bits as the same rows scored in one-round slices, the oracle's values."""
    T, D, V, U, B = 50, 32, 5000, 700, 32768 + 4099
    feats = SY.synth_din(B, T, V, U, seed=431)
    model = M.DIN(seed=97, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    assert model.engine.describe()["kernel"].startswith("k_din_fused")
    ids, dense = model.pack(feats)
    ti, td = _cuda(torch, ids), _cuda(torch, dense)
    full = model.predict_device(ti, td)
    model.engine.check_ids()
    for lo, hi in ((0, 32768), (32768, B), (16384 + 7, 16384 + 7 + 20000)):
        assert torch.equal(model.predict_device(ti[lo:hi].contiguous(), td[lo:hi].contiguous()), full[lo:hi])
    sl = slice(B - 3000, B)
    ref = O.din_forward({k: v[sl] for k, v in feats.items()}, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    assert np.abs(full[sl].cpu().numpy() - ref).max() <= TIGHT
    model.engine.close()


def test_din_attention_kernel_bad_ids_raise(torch):
    """History / candidate ids outside the table: flagged (TF raises InvalidArgumentError), no wild read."""
    T, D, V, U, B = 50, 32, 3000, 500, 257
    feats = SY.synth_din(B, T, V, U, seed=7)
    model = M.DIN(seed=8, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    ids, dense = model.pack(feats)
    eng = model.engine
    for col, val in ((1 + 49, V), (1 + 3, -1), (0, V + 5)):
        bad = ids.copy()
        bad[200, col] = val
        pooled = torch.empty((B, eng.n_aux), dtype=torch.float32, device="cuda")
        eng.din_pool(_cuda(torch, bad), pooled, None)
        with pytest.raises(ValueError):
            eng.check_ids()
    pooled = torch.empty((B, eng.n_aux), dtype=torch.float32, device="cuda")
    eng.din_pool(_cuda(torch, ids), pooled, None)
    eng.check_ids()


# --------------------------------------------------------------------------------------------
# BASELINE configs at (near) full size
# --------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def config2(torch):
    B = 65536
    feats = SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=SY.SEED)
    model = M.DeepFMv2(seed=31, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
    ids, dense = model.pack(feats)
    return model, feats, _cuda(torch, ids), _cuda(torch, dense)


def test_config2_deepfm_v2_full_batch_vs_oracle(torch, config2):
    model, feats, ids, dense = config2
    p = model.predict_device(ids, dense).cpu().numpy()
    model.engine.check_ids()
    ref = O.deepfm_v2_forward(feats, model.weights, dtype=np.float64, fields=SY.CONFIG2_FIELDS,
                              order=[k for k, _, _ in SY.CONFIG2_FIELDS])[:, 0]
    ref32 = O.deepfm_v2_forward(feats, model.weights, dtype=np.float32, fields=SY.CONFIG2_FIELDS,
                                order=[k for k, _, _ in SY.CONFIG2_FIELDS])[:, 0]
    assert np.isfinite(p).all()
    assert np.abs(p - ref32).max() <= TOL
    assert np.abs(p - ref).max() <= TOL
    assert 0.05 < ref.std()                     # the comparison is not vacuous (scores are spread out)


def test_config2_size_independent_properties(torch, config2):
    model, feats, ids, dense = config2
    p = model.predict_device(ids, dense)
    # determinism
    assert torch.equal(p, model.predict_device(ids, dense))
    # samples are independent: permuting rows permutes scores, bit for bit
    perm = torch.randperm(ids.shape[0], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    assert torch.equal(model.predict_device(ids[perm].contiguous(), dense[perm].contiguous()), p[perm])
    # ... and a slice scored alone equals the slice of the full batch (ragged last tile included)
    for lo, hi in ((0, 1000), (777, 40001), (65000, 65536)):
        assert torch.equal(model.predict_device(ids[lo:hi].contiguous(), dense[lo:hi].contiguous()), p[lo:hi])


def test_config2_pairwise_variant(torch):
    B = 16384
    feats = SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=5, dist="zipf")
    model = M.DeepFM(seed=32, emb_dim=16, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)
    p = model.predict(feats)[:, 0]
    ref = O.deepfm_forward(feats, model.weights, dtype=np.float64, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)[:, 0]
    assert np.abs(p - ref).max() <= TOL


def test_config3_din_t50_d32(torch):
    B, T, D = 8192, 50, 32
    feats = SY.synth_din(B, T, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=11)
    model = M.DIN(seed=33, emb_dim=D, hist_len=T, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
    p = model.predict(feats)[:, 0]
    ref, parts = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=SY.ML20M_MOVIE_IDS,
                               user_buckets=SY.ML20M_USER_IDS, return_parts=True)
    assert np.abs(p - ref[:, 0]).max() <= TOL
    ids, dense = model.pack(feats)
    ti, td = _cuda(torch, ids), _cuda(torch, dense)
    full = model.predict_device(ti, td)
    assert torch.equal(full[1000:5003], model.predict_device(ti[1000:5003].contiguous(), td[1000:5003].contiguous()))
    assert 0.2 < parts["att"].mean() < 0.8


def test_config4_deepfm_d64_wide_rows(torch):
    B = 8192
    fields = [("movieId", "id", SY.ML20M_MOVIE_IDS), ("userId", "id", 1_000_000),
              ("userGenre1", "genre", 19), ("movieGenre1", "genre", 19)]
    feats = SY.synth_fields(B, fields, seed=12)
    model = M.DeepFMv2(seed=34, emb_dim=64, fields=fields, proj_dim=16)
    p = model.predict(feats)[:, 0]
    ref = O.deepfm_v2_forward(feats, model.weights, dtype=np.float64, fields=fields, order=[k for k, _, _ in fields])[:, 0]
    assert np.abs(p - ref).max() <= TOL


def test_config5_wide_deep_hashed_cross(torch):
    B = 16384
    feats = SY.synth_embedding_mlp(B, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=13, rated_vocab=SY.ML20M_MOVIE_IDS)
    for kw in (dict(cross_buckets=10000, cross_dim=0), dict(cross_buckets=1_000_000, cross_dim=32)):
        model = M.WideNDeep(seed=35, emb_dim=32, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS, **kw)
        p = model.predict(feats)[:, 0]
        ref = O.wide_n_deep_forward(feats, model.weights, dtype=np.float64, movie_buckets=SY.ML20M_MOVIE_IDS,
                                    user_buckets=SY.ML20M_USER_IDS, cross_buckets=kw["cross_buckets"],
                                    rated_buckets=SY.ML20M_MOVIE_IDS)[:, 0]
        assert np.abs(p - ref).max() <= TOL


def test_concurrent_streams_share_a_handle(torch, config2):
    model, feats, ids, dense = config2
    ref = model.predict_device(ids, dense)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1 = torch.empty_like(ref)
    o2 = torch.empty_like(ref)
    for _ in range(5):
        with torch.cuda.stream(s1):
            model.predict_device(ids, dense, out=o1)
        with torch.cuda.stream(s2):
            model.predict_device(ids, dense, out=o2)
    torch.cuda.synchronize()
    assert torch.equal(o1, ref) and torch.equal(o2, ref)


# --------------------------------------------------------------------------------------------
# the execution paths of the DeepFM_v2 graph must agree with the oracle and with each other ([r6] k_deepfm_v2_chain is retired: a model the
# joint kernels refuse -- here by switch -- goes to k_rows_chain, then to the interpreter)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("env", [{}, {"SPRK_V2_HALF": "0"}, {"SPRK_V2_JOINT": "0"}, {"SPRK_V2_FOLD": "0"}, {"SPRK_FORCE_INTERPRETER": "1"}],
                         ids=["joint-split-f16", "joint-f32", "no-joint-form", "no-folded-form", "interpreter"])
def test_deepfm_v2_execution_paths(torch, env, monkeypatch):
    for k in ("SPRK_V2_FOLD", "SPRK_V2_REG", "SPRK_V2_JOINT", "SPRK_V2_HALF", "SPRK_FORCE_INTERPRETER"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B = 10007                                     # ragged: not a multiple of 16 or 64
    order = [k for k, _, _ in SY.CONFIG2_FIELDS]
    feats = SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=77, dist="zipf")
    model = M.DeepFMv2(seed=41, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
    p = model.predict(feats)[:, 0]                # engine is created here, under the env above
    ref = O.deepfm_v2_forward(feats, model.weights, dtype=np.float64, fields=SY.CONFIG2_FIELDS, order=order)[:, 0]
    assert np.abs(p - ref).max() <= TIGHT
    # 4-field D=16 and D=64 shapes (other template instantiations / interpreter)
    for D, vocab_u in ((16, 50000), (64, 200000)):
        fields = [("movieId", "id", 3000), ("userId", "id", vocab_u), ("userGenre1", "genre", 19), ("movieGenre1", "genre", 19)]
        f2 = SY.synth_fields(3001, fields, seed=78)
        m2 = M.DeepFMv2(seed=42, emb_dim=D, fields=fields, proj_dim=16)
        p2 = m2.predict(f2)[:, 0]
        r2 = O.deepfm_v2_forward(f2, m2.weights, dtype=np.float64, fields=fields, order=[k for k, _, _ in fields])[:, 0]
        assert np.abs(p2 - r2).max() <= TIGHT


def test_deepfm_v2_folded_tables_do_not_change_scores(torch, monkeypatch):
    """Folding the per-field Dense projections into the tables at finalize (k_v2_fold) must give the
    scores of the per-sample projection path (the projected rows are bit-equal; the two kernels sum deep0 in different orders)."""
    B = 20000
    feats = SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=91)
    outs = []
    for fold, joint in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("SPRK_V2_FOLD", fold)
        monkeypatch.setenv("SPRK_V2_JOINT", joint)
        monkeypatch.delenv("SPRK_FORCE_INTERPRETER", raising=False)
        model = M.DeepFMv2(seed=43, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
        outs.append(model.predict(feats)[:, 0])
    assert np.abs(outs[0] - outs[2]).max() <= TIGHT
    assert np.abs(outs[1] - outs[2]).max() <= TIGHT


@pytest.mark.parametrize("fields", [
    [("movieId", "id", 3000), ("userGenre1", "genre", 19)],
    [("movieId", "id", 3000), ("userGenre1", "genre", 19), ("userGenre2", "genre", 19), ("movieGenre1", "genre", 19)],
    [("movieId", "id", 3000), ("userId", "id", 9000), ("userGenre1", "genre", 19)],
    [("movieId", "id", 3000), ("userId", "id", 9000), ("userRatedMovie1", "id", 3000), ("userGenre1", "genre", 19), ("userGenre2", "genre", 19)],
], ids=["1big+1small", "1big+3small", "2big+1small", "3big+2small"])
def test_deepfm_v2_joint_field_splits(torch, monkeypatch, fields):
    """Other big/small field splits of k_deepfm_v2_joint (joint table over 1..3 small-vocabulary fields),
    missing ids (-1) included, against the fp64 oracle and the per-field path."""
    B = 5003
    feats = SY.synth_fields(B, fields, seed=55)
    order = [k for k, _, _ in fields]
    outs = []
    for joint, half in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("SPRK_V2_JOINT", joint)
        monkeypatch.setenv("SPRK_V2_HALF", half)
        model = M.DeepFMv2(seed=44, emb_dim=16, fields=fields, proj_dim=16)
        outs.append(model.predict(feats)[:, 0])
    ref = O.deepfm_v2_forward(feats, model.weights, dtype=np.float64, fields=fields, order=order)[:, 0]
    for o in outs:
        assert np.abs(o - ref).max() <= TIGHT


@pytest.mark.parametrize("fields", [
    SY.CONFIG2_FIELDS,
    [("movieId", "id", 3000), ("userId", "id", 9000), ("userGenre1", "genre", 19), ("movieGenre1", "genre", 19)],
    [("movieId", "id", 3000), ("userGenre1", "genre", 19), ("userGenre2", "genre", 19), ("movieGenre1", "genre", 19)],
], ids=["config2", "2big+2small", "1big+3small"])
def test_v2_joint1_bit_identical_to_joint(torch, monkeypatch, fields):
    """k_deepfm_v2_joint1 (one task per wave, four waves per SIMD: the strict one-batch launch, k_chain_v2j1.h) against
    k_deepfm_v2_joint (SPRK_V2J_ONE=0): same arithmetic in the same order => the same bits; ragged tails, a batch of one
    partial task, missing ids, unaligned buffers (a row-offset view), and the oracle."""
    order = [k for k, _, _ in fields]
    for B in (65536, 5003, 16, 7, 1):
        feats = SY.synth_fields(B, fields, seed=77 + B)
        outs = []
        # (joint1 with its weight fragments read behind the gathers, then in front of them -- the form finalize picks for tables larger
        # than the Infinity Cache --, then the looped kernel)
        for one, hoist in (("1", "0"), ("1", "1"), ("0", "0")):
            monkeypatch.setenv("SPRK_V2J_ONE", one)
            monkeypatch.setenv("SPRK_V2J1_HOIST", hoist)
            model = M.DeepFMv2(seed=46, emb_dim=16, fields=fields, proj_dim=16)
            ids, dense = model.pack(feats)
            ids_t, dense_t = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
            got = model.predict_device(ids_t, dense_t).cpu().numpy()
            if B > 16:                                                    # rows 1.. of the same buffers: not 16-byte aligned
                tail = model.predict_device(ids_t[1:], dense_t[1:]).cpu().numpy()
                assert np.array_equal(tail, got[1:])
            model.engine.check_ids()
            outs.append(got)
        assert np.array_equal(outs[0], outs[1]), "B=%d: the two forms of joint1 differ by %g" % (B, np.abs(outs[0] - outs[1]).max())
        assert np.array_equal(outs[0], outs[2]), "B=%d: joint1 and joint differ by %g" % (B, np.abs(outs[0] - outs[2]).max())
        n = min(B, 4096)
        ref = O.deepfm_v2_forward({k: v[:n] for k, v in feats.items()}, model.weights, dtype=np.float64, fields=fields, order=order)[:, 0]
        assert np.abs(outs[0][:n] - ref).max() <= TIGHT


def test_deepfm_v2_split_f16_is_fp32_class(torch, monkeypatch):
    """The split-f16 MFMA path (hi + lo halfs, f32 accumulate) must be as close to the fp64 oracle as the
    f32 MFMA path is -- on the pre-sigmoid scale too, with O(1) numerics so that the score is not saturated
    -- and must stay so for tables whose magnitudes are far from 1 (power-of-two operand scaling)."""
    B = 16384
    fields = SY.CONFIG2_FIELDS
    order = [k for k, _, _ in fields]
    feats = SY.synth_fields(B, fields, seed=61)
    for k in ("movieRatingCount", "userRatingCount", "releaseYear"):     # keep the logit in sigmoid's live range
        feats[k] = (np.asarray(feats[k], np.float64) % 7).astype(np.asarray(feats[k]).dtype)
    for table_scale in (1.0, 1.0 / 4096.0, 37.0):
        base = M.DeepFMv2(seed=45, emb_dim=16, fields=fields, proj_dim=16)
        w = dict(base.weights)
        for k, _, _ in fields:
            w["emb/" + k] = (w["emb/" + k] * table_scale).astype(np.float32)
        ref = O.deepfm_v2_forward(feats, w, dtype=np.float64, fields=fields, order=order)[:, 0]
        err = {}
        for half in ("1", "0"):
            monkeypatch.setenv("SPRK_V2_HALF", half)
            p = M.DeepFMv2(weights=w, emb_dim=16, fields=fields, proj_dim=16).predict(feats)[:, 0]
            err[half] = float(np.abs(p - ref).max())
        print("split-f16 vs f32 MFMA, table scale %g: max|err| vs fp64 oracle %.3g (split) %.3g (f32), score std %.3f"
              % (table_scale, err["1"], err["0"], ref.std()))
        assert 0.02 < ref.std()
        assert err["1"] <= TOL and err["0"] <= TOL, (table_scale, err)
        assert err["1"] <= 2 * err["0"] + 2e-6, (table_scale, err)


def test_deepfm_v2_outlier_row_keeps_fp32_class(torch, monkeypatch):
    """One static power-of-two scale per table comes from max |x|: a single huge row next to ordinary ones would push the
    ordinary rows' lo halves into f16 subnormals.  The finalize-time dynamic-range guard (more than 1 in 1024
    non-zero entries over 2^20 below the maximum) must then pick the f32 MFMA variant of the same kernel, and the scores
    of samples that do not touch the outlier must stay fp32-class."""
    B = 8192
    fields = SY.CONFIG2_FIELDS
    order = [k for k, _, _ in fields]
    feats = SY.synth_fields(B, fields, seed=67)
    for k in ("movieRatingCount", "userRatingCount", "releaseYear"):
        feats[k] = (np.asarray(feats[k], np.float64) % 7).astype(np.asarray(feats[k]).dtype)
    big_field = max(fields, key=lambda f: f[1])[0]
    feats[big_field] = np.where(np.asarray(feats[big_field]) == 3, 4, np.asarray(feats[big_field])).astype(np.asarray(feats[big_field]).dtype)
    base = M.DeepFMv2(seed=46, emb_dim=16, fields=fields, proj_dim=16)
    w = dict(base.weights)
    tab = w["emb/" + big_field].copy()
    tab[3] *= np.float32(3.0e8)                                     # the outlier row (id 3 is referenced by nobody)
    w["emb/" + big_field] = tab
    ref = O.deepfm_v2_forward(feats, w, dtype=np.float64, fields=fields, order=order)[:, 0]
    model = M.DeepFMv2(weights=w, emb_dim=16, fields=fields, proj_dim=16)
    p = model.predict(feats)[:, 0]
    d = model.engine.describe()
    print("outlier row: kernel", d.get("kernel"), "max|err|", float(np.abs(p - ref).max()))
    assert "k_deepfm_v2_joint" in d["kernel"] and "f32" in d["kernel"] and "split-f16" not in d["kernel"], d
    assert np.abs(p - ref).max() <= TIGHT
    monkeypatch.setenv("SPRK_HALF_RANGE_GUARD", "0")                  # what the guard prevents
    m2 = M.DeepFMv2(weights=w, emb_dim=16, fields=fields, proj_dim=16)
    p2 = m2.predict(feats)[:, 0]
    assert "split-f16" in m2.engine.describe()["kernel"]
    print("  guard off (split-f16 with subnormal lo halves): max|err| %.3g" % float(np.abs(p2 - ref).max()))
    assert np.abs(p2 - ref).max() > np.abs(p - ref).max()


def test_deepfm_v2_unaligned_views_and_tails(torch, config2):
    """ids/dense views that do not start on a 16-byte boundary take the element-wise staging path;
    every batch length mod 16 exercises the partial last task.  Pure data movement: bit-equal."""
    model, feats, ids, dense = config2
    p = model.predict_device(ids, dense)
    for lo in (1, 2, 3, 5):
        vi, vd = ids[lo:lo + 4099], dense[lo:lo + 4099]          # views, not copies
        assert vi.is_contiguous() and vd.is_contiguous()
        assert torch.equal(model.predict_device(vi, vd), p[lo:lo + 4099])
    for n in range(1, 34):
        assert torch.equal(model.predict_device(ids[:n].contiguous(), dense[:n].contiguous()), p[:n])


def test_forward_many_equals_forward(torch, config2):
    """sprk_forward_many (the predict-over-batches loop in one foreign call) == n calls of sprk_forward."""
    model, feats, ids, dense = config2
    eng = model.engine
    B = 8192
    chunks = [(ids[i * B:(i + 1) * B].contiguous(), dense[i * B:(i + 1) * B].contiguous()) for i in range(4)]
    outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in chunks]
    eng.forward_many([c[0] for c in chunks], [c[1] for c in chunks], outs)
    eng.check_ids()
    ref = model.predict_device(ids, dense)
    for i, o in enumerate(outs):
        assert torch.equal(o, ref[i * B:(i + 1) * B])


# --------------------------------------------------------------------------------------------
# first-Dense fold of the tile interpreter (embedding columns -> per-id tables of the layer's outputs)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["embedding_mlp", "wide_n_deep", "neural_cf", "neural_cf2", "deepfm", "din"])
def test_first_dense_fold_matches_unfolded(torch, samples, monkeypatch, name):
    g = np.load(os.path.join(GOLDEN, "oracle_%s.npz" % name))
    out = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("SPRK_TILE_FOLD", fold)
        out[fold] = make_model(name).predict(samples)[:, 0]
        assert np.abs(out[fold] - g["pred64"]).max() <= TIGHT
    assert np.abs(out["1"] - out["0"]).max() <= TIGHT
    # missing / out-of-vocabulary ids still are zero rows when the column is folded
    monkeypatch.setenv("SPRK_TILE_FOLD", "1")
    if name in ("embedding_mlp", "wide_n_deep"):
        s2 = dict(samples)
        s2["userGenre2"] = np.array([""] * 256, dtype=object)
        p_missing = make_model(name).predict(s2)[:, 0]
        monkeypatch.setenv("SPRK_TILE_FOLD", "0")
        np.testing.assert_allclose(make_model(name).predict(s2)[:, 0], p_missing, atol=TIGHT)


@pytest.mark.parametrize("T,D,B", [(50, 32, 4099), (5, 10, 777), (20, 16, 1), (50, 24, 33)])
def test_din_tail_paths(torch, monkeypatch, T, D, B):
    """The three ways the DIN tail (DIN.py:161-167) runs -- k_din_tail (register chained, folded embedding
    columns), the interpreter with the first-Dense fold, the plain interpreter -- against the fp64 oracle and
    each other, missing genres (-1) included, ragged batch sizes."""
    V, U = SY.ML20M_MOVIE_IDS if D == 32 else 4000, SY.ML20M_USER_IDS if D == 32 else 900
    feats = SY.synth_din(B, T, V, U, seed=21)
    out = {}
    for tag, env in (("fused", {"SPRK_TILE_FOLD": "1", "SPRK_DIN_TAIL": "1", "SPRK_DYN_F16": "1", "SPRK_DIN_FUSED": "1"}),     # k_din_fused (one launch)
                     ("chain", {"SPRK_TILE_FOLD": "1", "SPRK_DIN_TAIL": "1", "SPRK_DYN_F16": "1", "SPRK_DIN_FUSED": "0"}),     # attention -> pooled -> k_din_tail
                     ("chain_f32", {"SPRK_TILE_FOLD": "1", "SPRK_DIN_TAIL": "1", "SPRK_DYN_F16": "0"}),
                     ("fold", {"SPRK_TILE_FOLD": "1", "SPRK_DIN_TAIL": "0"}),
                     ("plain", {"SPRK_TILE_FOLD": "0", "SPRK_DIN_TAIL": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = M.DIN(seed=35, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        out[tag] = model.predict(feats)[:, 0]
    ref = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    for tag in out:
        assert np.abs(out[tag] - ref).max() <= TOL, tag
    assert np.abs(out["chain"] - out["plain"]).max() <= TIGHT
    assert np.abs(out["fused"] - out["plain"]).max() <= TIGHT
    assert np.abs(out["fused"] - out["chain"]).max() <= 3e-6
    assert np.abs(out["chain_f32"] - out["plain"]).max() <= TIGHT
    assert np.abs(out["fold"] - out["plain"]).max() <= TIGHT


@pytest.mark.parametrize("scale", [1e-4, 1.0, 300.0])
def test_din_tail_dynamic_f16_wide_activation_range(torch, monkeypatch, scale):
    """fc1 of k_din_tail runs on the f16 matrix pipe with a PER-SAMPLE power-of-two scale taken from the sample's
    largest hidden activation (dyn_split.h).  Stress it where a static scale would overflow or flush: embedding
    tables and numerics blown up / shrunk by `scale`, so the hidden activations span many binades across samples;
    the logit before the sigmoid is the quantity compared (a saturated sigmoid would hide errors)."""
    T, D, B = 20, 16, 2051
    V, U = 4000, 900
    feats = SY.synth_din(B, T, V, U, seed=23)
    rng = np.random.default_rng(5)
    # per-sample spread on top of the global scale: numerics over 6 decades
    feats = dict(feats)
    spread = (10.0 ** rng.uniform(-3, 3, size=B)).astype(np.float32)
    for k in list(feats):
        if np.asarray(feats[k]).dtype.kind == "f":
            feats[k] = (np.asarray(feats[k], dtype=np.float32) * spread).astype(np.float32)
    out = {}
    for tag in ("1", "0"):
        monkeypatch.setenv("SPRK_DYN_F16", tag)
        w0 = M.DIN(seed=36, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U).weights
        w0 = {k: (np.asarray(w) * np.float32(scale) if k.startswith("emb/") else np.asarray(w)) for k, w in w0.items()}
        model = M.DIN(weights=w0, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        out[tag] = model.predict(feats)[:, 0]
    ref = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    assert np.isfinite(out["1"]).all()
    e1, e0 = np.abs(out["1"] - ref).max(), np.abs(out["0"] - ref).max()
    print("scale %g: dyn-f16 err %.3e, f32-MFMA err %.3e" % (scale, e1, e0))
    if scale <= 1.0:                       # at 300x the logits reach 1e5: fp32 itself cannot hold 1e-4 on the sigmoid
        assert e1 <= TOL and e0 <= TOL
    # the f16-split path is no worse than twice the f32-MFMA path (+ fp32 rounding of the sigmoid)
    assert e1 <= 2 * e0 + 2e-6


# --------------------------------------------------------------------------------------------
# k_deepfm_pairs: the pairwise-dot DeepFM graph (DeepFM.py) as a register-chained kernel
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tied", [False, True], ids=["own-deep-tables", "tied-tables"])
@pytest.mark.parametrize("shape", ["reference", "config2", "config2-zipf-ragged"])
def test_deepfm_pairs_kernel_vs_interpreter_and_oracle(torch, monkeypatch, shape, tied):
    """``tied`` = False is the reference's graph (the deep part owns its movieId / userId tables, DeepFM.py:106: rows packed in
    the field's line at emb_dim 10, rows of their own at emb_dim 16); True shares one table per key.  Each through the
    one-task-per-wave kernel (the one-batch launch), the looped kernel (SPRK_V1_ONE=0), f32 MFMA, and the interpreter."""
    if shape == "reference":
        fields, pairs, D, B, dist = None, None, 10, 2049, "uniform"       # DeepFM.py literals: 4 fields, 4 pairs, emb_dim 10
        feats = SY.synth_fields(B, M._default_fields(), seed=71)
    else:
        fields, pairs, D = SY.CONFIG2_FIELDS, SY.CONFIG2_PAIRS, 16
        B, dist = (16384, "uniform") if shape == "config2" else (10007, "zipf")
        feats = SY.synth_fields(B, fields, seed=72, dist=dist)
    out = {}
    for chain, dyn, one in (("1", "1", "1"), ("1loop", "1", "0"), ("1f32", "0", "1"), ("0", "1", "1")):   # pairs1 / looped / f32 MFMA / interpreter
        monkeypatch.setenv("SPRK_V1_CHAIN", chain[0])
        monkeypatch.setenv("SPRK_DYN_F16", dyn)
        monkeypatch.setenv("SPRK_V1_ONE", one)
        model = M.DeepFM(seed=46, emb_dim=D, fields=fields, pairs=pairs, share_deep_tables=tied)
        if chain != "0":
            assert model.engine.describe()["kernel"].startswith("k_deepfm_pairs")
        out[chain] = model.predict(feats)[:, 0]
    assert ("deep_emb/movieId" in model.weights) != tied
    kw = {} if fields is None else {"fields": fields, "pairs": pairs}
    ref = O.deepfm_forward(feats, model.weights, dtype=np.float64, share_deep_tables=tied, **kw)[:, 0]
    assert np.abs(out["1"] - ref).max() <= TIGHT
    assert np.array_equal(out["1"], out["1loop"]), "one-task and looped kernels differ by %g" % np.abs(out["1"] - out["1loop"]).max()
    assert np.abs(out["1f32"] - ref).max() <= TIGHT
    assert np.abs(out["1"] - ref).max() <= 2 * np.abs(out["1f32"] - ref).max() + 2e-6
    assert np.abs(out["0"] - ref).max() <= TIGHT
    assert 0.02 < ref.std()


@pytest.mark.parametrize("B", [5003, 17, 70001])
def test_deepfm_pairs_wide_rows_emb_dim_64(torch, monkeypatch, B):
    """BASELINE config 4's pair-dot graph (DeepFM.py:100-103 at emb_dim 64): 256-byte rows, four 16-byte pieces per lane,
    deep0's K = 128 embedding columns as four split-f16 K blocks -- against the fp64 oracle and the plan interpreter."""
    fields = [("movieId", "id", 50000), ("userId", "id", 30000), ("userGenre1", "genre", 19), ("movieGenre1", "genre", 19)]
    feats = SY.synth_fields(B, fields, seed=77, missing=0.1)
    model = M.DeepFM(seed=41, emb_dim=64, fields=fields)
    assert model.engine.describe()["kernel"].startswith("k_deepfm_pairs<NF=4,NV=16>")
    p = model.predict(feats)[:, 0]
    n = min(B, 8192)
    ref = O.deepfm_forward({k: v[:n] for k, v in feats.items()}, model.weights, dtype=np.float64, fields=fields, pairs=model.pairs)[:, 0]
    assert np.abs(p[:n] - ref).max() <= TIGHT and (B < 100 or ref.std() > 0.02)
    for env in ({"SPRK_DYN_F16": "0"}, {"SPRK_V1_CHAIN": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = M.DeepFM(weights=model.weights, emb_dim=64, fields=fields)
        q = other.predict(feats)[:, 0]
        for k in env:
            monkeypatch.delenv(k)
        assert np.abs(p - q).max() <= TIGHT, env


def test_deepfm_pairs_kernel_properties(torch):
    """Determinism, slice invariance (ragged tails), missing ids = zero rows, out-of-range ids raise."""
    B = 5000
    feats = SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=73)
    model = M.DeepFM(seed=47, emb_dim=16, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)
    ids, dense = model.pack(feats)
    ti, td = _cuda(torch, ids), _cuda(torch, dense)
    p = model.predict_device(ti, td)
    assert torch.equal(p, model.predict_device(ti, td))
    for lo, hi in ((0, 1), (3, 20), (100, 1133), (4990, 5000)):
        assert torch.equal(model.predict_device(ti[lo:hi].contiguous(), td[lo:hi].contiguous()), p[lo:hi])
    # a missing genre (-1) must equal a zero embedding row and zero first-order weight
    col = [k for k, _, _ in SY.CONFIG2_FIELDS].index("userGenre1")
    ids2 = ids.copy()
    ids2[:, col] = -1
    w = dict(model.weights)
    w["emb/userGenre1"] = np.zeros_like(w["emb/userGenre1"])
    fo = M.first_order_offsets(SY.CONFIG2_FIELDS)
    hk = w["head/kernel"].copy()
    hk[fo["userGenre1"]:fo["userGenre1"] + 19] = 0
    w["head/kernel"] = hk
    zeroed = M.DeepFM(weights=w, emb_dim=16, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)
    got = model.predict_device(_cuda(torch, ids2), td).cpu().numpy()
    want = zeroed.predict_device(ti, td).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-6
    bad = ids.copy()
    bad[17, 0] = SY.ML20M_MOVIE_IDS + 3
    model.predict_device(_cuda(torch, bad), td)
    with pytest.raises(ValueError):
        model.engine.check_ids()


def test_forward_many_fan_out_over_streams(torch, monkeypatch):
    """sprk_set_many_streams(2): sprk_forward_many alternates independent batches over two helper streams (forked from and
    joined back into the caller's stream); results equal the strictly ordered run bit for bit."""
    B, n = 4099, 7
    feats = [SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=80 + i) for i in range(n)]
    res = {}
    for streams in ("0", "2"):
        model = M.DeepFMv2(seed=48, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
        assert model.engine.set_many_streams(int(streams))
        packed = [model.pack(f) for f in feats]
        ids = [_cuda(torch, p[0]) for p in packed]
        dense = [_cuda(torch, p[1]) for p in packed]
        outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        model.engine.forward_many(ids, dense, outs)
        torch.cuda.current_stream().synchronize()            # the join makes the caller's stream wait for every helper stream
        model.engine.check_ids()
        res[streams] = [o.cpu().numpy() for o in outs]
    for a, b in zip(res["0"], res["2"]):
        np.testing.assert_array_equal(a, b)
    # DIN (attention kernel -> pooled vectors in the workspace -> tail kernel): one workspace slice per stream
    Bd, T = 2049, 50
    din = M.DIN(seed=49, emb_dim=32, hist_len=T, movie_buckets=5000, user_buckets=7000)
    eng = din.engine
    fd = [SY.synth_din(Bd, T, 5000, 7000, seed=90 + i) for i in range(5)]
    packed = [din.pack(f) for f in fd]
    ids = [_cuda(torch, p[0]) for p in packed]
    dense = [_cuda(torch, p[1]) for p in packed]
    res = {}
    for streams in (0, 2):
        eng.set_many_streams(streams)
        ws = torch.empty(eng.many_workspace_bytes(Bd, max(streams, 1)) // 4, dtype=torch.float32, device="cuda")
        outs = [torch.full((Bd,), -1.0, dtype=torch.float32, device="cuda") for _ in range(5)]
        eng.forward_many(ids, dense, outs, ws)
        torch.cuda.current_stream().synchronize()
        eng.check_ids()
        res[streams] = [o.cpu().numpy() for o in outs]
    for a, b in zip(res[0], res[2]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("B,n,k", [(4099, 7, 4), (65536, 5, 16), (1000, 17, 16), (16, 3, 2), (33, 70, 64), (2048, 2, 16)])
def test_forward_many_several_batches_per_launch(torch, B, n, k):
    """sprk_set_many_batches(k): one launch of the fused DeepFM_v2 kernel scores up to k batches (own buffers each); scores
    equal the launch-per-batch run bit for bit, ragged B and a ragged last launch included; unaligned buffers and other
    models fall back to launch-per-batch."""
    feats = [SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=180 + i, dist="zipf" if i % 2 else "uniform") for i in range(n)]
    model = M.DeepFMv2(seed=58, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
    eng = model.engine
    packed = [model.pack(f) for f in feats]
    ids = [_cuda(torch, p[0]) for p in packed]
    dense = [_cuda(torch, p[1]) for p in packed]
    res = {}
    for kk in (1, k):
        assert eng.set_many_batches(kk)
        outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        eng.forward_many(ids, dense, outs)
        torch.cuda.synchronize()
        eng.check_ids()
        res[kk] = [o.cpu().numpy() for o in outs]
    for a, b in zip(res[1], res[k]):
        np.testing.assert_array_equal(a, b)
    ref = O.deepfm_v2_forward(feats[-1], model.weights, dtype=np.float64, fields=SY.CONFIG2_FIELDS,
                              order=[f for f, _, _ in SY.CONFIG2_FIELDS])[:, 0]
    assert np.abs(res[k][-1] - ref).max() <= TIGHT
    # an out-of-range id in a later batch of a launch is still flagged
    bad = packed[-1][0].copy()
    bad[B // 2, 0] = 10 ** 6
    eng.forward_many(ids[:-1] + [_cuda(torch, bad)], dense, [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(n)])
    with pytest.raises(ValueError):
        eng.check_ids()
    # unaligned buffers: served launch by launch, same scores
    if B >= 33:
        big_i = torch.empty(B * 6 + 3, dtype=torch.int32, device="cuda")
        off_ids = [big_i[1:1 + B * 6].view(B, 6).copy_(i) for i in ids[:1]] + ids[1:]
        outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        eng.forward_many(off_ids, dense, outs)
        torch.cuda.synchronize()
        for a, o in zip(res[1], outs):
            np.testing.assert_array_equal(a, o.cpu().numpy())
    with pytest.raises(ValueError):
        eng.set_many_batches(65)
    # the C ABI keeps the setters as the DEFAULTS of sprk_forward_many; the per-call form refuses a bad shape itself
    import ctypes as C
    ids_a, dense_a, out_a = eng._many_arrays(ids[:2], dense[:2], outs[:2])
    with pytest.raises(L.SparrowHipError):
        L.check(eng.lib.sprk_forward_many_opts(eng.handle, 2, ids_a, dense_a, out_a, B, None, 0, None, 65, 0))
    L.check(eng.lib.sprk_set_many_batches(eng.handle, 2))
    L.check(eng.lib.sprk_forward_many(eng.handle, 2, ids_a, dense_a, out_a, B, None, 0, None))
    torch.cuda.synchronize()
    for a, o in zip(res[1][:2], outs[:2]):
        np.testing.assert_array_equal(a, o.cpu().numpy())


@pytest.mark.parametrize("shape,B,n,k", [("config2", 4099, 7, 4), ("reference", 1000, 19, 16), ("config2", 16, 3, 2)])
def test_forward_many_several_batches_per_launch_pairs(torch, shape, B, n, k):
    """k_deepfm_pairs under sprk_set_many_batches(k): bit-identical to a launch per batch."""
    if shape == "reference":
        fields, pairs, D = None, None, 10
        feats = [SY.synth_fields(B, M._default_fields(), seed=270 + i) for i in range(n)]
    else:
        fields, pairs, D = SY.CONFIG2_FIELDS, SY.CONFIG2_PAIRS, 16
        feats = [SY.synth_fields(B, fields, seed=270 + i, dist="zipf" if i % 2 else "uniform") for i in range(n)]
    model = M.DeepFM(seed=47, emb_dim=D, fields=fields, pairs=pairs)
    eng = model.engine
    packed = [model.pack(f) for f in feats]
    ids = [_cuda(torch, p[0]) for p in packed]
    dense = [_cuda(torch, p[1]) for p in packed]
    res = {}
    for kk in (1, k):
        eng.set_many_batches(kk)
        outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        eng.forward_many(ids, dense, outs)
        torch.cuda.synchronize()
        eng.check_ids()
        res[kk] = [o.cpu().numpy() for o in outs]
    for a, b in zip(res[1], res[k]):
        np.testing.assert_array_equal(a, b)
    kw = {} if fields is None else {"fields": fields, "pairs": pairs}
    ref = O.deepfm_forward(feats[-1], model.weights, dtype=np.float64, **kw)[:, 0]
    assert np.abs(res[k][-1] - ref).max() <= TIGHT


@pytest.mark.parametrize("mode", ["two_launch", "fused"])
@pytest.mark.parametrize("Bd,n,k,streams", [(2049, 5, 4, 0), (4096, 9, 16, 0), (1000, 7, 3, 2), (16, 4, 2, 2)])
def test_forward_many_several_batches_per_launch_din(torch, monkeypatch, Bd, n, k, streams, mode):
    """DIN with sprk_set_many_batches(k), bit-identical to a launch per batch in both forms: two_launch = the group's attention launch,
    then ONE k_din_tail launch for the k batches (a workspace slice per batch of the group; groups alternate over the helper streams
    when slices allow) against attention + tail per batch (SPRK_DIN_FUSED=0); fused = k_din_fused<MB> walking the k batches' tasks as
    one grid (SPRK_DIN_FUSED_MB=1) against one k_din_fused launch per batch -- other (task, time slice) shapes, same bits."""
    monkeypatch.setenv("SPRK_DIN_FUSED", "0" if mode == "two_launch" else "1")
    monkeypatch.setenv("SPRK_DIN_FUSED_MB", "0" if mode == "two_launch" else "1")
    T = 50
    din = M.DIN(seed=59, emb_dim=32, hist_len=T, movie_buckets=5000, user_buckets=7000)
    eng = din.engine
    fd = [SY.synth_din(Bd, T, 5000, 7000, seed=190 + i) for i in range(n)]
    packed = [din.pack(f) for f in fd]
    ids = [_cuda(torch, p[0]) for p in packed]
    dense = [_cuda(torch, p[1]) for p in packed]
    res = {}
    for kk in (1, k):
        eng.set_many_batches(kk)
        eng.set_many_streams(streams)
        ws = torch.empty(eng.many_workspace_bytes(Bd, max(streams, 1) * kk) // 4, dtype=torch.float32, device="cuda")
        outs = [torch.full((Bd,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        eng.forward_many(ids, dense, outs, ws)
        torch.cuda.synchronize()
        eng.check_ids()
        res[kk] = [o.cpu().numpy() for o in outs]
    for a, b in zip(res[1], res[k]):
        np.testing.assert_array_equal(a, b)
    ref = O.din_forward(fd[-1], din.weights, dtype=np.float64, hist_len=T, movie_buckets=5000, user_buckets=7000)[:, 0]
    assert np.abs(res[k][-1] - ref).max() <= TOL
    # a workspace with room for one slice only: served batch by batch, same scores
    eng.set_many_batches(k)
    eng.set_many_streams(0)
    ws1 = torch.empty(eng.many_workspace_bytes(Bd, 1) // 4, dtype=torch.float32, device="cuda")
    outs = [torch.full((Bd,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    eng.forward_many(ids, dense, outs, ws1)
    torch.cuda.synchronize()
    for a, o in zip(res[1], outs):
        np.testing.assert_array_equal(a, o.cpu().numpy())


# --------------------------------------------------------------------------------------------
# k_mlp_rows: EmbeddingMLP / Wide&Deep graphs as a register-chained kernel
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["embedding_mlp", "wide_indicator", "wide_cross_rows"])
def test_mlp_chain_vs_interpreter_and_oracle(torch, monkeypatch, kind):
    B = 6007                                                   # ragged
    V, U = 20000, 30000
    feats = SY.synth_embedding_mlp(B, V, U, seed=91, rated_vocab=V if kind != "embedding_mlp" else None)
    out = {}
    for chain, dyn in (("1", "1"), ("1f32", "0"), ("0", "1")):   # fused kernel with layer 2 on split-f16 / f32 MFMA; interpreter
        monkeypatch.setenv("SPRK_MLP_CHAIN", chain[0])
        monkeypatch.setenv("SPRK_DYN_F16", dyn)
        if kind == "embedding_mlp":
            model = M.EmbeddingMLP(seed=51, emb_dim=32, movie_buckets=V, user_buckets=U)
        else:
            model = M.WideNDeep(seed=52, emb_dim=32, movie_buckets=V, user_buckets=U,
                                **(dict(cross_buckets=10000, cross_dim=0) if kind == "wide_indicator" else dict(cross_buckets=200000, cross_dim=32)))
        out[chain] = model.predict(feats)[:, 0]
    if kind == "embedding_mlp":
        ref = O.embedding_mlp_forward(feats, model.weights, dtype=np.float64, movie_buckets=V, user_buckets=U)[:, 0]
    else:
        ref = O.wide_n_deep_forward(feats, model.weights, dtype=np.float64, movie_buckets=V, user_buckets=U,
                                    cross_buckets=model.cross_buckets, rated_buckets=V)[:, 0]
    assert np.abs(out["1"] - ref).max() <= TIGHT
    assert np.abs(out["1f32"] - ref).max() <= TIGHT
    assert np.abs(out["1"] - ref).max() <= 2 * np.abs(out["1f32"] - ref).max() + 2e-6
    assert np.abs(out["0"] - ref).max() <= TIGHT
    assert 0.02 < ref.std()


@pytest.mark.parametrize("kind,B", [("embedding_mlp_ref", 4099), ("embedding_mlp", 131072), ("wide_indicator", 6007), ("wide_cross_rows", 1), ("wide_cross_rows", 33)])
def test_mlp_rows_kernel_vs_interpreter_and_oracle(torch, monkeypatch, kind, B):
    """k_mlp_rows (every embedding column folded through the first Dense, genre tables in LDS, gathers a task ahead)
    against the plan interpreter (SPRK_MLP_CHAIN=0) and the fp64 oracle; missing genre / history ids, ragged sizes,
    a batch large enough that every wave loops over several tasks, unaligned views."""
    V, U = (1001, 30001) if kind == "embedding_mlp_ref" else (20000, 30000)
    D = 10 if kind == "embedding_mlp_ref" else 32
    feats = SY.synth_embedding_mlp(B, V, U, seed=93, rated_vocab=V if kind.startswith("wide") else None)

    def make():
        if kind.startswith("embedding_mlp"):
            return M.EmbeddingMLP(seed=51, emb_dim=D, movie_buckets=V, user_buckets=U)
        return M.WideNDeep(seed=52, emb_dim=D, movie_buckets=V, user_buckets=U,
                           **(dict(cross_buckets=10000, cross_dim=0) if kind == "wide_indicator" else dict(cross_buckets=200000, cross_dim=32)))
    model = make()
    assert model.engine.describe()["kernel"].startswith("k_mlp_rows<8,8,NBIG=2,NSMALL=8>")
    p = model.predict(feats)[:, 0]
    monkeypatch.setenv("SPRK_MLP_CHAIN", "0")
    old = make()
    assert old.engine.describe()["kernel"].startswith("k_tile_forward")
    q = old.predict(feats)[:, 0]
    monkeypatch.delenv("SPRK_MLP_CHAIN")
    n = min(B, 8192)
    sub = {k: v[:n] for k, v in feats.items()}
    if kind.startswith("embedding_mlp"):
        ref = O.embedding_mlp_forward(sub, model.weights, dtype=np.float64, movie_buckets=V, user_buckets=U)[:, 0]
    else:
        ref = O.wide_n_deep_forward(sub, model.weights, dtype=np.float64, movie_buckets=V, user_buckets=U,
                                    cross_buckets=model.cross_buckets, rated_buckets=V)[:, 0]
    assert np.abs(p[:n] - ref).max() <= TIGHT
    assert np.abs(p - q).max() <= TIGHT
    # slices / unaligned row views score the same as inside the batch
    ids, dense = model.pack(feats)
    ti, td = _cuda(torch, ids), _cuda(torch, dense)
    full = model.predict_device(ti, td)
    for lo, hi in ((1, min(B, 40)), (min(B - 1, 3), min(B, 3 + 1000)), (max(0, B - 517), B)):
        if hi > lo:
            assert torch.equal(model.predict_device(ti[lo:hi], td[lo:hi]), full[lo:hi]), (lo, hi)
    # out-of-range id raises, then the engine works again
    if B > 10:
        bad = dict(feats)
        bad["movieId"] = feats["movieId"].copy()
        bad["movieId"][7] = V
        with pytest.raises(ValueError):
            model.predict(bad)
        np.testing.assert_array_equal(model.predict(feats)[:, 0], p)


@pytest.mark.parametrize("kind,B,n,k", [("embedding_mlp_ref", 4099, 5, 4), ("wide_cross_rows", 1000, 19, 16), ("wide_indicator", 16, 3, 2),
                                        ("embedding_mlp", 20000, 3, 64)])
def test_forward_many_several_batches_per_launch_mlp_rows(torch, monkeypatch, kind, B, n, k):
    """k_mlp_rows_many under sprk_set_many_batches(k) (up to 16 batches per launch, own buffers each): bit-identical to a launch per batch --
    ragged B, a ragged last launch, k above the kernel's 16 --, a bad id in a later batch of a launch is flagged, unaligned buffers and
    SPRK_MLP_ROWS_MANY=0 go launch by launch with the same scores."""
    V, U = (1001, 30001) if kind == "embedding_mlp_ref" else (20000, 30000)
    D = 10 if kind == "embedding_mlp_ref" else 32
    feats = [SY.synth_embedding_mlp(B, V, U, seed=400 + i, rated_vocab=V if kind.startswith("wide") else None) for i in range(n)]

    def make():
        if kind.startswith("embedding_mlp"):
            return M.EmbeddingMLP(seed=51, emb_dim=D, movie_buckets=V, user_buckets=U)
        return M.WideNDeep(seed=52, emb_dim=D, movie_buckets=V, user_buckets=U,
                           **(dict(cross_buckets=10000, cross_dim=0) if kind == "wide_indicator" else dict(cross_buckets=200000, cross_dim=32)))
    model = make()
    eng = model.engine
    assert eng.describe()["kernel"].startswith("k_mlp_rows<")
    packed = [model.pack(f) for f in feats]
    ids = [_cuda(torch, p[0]) for p in packed]
    dense = [_cuda(torch, p[1]) for p in packed]
    res = {}
    for kk in (1, k):
        eng.set_many_batches(kk)
        outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        eng.forward_many(ids, dense, outs)
        torch.cuda.synchronize()
        eng.check_ids()
        res[kk] = [o.cpu().numpy() for o in outs]
    for a, b in zip(res[1], res[k]):
        np.testing.assert_array_equal(a, b)
    for i in (0, n - 1):
        np.testing.assert_array_equal(res[k][i], model.predict_device(ids[i], dense[i]).cpu().numpy().reshape(-1))
    m = min(B, 4096)
    sub = {kk: v[:m] for kk, v in feats[-1].items()}
    if kind.startswith("embedding_mlp"):
        ref = O.embedding_mlp_forward(sub, model.weights, dtype=np.float64, movie_buckets=V, user_buckets=U)[:, 0]
    else:
        ref = O.wide_n_deep_forward(sub, model.weights, dtype=np.float64, movie_buckets=V, user_buckets=U,
                                    cross_buckets=model.cross_buckets, rated_buckets=V)[:, 0]
    assert np.abs(res[k][-1][:m] - ref).max() <= TIGHT
    # an out-of-range id in a later batch of a launch is still flagged
    bad = packed[-1][0].copy()
    bad[B // 2, [c.key for c in model.id_columns].index("movieId")] = 10 ** 6
    eng.forward_many(ids[:-1] + [_cuda(torch, bad)], dense, [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(n)])
    with pytest.raises(ValueError):
        eng.check_ids()
    # unaligned buffers: launch by launch, same scores
    if B >= 33:
        F = packed[0][0].shape[1]
        big_i = torch.empty(B * F + 3, dtype=torch.int32, device="cuda")
        off_ids = [big_i[1:1 + B * F].view(B, F).copy_(ids[0])] + ids[1:]
        outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        eng.forward_many(off_ids, dense, outs)
        torch.cuda.synchronize()
        for a, o in zip(res[1], outs):
            np.testing.assert_array_equal(a, o.cpu().numpy())
    # the switch: the same scores launch by launch
    monkeypatch.setenv("SPRK_MLP_ROWS_MANY", "0")
    other = make()
    other.engine.set_many_batches(k)
    outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
    other.engine.forward_many(ids, dense, outs)
    torch.cuda.synchronize()
    for a, o in zip(res[1], outs):
        np.testing.assert_array_equal(a, o.cpu().numpy())


# --------------------------------------------------------------------------------------------
# DIEN (DIEN.py:163-259): k_dien_seq (GRU with the Embedding mask -> attention gate -> AUGRU) + the DIN tail kernels
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("D,T,B,holes", [(10, 5, 4099, "tail"), (16, 20, 1000, "anywhere"), (10, 50, 777, "anywhere"),
                                         (10, 1, 65, "tail"), (16, 5, 1, "tail")])
def test_dien_vs_oracle(torch, monkeypatch, D, T, B, holes):
    V, U = 3000, 900
    feats = SY.synth_din(B, T, V, U, seed=300 + T)
    h = feats["userRatedMovies"]
    if holes == "anywhere":
        h[np.random.default_rng(T).random(h.shape) < 0.3] = 0
    h[0] = 0                                                         # no history at all
    if B > 2:
        h[2, :T // 2] = 0                                            # leading holes
    out = {}
    for tail in ("1", "0"):
        monkeypatch.setenv("SPRK_DIN_TAIL", tail)
        model = M.DIEN(seed=61, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        ids, dense = model.pack(feats)
        eng = model.engine
        aux = torch.full((B, eng.n_aux), float("nan"), dtype=torch.float32, device="cuda")
        eng.din_pool(_cuda(torch, ids), aux, None)
        eng.check_ids()
        out[tail] = (aux.cpu().numpy(), model.predict(feats)[:, 0])
    ref, parts = O.dien_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U,
                                return_parts=True)
    for tail in out:
        a, sc = out[tail]
        assert np.abs(a[:, :D] - parts["augru"]).max() <= TIGHT
        assert not a[:, D:].any()
        assert np.abs(sc - ref[:, 0]).max() <= TOL
    assert np.abs(out["1"][1] - out["0"][1]).max() <= TIGHT


@pytest.mark.parametrize("D,T,B", [(10, 5, 65536 + 5), (16, 7, 4099), (10, 3, 15)])
def test_dien_matrix_pipe_stage_equals_the_lane_per_sample_stage(torch, monkeypatch, D, T, B):
    """k_dien_seq_mfma (16 samples per wave, every Dense of the recurrence as four f32 MFMAs) against k_dien_seq (one lane per
    sample, SPRK_DIEN_MFMA=0): same fp32 arithmetic, another association -- the AUGRU state within 2e-6, masked slots, ragged
    tile, bad ids flagged by both."""
    V, U = 3000, 900
    feats = SY.synth_din(B, T, V, U, seed=41 + T)
    h = feats["userRatedMovies"]
    h[np.random.default_rng(T).random(h.shape) < 0.25] = 0
    h[0] = 0
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("SPRK_DIEN_MFMA", sw)
        model = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        mfma = sw == "1" and os.environ.get("SPRK_DYN_F16") != "0"
        assert model.engine.describe()["stage"] == ("k_dien_seq_mfma" if mfma else "k_dien_seq")
        ids, dense = model.pack(feats)
        aux = torch.full((B, model.engine.n_aux), float("nan"), dtype=torch.float32, device="cuda")
        model.engine.din_pool(_cuda(torch, ids), aux, None)
        model.engine.check_ids()
        out[sw] = (aux.cpu().numpy(), model.predict(feats)[:, 0])
        bad = ids.copy()
        bad[B // 2, 2] = V                                           # a history id outside the table
        model.engine.din_pool(_cuda(torch, bad), aux, None)
        with pytest.raises(ValueError):
            model.engine.check_ids()
    assert np.abs(out["1"][0] - out["0"][0]).max() <= 2e-6 and not out["1"][0][:, D:].any()
    assert np.abs(out["1"][1] - out["0"][1]).max() <= 2e-6 and out["1"][1].std() > 0.005
    n = min(B, 4096)
    ref, parts = O.dien_forward({k: v[:n] for k, v in feats.items()}, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V,
                                user_buckets=U, return_parts=True)
    assert np.abs(out["1"][0][:n, :D] - parts["augru"]).max() <= TIGHT


@pytest.mark.parametrize("kind,D,T", [("din", 10, 5), ("din", 16, 12), ("dien", 10, 5), ("din", 32, 9), ("din", 20, 50)])
def test_tail_with_raw_embedding_rows_equals_the_folded_tail(torch, monkeypatch, kind, D, T):
    """k_din_tail UNF (emb_dim <= 32: the embedding columns as raw split-f16 rows, fc0's share of them on the matrix pipe) against
    the folded 512-byte rows (SPRK_TAIL_UNF=0) and the oracle: missing genre ids, ragged batch, several batches per launch."""
    V, U, B = 3000, 900, 20011
    feats = SY.synth_din(B, T, V, U, seed=77 + T)
    feats["userGenre1"][::7] = -1                                   # no id: the all-zero row
    monkeypatch.setenv("SPRK_DIN_FUSED", "0")                       # (k_din_tail itself: the two-launch path)
    monkeypatch.setenv("SPRK_DIEN_FUSED", "0")
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("SPRK_TAIL_UNF", sw)
        cls = M.DIN if kind == "din" else M.DIEN
        model = cls(seed=71, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        k = model.engine.describe()["kernel"]
        # (the A/B sweeps of scripts/r03/7*_env_switches*.sh run this file under switches that take the fused tail away altogether)
        fused_tail = (os.environ.get("SPRK_DIN_TAIL") != "0" and os.environ.get("SPRK_TILE_FOLD") != "0" and
                      os.environ.get("SPRK_FORCE_INTERPRETER") != "1")
        want_unf = sw == "1" and os.environ.get("SPRK_TAIL_POOLED_F16") != "0" and os.environ.get("SPRK_DYN_F16") != "0"
        if fused_tail:
            assert k.startswith("k_din_tail<") and (",UNF>" in k) == want_unf, k   # (UNF is set up together with the pooled fragments)
        out[sw] = model.predict(feats)[:, 0]
        if sw == "1":
            ids, dense = model.pack(feats)
            ti, td = _cuda(torch, ids), _cuda(torch, dense)
            full = model.predict_device(ti, td)
            assert torch.equal(model.predict_device(ti[5:1029], td[5:1029]), full[5:1029])
            bad = dict(feats)
            bad["userId"] = feats["userId"].copy()
            bad["userId"][11] = U                                    # outside userId's buckets
            with pytest.raises(ValueError):
                model.predict(bad)
            np.testing.assert_array_equal(model.predict(feats)[:, 0], out[sw])
    fwd = O.din_forward if kind == "din" else O.dien_forward
    n = 4096
    ref = fwd({k: v[:n] for k, v in feats.items()}, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    assert np.abs(out["1"] - out["0"]).max() <= 3e-6 and np.abs(out["1"][:n] - ref).max() <= TOL and out["1"].std() > 0.005


@pytest.mark.parametrize("D,T,B", [(10, 5, 65536 + 21), (16, 20, 4099), (10, 1, 65), (16, 3, 1), (10, 50, 777)])
def test_dien_in_one_launch_equals_the_two_launches(torch, monkeypatch, D, T, B):
    """[r5] k_dien_fused (the recurrence, then k_din_tail's register chain as the same wave's epilogue) against k_dien_seq_mfma -> final
    states in HBM -> k_din_tail (SPRK_DIEN_FUSED=0): the same inlined code on the same operands, so the scores may differ only where the
    compiler contracted a multiply-add differently (<= 1e-6); against the fp64 oracle within the bar.  Masked slots, a user without history,
    missing genre ids, a ragged last tile, more than one tile per wave (B > 65 536), bad ids in the history, the candidate and a tail
    column all flagged."""
    V, U = 3000, 900
    feats = SY.synth_din(B, T, V, U, seed=900 + T)
    h = feats["userRatedMovies"]
    h[np.random.default_rng(T).random(h.shape) < 0.25] = 0
    h[0] = 0
    feats["movieGenre1"][::5] = -1
    out = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("SPRK_DIEN_FUSED", sw)
        model = M.DIEN(seed=65, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
        k = model.engine.describe()["kernel"]
        assert k.startswith("k_dien_fused<D=%d" % D if sw == "1" else "k_din_tail<8,4,1,UNF"), k
        out[sw] = model.predict(feats)[:, 0]
        np.testing.assert_array_equal(model.predict(feats)[:, 0], out[sw])
        if B > 16:
            ids, dense = model.pack(feats)
            ti, td = _cuda(torch, ids), _cuda(torch, dense)
            full = model.predict_device(ti, td)
            assert torch.equal(model.predict_device(ti[5:B - 3], td[5:B - 3]), full[5:B - 3])    # launch-shape invariant
            for key, row, val in (("userRatedMovies", B // 2, V), ("movieId", B - 1, V), ("userId", 11, U), ("userRatedMovies", 3, -2)):
                bad = dict(feats)
                bad[key] = feats[key].copy()
                if bad[key].ndim == 2:
                    bad[key][row, T - 1] = val
                else:
                    bad[key][row] = val
                with pytest.raises(ValueError):
                    model.predict(bad)
            np.testing.assert_array_equal(model.predict(feats)[:, 0], out[sw])
        model.engine.close()
    assert np.abs(out["1"] - out["0"]).max() <= 1e-6, np.abs(out["1"] - out["0"]).max()
    n = min(B, 4096)
    ref = O.dien_forward({k: v[:n] for k, v in feats.items()}, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    assert np.abs(out["1"][:n] - ref).max() <= TOL and (B < 64 or out["1"].std() > 0.005)


@pytest.mark.parametrize("D,T,B", [(16, 7, 65536), (10, 5, 65536 + 21), (16, 20, 20000)])
def test_dien_is_the_same_every_launch_and_the_oracles(torch, monkeypatch, D, T, B):
    """[r5] Both DIEN paths at FULL occupancy (four waves per SIMD), 40 launches each: every launch bit for bit the first, the one-launch kernel
    within 1e-6 of the two launches, and both within the bar of the fp64 oracle on every 8th tile.  Until round 5 k_dien_seq_mfma<16, 32> scored
    ~40 of 4 096 tiles per launch off by up to 6e-5 -- other tiles every launch -- once four of its workgroups shared a CU, and the suite only
    ran emb_dim 16 at batches that leave the CUs a quarter full (k_dien_fused.h, "an open issue, fenced": the cause is not known; the fence
    showed 0 of 1.4 M tiles)."""
    V, U = 3000, 900
    feats = SY.synth_din(B, T, V, U, seed=41 + T)
    h = feats["userRatedMovies"]
    h[np.random.default_rng(T).random(h.shape) < 0.25] = 0
    monkeypatch.setenv("SPRK_DIEN_FUSED", "0")
    two = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    assert two.engine.describe()["kernel"].startswith("k_din_tail")      # (the engine is built on first use: under THIS setting)
    monkeypatch.setenv("SPRK_DIEN_FUSED", "1")
    one = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    assert one.engine.describe()["kernel"].startswith("k_dien_fused")
    ids, dense = one.pack(feats)
    ti, td = _cuda(torch, ids), _cuda(torch, dense)
    first = {}
    for name, model in (("two", two), ("one", one)):
        first[name] = model.predict_device(ti, td).clone()
        bad = 0
        for _ in range(40):
            bad += int((model.predict_device(ti, td) != first[name]).sum().item())
        assert bad == 0, "%s: %d scores of 40 launches differ from the first launch" % (name, bad)
    # (the same inlined code in another kernel: the compiler may contract a multiply-add differently -- a last bit here and there)
    assert float((first["one"] - first["two"]).abs().max().item()) <= 1e-6
    rows = np.concatenate([np.arange(t * 16, min(t * 16 + 16, B)) for t in range(0, (B + 15) // 16, 8)])
    ref = O.dien_forward({k: v[rows] for k, v in feats.items()}, one.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    for name in first:
        assert np.abs(first[name].cpu().numpy()[rows] - ref).max() <= TIGHT, name


def test_dien_reference_schema_and_bad_ids(torch, samples):
    model = M.DIEN(seed=6)
    got = model.predict(samples)[:, 0]
    ref = O.dien_forward(samples, model.weights, dtype=np.float64)[:, 0]
    assert np.abs(got - ref).max() <= TOL
    ids, dense = model.pack(samples)
    bad = ids.copy()
    bad[100, 3] = 1001                                               # a history id outside Embedding(1001, ...)
    eng = model.engine
    aux = torch.empty((256, eng.n_aux), dtype=torch.float32, device="cuda")
    eng.din_pool(_cuda(torch, bad), aux, None)
    with pytest.raises(ValueError):
        eng.check_ids()
    with pytest.raises(L.SparrowHipError):                           # DIEN has no attention-weights output
        eng.din_pool(_cuda(torch, ids), aux, torch.empty((256, 5), dtype=torch.float32, device="cuda"))

"""BASELINE.json configs 3, 4 and 5 AT THEIR STATED SIZES (``-m gpu``):

* config 3  DIN, hist_len 50, emb_dim 32, batch 32 768
* config 4  DeepFM emb_dim 64 with a 27 M-row x 64 item table (6.9 GB, drawn on the device) next to the 138 k-user table,
            batch 65 536 -- both the sum-of-squares graph (DeepFM_v2: folded rows) and the pair-dot graph (DeepFM.py:100-103:
            the 256-byte embedding rows themselves are gathered)
* config 5  Wide&Deep with the 10 M-bucket x 32 hashed cross table (1.28 GB, hash computed on the device), batch 131 072

The models are the ones ``bench.py --workload ...`` runs (same builder).  The host stays cheap: a few thousand rows sampled
ACROSS the batch are compared with the fp64 oracle (for the device-resident tables the oracle sees only the rows the sample
references, ``bench.compact_for_oracle``); the whole batch goes through size-independent properties -- determinism, and a
slice scored alone equals the slice of the full batch, bit for bit."""
import os
import sys

import numpy as np
import pytest

from tests.conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-4
SAMPLE = 4096


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available()
    return t


def _check(torch, name, B, expect_kernel, sample=SAMPLE):
    model, feats, desc, roof = bench.build_workload(name, B, "uniform", NB=1)
    f = feats[0]
    eng = model.engine
    d = eng.describe()
    assert d["kernel"].split("<")[0] == expect_kernel, d
    ids, dense = model.pack(f)
    assert ids.shape[0] == B
    ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    p = model.predict_device(ti, td)
    eng.check_ids()
    assert torch.equal(p, model.predict_device(ti, td))                               # deterministic
    for lo, hi in ((0, 1000), (B // 2 - 333, B // 2 + 4001), (B - 517, B)):          # slice invariance incl. a ragged tail
        assert torch.equal(model.predict_device(ti[lo:hi].contiguous(), td[lo:hi].contiguous()), p[lo:hi])
    idx = np.sort(np.random.default_rng(1).choice(B, size=sample, replace=False))
    ref = bench.oracle_forward(name, model, {k: np.asarray(v)[idx] for k, v in f.items()}, dtype=np.float64)[:, 0]
    got = p.cpu().numpy()[idx]
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= TOL
    assert ref.std() > 0.02                                                            # scores are spread out: not a vacuous comparison
    tables = eng.table_bytes()
    eng.close()
    return tables


def test_config3_din_at_stated_size(torch):
    _check(torch, "din_c3", 32768, "k_din_fused")


def test_config4_deepfm_v2_27m_row_table(torch):
    tables = _check(torch, "deepfm_v2_c4", 65536, "k_deepfm_v2_joint")
    assert tables > 6.9e9                                                              # the 27 M x 64 table really lives on the device


def test_config4_pair_dot_deepfm_27m_row_table(torch):
    """DeepFM.py:100-103 at emb_dim 64: the fold does not apply, 256-byte rows are gathered out of the 6.9 GB table."""
    tables = _check(torch, "deepfm_c4", 65536, "k_deepfm_pairs")
    assert tables > 6.9e9


def test_config5_wide_and_deep_10m_bucket_cross(torch):
    tables = _check(torch, "widedeep_c5", 131072, "k_mlp_rows")
    assert tables > 1.28e9


def test_config5_hashed_buckets_bit_exact_at_stated_size(torch):
    """The cross hash alone, 131 072 pairs into 10 M buckets: device == oracle, bit for bit."""
    import ctypes as C
    from oracle import ctr_oracle as O
    from sparrowrecsys_amd import _lib as L, synthetic as SY
    B = 131072
    rng = np.random.default_rng(2)
    a = rng.integers(0, SY.ML20M_MOVIE_IDS, B).astype(np.int32)
    b = rng.integers(0, SY.ML20M_MOVIE_IDS, B).astype(np.int32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    out = torch.empty(B, dtype=torch.int64, device="cuda")
    lib = L.load_library()
    L.check(lib.sprk_cross_hash(C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), B, 10_000_000, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), O.crossed_bucket_np([a.astype(np.int64), b.astype(np.int64)], 10_000_000))


@pytest.mark.parametrize("name,B", [("deepfm_v2_c2", 65536), ("deepfm_c2", 65536), ("din_c3", 32768), ("widedeep_c5", 131072), ("deepfm_v2_ref", 65536),
                                    ("neuralcf_ref", 65536), ("deepfm_ref", 65536), ("din_ref", 65536), ("embedding_mlp_ref", 65536), ("dien_ref", 65536)])
def test_every_workload_scores_the_same_every_launch(torch, name, B):
    """[r5] Twenty launches of every bench workload at its stated batch -- every CU full, as many waves per SIMD as the kernel runs with -- bit for
    bit the first.  `_check` above repeats a launch once and compares 4 096 sampled rows with the oracle at 1e-4; a kernel that scores one tile in
    a hundred 6e-5 off, other tiles every launch, passes both -- k_dien_seq_mfma<16, 32> did until round 5 (k_dien_fused.h)."""
    model, feats, desc, roof = bench.build_workload(name, B, "uniform", NB=1)
    ids, dense = model.pack(feats[0])
    ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    first = model.predict_device(ti, td).clone()
    bad = 0
    for _ in range(20):
        bad += int((model.predict_device(ti, td) != first).sum().item())
    model.engine.close()
    assert bad == 0, "%s: %d scores of 20 launches differ from the first launch" % (name, bad)


@pytest.mark.parametrize("name,B,k", [("deepfm_v2_c2", 65536, 64), ("deepfm_c2", 65536, 16), ("din_c3", 32768, 16), ("widedeep_c5", 131072, 16),
                                      ("embedding_mlp_ref", 65536, 16), ("deepfm_v2_ref", 65536, 64)])
def test_several_batches_per_launch_at_stated_sizes(torch, name, B, k):
    """[r6] sprk_forward_many at `bench.py`'s launch shape -- k batches of the stated size per launch, every wave walking several tasks of several
    batches -- against the same batches one launch each: bit for bit, for every workload whose graph has a several-batches kernel."""
    n = 5
    model, feats, desc, roof = bench.build_workload(name, B, "uniform", NB=n)
    eng = model.engine
    packed = [model.pack(f) for f in feats]
    ids = [torch.from_numpy(p[0]).cuda() for p in packed]
    dense = [torch.from_numpy(p[1]).cuda() for p in packed]
    ws = torch.empty(max(eng.many_workspace_bytes(B, k) // 4, 1), dtype=torch.float32, device="cuda")
    res = {}
    for kk in (1, k):
        eng.set_many_batches(kk)
        outs = [torch.full((B,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
        eng.forward_many(ids, dense, outs, ws)
        torch.cuda.synchronize()
        eng.check_ids()
        res[kk] = outs
    bad = sum(int((a != b).sum().item()) for a, b in zip(res[1], res[k]))
    eng.set_many_batches(1)
    model.engine.close()
    assert bad == 0, "%s: %d scores differ between %d batches per launch and a launch per batch" % (name, bad, k)


# (workload, batch, the kernel its scores come from): every split-f16 register chain of the library, at the occupancy it runs with
EVERY_TILE = [("deepfm_v2_c2", 65536, "k_deepfm_v2_joint"), ("deepfm_c2", 65536, "k_deepfm_pairs"), ("din_c3", 32768, "k_din_fused"),
              ("widedeep_c5", 131072, "k_mlp_rows"), ("deepfm_v2_ref", 65536, "k_rows_chain"), ("din_ref", 65536, "k_din_tail"),
              ("dien_ref", 65536, "k_dien_fused"), ("embedding_mlp_ref", 65536, "k_mlp_rows")]


@pytest.mark.parametrize("name,B,kernel", EVERY_TILE)
def test_every_tile_against_the_oracle_at_full_occupancy(torch, name, B, kernel):
    """[r6, VERDICT r05 item 1] The fp64 oracle on EVERY 16-sample tile of one launch at the stated batch (not a 4 096-row sample, not every 8th
    tile) for each kernel built on the split-f16 `mm` idiom.  Round 5's flaky DIEN tiles were ~1 % of the tiles of a launch, 6e-5 off -- under the
    1e-4 bar and invisible to a sample; their cause ([r6]: a packed-f32 instruction with op_sel on a VGPR src1 losing its high dword next to another
    wave's 16x16x32 MFMAs, k_dien_fused.h) is kept out of every kernel by scripts/isa/isa_pk_opsel.py, and this is the one-off whole-batch check that
    nothing else of that size hides in the other chains.  Bar: 3e-5 on every score (fp32-class), and no tile stands out of the batch's own error
    distribution (a whole tile off by one vector is what the erratum looked like)."""
    model, feats, desc, roof = bench.build_workload(name, B, "uniform", NB=1)
    f = feats[0]
    assert model.engine.describe()["kernel"].split("<")[0].startswith(kernel), model.engine.describe()
    ids, dense = model.pack(f)
    got = model.predict_device(torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()).cpu().numpy()
    model.engine.check_ids()
    err = np.empty(B, np.float64)
    step = 8192
    for lo in range(0, B, step):
        sl = slice(lo, min(B, lo + step))
        ref = bench.oracle_forward(name, model, {k: np.asarray(v)[sl] for k, v in f.items()}, dtype=np.float64)[:, 0]
        err[sl] = np.abs(got[sl] - ref)
    model.engine.close()
    assert err.max() <= 3e-5, "%s: max |err| %.3g at row %d" % (name, err.max(), int(err.argmax()))
    tile = err[:B // 16 * 16].reshape(-1, 16).mean(axis=1)                     # a tile's mean error: rounding noise averages out, a stale operand does not
    assert tile.max() <= max(8 * np.median(tile), 2e-6), "%s: tile %d mean |err| %.3g, median tile %.3g" % (name, int(tile.argmax()), tile.max(), np.median(tile))

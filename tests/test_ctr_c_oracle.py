"""oracle/ctr_c.c (plain-C DeepFM_v2 forward, bench.py's cpu_baseline leg) against the numpy oracle, and the cpu_baseline
helpers of bench.py on a machine without a GPU."""
import os
import sys

import numpy as np
import pytest

from oracle import ctr_c
from oracle import ctr_oracle as O
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
from tests.conftest import ROOT


@pytest.mark.parametrize("fields,D,K", [(SY.CONFIG2_FIELDS, 16, 16), (None, 10, 64)])
def test_c_forward_matches_numpy_oracle(fields, D, K):
    B = 3001
    if fields is None:
        fields = M._default_fields()                                  # the reference's own four fields, emb_dim 10, Dense(64)
    model = M.DeepFMv2(seed=7, emb_dim=D, fields=fields, proj_dim=K)
    feats = SY.synth_fields(B, fields, seed=11)
    ids, dense = model.pack(feats)
    assert (ids == -1).any()
    cm = ctr_c.DeepFMv2C(model.weights, fields)
    got1 = cm.forward(ids, dense, threads=1)
    got4 = cm.forward(ids, dense, threads=4)
    assert np.array_equal(got1, got4)                                  # rows are independent: thread count cannot matter
    ref = O.deepfm_v2_forward(feats, model.weights, dtype=np.float64, fields=fields, order=[k for k, _, _ in fields])[:, 0]
    assert np.abs(got1 - ref).max() <= 2e-5
    assert ref.std() > 0.02


@pytest.mark.parametrize("T,D,B", [(50, 32, 1500), (5, 10, 777), (20, 16, 1)])
def test_c_din_forward_matches_numpy_oracle(T, D, B):
    V, U = 5000, 7000
    model = M.DIN(seed=9, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    feats = SY.synth_din(B, T, V, U, seed=13)
    ids, dense = model.pack(feats)
    cm = ctr_c.DinC(model)
    got = cm.forward(ids, dense, threads=1)
    assert np.array_equal(got, cm.forward(ids, dense, threads=3))
    ref = O.din_forward(feats, model.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    assert np.abs(got - ref).max() <= 2e-5


def test_bench_cpu_baseline_legs_run_without_a_gpu():
    sys.path.insert(0, ROOT)
    import bench
    B = 4096
    model = M.DeepFMv2(seed=101, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
    feats = [SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=3)]
    r = bench.cpu_baseline("deepfm_v2_c2", model, feats, 0.4)
    assert r["kind"] == "port" and r["unit"] == "samples/s" and r["value"] > 0 and r["cores"] >= 1
    assert "oracle/ctr_c.c" in r["sample"] and r["numpy_oracle_samples_per_sec"] > 0
    r2 = bench.cpu_baseline_numpy("deepfm_v2_c2", model, feats, 0.2)
    assert r2["kind"] == "port" and "numpy oracle" in r2["sample"]
    note = []
    assert bench.cpu_baseline_c("widedeep_c5", model, feats, 0.1, note) is None and note   # no C restatement: the numpy leg is used, and says why
    assert r["ran"] == "oracle/ctr_c.c"
    din = M.DIN(seed=103, emb_dim=32, hist_len=50, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
    fd = [SY.synth_din(1024, 50, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=4)]
    r3 = bench.cpu_baseline("din_c3", din, fd, 0.4)
    assert r3["kind"] == "port" and "DIN forward (oracle/ctr_c.c" in r3["sample"] and r3["value"] > 0

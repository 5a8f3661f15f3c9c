"""SURVEY 8(a) A14 pinned against TensorFlow's OWN golden vectors.

tensorflow/python/kernel_tests/sparse_cross_op_test.py crosses three string features and asserts the hashed output
("Check actual hashed output to prevent unintentional hashing changes"): 1971693436396284976 with the default hash key,
4847552627144134031 with key + 1, and 83 / 31 with num_buckets = 100.  Reproducing them needs every piece the reference's
crossed_column (WideNDeep.py:72-73) relies on: FingerprintCat64, the default key 0xDECAFCAFFE, left-to-right chaining and
the modulo on the unsigned value.  The same FingerprintCat64 is then shown to be the one the oracle, the golden fixture
and (under -m gpu) the HIP kernel use."""
import os

import numpy as np
import pytest

from oracle import ctr_oracle as O
from oracle import farmhash64 as FH
from tests.conftest import GOLDEN

STRINGS = ["batch1-FC1-F1", "batch1-FC2-F1", "batch1-FC3-F1"]


def test_tensorflow_sparse_cross_hashed_known_answers():
    assert FH.cross_hashed(STRINGS, 0) == 1971693436396284976                       # test_hashed_zero_bucket_no_hash_key
    assert FH.cross_hashed(STRINGS, 0, FH.DEFAULT_HASH_KEY + 1) == 4847552627144134031   # test_hashed_zero_bucket
    assert FH.cross_hashed(STRINGS, 100) == 83                                      # test_hashed_no_hash_key
    assert FH.cross_hashed(STRINGS, 100, FH.DEFAULT_HASH_KEY + 1) == 31             # test_hashed_output


def test_modulo_is_unsigned():
    # the raw hash of the first vector is >= 2^63: a signed modulo would give a different (negative) bucket
    h = FH.DEFAULT_HASH_KEY
    for s in STRINGS:
        h = FH.fingerprint_cat64(h, FH.fingerprint64(s.encode()))
    assert h >= 1 << 63 and h % FH.INT64_MAX == 1971693436396284976 and h % 100 == 83


def test_oracle_uses_the_pinned_fingerprint_cat64():
    rng = np.random.default_rng(5)
    for _ in range(200):
        a, b = int(rng.integers(0, 1 << 63)), int(rng.integers(0, 1 << 63))
        assert O.fingerprint_cat64(a, b) == FH.fingerprint_cat64(a, b)
    assert O.CROSS_HASH_KEY == FH.DEFAULT_HASH_KEY
    # the vectors themselves through the ORACLE's chain, with the strings' fingerprints as the int64 features
    fps = [np.array([FH.fingerprint64(s.encode())], dtype=np.uint64).astype(np.int64) for s in STRINGS]
    assert int(O.crossed_bucket(fps, 100)[0]) == 83
    assert int(O.crossed_bucket_np(fps, 100)[0]) == 83
    assert int(O.crossed_bucket(fps, 100, hash_key=FH.DEFAULT_HASH_KEY + 1)[0]) == 31


def test_golden_cross_fixture_matches_the_pinned_hash():
    g = np.load(os.path.join(GOLDEN, "cross_hash.npz"))
    for a, b, b1, b2 in zip(g["a"], g["b"], g["b10000"], g["b10m"]):
        assert FH.cross_hashed([int(a), int(b)], 10000) == int(b1)                  # WideNDeep.py:73
        assert FH.cross_hashed([int(a), int(b)], 10_000_000) == int(b2)             # BASELINE config 5


@pytest.mark.gpu
def test_hip_cross_hash_equals_the_pinned_chain():
    """k_cross_hash (ids are int32 on the device) against the python chain that reproduces TensorFlow's vectors."""
    import ctypes as C

    import torch

    from sparrowrecsys_amd import _lib as L
    lib = L.load_library()
    rng = np.random.default_rng(9)
    a = rng.integers(0, 1 << 31, size=512).astype(np.int32)
    b = rng.integers(0, 1 << 31, size=512).astype(np.int32)
    a[:4], b[:4] = [0, 1, 1000, 131262], [0, 0, 1000, 131262]
    for buckets in (10000, 10_000_000, (1 << 31) - 1):
        want = np.array([FH.cross_hashed([int(x), int(y)], buckets) for x, y in zip(a, b)], dtype=np.int64)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        out = torch.empty(len(a), dtype=torch.int64, device="cuda")
        L.check(lib.sprk_cross_hash(C.c_void_p(ta.data_ptr()), C.c_void_p(tb.data_ptr()), len(a), buckets, C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)

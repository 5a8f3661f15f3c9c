"""Native CSV ingest (sprk_pack_csv) against the Python restatement of make_csv_dataset + feature columns
(schema.read_samples_csv + pack_ids + pack_dense): bit-identical packed arrays."""
import io
import os
import time

import numpy as np
import pytest

from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import schema as S
from sparrowrecsys_amd.ingest import pack_csv, pack_csv_file

REF_CSV = "/root/reference/src/main/resources/webroot/sampledata/testSamples.csv"


def _python_pack(text, cols, keys, tmp_path):
    p = tmp_path / "x.csv"
    p.write_text(text)
    feats = S.read_samples_csv(str(p))
    return S.pack_ids(feats, cols), S.pack_dense(feats, keys)


def test_na_rules_quotes_and_ragged_rows(tmp_path):
    cols = M.EmbeddingMLP(seed=1).id_columns
    header = ["movieId", "userId", "rating", "timestamp", "releaseYear", "movieGenre1", "movieGenre2", "movieGenre3",
              "movieRatingCount", "movieAvgRating", "movieRatingStddev", "userRatedMovie1", "userRatingCount", "userAvgRating",
              "userRatingStddev", "userGenre1", "userGenre2", "userGenre3", "userGenre4", "userGenre5"]
    rows = [
        ["1", "15555", "3.0", "900953740", "1995", "Adventure", "Animation", "Children", "10759", "3.91", "0.89", "",
         "32", "3.47", "0.76", "Crime", "Drama", "", "", ""],
        ["999", "1", "5.0", "1", "", "Sci-Fi", "", "NotAGenre", "", "", "", "25", "", "", "", "Film-Noir", "Musical", "IMAX", "War", "Western"],
        ["12", "7"],                                                                   # wrong width: dropped
        ["3", "30000", "1", "2", "2001.0", '"Comedy"', "Romance", "", "7", "2.5", "1.25", "1000.0", "3", "4", "0.5",
         "Action", "", "", "", "Thriller"],
    ]
    text = ",".join(header) + "\n" + "\n".join(",".join(r) for r in rows) + "\n"
    ids, dense = pack_csv(text, cols)
    want_ids, want_dense = _python_pack(text, cols, S.NUMERIC_KEYS, tmp_path)
    assert ids.shape == (3, len(cols))
    np.testing.assert_array_equal(ids, want_ids)
    np.testing.assert_array_equal(dense, want_dense)
    # \r\n line ends and no trailing newline
    ids2, dense2 = pack_csv(text.replace("\n", "\r\n").rstrip("\r\n"), cols)
    np.testing.assert_array_equal(ids2, ids)
    np.testing.assert_array_equal(dense2, dense)
    # max_rows
    assert pack_csv(text, cols, max_rows=2)[0].shape[0] == 2


def test_out_of_range_id_and_missing_column_raise():
    cols = [S.IdColumn("movieId", "id", 1001), S.IdColumn("userGenre1", "genre", 19)]
    text = "movieId,userGenre1,releaseYear\n5,Drama,1990\n1001,Drama,1990\n"
    with pytest.raises(ValueError):
        pack_csv(text, cols, ["releaseYear"])
    with pytest.raises(Exception):
        pack_csv("movieId,releaseYear\n5,1990\n", cols, ["releaseYear"])
    ids, dense = pack_csv("movieId,userGenre1,releaseYear\n5,Drama,1990\n", cols, ["releaseYear"])
    assert ids.tolist() == [[5, 10]] and dense.tolist() == [[1990.0]]


@pytest.mark.skipif(not os.path.exists(REF_CSV), reason="reference sample file not present")
def test_reference_test_samples_bit_identical_and_faster():
    """The reference's own testSamples.csv, every row: native pack == Python pack, for three models' column sets."""
    for model in (M.EmbeddingMLP(seed=1), M.DeepFMv2(seed=1), M.DIN(seed=1)):
        cols = model.id_columns
        t0 = time.perf_counter()
        ids, dense = pack_csv_file(REF_CSV, cols)
        t_native = time.perf_counter() - t0
        t0 = time.perf_counter()
        feats = S.read_samples_csv(REF_CSV)
        want_ids, want_dense = S.pack_ids(feats, cols), S.pack_dense(feats)
        t_python = time.perf_counter() - t0
        assert ids.shape == want_ids.shape and ids.shape[0] > 10000
        np.testing.assert_array_equal(ids, want_ids)
        np.testing.assert_array_equal(dense, want_dense)
        print("%s: %d rows, native %.0f k rows/s, python %.0f k rows/s" % (type(model).__name__, ids.shape[0],
              ids.shape[0] / t_native / 1e3, ids.shape[0] / t_python / 1e3))
        assert t_native < t_python


def _synthetic_csv(n_rows, seed=0, bad_at=None, ragged_every=0):
    """A MovieLens-schema CSV body of n_rows lines (> 1 MiB so the packer really splits it)."""
    rng = np.random.default_rng(seed)
    genres = ["Drama", "Comedy", "Action", "", "Sci-Fi", "NotAGenre"]
    lines = ["movieId,userId,userGenre1,movieGenre1,releaseYear,movieAvgRating"]
    mid = rng.integers(0, 1001, n_rows)
    uid = rng.integers(0, 30001, n_rows)
    g1 = rng.integers(0, len(genres), n_rows)
    g2 = rng.integers(0, len(genres), n_rows)
    yr = rng.integers(1900, 2020, n_rows)
    rt = rng.random(n_rows) * 5
    for i in range(n_rows):
        m = "" if i % 97 == 0 else str(mid[i])                          # NA -> 0
        if bad_at is not None and i == bad_at:
            m = "1001"                                                  # outside movieId's buckets
        row = '%s,%d,%s,"%s",%d,%.2f' % (m, uid[i], genres[g1[i]], genres[g2[i]], yr[i], rt[i])
        if ragged_every and i % ragged_every == 5:
            row += ",extra"                                             # wrong width: dropped (ignore_errors)
        lines.append(row)
    return ("\n".join(lines) + "\n").encode()


COLS4 = [S.IdColumn("movieId", "id", 1001), S.IdColumn("userId", "id", 30001), S.IdColumn("userGenre1", "genre", 19),
         S.IdColumn("movieGenre1", "genre", 19)]


@pytest.mark.parametrize("threads", [2, 3, 8, 64])
def test_multithreaded_pack_is_identical(threads):
    text = _synthetic_csv(60000, seed=1, ragged_every=1000)
    assert len(text) > (1 << 20)
    ids1, dense1 = pack_csv(text, COLS4, ["releaseYear", "movieAvgRating"])
    idsN, denseN = pack_csv(text, COLS4, ["releaseYear", "movieAvgRating"], threads=threads)
    assert ids1.shape[0] == 60000 - 60
    np.testing.assert_array_equal(ids1, idsN)
    np.testing.assert_array_equal(dense1, denseN)
    # max_rows cuts the same prefix
    idsC, denseC = pack_csv(text, COLS4, ["releaseYear", "movieAvgRating"], max_rows=12345, threads=threads)
    np.testing.assert_array_equal(idsC, ids1[:12345])
    np.testing.assert_array_equal(denseC, dense1[:12345])


def test_multithreaded_pack_reports_the_first_error_like_one_thread():
    text = _synthetic_csv(60000, seed=2, bad_at=41000)
    msgs = []
    for threads in (1, 4, 16):
        with pytest.raises(ValueError) as e:
            pack_csv(text, COLS4, ["releaseYear"], threads=threads)
        msgs.append(str(e.value))
    assert len(set(msgs)) == 1 and "row 41000" in msgs[0] and "movieId id 1001" in msgs[0]
    # the bad row lies beyond max_rows: never looked at, with any thread count
    for threads in (1, 4, 16):
        ids, _ = pack_csv(text, COLS4, ["releaseYear"], max_rows=40000, threads=threads)
        assert ids.shape[0] == 40000


def test_multithreaded_pack_throughput_note():
    text = _synthetic_csv(200000, seed=3)
    t = {}
    for threads in (1, 8):
        t0 = time.perf_counter()
        ids, _ = pack_csv(text, COLS4, ["releaseYear", "movieAvgRating"], threads=threads)
        t[threads] = time.perf_counter() - t0
    print("pack_csv: 1 thread %.2f M rows/s, 8 threads %.2f M rows/s" % (0.2 / t[1], 0.2 / t[8]))
    assert ids.shape[0] == 200000

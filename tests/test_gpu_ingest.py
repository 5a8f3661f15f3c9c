"""Device CSV tokenizer (``sprk_pack_csv_device``, k_csv_pack.h) against the host tokenizer (``sprk_pack_csv``, itself pinned
bit for bit on the Python restatement of make_csv_dataset + feature columns over all 22 440 rows of the reference's
testSamples.csv, tests/test_ingest.py): identical packed arrays, identical first error (``-m gpu``)."""
import os

import numpy as np
import pytest

from sparrowrecsys_amd import _lib as L
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import schema as S
from sparrowrecsys_amd.ingest import last_device_path, pack_csv, pack_csv_device

pytestmark = pytest.mark.gpu
# (the A/B sweep of scripts/r03/70_env_switches.sh runs this file with SPRK_CSV_TWO_PASS=1: the exact sequence then always runs)
ONE_PASS = 2 if os.environ.get("SPRK_CSV_TWO_PASS") == "1" else 1
EXCERPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test_samples_512.csv")

HEADER = ["movieId", "userId", "rating", "timestamp", "releaseYear", "movieGenre1", "movieGenre2", "movieGenre3",
          "movieRatingCount", "movieAvgRating", "movieRatingStddev", "userRatedMovie1", "userRatingCount", "userAvgRating",
          "userRatingStddev", "userGenre1", "userGenre2", "userGenre3", "userGenre4", "userGenre5"]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "gpu tests need a HIP device"
    return t


def _same(text, cols, keys=S.NUMERIC_KEYS, **kw):
    ids, dense = pack_csv(text, cols, keys, **kw)
    dids, ddense = pack_csv_device(text, cols, keys, **kw)
    assert dids.is_cuda and ddense.is_cuda and dids.dtype.is_floating_point is False
    np.testing.assert_array_equal(dids.cpu().numpy(), ids)
    np.testing.assert_array_equal(ddense.cpu().numpy().view(np.uint32), dense.view(np.uint32))      # bits, not values
    return ids, dense


def test_na_rules_quotes_ragged_rows_line_ends(torch):
    cols = M.EmbeddingMLP(seed=1).id_columns
    rows = [
        ["1", "15555", "3.0", "900953740", "1995", "Adventure", "Animation", "Children", "10759", "3.91", "0.89", "",
         "32", "3.47", "0.76", "Crime", "Drama", "", "", ""],
        ["999", "1", "5.0", "1", "", "Sci-Fi", "", "NotAGenre", "", "", "", "25", "", "", "", "Film-Noir", "Musical", "IMAX", "War", "Western"],
        ["12", "7"],                                                                   # wrong width: dropped
        ["3", "30000", "1", "2", "2001.0", '"Comedy"', "Romance", '""', "7", "2.5", "1.25", '"1000.0"', "3", "4", "0.5",
         "Action", "", "", "", "Thriller"],
        [""] * 20,                                                                     # all empty: ids 0 / -1, numerics 0.0
        ["7", "8", "1", "2", "1e3", "Drama", "Drama", "Drama", "+5", "-0", "0.000123", "007", "1.50E+2", "12345.678901234", "1E-2",
         '"Children"junk', "Documentary", "documentary", "Film-Noir", "IMAX"],
    ]
    text = ",".join(HEADER) + "\n" + "\n".join(",".join(r) for r in rows) + "\n"
    ids, dense = _same(text, cols)
    assert ids.shape[0] == 5
    _same(text.replace("\n", "\r\n").rstrip("\r\n"), cols)                            # \r\n line ends, no trailing newline
    _same(text + "\n\n", cols)                                                         # empty lines
    _same(text, cols, max_rows=2)
    _same(",".join(HEADER) + "\n", cols)                                               # header only
    _same(",".join(HEADER), cols)
    # a column named twice and a column list that skips most of the file
    two = [S.IdColumn("movieId", "id", 1001), S.IdColumn("userGenre1", "genre", 19), S.IdColumn("movieId", "id", 5000)]
    _same(text, two, ["releaseYear", "movieAvgRating", "releaseYear"])


def test_reference_sample_rows(torch):
    """512 rows of the reference's own testSamples.csv (Spark wrote its empty strings as ""), three models' column sets."""
    text = open(EXCERPT, "rb").read()
    for model in (M.EmbeddingMLP(seed=1), M.DeepFMv2(seed=1), M.DIN(seed=1), M.WideNDeep(seed=1)):
        ids, dense = _same(text, model.id_columns)
        assert ids.shape[0] == 512


def _synthetic_csv(n_rows, seed, ragged=20):
    rng = np.random.default_rng(seed)
    genres = S.GENRE_VOCAB + ["", "(no genres listed)"]
    cols = []
    cols.append(rng.integers(1, 1000, n_rows).astype(str))                                  # movieId
    cols.append(rng.integers(1, 30000, n_rows).astype(str))                                 # userId
    cols.append(np.char.mod("%.1f", rng.integers(1, 11, n_rows) / 2.0))                     # rating
    cols.append(rng.integers(800000000, 1500000000, n_rows).astype(str))                    # timestamp
    cols.append(np.where(rng.random(n_rows) < 0.03, "", rng.integers(1900, 2020, n_rows).astype(str)))   # releaseYear
    for _ in range(3):
        cols.append(np.array(genres, dtype=object)[rng.integers(0, len(genres), n_rows)])
    cols.append(rng.integers(0, 70000, n_rows).astype(str))
    cols.append(np.char.mod("%.2f", rng.random(n_rows) * 5))
    cols.append(np.where(rng.random(n_rows) < 0.05, '""', np.char.mod("%.2f", rng.random(n_rows) * 2)))
    cols.append(np.where(rng.random(n_rows) < 0.1, "", rng.integers(1, 1000, n_rows).astype(str)))       # userRatedMovie1
    cols.append(rng.integers(0, 3000, n_rows).astype(str))
    cols.append(np.char.mod("%.6g", rng.random(n_rows) * 5))
    cols.append(np.char.mod("%.3e", rng.random(n_rows) * 3))
    for _ in range(5):
        cols.append(np.array(genres, dtype=object)[rng.integers(0, len(genres), n_rows)])
    lines = [",".join(HEADER)]
    lines += [",".join(map(str, r)) for r in zip(*cols)]
    for k in rng.integers(1, n_rows, ragged):                                              # ragged rows: dropped
        lines[k] = lines[k] + ",extra"
    return ("\n".join(lines) + "\n").encode()


def test_large_synthetic_file_bit_identical(torch):
    """200 000 rows (27 MB of text, ~6 600 chunks, every scan tile and workgroup boundary crossed many times)."""
    cols = M.EmbeddingMLP(seed=1).id_columns
    text = _synthetic_csv(200000, 5)
    ids, dense = _same(text, cols)
    assert ids.shape[0] == 200000 - 20
    assert last_device_path() == 2                                 # (ragged rows: the exact keep -> scan -> parse sequence)
    # the device buffer may be handed over directly
    buf = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
    dids, ddense = pack_csv_device(buf, cols)
    np.testing.assert_array_equal(dids.cpu().numpy(), ids)
    np.testing.assert_array_equal(ddense.cpu().numpy(), dense)


def test_errors_name_the_first_bad_row(torch):
    cols = [S.IdColumn("movieId", "id", 1001), S.IdColumn("userGenre1", "genre", 19)]
    good = "movieId,userGenre1,releaseYear\n5,Drama,1990\n6,War,1991\n"
    text = good + "1001,Drama,1990\n7,Drama,1\n2000,Drama,1\n"
    with pytest.raises(ValueError) as host:
        pack_csv(text, cols, ["releaseYear"])
    with pytest.raises(ValueError) as dev:
        pack_csv_device(text, cols, ["releaseYear"])
    assert str(dev.value) == str(host.value) and "row 2" in str(dev.value) and "1001" in str(dev.value)
    ids, _ = _same(text, cols, ["releaseYear"], max_rows=2)                               # rows past max_rows are never looked at
    assert ids.tolist() == [[5, 10], [6, 5]]
    with pytest.raises(Exception):
        pack_csv_device("movieId,releaseYear\n5,1990\n", cols, ["releaseYear"])             # no userGenre1 column
    # number spellings outside strtod's exact fast path: the device tokenizer refuses and names the row; the host one parses
    for bad in ("inf", " 5", "12345678901234567", "0x10", "1e400", "abc", '"1""2"'):
        t = good + "8,Drama,%s\n" % bad
        with pytest.raises(L.SparrowHipError) as e:
            pack_csv_device(t, cols, ["releaseYear"])
        assert e.value.code == L.EKIND and "row 2" in str(e.value) and "releaseYear" in str(e.value), (bad, str(e.value))
    ok = good + "8,Drama,123456789012345\n9,Drama,1234567890123450000\n"                     # 15 digits; zeros beyond them
    _same(ok, cols, ["releaseYear"])


def test_predict_csv_equals_predict_on_parsed_features(torch, tmp_path):
    """File -> device tokenizer -> forward, against the host route (read_samples_csv -> pack -> forward): same scores."""
    path = tmp_path / "s.csv"
    path.write_bytes(open(EXCERPT, "rb").read())
    for model in (M.DeepFMv2(seed=3), M.NeuralCF(seed=4), M.DIN(seed=5), M.WideNDeep(seed=6)):
        want = model.predict(S.read_samples_csv(str(path)))
        got = model.predict_csv(str(path), batch_size=200)                  # 200, 200, 112 rows
        assert got.shape == want.shape == (512, 1)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(model.predict_csv(open(EXCERPT, "rb").read(), max_rows=17), want[:17])


def test_short_decimal_fast_path_every_shape(torch):
    """The branch-free parser for fields of at most 8 characters against strtod + (float) / (long long) on the host: every
    placement of the sign and the dot, leading / trailing zeros, 1 .. 8 characters, random digits -- as a dense column (float
    bits) and as an identity column (truncation)."""
    rng = np.random.default_rng(77)
    vals = [".5", "5.", "-0", "+0", "0", "00000000", "99999999", "-9999999", "+1234.56", "-1234.56", "1234.567", ".0000001", "0.000001",
            "1000000.", "-.5", "+.5", "7", "-7", "0012.300", "9.999999", "4.", "000.000"]
    for _ in range(20000):
        n = int(rng.integers(1, 9))
        sign = ["", "-", "+"][int(rng.integers(0, 3))] if n > 1 else ""
        body = n - len(sign)
        digits = "".join(rng.choice(list("0123456789"), size=body))
        if body >= 2 and rng.random() < 0.6:
            d = int(rng.integers(0, body))
            digits = digits[:d] + "." + digits[d + 1:]
        if digits == ".":
            digits = "1"
        vals.append(sign + digits)
    text = "x,movieId\n" + "".join("%s,%d\n" % (v, i % 1000) for i, v in enumerate(vals))
    cols = [S.IdColumn("movieId", "id", 1001)]
    _same(text, cols, ["x"])
    # as identity ids: values in [0, 2^31) only (negative ones are range errors on both sides)
    pos = [v for v in vals if not v.startswith("-") or float(v) == 0.0]
    text = "x,movieId\n" + "".join("%s,%d\n" % (v, i % 1000) for i, v in enumerate(pos))
    _same(text, [S.IdColumn("x", "id", 2 ** 31 - 1), S.IdColumn("movieId", "id", 1001)], [])


def test_lines_too_long_for_the_lds_piece_are_read_in_place(torch):
    """A workgroup stages its 256 lines in LDS when they fit (at most 48 KB); a file with a few enormous free-text fields
    takes the in-place reader for those workgroups -- same bits -- and max_rows = 0 / a header-only text give no rows."""
    cols = M.EmbeddingMLP(seed=1).id_columns
    hdr = HEADER + ["comment"]
    rng = np.random.default_rng(9)
    lines = [",".join(hdr)]
    for i in range(1500):
        row = [str(int(rng.integers(1, 1000))), str(int(rng.integers(1, 30000))), "3.5", "1", "1999", "Drama", "", "War",
               str(int(rng.integers(0, 5000))), "%.2f" % rng.random(), "0.5", str(int(rng.integers(1, 1000))), "12", "3.25", "1.5",
               "Comedy", "", "", "", "IMAX"]
        junk = "x" * (70000 if i % 400 == 7 else int(rng.integers(0, 300)))          # a 70 KB field: beyond any LDS piece
        lines.append(",".join(row + [junk]))
    text = "\n".join(lines) + "\n"
    ids, dense = _same(text, cols)
    assert ids.shape[0] == 1500
    _same(text, cols, max_rows=0)
    _same(text, cols, max_rows=9)


# --------------------------------------------------------------------------------------------
# which kernels ran (sprk_csv_last_path): the optimistic pass alone on texts whose every line is a row, the exact sequence otherwise
# --------------------------------------------------------------------------------------------
def test_optimistic_pass_runs_alone_when_every_line_is_a_row(torch, monkeypatch):
    """The reference's sample rows and a 200 000-row synthetic file without ragged rows take ONE parse pass
    (sprk_csv_last_path() == 1) and give the host tokenizer's bits: with max_rows, \\r\\n line ends, no trailing newline;
    SPRK_CSV_TWO_PASS=1 (the exact keep -> scan -> parse sequence) gives the same arrays."""
    text = open(EXCERPT, "rb").read()
    for model in (M.EmbeddingMLP(seed=1), M.DeepFMv2(seed=1), M.DIN(seed=1), M.WideNDeep(seed=1)):
        _same(text, model.id_columns)
        assert last_device_path() == ONE_PASS
    cols = M.EmbeddingMLP(seed=1).id_columns
    text = _synthetic_csv(200000, 6, ragged=0)
    ids, dense = _same(text, cols)
    assert ids.shape[0] == 200000 and last_device_path() == ONE_PASS
    for kw in (dict(max_rows=1), dict(max_rows=12345), dict(max_rows=199999)):
        _same(text, cols, **kw)
        assert last_device_path() == ONE_PASS
    _same(text.rstrip(b"\n"), cols)
    _same(text.replace(b"\n", b"\r\n"), cols)
    _same(text.replace(b"\n", b"\r\n")[:-2], cols)
    assert last_device_path() == ONE_PASS
    monkeypatch.setenv("SPRK_CSV_TWO_PASS", "1")
    a, b = _same(text, cols)
    assert last_device_path() == 2
    np.testing.assert_array_equal(a, ids)


def test_every_alignment_of_rows_to_the_4kb_chunks(torch):
    """Rows, quoted empty strings, \\r\\n pairs and the end of the text at every offset relative to a 4-KB chunk boundary of the
    newline passes: the first row carries a pad field of 0 .. 47 bytes that shifts everything behind it; lengths that are exact
    multiples of 4 096 with and without the final newline."""
    cols = [S.IdColumn("movieId", "id", 100000), S.IdColumn("userGenre1", "genre", 19), S.IdColumn("q", "id", 10)]
    rng = np.random.default_rng(5)
    genres = S.GENRE_VOCAB + ["", '""', '"Drama"', "(none)"]
    body = ["%d,%s,%s,%s,p" % (rng.integers(0, 100000), genres[int(rng.integers(0, len(genres)))],
                             ["1990", "1990.5", "", '"1995"', "-3.25e1"][int(rng.integers(0, 5))], ['""', "", "7", '"3"'][int(rng.integers(0, 4))])
            for _ in range(420)]
    head = "movieId,userGenre1,releaseYear,q,pad"
    for eol in ("\n", "\r\n"):
        for shift in range(48):
            rows = [body[0] + "x" * shift] + body[1:]
            text = head + eol + eol.join(rows) + (eol if shift % 2 else "")
            ids, _ = _same(text, cols, ["releaseYear"])
            assert ids.shape[0] == 420 and last_device_path() == ONE_PASS, (eol, shift)
    for trailing in (True, False):
        for extra in (-1, 0, 1):
            base_text = head + "\n" + "\n".join(body) + ("\n" if trailing else "")
            want = ((len(base_text) + 4095) // 4096) * 4096 + extra
            pad = want - len(base_text)
            rows = [body[0] + "x" * pad] + body[1:]
            text = head + "\n" + "\n".join(rows) + ("\n" if trailing else "")
            assert len(text) == want
            ids, _ = _same(text, cols, ["releaseYear"])
            assert ids.shape[0] == 420 and last_device_path() == ONE_PASS, (trailing, extra, pad)


def test_quotes_the_host_splitter_reads_in_its_own_way(torch):
    """A comma / an escaped quote inside quotes, text after a closing quote, an unclosed quote, stray quotes, a 1.5-KB field, a
    56-digit number, '\\r' inside quotes, every field quoted -- at the start, in the middle (straddling a chunk boundary for some
    placements) and at the end of a 6-KB text: the device arrays are the host's bits, and the exact sequence runs exactly when the
    host drops a line."""
    cols = [S.IdColumn("movieId", "id", 1001), S.IdColumn("userGenre1", "genre", 19)]
    head = "movieId,userGenre1,releaseYear,note\n"
    good = "".join("%d,Drama,19%02d,n%d\n" % (i % 1000, i % 100, i) for i in range(300))
    rows = good.splitlines(keepends=True)
    kept = ['5,"Drama,War",1990,x', '5,"Dra""ma",1990,x', '5,"Drama"junk,1990,x', '5,Dra"ma,1990,x', '5,Drama",1990,x',
            "5,Drama,1990," + "y" * 1500, "5,Drama," + "0" * 55 + "7,x", '5,Drama,1990,"a\r"', '5,"Drama\r",1990,x', '"5","Drama","1990",""',
            '5,Drama,1990,"x,y,z"']
    dropped = ['5,"Drama,1990,x', '5,",1990,x', "", "5,Drama,1990", "5,Drama,1990,x,extra", "\r"]
    for line in kept + dropped:
        for where in (0, 150, 190, 200, 210, 300):
            text = head + "".join(rows[:where]) + line + "\n" + "".join(rows[where:])
            ids, _ = _same(text, cols, ["releaseYear"])
            assert ids.shape[0] == 300 + (line in kept), (line[:30], where)
            assert last_device_path() == (ONE_PASS if line in kept else 2), (line[:30], where)
    # one column: every non-empty line is a row
    one = "movieId\n" + "".join("%d\n" % (i % 1000) for i in range(3000))
    _same(one, cols[:1], [])
    assert last_device_path() == ONE_PASS
    _same(one.replace("\n", "\r\n"), cols[:1], [])
    assert last_device_path() == ONE_PASS
    cut = one.index("\n", 5000) + 1
    _same(one[:cut] + "\n" + one[cut:], cols[:1], [])                       # an empty line: not a row
    assert last_device_path() == 2
    crlf = one.replace("\n", "\r\n")
    cut = crlf.index("\n", 6000) + 1
    _same(crlf[:cut] + "\r\n" + crlf[cut:], cols[:1], [])
    assert last_device_path() == 2
    # an id outside its buckets in the middle of a clean text: the host's message
    bad = head + good + "1001,Drama,1990,x\n" + good
    with pytest.raises(ValueError) as host:
        pack_csv(bad, cols, ["releaseYear"])
    with pytest.raises(ValueError) as dev:
        pack_csv_device(bad, cols, ["releaseYear"])
    assert str(dev.value) == str(host.value) and "row 300" in str(dev.value)

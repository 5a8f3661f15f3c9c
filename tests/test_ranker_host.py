"""Host side of the "emb" ranker without a GPU: the embedding text format (Utility.parseEmbStr / DataManager.loadMovieEmb)
and the product parser's agreement with the oracle's."""
import os

import numpy as np
import pytest

from oracle import emb_rank_oracle as EO
from sparrowrecsys_amd import ranker as R

REF = "/root/reference/src/main/resources/webroot/modeldata/"


def test_load_emb_file_format(tmp_path):
    p = tmp_path / "emb.csv"
    p.write_text("710:-1.1897237 0.48152843 -0.6113423\n205:0.5 0.25 1e-3\nbroken line without colon\n7:1:2\n\n45:1 2 3\r\n")
    emb = R.load_emb_file(str(p))
    assert sorted(emb) == [45, 205, 710]                       # lines that do not split into two parts on ':' are skipped
    assert emb[710].dtype == np.float32 and emb[710][0] == np.float32(-1.1897237)
    assert emb[45].tolist() == [1.0, 2.0, 3.0]                  # trailing CR of a CRLF file is not part of the last float
    assert np.array_equal(emb[205], EO.parse_emb_str("0.5 0.25 1e-3"))


def test_emb_ranker_needs_a_gpu_and_never_falls_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        R.EmbRanker({1: np.ones(4, dtype=np.float32)})


@pytest.mark.skipif(not os.path.exists(REF + "item2vecEmb.csv"), reason="reference modeldata not present")
def test_reference_embedding_files_parse_identically():
    movies = R.load_emb_file(REF + "item2vecEmb.csv")
    users = R.load_emb_file(REF + "userEmb.csv")
    assert len(movies) == 881 and len(users) == 29776
    assert all(v.shape == (10,) for v in movies.values())
    with open(REF + "item2vecEmb.csv") as fh:
        for i, line in enumerate(fh):
            k, v = line.rstrip("\r\n").split(":")
            assert np.array_equal(movies[int(k)], EO.parse_emb_str(v))
            if i == 50:
                break

"""Generates tests/golden/emb_rank.npz (run in the build container, where /root/reference is mounted):
the first 256 movie embeddings of the reference's webroot/modeldata/item2vecEmb.csv, the first 32 user embeddings of
userEmb.csv (both parsed from the `id:f f f` text), and the ORACLE's scores / ranking of every user against every one
of those movies (oracle/emb_rank_oracle.py; the JVM reference cannot run here).

    python tests/golden/make_emb_rank_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import emb_rank_oracle as EO  # noqa: E402
from sparrowrecsys_amd.ranker import load_emb_file  # noqa: E402

REF = "/root/reference/src/main/resources/webroot/modeldata/"


def first(path, n):
    out = []
    with open(path) as fh:
        for line in fh:
            parts = line.rstrip("\r\n").split(":")
            if len(parts) == 2:
                out.append((int(parts[0]), EO.parse_emb_str(parts[1])))
            if len(out) == n:
                break
    return out


def main():
    movies, users = first(REF + "item2vecEmb.csv", 256), first(REF + "userEmb.csv", 32)
    host = load_emb_file(REF + "item2vecEmb.csv")                     # the product's parser agrees with the oracle's
    assert all(np.array_equal(host[m], v) for m, v in movies)
    item_ids = np.array([m for m, _ in movies], dtype=np.int32)
    item_emb = np.stack([v for _, v in movies])
    user_ids = np.array([u for u, _ in users], dtype=np.int32)
    user_emb = np.stack([v for _, v in users])
    cand = np.tile(np.arange(len(movies), dtype=np.int32), (len(users), 1))
    scores = EO.scores(item_emb, None, user_emb, None, cand)
    # spot-check the vectorised oracle against the scalar definition
    for u in (0, 7, 31):
        for c in (0, 100, 255):
            assert scores[u, c] == EO.calculate_similarity(user_emb[u], item_emb[c])
    np.savez_compressed(os.path.join(HERE, "emb_rank.npz"), item_ids=item_ids, item_emb=item_emb, user_ids=user_ids,
                        user_emb=user_emb, scores=scores, order=EO.rank(scores))
    print("emb_rank.npz: %d movies x %d users, best score %.6f" % (len(movies), len(users), np.nanmax(scores)))


if __name__ == "__main__":
    main()

"""The reference's "emb" ranker on the MI355X: cosine similarity of a user (or movie) embedding against candidate
movies and the ranked candidate list -- `RecForYouProcess.ranker(user, candidates, "emb")`
(RecForYouProcess.java:69-92,100-105), `SimilarMovieProcess.ranker(movie, candidates, "emb")`
(SimilarMovieProcess.java:121-136,167-172), `Embedding.calculateSimilarity` (Embedding.java:33-47).

Host side only parses the reference's embedding files and moves arrays; the scores and the ranking come from
`sprk_emb_rank` (HIP, include/sparrow_hip.h).  There is no CPU fallback: without the library / a GPU it raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L


def parse_emb_str(s: str) -> np.ndarray:
    """Utility.parseEmbStr (Utility.java:6-13): whitespace-separated floats."""
    return np.array(s.split(" "), dtype=np.float32)


def load_emb_file(path: str) -> Dict[int, np.ndarray]:
    """`id:f f f ...` per line -- DataManager.loadMovieEmb / loadUserEmb (DataManager.java:92-108,146-162):
    lines that do not split into exactly two parts on ':' are skipped."""
    out: Dict[int, np.ndarray] = {}
    with open(path, "r") as fh:
        for line in fh:
            parts = line.rstrip("\r\n").split(":")
            if len(parts) == 2:
                out[int(parts[0])] = parse_emb_str(parts[1])
    return out


class EmbRanker:
    """Movie embedding table resident in HBM + the ranker over it.

    `movie_emb`: {movieId: vector} (e.g. load_emb_file('item2vecEmb.csv')).  Movies passed as candidates that are
    not in the table score -1.0, like a Movie whose getEmb() is null (Embedding.java:34-37)."""

    def __init__(self, movie_emb: Dict[int, np.ndarray], device: str = "cuda:0"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("EmbRanker needs an MI355X (no CPU fallback)")
        self._torch = torch
        self._lib = L.load_library()
        self.device = torch.device(device)
        ids = sorted(movie_emb)
        if not ids:
            raise ValueError("empty embedding table")
        self.D = len(movie_emb[ids[0]])
        table = np.zeros((len(ids), self.D), dtype=np.float32)
        has = np.ones(len(ids), dtype=np.uint8)
        for i, m in enumerate(ids):
            v = np.asarray(movie_emb[m], dtype=np.float32)
            if v.shape != (self.D,):
                has[i] = 0                       # size mismatch -> -1 (Embedding.java:35)
            else:
                table[i] = v
        self.row_of = {m: i for i, m in enumerate(ids)}
        self.table = torch.from_numpy(table).to(self.device)
        self.has = torch.from_numpy(has).to(self.device)

    def rows(self, movie_ids: Iterable[int]) -> np.ndarray:
        return np.array([self.row_of.get(int(m), -1) for m in movie_ids], dtype=np.int32)

    def score_many(self, query_emb, cand_rows, query_has=None, want_order: bool = True) -> Tuple[object, Optional[object]]:
        """query_emb [Q, D] float32, cand_rows [Q, C] int32 table rows (-1 = no embedding) -- numpy or device tensors.
        Returns (scores [Q, C] float64, order [Q, C] int32 | None) as device tensors."""
        torch = self._torch
        q = torch.as_tensor(query_emb, dtype=torch.float32).to(self.device).contiguous()
        c = torch.as_tensor(cand_rows, dtype=torch.int32).to(self.device).contiguous()
        if q.dim() != 2 or c.dim() != 2 or q.shape[0] != c.shape[0] or q.shape[1] != self.D:
            raise ValueError("query_emb must be [Q, %d] and cand_rows [Q, C]" % self.D)
        qh = None if query_has is None else torch.as_tensor(query_has, dtype=torch.uint8).to(self.device).contiguous()
        Q, Cn = c.shape
        scores = torch.empty((Q, Cn), dtype=torch.float64, device=self.device)
        order = torch.empty((Q, Cn), dtype=torch.int32, device=self.device) if want_order else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self._lib.sprk_emb_rank(
            C.c_void_p(self.table.data_ptr()), C.c_void_p(self.has.data_ptr()), C.c_int32(self.table.shape[0]), C.c_int32(self.D),
            C.c_int32(self.D), C.c_void_p(q.data_ptr()), C.c_void_p(qh.data_ptr() if qh is not None else None),
            C.c_int32(Q), C.c_int32(self.D), C.c_void_p(c.data_ptr()), C.c_int32(Cn), C.c_void_p(scores.data_ptr()),
            C.c_void_p(order.data_ptr() if order is not None else None), C.c_void_p(stream)))
        return scores, order

    def rank(self, query: Optional[np.ndarray], candidate_ids: Sequence[int]) -> list:
        """ranker(user, candidates, "emb") (RecForYouProcess.java:69-92): the candidate movie ids, best first.
        `query` None = the user has no embedding (every score -1, candidates keep their order)."""
        cand = self.rows(candidate_ids)[None, :]
        if query is None:
            q, qh = np.zeros((1, self.D), dtype=np.float32), np.zeros(1, dtype=np.uint8)
        else:
            q, qh = np.asarray(query, dtype=np.float32)[None, :], np.ones(1, dtype=np.uint8)
            if q.shape[1] != self.D:
                q, qh = np.zeros((1, self.D), dtype=np.float32), np.zeros(1, dtype=np.uint8)    # size mismatch -> -1
        _, order = self.score_many(q, cand, qh)
        ids = list(candidate_ids)
        return [ids[i] for i in order[0].cpu().tolist()]

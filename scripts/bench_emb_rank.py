"""Throughput of the GPU "emb" ranker (sprk_emb_rank) on the reference's shape: every query ranks the 800-movie
candidate pool of RecForYouProcess.getRecList (RecForYouProcess.java:34) out of an 881-movie, D = 10 table
(item2vecEmb.csv).  One JSON line like bench.py's; cpu_baseline = the numpy oracle on a bounded sample.

    python scripts/bench_emb_rank.py [--queries 4096] [--steps 50] [--no-rank]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--cands", type=int, default=800)
    ap.add_argument("--items", type=int, default=881)
    ap.add_argument("--dim", type=int, default=10)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-rank", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    a = ap.parse_args()
    import torch
    from sparrowrecsys_amd.ranker import EmbRanker
    from oracle import emb_rank_oracle as EO
    rng = np.random.default_rng(1)
    items = rng.normal(size=(a.items, a.dim)).astype(np.float32)
    q = rng.normal(size=(a.queries, a.dim)).astype(np.float32)
    cand = np.stack([rng.permutation(a.items)[:a.cands] if a.cands <= a.items else rng.integers(0, a.items, a.cands)
                     for _ in range(a.queries)]).astype(np.int32)
    r = EmbRanker({i: items[i] for i in range(a.items)})
    qd, cd = torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda()
    want = not a.no_rank
    for _ in range(a.warmup):
        s, o = r.score_many(qd, cd, want_order=want)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        s, o = r.score_many(qd, cd, want_order=want)
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / a.steps
    # parity on the first queries
    n = min(64, a.queries)
    ref = EO.scores(items, None, q[:n], None, cand[:n])
    ok = bool(np.array_equal(s[:n].cpu().numpy().view(np.uint64), ref.view(np.uint64)))
    if want:
        ok = ok and bool(np.array_equal(o[:n].cpu().numpy(), EO.rank(ref)))
    per = 4 + 4 * a.dim + 8 + (4 if want else 0)
    units = a.queries * a.cands
    # cpu baseline: oracle on a bounded sample
    nq, t, done = 32, 0.0, 0
    tc = time.perf_counter()
    while time.perf_counter() - tc < a.cpu_seconds:
        sc = EO.scores(items, None, q[:nq], None, cand[:nq])
        if want:
            EO.rank(sc)
        done += nq * a.cands
    t = time.perf_counter() - tc
    print(json.dumps({
        "metric": "emb_ranker_candidates_per_sec", "value": units / (ms * 1e-3), "unit": "candidates/s", "n_gpus": 1,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "wall_ms_per_step": wall * 1e3 / a.steps,
        "higher_is_better": True, "dtype": "f64 (f32 products)", "data": "synthetic",
        "config": {"workload": "emb_rank: %d queries x %d candidates, table %d x %d, %s" %
                   (a.queries, a.cands, a.items, a.dim, "score + rank" if want else "score only")},
        "parity_checked": ok,
        "roofline": {"bound": "hbm", "achieved": units * per / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": units * per / (ms * 1e-3) / 8e12, "traffic": None, "bytes_per_candidate": per},
        "cpu_baseline": {"value": done / t, "unit": "candidates/s", "cores": 1, "kind": "port",
                         "sample": "numpy oracle, %d queries per call for %.0f s" % (nq, t)}}))


if __name__ == "__main__":
    main()

"""Host ingest throughput: sprk_pack_csv_mt on a synthetic MovieLens-schema CSV (no GPU involved).

    python scripts/bench_ingest.py [--rows 2000000] [--threads 1,8,32,64]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2000000)
    ap.add_argument("--threads", default="1,8,32,64")
    a = ap.parse_args()
    from tests.test_ingest import _synthetic_csv, COLS4
    from sparrowrecsys_amd.ingest import pack_csv
    base = _synthetic_csv(200000, seed=3)
    head, body = base.split(b"\n", 1)
    text = head + b"\n" + body * max(1, a.rows // 200000)
    n = text.count(b"\n") - 1
    out = {"rows": n, "mbytes": round(len(text) / 1e6, 1), "host_cpus": os.cpu_count(), "rows_per_sec": {}}
    ref = None
    for th in [int(x) for x in a.threads.split(",")]:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ids, dense = pack_csv(text, COLS4, ["releaseYear", "movieAvgRating"], max_rows=n, threads=th)
            best = min(best, time.perf_counter() - t0)
        if ref is None:
            ref = (ids.copy(), dense.copy())
        else:
            assert (ids == ref[0]).all() and (dense == ref[1]).all()
        out["rows_per_sec"][str(th)] = round(n / best)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

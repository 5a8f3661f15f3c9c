"""End to end, the reference's own workflow -- model.predict(get_dataset(csv)) (DeepFM.py:14-22,131-133) -- three ways:
  device : file bytes -> one copy of the raw text to the GPU -> sprk_pack_csv_device -> forward   (CTRModel.predict_csv)
  native : sprk_pack_csv_mt on the host's threads -> copy of the packed arrays -> forward
  python : schema.read_samples_csv + pack_ids / pack_dense (the restatement of make_csv_dataset) -> forward, on a subset
The text is the reference's sample rows (tests/golden/test_samples_512.csv) repeated.

    python scripts/bench_predict_csv.py [--rows 2000000] [--model deepfm_v2|din|neuralcf]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2000000)
    ap.add_argument("--model", default="deepfm_v2", choices=["deepfm_v2", "din", "neuralcf"])
    ap.add_argument("--threads", type=int, default=64)
    a = ap.parse_args()
    import numpy as np
    import torch
    from sparrowrecsys_amd import models as M, schema as S
    from sparrowrecsys_amd.ingest import pack_csv
    here = os.path.dirname(os.path.abspath(__file__))
    base = open(os.path.join(here, "..", "tests", "golden", "test_samples_512.csv"), "rb").read()
    head, body = base.split(b"\n", 1)
    reps = max(1, a.rows // 512)
    text = head + b"\n" + body * reps
    n = 512 * reps
    model = {"deepfm_v2": lambda: M.DeepFMv2(seed=1), "din": lambda: M.DIN(seed=1), "neuralcf": lambda: M.NeuralCF(seed=1)}[a.model]()
    out = {"model": a.model, "rows": n, "text_mbytes": round(len(text) / 1e6, 1), "host_cpus": os.cpu_count()}

    def timed(fn, reps=3):
        best, res = 1e9, None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best, res

    model.predict_csv(base)                                           # engine creation, scratch sizing
    with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as f:
        f.write(text)
        big = f.name
    t_dev, p_dev = timed(lambda: model.predict_csv(big))              # from the file (page cache): pinned read, one copy of the text
    out["device"] = {"seconds": round(t_dev, 4), "rows_per_sec": round(n / t_dev)}
    t_devb, p_devb = timed(lambda: model.predict_csv(text))           # from bytes already in (pageable) host memory
    out["device_from_bytes"] = {"seconds": round(t_devb, 4), "rows_per_sec": round(n / t_devb)}
    assert np.array_equal(p_dev, p_devb)

    def native_file():
        with open(big, "rb") as fh:
            return fh.read()
    t_read, _ = timed(native_file, reps=2)
    out["file_read_seconds"] = round(t_read, 4)
    os.unlink(big)

    def native():
        ids, dense = pack_csv(text, model.id_columns, list(model.numeric_keys), max_rows=n, threads=a.threads)
        outs = []
        for lo in range(0, n, 65536):
            outs.append(model.predict_device(torch.from_numpy(ids[lo:lo + 65536]).cuda(), torch.from_numpy(dense[lo:lo + 65536]).cuda()))
        return torch.cat(outs).cpu().numpy().reshape(-1, 1)
    t_nat, p_nat = timed(native, reps=2)
    out["native_host_tokenizer"] = {"seconds": round(t_nat, 4), "rows_per_sec": round(n / t_nat), "threads": a.threads}
    assert np.array_equal(p_dev, p_nat), "device and host routes disagree"

    sub = head + b"\n" + body * 20                                    # 10 240 rows through the Python restatement
    with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as f:
        f.write(sub)
        path = f.name
    t_py, p_py = timed(lambda: model.predict(S.read_samples_csv(path)), reps=1)
    os.unlink(path)
    out["python_feature_columns"] = {"seconds": round(t_py, 4), "rows": 10240, "rows_per_sec": round(10240 / t_py)}
    assert np.array_equal(p_py, p_dev[:10240])
    out["speedup_device_vs_native"] = round(t_nat / t_dev, 1)
    out["speedup_device_vs_python"] = round((n / t_dev) / (10240 / t_py), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Generates the committed golden fixtures under tests/golden/ (run in the build container, where
/root/reference is mounted; the GPU box only ever reads the generated files).

    python tests/golden/make_golden.py

Fixtures
* samples_256.npz      first 256 rows of the reference's webroot/sampledata/testSamples.csv (raw
                       string columns, file order) -- BASELINE config 1's batch.
* oracle_<model>.npz   oracle (numpy restatement) float32 + float64 predictions for those rows with
                       ``Model(seed=SEEDS[model]).init_weights`` weights (regenerated from the seed
                       at test time; sha256 of the weights is stored to detect RNG drift).
* neuralcf_ckpt.npz    weights of the reference's TRAINED checkpoints modeldata/neuralcf/{001,002}
                       restricted to the rows the first 2048 test samples touch, plus the oracle's
                       predictions / labels for those samples (the known answers SURVEY.md section 4
                       lists were reproduced on the full file by tests/test_oracle_pins.py).
* cross_hash.npz       FingerprintCat64 buckets for a fixed id grid (python-int implementation).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ctr_oracle as O  # noqa: E402
from sparrowrecsys_amd import models as M  # noqa: E402
from sparrowrecsys_amd import schema as S  # noqa: E402
from sparrowrecsys_amd.tensorbundle import model_variables  # noqa: E402

REF = "/root/reference/src/main/resources/webroot/"

SEEDS = {"embedding_mlp": 11, "wide_n_deep": 12, "neural_cf": 13, "neural_cf2": 14, "deepfm": 15,
         "deepfm_v2": 16, "din": 17}


def make_model(name):
    return {"embedding_mlp": lambda: M.EmbeddingMLP(seed=SEEDS[name]),
            "wide_n_deep": lambda: M.WideNDeep(seed=SEEDS[name]),
            "neural_cf": lambda: M.NeuralCF(seed=SEEDS[name]),
            "neural_cf2": lambda: M.NeuralCF(seed=SEEDS[name], arch=2),
            "deepfm": lambda: M.DeepFM(seed=SEEDS[name]),
            "deepfm_v2": lambda: M.DeepFMv2(seed=SEEDS[name]),
            "din": lambda: M.DIN(seed=SEEDS[name])}[name]()


def weights_digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def load_samples(path=os.path.join(HERE, "samples_256.npz")):
    z = np.load(path)
    return {k: z[k].astype(object) for k in z.files}


def ncf_weights(v):
    return {"emb/movieId": v["layer_with_weights-0/movieId_embedding.Sembedding_weights"],
            "emb/userId": v["layer_with_weights-1/userId_embedding.Sembedding_weights"],
            "dense0/kernel": v["layer_with_weights-2/kernel"], "dense0/bias": v["layer_with_weights-2/bias"],
            "dense1/kernel": v["layer_with_weights-3/kernel"], "dense1/bias": v["layer_with_weights-3/bias"],
            "head/kernel": v["layer_with_weights-4/kernel"], "head/bias": v["layer_with_weights-4/bias"]}


def main():
    feats = S.read_samples_csv(REF + "sampledata/testSamples.csv", limit=256)
    np.savez_compressed(os.path.join(HERE, "samples_256.npz"), **{k: v.astype(str) for k, v in feats.items()})
    feats = load_samples()
    for name in SEEDS:
        model = make_model(name)
        fn = O.FORWARDS[name]
        p32 = fn(feats, model.weights, dtype=np.float32)[:, 0]
        p64 = fn(feats, model.weights, dtype=np.float64)[:, 0]
        np.savez_compressed(os.path.join(HERE, "oracle_%s.npz" % name), pred32=p32, pred64=p64,
                            digest=np.array(weights_digest(model.weights)))
        print("%-14s mean %.4f std %.4f |p32-p64| %.2e" % (name, p64.mean(), p64.std(), np.abs(p32 - p64).max()))

    # trained NeuralCF checkpoints, restricted to the rows the first 2048 samples touch
    big = S.read_samples_csv(REF + "sampledata/testSamples.csv", limit=2048)
    users = np.unique(S.to_int_column(big["userId"]))
    out = {"movieId": S.to_int_column(big["movieId"]).astype(np.int32),
           "userId": S.to_int_column(big["userId"]).astype(np.int32),
           "label": S.to_int_column(big["label"]).astype(np.int8), "users": users.astype(np.int32)}
    for ver in ("001", "002"):
        w = ncf_weights(model_variables(REF + "modeldata/neuralcf/%s/variables" % ver))
        pred = O.neural_cf_forward(big, w)[:, 0]
        out["pred_" + ver] = pred
        out["user_rows_" + ver] = w["emb/userId"][users]
        for k, a in w.items():
            if k != "emb/userId":
                out[ver + "/" + k] = a
        print("neuralcf/%s first3 %s" % (ver, pred[:3]))
    np.savez_compressed(os.path.join(HERE, "neuralcf_ckpt.npz"), **out)

    a = np.array([0, 1, 2, 10, 999, 1000, 131262, 2 ** 31 - 1] * 8, dtype=np.int64)
    b = np.repeat(np.array([0, 1, 5, 318, 1000, 77777, 131262, 2 ** 31 - 1], dtype=np.int64), 8)
    np.savez_compressed(os.path.join(HERE, "cross_hash.npz"), a=a, b=b,
                        b10000=O.crossed_bucket([a, b], 10000), b10m=O.crossed_bucket([a, b], 10_000_000))
    print("done")


if __name__ == "__main__":
    main()

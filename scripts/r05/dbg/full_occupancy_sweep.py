"""Every DIN / DIEN dispatch shape at B = 65 536 (every CU full), 20 launches each, bit for bit the first; and the first against the fp64 oracle on every
16th tile.  (The bench workloads are in tests/test_gpu_stated_sizes.py::test_every_workload_scores_the_same_every_launch.)"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
from oracle import ctr_oracle as O
B, V, U = 65536, 5000, 7000
cases = [("din", 50, 32, {}), ("din", 20, 16, {}), ("din", 12, 10, {}), ("din", 5, 10, {}), ("din", 64, 32, {}), ("din", 30, 20, {}),
         ("din", 50, 32, {"SPRK_DIN_FUSED": "0"}), ("din", 20, 16, {"SPRK_DIN_FUSED": "0"}), ("din", 5, 10, {"SPRK_DIN_FUSED_MIN_T": "1"}),
         ("dien", 5, 10, {}), ("dien", 7, 16, {}), ("dien", 20, 16, {}), ("dien", 50, 10, {}),
         ("dien", 5, 10, {"SPRK_DIEN_FUSED": "0"}), ("dien", 7, 16, {"SPRK_DIEN_FUSED": "0"}), ("dien", 20, 16, {"SPRK_DIEN_FUSED": "0"}),
         ("dien", 7, 16, {"SPRK_DIEN_MFMA": "0"})]
worst = 0
for kind, T, D, env in cases:
    for k in ("SPRK_DIN_FUSED", "SPRK_DIN_FUSED_MIN_T", "SPRK_DIEN_FUSED", "SPRK_DIEN_MFMA"): os.environ.pop(k, None)
    os.environ.update(env)
    feats = SY.synth_din(B, T, V, U, seed=7 + T + D)
    h = feats["userRatedMovies"]
    h[np.random.default_rng(T).random(h.shape) < 0.2] = 0
    m = (M.DIN if kind == "din" else M.DIEN)(seed=5, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
    d = m.engine.describe()
    ids, dense = m.pack(feats)
    ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    first = m.predict_device(ti, td).clone()
    bad = sum(int((m.predict_device(ti, td) != first).sum().item()) for _ in range(20))
    rows = np.concatenate([np.arange(t * 16, t * 16 + 16) for t in range(0, B // 16, 16)])
    fwd = O.din_forward if kind == "din" else O.dien_forward
    ref = fwd({k: v[rows] for k, v in feats.items()}, m.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
    err = float(np.abs(first.cpu().numpy()[rows] - ref).max())
    worst += bad
    print("%-4s T=%-2d D=%-2d %-28s kernel=%-34s stage=%-16s differing scores in 20 launches: %d   max |err| vs fp64 on %d rows: %.2e" % (
        kind, T, D, env or "", d["kernel"][:34], d["stage"], bad, rows.size, err), flush=True)
    m.engine.close()
print("TOTAL differing:", worst)

"""k_dien_fused over many launches, every one bit for bit against the first (and the first against the two launches): counts the 16-sample
tiles that ever differ."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
D, T, B, RUNS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
V, U = 3000, 900
feats = SY.synth_din(B, T, V, U, seed=41 + T)
h = feats["userRatedMovies"]
h[np.random.default_rng(T).random(h.shape) < 0.25] = 0
os.environ["SPRK_DIEN_FUSED"] = "0"
m0 = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
assert m0.engine.describe()["kernel"].startswith("k_din_tail"), m0.engine.describe()      # (engines are built on first use)
os.environ["SPRK_DIEN_FUSED"] = "1"
m = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
assert m.engine.describe()["kernel"].startswith("k_dien_fused"), m.engine.describe()
ids, dense = m.pack(feats)
ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
two = m0.predict_device(ti, td).clone()
ref = m.predict_device(ti, td).clone()
print("first launch against the two launches: max |diff| %.3g, %d scores differ" % (float((ref - two).abs().max().item()), int((ref != two).sum().item())))
bad_tiles, bad_runs = 0, 0
for rep in range(RUNS):
    got = m.predict_device(ti, td)
    ne = (got != ref).reshape(-1)
    n = int(ne.sum().item())
    if n:
        bad_runs += 1
        bad_tiles += len(set((torch.nonzero(ne).reshape(-1) // 16).tolist()))
print("D=%d T=%d B=%d: %d launches, %d differ from the first, %d tiles of %d" % (D, T, B, RUNS, bad_runs, bad_tiles, RUNS * ((B + 15) // 16)))

# the two-launch path over the same number of launches, and both against the fp64 oracle on every 8th tile
bad_runs2 = 0
for rep in range(RUNS):
    if not torch.equal(m0.predict_device(ti, td), two): bad_runs2 += 1
from oracle import ctr_oracle as O
rows = np.concatenate([np.arange(t * 16, min(t * 16 + 16, B)) for t in range(0, (B + 15) // 16, 8)])
oref = O.dien_forward({k: v[rows] for k, v in feats.items()}, m.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
print("two launches: %d of %d launches differ from their first;  against the fp64 oracle on %d rows: one launch %.3g, two launches %.3g" % (
    bad_runs2, RUNS, rows.size, np.abs(ref.cpu().numpy()[rows] - oref).max(), np.abs(two.cpu().numpy()[rows] - oref).max()))

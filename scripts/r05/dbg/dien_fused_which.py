import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
D, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
V, U = 3000, 900
feats = SY.synth_din(B, T, V, U, seed=41 + T)
h = feats["userRatedMovies"]
h[np.random.default_rng(T).random(h.shape) < 0.25] = 0
os.environ["SPRK_DIEN_FUSED"] = "0"
m0 = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
assert m0.engine.describe()["kernel"].startswith("k_din_tail")
os.environ["SPRK_DIEN_FUSED"] = "1"
m = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
ids, dense = m.pack(feats)
ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
two = m0.predict_device(ti, td).cpu().numpy()
one = m.predict_device(ti, td).cpu().numpy()
d = np.abs(one - two)
idx = np.nonzero(d > 1e-6)[0]
tiles = sorted(set(idx // 16))
print("tiles with a score off by > 1e-6:", len(tiles), "of", (B + 15) // 16, " 1-ulp-only differences:", int(((d > 0) & (d <= 1e-6)).sum()))
print("tile index:", tiles[:40])
print("wave (tile % 16):", np.bincount(np.array(tiles) % 16, minlength=16))
print("workgroup % 8:", np.bincount((np.array(tiles) // 16) % 8, minlength=8))
# data properties of the bad tiles against all tiles
hist = feats["userRatedMovies"]
def props(ts):
    ts = np.array(ts)
    rows = (ts[:, None] * 16 + np.arange(16)[None, :]).reshape(-1)
    rows = rows[rows < B]
    hh = hist[rows]
    return dict(masked_frac=float((hh == 0).mean()), any_all_masked=float(((hh == 0).all(axis=1)).reshape(-1, 16).any(axis=1).mean()) if rows.size % 16 == 0 else -1,
                last_masked=float((hh[:, -1] == 0).reshape(-1, 16).any(axis=1).mean()) if rows.size % 16 == 0 else -1,
                first_masked_all=float((hh[:, 0] == 0).reshape(-1, 16).all(axis=1).mean()) if rows.size % 16 == 0 else -1)
print("bad tiles:", props(tiles))
print("all tiles:", props(list(range(B // 16))))
for t in tiles[:3]:
    print("tile", t, "hist zero pattern (rows = samples):")
    print((hist[t * 16:(t + 1) * 16] == 0).astype(int))
    print(" diffs x1e5:", np.round(1e5 * (one - two)[t * 16:(t + 1) * 16], 1))
from oracle import ctr_oracle as O
rows = np.concatenate([np.arange(t * 16, t * 16 + 16) for t in tiles[:6]] + [np.arange(0, 64)])
sub = {k: v[rows] for k, v in feats.items()}
ref = O.dien_forward(sub, m.weights, dtype=np.float64, hist_len=T, movie_buckets=V, user_buckets=U)[:, 0]
print("against the fp64 oracle on the first 6 differing tiles (+ tiles 0..3): max |one - ref| %.3g (bad tiles) %.3g (tiles 0..3);  max |two - ref| %.3g / %.3g" % (
    np.abs(one[rows][:96] - ref[:96]).max(), np.abs(one[rows][96:] - ref[96:]).max(), np.abs(two[rows][:96] - ref[:96]).max(), np.abs(two[rows][96:] - ref[96:]).max()))
one2 = m.predict_device(ti, td).cpu().numpy()
print("second fused launch equals the first:", np.array_equal(one, one2), " sha of the differing tile list:", hash(tuple(int(t) for t in tiles)) & 0xffffffff)

"""k_dien_seq_mfma (the two-launch DIEN path, SPRK_DIEN_FUSED=0) over many launches at full occupancy: every launch bit for bit against the
first; counts the 16-sample tiles that differ.  usage: dien_seq_stress.py D T B RUNS [label]"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["SPRK_DIEN_FUSED"] = "0"
import torch
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
D, T, B, RUNS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
label = sys.argv[5] if len(sys.argv) > 5 else ""
V, U = 3000, 900
feats = SY.synth_din(B, T, V, U, seed=41 + T)
h = feats["userRatedMovies"]
h[np.random.default_rng(T).random(h.shape) < 0.25] = 0
m = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
ids, dense = m.pack(feats)
ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
outs = [m.predict_device(ti, td).clone() for _ in range(RUNS)]
assert m.engine.describe()["kernel"].startswith("k_din_tail"), m.engine.describe()
st = torch.stack(outs).reshape(RUNS, -1)
maj = st.median(dim=0).values                                     # (a tile is wrong in ~1 % of the launches: the median is the clean value)
ne = st != maj
tiles = 0
for r in range(RUNS):
    idx = torch.nonzero(ne[r]).reshape(-1)
    tiles += len(set((idx // 16).tolist()))
print("%-10s D=%d T=%d B=%d: %d launches, %d tiles of %d differ from the per-sample median (%.2f per launch), max |diff| %.3g" % (
    label, D, T, B, RUNS, tiles, RUNS * ((B + 15) // 16), tiles / RUNS, float((st - maj).abs().max().item())))

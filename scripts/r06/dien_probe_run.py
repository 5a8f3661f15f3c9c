"""Run a probed k_dien_seq_mfma build (scripts/r06/probe_*.s: the probe ORs bits into the engine's error flag) RUNS times one launch at a time and
print the flag of every launch next to the number of tiles that differ from the per-sample median.  usage: dien_probe_run.py D T B RUNS [label]"""
import os, re, sys, numpy as np
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
outs, flags = [], []
for _ in range(RUNS):
    outs.append(m.predict_device(ti, td).clone())
    try:
        m.engine.check_ids()
        flags.append(0)
    except ValueError as e:
        mm = re.search(r"flag 0x([0-9a-f]+)", str(e))
        flags.append(int(mm.group(1), 16) if mm else -1)
st = torch.stack(outs).reshape(RUNS, -1)
maj = st.median(dim=0).values
ne = st != maj
per = [len(set((torch.nonzero(ne[r]).reshape(-1) // 16).tolist())) for r in range(RUNS)]
print("%-10s D=%d T=%d B=%d: %d launches; bad tiles per launch %s" % (label, D, T, B, RUNS, per))
print("%-10s flags per launch (2 = bias, 4 = un pair, 8 = chain result differ from a re-read / recomputation): %s" % (label, flags))

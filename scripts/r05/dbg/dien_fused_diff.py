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
h[0] = 0
os.environ["SPRK_DIEN_FUSED"] = "0"
m0 = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
ref = m0.predict(feats)[:, 0]
print("two-launch rerun equal:", all(np.array_equal(m0.predict(feats)[:, 0], ref) for _ in range(6)))
os.environ["SPRK_DIEN_FUSED"] = "1"
m = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
print(m.engine.describe()["kernel"])
ids, dense = m.pack(feats)
ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
for rep in range(8):
    got = m.predict_device(ti, td).cpu().numpy().reshape(-1)
    d = np.abs(got - ref)
    idx = np.nonzero(d > 0)[0]
    print("run", rep, "max diff %.3g" % d.max(), "n != :", idx.size, "idx", idx[:12], "r", (idx % 16)[:12], "tile%16 (wave)", ((idx // 16) % 16)[:12], "wg", (idx // 256)[:12])
if os.environ.get("DNF_DBG"):
    NA = m.engine.n_aux
    aux = torch.empty((B, NA), dtype=torch.float32, device="cuda")
    m0.engine.din_pool(ti, aux, None)
    aux = aux.cpu().numpy()
    np.set_printoptions(linewidth=250, precision=2, suppress=False)
    shown = 0
    for rep in range(10):
        ws = torch.zeros((6, B, NA), dtype=torch.float32, device="cuda")
        got = m.predict_device(ti, td, workspace=ws.reshape(-1)).cpu().numpy().reshape(-1)
        st = ws.cpu().numpy()[0]
        d = st - aux
        tiles = sorted(set(int(t) for t in np.nonzero(np.abs(d).max(axis=1) > 0)[0] // 16))
        print("dbg run", rep, "state tiles wrong", tiles[:10])
        for t in tiles[:2]:
            if shown < 6:
                shown += 1
                blk = d[t * 16:(t + 1) * 16]
                print(" tile", t, "relative diff (rows = samples, columns = features) x 1e4:")
                print(np.round(1e4 * blk / (np.abs(aux[t * 16:(t + 1) * 16]).max() + 1e-30), 1))

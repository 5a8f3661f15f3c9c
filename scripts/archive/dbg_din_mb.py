"""Pooled vectors of the several-batches-per-launch attention kernel against the one-batch kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sparrowrecsys_amd import models as M, synthetic as SY

np.set_printoptions(linewidth=220, precision=4, suppress=True)
Bd, n, k, T = 100, 3, 3, 50
din = M.DIN(seed=59, emb_dim=32, hist_len=T, movie_buckets=5000, user_buckets=7000)
eng = din.engine
fd = [SY.synth_din(Bd, T, 5000, 7000, seed=190 + i) for i in range(n)]
packed = [din.pack(f) for f in fd]
ids = [torch.from_numpy(p[0]).cuda() for p in packed]
dense = [torch.from_numpy(p[1]).cuda() for p in packed]
Dp = eng.n_aux
ref = []
for j in range(n):
    p = torch.zeros((Bd, Dp), dtype=torch.float32, device="cuda")
    eng.din_pool(ids[j], p)
    torch.cuda.synchronize()
    ref.append(p.cpu().numpy())
eng.set_many_batches(k)
eng.set_many_streams(0)
need = eng.many_workspace_bytes(Bd, 1)
ws = torch.zeros(need * k // 4, dtype=torch.float32, device="cuda")
outs = [torch.full((Bd,), -1.0, dtype=torch.float32, device="cuda") for _ in range(n)]
eng.forward_many(ids, dense, outs, ws)
torch.cuda.synchronize()
w = ws.cpu().numpy()
for j in range(n):
    got = w[j * need // 4: j * need // 4 + Bd * Dp].reshape(Bd, Dp)
    d = np.abs(got - ref[j])
    print("batch", j, "pooled max diff", d.max(), "rows differing", int((d.max(1) > 1e-6).sum()), "cols differing", np.nonzero(d.max(0) > 1e-6)[0])
    if d.max() > 1e-6:
        r = int(np.argmax(d.max(1)))
        print(" row", r, "\n  got", got[r], "\n  ref", ref[j][r], "\n  ratio", got[r] / np.where(ref[j][r] == 0, 1, ref[j][r]))
        # is it another row's pooled vector?
        allref = np.concatenate(ref)
        near = np.abs(allref - got[r]).max(1)
        print("  closest reference row:", int(np.argmin(near)), "dist", near.min(), "(expected index", j * Bd + r, ")")

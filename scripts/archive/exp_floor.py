"""Dispatch floor of a dependent launch chain: 2000 forwards of a 16-row NeuralCF batch (one 256-thread workgroup each)
enqueued by ONE sprk_forward_many call, timed with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sparrowrecsys_amd import models as M
m = M.NeuralCF(seed=1)
eng = m.engine
for B in (16, 4096):
    feats = {"userId": np.arange(1, B + 1) % 30000 + 1, "movieId": np.arange(1, B + 1) % 1000 + 1}
    ids, dense = m.pack(feats)
    ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(8)]
    n = 2000
    args = ([ti] * n, [td] * n, [outs[i % 8] for i in range(n)])
    eng.forward_many(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.forward_many(*args); e1.record(); torch.cuda.synchronize()
    print("NeuralCF B=%d: %.2f us per launch" % (B, e0.elapsed_time(e1) * 1e3 / n))

#!/bin/bash
# DIEN: parity tests, then a timing of the sequence stage + tail at B = 32 768 (reference shape D = 10, T = 5 and T = 50).
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests -m gpu -q -x -k "dien or emb_rank" 2>&1 | tail -8 | tee gpurun_out/pytest_dien.log
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/dien_time.log
import numpy as np, torch, time
from sparrowrecsys_amd import models as M, synthetic as SY
for T in (5, 50):
    B = 32768
    f = SY.synth_din(B, T, 1001, 30001, seed=1)
    m = M.DIEN(seed=2, emb_dim=10, hist_len=T)
    ids, dense = m.pack(f)
    ids, dense = torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()
    eng = m.engine
    aux = torch.empty((B, eng.n_aux), device="cuda")
    for name, fn in (("stage", lambda: eng.din_pool(ids, aux, None)), ("forward", lambda: m.predict_device(ids, dense))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print("DIEN T=%d B=%d %s: %.1f us" % (T, B, name, e0.elapsed_time(e1) * 20))
PY

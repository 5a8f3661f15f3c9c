import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ctr_oracle as O
from sparrowrecsys_amd import models as M, synthetic as SY
for dyn in ("1", "0"):
    os.environ["SPRK_DYN_F16"] = dyn
    model = M.DeepFM(seed=32, emb_dim=16, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)
    eng = model.engine
    B, n = 4099, 4
    feats = [SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=60 + i) for i in range(n)]
    packs = [model.pack(f) for f in feats]
    ids = [torch.from_numpy(a).cuda() for a, _ in packs]
    dense = [torch.from_numpy(b).cuda() for _, b in packs]
    one = [torch.empty(B, device="cuda") for _ in range(n)]
    many = [torch.empty(B, device="cuda") for _ in range(n)]
    for i in range(n): eng.forward(ids[i], dense[i], one[i])
    eng.set_many_batches(4); eng.forward_many(ids, dense, many); eng.set_many_batches(1)
    torch.cuda.synchronize()
    for i in range(n):
        ref = O.deepfm_forward(feats[i], model.weights, dtype=np.float64, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)[:, 0]
        a, b = one[i].cpu().numpy(), many[i].cpu().numpy()
        bad = np.nonzero(a != b)[0]
        print("dyn", dyn, "batch", i, "one-vs-oracle %.2e many-vs-oracle %.2e one-vs-many %.2e mismatches %d first %s" % (np.abs(a-ref).max(), np.abs(b-ref).max(), np.abs(a-b).max(), len(bad), bad[:20]))
    eng.close()

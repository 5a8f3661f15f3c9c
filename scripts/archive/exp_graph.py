"""Does a HIP graph of the predict-over-batches loop beat enqueueing it launch by launch?  (config 2, B = 65 536)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sparrowrecsys_amd import models as M, synthetic as SY
B, NB = 65536, 8
model = M.DeepFMv2(seed=101, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
eng = model.engine
batches = []
for i in range(NB):
    ids, dense = model.pack(SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=SY.SEED + i))
    batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()))
outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(NB)]
def many(n):
    idx = [i % NB for i in range(n)]
    eng.forward_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx])
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for streams in (0, 2):
    eng.set_many_streams(streams)
    n = 256
    t = timeit(lambda: many(n), 8) / (8 * n)
    print("streams=%d  eager forward_many: %.2f us/step" % (streams, t))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        many(n); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            many(n)
    torch.cuda.synchronize()
    t = timeit(g.replay, 8) / (8 * n)
    print("streams=%d  hipGraph replay:    %.2f us/step" % (streams, t))
    ref = outs[0].clone(); g.replay(); torch.cuda.synchronize()
    assert torch.equal(ref, outs[0])

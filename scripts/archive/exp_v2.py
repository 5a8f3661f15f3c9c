#!/usr/bin/env python
"""Experiment driver for the fused DeepFM_v2 kernel (not part of the product path).

  python scripts/exp_v2.py sweep      # A/B the SPRK_V2_* switches at several batch sizes
  python scripts/exp_v2.py trace      # per-wave phase timeline (sprk_debug_set_trace)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make(env):
    import torch
    torch.cuda.init()
    from sparrowrecsys_amd import models as M, synthetic as SY
    for k in ("SPRK_V2_REG", "SPRK_V2_XFLAGS", "SPRK_V2_GRID_CAP", "SPRK_V2_FOLD", "SPRK_V2_WGS_PER_CU", "SPRK_FORCE_INTERPRETER", "SPRK_V2_PRIO", "SPRK_V2_WAVES"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    model = M.DeepFMv2(seed=101, emb_dim=16, fields=SY.CONFIG2_FIELDS, proj_dim=16)
    model.engine
    return model


def time_forward(model, batches, outs, steps=200, warmup=20):
    import torch
    eng = model.engine
    n = len(batches)
    for i in range(warmup):
        eng.forward(batches[i % n][0], batches[i % n][1], outs[i % n])
    torch.cuda.synchronize()
    best = None
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            eng.forward(batches[i % n][0], batches[i % n][1], outs[i % n])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / steps
        best = us if best is None else min(best, us)
    return best


def inputs(model, B, nb=4, dist="uniform"):
    import torch
    from sparrowrecsys_amd import synthetic as SY
    batches, outs = [], []
    for i in range(nb):
        f = SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=SY.SEED + i, dist=dist)
        ids, dense = model.pack(f)
        batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()))
        outs.append(torch.empty(B, dtype=torch.float32, device="cuda"))
    return batches, outs


def sweep():
    variants = [
        ("reg", {}),
        ("reg,hotrow", {"SPRK_V2_XFLAGS": 16}),
        ("reg,nocompute", {"SPRK_V2_XFLAGS": 8}),
        ("unfolded", {"SPRK_V2_FOLD": 0}),
    ]
    sizes = [16384, 65536, 262144, 1048576]
    # clock ramp check: the same launch timed over longer and longer runs
    model = make({})
    bi = inputs(model, 65536)
    for steps in (400,):
        print("steps=%-6d %8.2f us/launch" % (steps, time_forward(model, *bi, steps=steps)), flush=True)
    model.engine.close()
    res = {}
    cache = {}
    for name, env in variants:
        model = make(env)
        for B in sizes:
            if B not in cache:
                cache[B] = inputs(model, B)
            us = time_forward(model, *cache[B])
            res.setdefault(name, {})[B] = round(us, 2)
            print("%-10s B=%-8d %8.2f us  %6.2f Gsamples/s" % (name, B, us, B / us / 1e3), flush=True)
        model.engine.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "exp_v2_sweep.json"), "w"), indent=1)


def trace(env, B=65536, dist='uniform'):
    import torch
    from sparrowrecsys_amd import _lib as L
    model = make(env)
    eng = model.engine
    batches, outs = inputs(model, B, nb=2, dist=dist)
    nw = 256 * 4 * 8 * 2
    buf = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
    for i in range(5):
        eng.forward(batches[i % 2][0], batches[i % 2][1], outs[i % 2])
    torch.cuda.synchronize()
    L.check(eng.lib.sprk_debug_set_trace(eng.handle, C.c_void_p(buf.data_ptr()), buf.numel() * 8))
    eng.forward(batches[0][0], batches[0][1], outs[0])
    torch.cuda.synchronize()
    L.check(eng.lib.sprk_debug_set_trace(eng.handle, None, 0))
    t = buf.cpu().numpy().reshape(nw, 16)
    t = t[t[:, 0] != 0]
    print("%s B=%d dist=%s waves traced: %d  untraced %.2f us/launch" % (env, B, dist, len(t), time_forward(model, batches, outs)))
    names = ["entry->ids", "ids->rowsIssued", "issued->imgIssued", "imgIssued->barrier", "barrier->rows",
             "rows->scored1", "scored1->exit", "entry->exit"]
    d = [t[:, i + 1] - t[:, i] for i in range(7)] + [t[:, 7] - t[:, 0]]
    for n, x in zip(names, d):
        x = x.astype(np.float64)
        print("  %-20s cycles p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (n, np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max()))
    t2 = t[t[:, 10] != 0]
    if len(t2):
        x = (t2[:, 10] - t2[:, 6]).astype(np.float64)
        print("  %-20s cycles p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f  (%d waves)" % ("scored1->scored2", np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max(), len(t2)))
    t3 = t[t[:, 11] != 0]
    if len(t3):
        for nm, x in (("rows1->gather2", t3[:, 11] - t3[:, 5]), ("gather2->scored1", t3[:, 6] - t3[:, 11])):
            x = x.astype(np.float64)
            print("  %-20s cycles p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f  (%d waves)" % (nm, np.percentile(x, 10), np.percentile(x, 50), np.percentile(x, 90), x.max(), len(t3)))
    w = (t[:, 8] - t[:, 8].min()).astype(np.float64) * 10.0   # 100 MHz ticks -> ns
    print("  wave entry skew (wall clock, ns): p50 %.0f p90 %.0f max %.0f" % (np.percentile(w, 50), np.percentile(w, 90), w.max()))
    span_wall = (t[:, 9].max() - t[:, 8].min()) * 10.0
    dm = (t[:, 7] - t[:, 0]).astype(np.float64)
    dw = (t[:, 9] - t[:, 8]).astype(np.float64) * 10.0
    print("  launch span first entry -> last exit: %.0f ns; memtime ticks per ns (median over waves): %.3f" % (span_wall, np.median(dm / np.maximum(dw, 1))))
    eng.close()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "sweep"
    if mode == "sweep":
        sweep()
    else:
        trace({"SPRK_V2_WGS_PER_CU": 1}, 65536)
        trace({"SPRK_V2_WGS_PER_CU": 1}, 32768)

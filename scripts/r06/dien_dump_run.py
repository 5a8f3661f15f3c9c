"""The failing k_dien_seq_mfma<16,32> build with a dump behind block 7's consumers (scripts/r06/dump_block7.s): per (tile, step, lane) the chain
result v[36:39], the bias v[48:51], the un pair v[0:1] and pre_z (v[4:5], v[12:13]).  RUNS launches through sprk_din_pool into one big buffer
each; which dumped field is the FIRST to differ from the per-element median in the tiles whose final state differs?
usage: dien_dump_run.py RUNS"""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
os.environ["SPRK_DIEN_FUSED"] = "0"
import torch
from sparrowrecsys_amd import models as M
from sparrowrecsys_amd import synthetic as SY
RUNS = int(sys.argv[1])
D, T, B, V, U = 16, 7, 65536, 3000, 900
feats = SY.synth_din(B, T, V, U, seed=41 + T)
h = feats["userRatedMovies"]
h[np.random.default_rng(T).random(h.shape) < 0.25] = 0
m = M.DIEN(seed=63, emb_dim=D, hist_len=T, movie_buckets=V, user_buckets=U)
ids, dense = m.pack(feats)
ti = torch.from_numpy(ids).cuda()
eng = m.engine
NA = eng.n_aux
assert B * NA * 4 <= 0x1000000
ntiles = B // 16
REC = 4096
nbytes = 0x1000000 + ntiles * T * REC
bufs = []
st = torch.cuda.current_stream().cuda_stream
for r in range(RUNS):
    buf = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    rc = eng.lib.sprk_din_pool(eng.handle, C.c_void_p(ti.data_ptr()), C.c_void_p(buf.data_ptr()), C.c_void_p(None), B, C.c_void_p(st))
    assert rc == 0, rc
    torch.cuda.synchronize()
    bufs.append(buf.cpu().numpy().view(np.uint32))
pooled = np.stack([b[:B * NA].reshape(ntiles, 16 * NA) for b in bufs])              # [RUNS, tiles, ..]
dump = np.stack([b[0x1000000 // 4:].reshape(ntiles, T, REC // 4) for b in bufs])      # [RUNS, tiles, T, 1024 words]
def median_u32(a):
    return np.sort(a, axis=0)[a.shape[0] // 2]
pm, dm = median_u32(pooled), median_u32(dump)
fields = {"chain result v[36:39]": (0, 256), "bias v[48:51]": (256, 512), "un pair v[0:1] / pre_z v[4:5]": (512, 768), "pre_z v[12:13]": (768, 1024)}
total_bad = 0
first = {}
for r in range(RUNS):
    bad = np.nonzero((pooled[r] != pm).any(axis=1))[0]
    total_bad += len(bad)
    for tl in bad:
        dd = dump[r, tl] != dm[tl]                                    # [T, 1024]
        steps = np.nonzero(dd.any(axis=1))[0]
        if len(steps) == 0:
            key = "no dumped field differs"
        else:
            t0 = steps[0]
            names = []
            for nm, (a, b) in fields.items():
                w = dd[t0, a:b]
                if nm.startswith("un pair"):
                    wl = w.reshape(64, 4)
                    if wl[:, :2].any(): names.append("un pair v[0:1]")
                    if wl[:, 2:].any(): names.append("pre_z v[4:5]")
                elif nm.startswith("pre_z"):
                    if w.reshape(64, 4)[:, :2].any(): names.append(nm)
                elif w.any():
                    names.append(nm)
            key = "step %d: %s" % (t0, " + ".join(names))
        first[key] = first.get(key, 0) + 1
print("%d launches, %d bad tiles (final state differs from the median)" % (RUNS, total_bad))
for k, v in sorted(first.items(), key=lambda kv: -kv[1]):
    print("  %5d  %s" % (v, k))
# one example in detail
for r in range(RUNS):
    bad = np.nonzero((pooled[r] != pm).any(axis=1))[0]
    for tl in bad[:3]:
        dd = dump[r, tl] != dm[tl]
        steps = np.nonzero(dd.any(axis=1))[0]
        if len(steps) == 0: continue
        t0 = steps[0]
        idx = np.nonzero(dd[t0])[0]
        print("launch %d tile %d step %d: %d words differ; word indices (first 24): %s" % (r, tl, t0, len(idx), idx[:24].tolist()))
        f = dump[r, tl, t0].view(np.float32); g = dm[tl, t0].view(np.float32)
        for i in idx[:8]:
            print("     word %4d (lane %2d, element %d): got %-14g median %-14g" % (i, (i % 256) // 4, i % 4, f[i], g[i]))
    if len(bad): break

# [detail] one bad tile: every dumped register of lanes 47..50 at the first bad step, and what the stale-operand hypotheses predict
img = None
for r in range(RUNS):
    bad = np.nonzero((pooled[r] != pm).any(axis=1))[0]
    done = 0
    for tl in bad:
        dd = dump[r, tl] != dm[tl]
        steps = np.nonzero(dd.any(axis=1))[0]
        if len(steps) == 0: continue
        t0 = steps[0]
        rec = dump[r, tl, t0].view(np.float32)
        for lane in (47, 48, 63):
            res = rec[lane * 4:lane * 4 + 4]; bias = rec[256 + lane * 4:256 + lane * 4 + 4]
            un = rec[512 + lane * 4:512 + lane * 4 + 2]; pz01 = rec[512 + lane * 4 + 2:512 + lane * 4 + 4]; pz23 = rec[768 + lane * 4:768 + lane * 4 + 2]
            want = np.float32(res) * np.float32(un[1]) + np.float32(bias)
            print("launch %d tile %d step %d lane %2d: result %s  un pair %s  bias %s\n      pre_z got %s %s   fma(result, un hi, bias) = %s   result * un LO + bias = %s" % (
                r, tl, t0, lane, res, un, bias, pz01, pz23, want, np.float32(res) * np.float32(un[0]) + np.float32(bias)))
        done += 1
        if done >= 2: break
    if done: break

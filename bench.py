#!/usr/bin/env python
"""bench.py -- CTR samples/sec of the HIP forward on synthetic MovieLens-20M-shaped batches.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One "step" = one pass of the hot path over one batch (ids + dense already resident in HBM ->
scores in HBM); at N>1 every rank scores its own B-row shard (weak scaling: global batch N*B) and its
score slices are all-gathered over RCCL, --gather-group steps per collective on a second stream (every
step's scores have been exchanged on every rank when the timed region ends).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  deepfm_v2_c2 (default) configs[1]: DeepFM (sum-of-squares FM cross, DeepFM_v2 graph), 6 sparse
               fields, emb_dim 16, projection width 16, B = 65 536 per GPU
  deepfm_c2    the pairwise-dot DeepFM graph on the same fields
  din_c3       configs[2]: DIN, hist_len 50, emb_dim 32, B = 32 768 per GPU
  widedeep_c5  configs[4], one GPU's share: Wide&Deep with the 10 M-bucket x 32 hashed cross table, B = 131 072
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12   # FLOP/s, fp32-input MFMA


def build_workload(name, B, dist_name, seed_offset=0, big_vocab=0, NB=8):
    """NB: distinct input batches (own ids / dense / score buffers each) the steps cycle through."""
    from sparrowrecsys_amd import models as M, synthetic as SY
    if name in ("deepfm_v2_c2", "deepfm_c2"):
        F, D = 6, 16
        fields = SY.CONFIG2_FIELDS
        if big_vocab:
            # same graph, the three identity fields' tables blown up past the 256 MB Infinity Cache: every row gather is a
            # real HBM access (config 4's "true HBM-resident gather" at config 2's widths)
            fields = [(k, kind, big_vocab if kind == "id" else v) for k, kind, v in fields]
        if name == "deepfm_v2_c2" and big_vocab:
            model = M.DeepFMv2(seed=101, emb_dim=D, fields=fields, proj_dim=16)
            desc = "DeepFM sum-of-squares FM (DeepFM_v2 graph), F=6, emb_dim=16, proj=16, deep 32-16, identity tables of %d rows each" % big_vocab
            feats = [SY.synth_fields(B, fields, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name) for i in range(NB)]
            bytes_per_sample = F * 4 + F * D * 4 + F * 4 + 7 * 4 + 4
            roof = {"bound": "hbm", "kernel": "k_deepfm_v2_joint", "bytes_per_sample": bytes_per_sample}
            model._bench_fields = fields
            return model, feats, desc, roof
        if name == "deepfm_v2_c2":
            model = M.DeepFMv2(seed=101, emb_dim=D, fields=SY.CONFIG2_FIELDS, proj_dim=16)
            desc = "DeepFM sum-of-squares FM (DeepFM_v2 graph), F=6 sparse fields, emb_dim=16, proj=16, deep 32-16"
        else:
            model = M.DeepFM(seed=101, emb_dim=D, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)
            desc = "DeepFM pairwise-dot FM (DeepFM graph), F=6 sparse fields, emb_dim=16, 8 pairs, deep 64-64"
        feats = [SY.synth_fields(B, SY.CONFIG2_FIELDS, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name) for i in range(NB)]
        # fused kernel: ids + embedding rows + first-order weights + numerics in, one score out
        bytes_per_sample = F * 4 + F * D * 4 + F * 4 + 7 * 4 + 4
        v2_kernel = "k_deepfm_v2_chain" if (os.environ.get("SPRK_V2_JOINT") == "0" or os.environ.get("SPRK_V2_FOLD") == "0") else "k_deepfm_v2_joint"
        if os.environ.get("SPRK_FORCE_INTERPRETER") == "1":
            v2_kernel = "k_tile_forward"
        v1_kernel = "k_tile_forward" if os.environ.get("SPRK_V1_CHAIN") == "0" else "k_deepfm_pairs"
        roof = {"bound": "hbm", "kernel": v2_kernel if name == "deepfm_v2_c2" else v1_kernel,
                "bytes_per_sample": bytes_per_sample}
    elif name == "din_c3":
        T, D = 50, 32
        model = M.DIN(seed=103, emb_dim=D, hist_len=T, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS)
        desc = "DIN, hist_len=50, emb_dim=32, attention 128->32->1, tail 167->128->64->1"
        feats = [SY.synth_din(B, T, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name) for i in range(NB)]
        flops = T * (2 * 4 * D * 32 + 2 * 32 + 3 * 32)            # the reference's count: K = 4D per (b,t)
        # what k_din_attn issues on the matrix pipe: K = D per (b,t) after folding the c-only and h-only
        # blocks (A_b = W12 + W4 diag(c)), over whole 16-row groups (T=50 -> 64 columns)
        executed = ((T + 15) // 16) * 16 * 2 * D * 32
        legacy = os.environ.get("SPRK_DIN_LEGACY") == "1"
        roof = {"bound": "mfma", "kernel": "k_din_pool" if legacy else "k_din_attn", "flops_per_sample": flops,
                "executed_flops_per_sample": flops if legacy else executed,
                "bytes_per_sample": (T + 1) * 4 + (T + 1) * D * 4 + D * 4}
    elif name == "widedeep_c5":
        # BASELINE configs[4], one GPU's share: Wide&Deep, hashed cross (movieId x userRatedMovie1) computed on device into a
        # 10 M-bucket x 32 embedding table (1.28 GB), emb_dim 32, deep 128-128
        D, CB = 32, 10_000_000
        model = M.WideNDeep(seed=105, emb_dim=D, movie_buckets=SY.ML20M_MOVIE_IDS, user_buckets=SY.ML20M_USER_IDS,
                            cross_buckets=CB, cross_dim=D)
        desc = "Wide&Deep, 10 embedding columns emb_dim=32 + hashed cross 10M buckets x 32 (on-device FingerprintCat64), deep 128-128"
        feats = [SY.synth_embedding_mlp(B, SY.ML20M_MOVIE_IDS, SY.ML20M_USER_IDS, seed=SY.SEED + 1000 * seed_offset + i, dist=dist_name,
                                        rated_vocab=SY.ML20M_MOVIE_IDS) for i in range(NB)]
        # 8 B ids + 128 B cross row per sample for the wide part; + the deep part's 11 ids, 10 rows, numerics, score
        roof = {"bound": "hbm", "kernel": "k_tile_forward" if os.environ.get("SPRK_MLP_CHAIN") == "0" else "k_mlp_chain",
                "bytes_per_sample": 8 + D * 4 + 9 * 4 + 10 * D * 4 + 7 * 4 + 4}
    else:
        raise SystemExit("unknown workload %r" % name)
    return model, feats, desc, roof


def oracle_forward(name, model, feats):
    from oracle import ctr_oracle as O
    from sparrowrecsys_amd import synthetic as SY
    if name == "deepfm_v2_c2":
        fields = getattr(model, "_bench_fields", SY.CONFIG2_FIELDS)
        return O.deepfm_v2_forward(feats, model.weights, dtype=np.float32, fields=fields, order=[k for k, _, _ in fields])
    if name == "deepfm_c2":
        return O.deepfm_forward(feats, model.weights, dtype=np.float32, fields=SY.CONFIG2_FIELDS, pairs=SY.CONFIG2_PAIRS)
    if name == "widedeep_c5":
        return O.wide_n_deep_forward(feats, model.weights, dtype=np.float32, movie_buckets=model.movie_buckets,
                                     user_buckets=model.user_buckets, cross_buckets=model.cross_buckets,
                                     rated_buckets=model.rated_buckets)
    return O.din_forward(feats, model.weights, dtype=np.float32, hist_len=model.hist_len,
                         movie_buckets=model.movie_buckets, user_buckets=model.user_buckets)


def cpu_baseline_c(name, model, feats, budget_s):
    """deepfm_v2_c2 / din_c3: the plain-C restatement of the forward (oracle/ctr_c.c, OpenMP over samples, built with
    -march=native on THIS host) on the packed ids / dense of the same synthetic batch -- a fairer stand-in for the
    reference's TF2 CPU forward than the numpy oracle, whose time goes into Python feature handling.  Checked against
    the numpy oracle before it is timed.  Returns None when it cannot be used (other workloads, no compiler)."""
    if name not in ("deepfm_v2_c2", "din_c3"):
        return None
    try:
        from oracle import ctr_c
        from sparrowrecsys_amd import synthetic as SY
        ctr_c.load(native=True)
        if name == "deepfm_v2_c2":
            cm = ctr_c.DeepFMv2C(model.weights, getattr(model, "_bench_fields", SY.CONFIG2_FIELDS))
            what = "DeepFM_v2"
        else:
            cm = ctr_c.DinC(model)
            what = "DIN"
        ids, dense = model.pack(feats[0])
        n = ids.shape[0]
        threads = max(1, min(os.cpu_count() or 1, 128))
        got = cm.forward(ids[:2048], dense[:2048], threads=1)
        ref = oracle_forward(name, model, {k: v[:2048] for k, v in feats[0].items()})[:, 0]
        if not (np.abs(got - ref).max() <= 5e-5):
            return None
        out = np.empty(n, dtype=np.float32)
        cm.forward(ids, dense, threads=threads, out=out)     # warm-up (threads, page faults)
        t0 = time.perf_counter()
        done = 0
        while True:
            cm.forward(ids, dense, threads=threads, out=out)
            done += n
            el = time.perf_counter() - t0
            if el >= budget_s or done >= 4096 * n:
                break
        return {"value": done / el, "unit": "samples/s", "cores": threads, "kind": "port",
                "sample": "plain-C restatement of the %s forward (oracle/ctr_c.c, OpenMP, -march=native; TensorFlow unavailable), "
                          "%d passes over one packed batch of %d rows, %.1f s" % (what, done // n, n, el),
                "host_cpus": os.cpu_count()}
    except Exception:
        return None


def cpu_baseline(name, model, feats, budget_s):
    """The CPU restatement ("port": TensorFlow is not installable) timed on this host's cores over a bounded sample of
    the same workload: the C restatement where there is one (DeepFM_v2), with the numpy oracle's rate reported next to
    it; the numpy oracle otherwise."""
    c = cpu_baseline_c(name, model, feats, 0.5 * budget_s)
    if c is not None:
        npy = cpu_baseline_numpy(name, model, feats, 0.5 * budget_s)
        c["numpy_oracle_samples_per_sec"] = npy["value"]
        c["numpy_oracle_threads"] = npy["cores"]
        return c
    return cpu_baseline_numpy(name, model, feats, budget_s)


def cpu_baseline_numpy(name, model, feats, budget_s):
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = 1
    f = feats[0]
    n = len(next(iter(f.values())))
    sample = min(n, 16384)
    fs = {k: v[:sample] for k, v in f.items()}
    oracle_forward(name, model, fs)                      # warm-up
    t0 = time.perf_counter()
    done = 0
    while True:
        oracle_forward(name, model, fs)
        done += sample
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 64 * sample:
            break
    return {"value": done / el, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "numpy oracle (CPU restatement; TensorFlow unavailable), %d passes of %d rows of the same synthetic batch, %.1f s"
                      % (done // sample, sample, el),
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="deepfm_v2_c2")
    ap.add_argument("--batch", type=int, default=0, help="rows per GPU (default: the config's batch)")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf", "hot"], help="id distribution")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle spot check of the outputs")
    ap.add_argument("--gather-group", type=int, default=64,
                    help="N>1: batches whose score slices share one RCCL all-gather (overlapped with the next group)")
    ap.add_argument("--overlap-streams", type=int, default=2,
                    help="fan sprk_forward_many's independent batches over S helper HIP streams (2..4; 0 = strict stream order) in "
                         "the TIMED region: a dependent launch chain costs ~3.3 us per launch even for an empty kernel, so the "
                         "predict-over-batches loop is launch bound without it.  The roofline block is always measured in strict "
                         "order (one kernel at a time), in its own loop after the timed region.")
    ap.add_argument("--launch-batches", type=int, default=16,
                    help="batches ONE kernel launch scores in the timed region (sprk_set_many_batches, up to 64; 1 = a launch per "
                         "batch; deepfm_v2_c2 only).  Each batch keeps its own buffers of --batch rows; the ~3.3 us launch floor is "
                         "spent once per N batches.  With N > 1 the launches of the timed region run in strict order (no stream "
                         "fan-out).  The `roofline` block stays ONE batch per launch; `roofline_timed_region` describes the "
                         "N-batch launches.")
    ap.add_argument("--input-batches", type=int, default=0,
                    help="distinct synthetic input batches the steps cycle through (default: 32 for the headline workload -- no "
                         "two batches of one 16-batch launch share buffers -- 16 for din_c3, 8 otherwise)")
    ap.add_argument("--big-vocab", type=int, default=0,
                    help="deepfm_v2_c2 only: rows of each identity table (e.g. 8388608 = 1 GiB of folded rows per table, "
                         "far beyond the Infinity Cache); default 0 = the MovieLens-20M-shaped vocabularies of the config")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (gloo: functional test of the N>1 path with ranks sharing one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    # SPRK_BENCH_FORCE_DIST=1: run the N>1 code path (process group, grouped all-gather on the side stream) with
    # WORLD_SIZE 1 -- a functional check of the RCCL calls on a one-GPU box
    dist_on = world > 1 or os.environ.get("SPRK_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group("gloo")

    B = args.batch or {"din_c3": 32768, "widedeep_c5": 131072}.get(args.workload, 65536)
    nb_in = args.input_batches or {"deepfm_v2_c2": 32, "deepfm_c2": 16, "din_c3": 16}.get(args.workload, 8)
    if args.batch and args.batch > 262144:
        nb_in = min(nb_in, 8)
    model, feats, desc, roof = build_workload(args.workload, B, args.dist, seed_offset=rank, big_vocab=args.big_vocab, NB=nb_in)
    eng = model.engine
    lb = 1
    if args.launch_batches > 1 and os.environ.get("SPRK_FORCE_INTERPRETER") != "1":
        if args.workload == "deepfm_v2_c2" and os.environ.get("SPRK_V2_JOINT") != "0" and os.environ.get("SPRK_V2_FOLD") != "0":
            lb = args.launch_batches
            args.overlap_streams = 0       # several batches per launch: strict order measured faster than the fan-out
        elif args.workload == "deepfm_c2" and os.environ.get("SPRK_V1_CHAIN") != "0":
            lb = min(args.launch_batches, 16)
            args.overlap_streams = 0
        elif args.workload == "din_c3" and os.environ.get("SPRK_DIN_LEGACY") != "1" and os.environ.get("SPRK_DIN_TAIL") != "0":
            lb = min(args.launch_batches, 16)   # groups of batches: one attention + one tail launch each, alternating streams
    if lb > 1:
        eng.set_many_batches(lb)
    fan = args.overlap_streams if (args.overlap_streams >= 2 and eng.set_many_streams(args.overlap_streams)) else 0
    batches = []
    for f in feats:
        ids, dense = model.pack(f)
        batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(dense).cuda()))
    NB = len(batches)
    outs = [torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(NB)]
    ws = torch.empty(max(eng.many_workspace_bytes(B, max(fan, 1) * lb) // 4, 1), dtype=torch.float32, device="cuda")
    gs = None
    if dist_on:
        from sparrowrecsys_amd.dist import GroupedScoreGather
        gs = GroupedScoreGather(B, max(1, args.gather_group), torch.device("cuda", local_dev))

    group_run = [None, None]

    def run_steps(first, count):
        """`count` steps starting at step index `first`.  One sprk_forward_many call enqueues a run of
        forwards (the same kernel launches, without a Python/ctypes round trip per launch, which at ~7 us
        would out-last the kernel).  N>1: every step's score slice lands in a GroupedScoreGather ring slot;
        each full group of --gather-group steps is exchanged by ONE RCCL all-gather on a second stream
        while the next group is scored (a per-step all-gather of 256 KiB costs more launch latency than
        the forward it follows)."""
        if gs is None:
            idx = [i % NB for i in range(first, first + count)]
            eng.forward_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
            return
        i = first
        while i < first + count:
            if gs.fill == 0 and first + count - i >= gs.G:
                # a whole group: prepared pointer arrays (one per ring slot), one foreign call, one collective
                slot = gs.slot
                outs_g = gs.group_outs()
                if group_run[slot] is None:
                    idx = [j % NB for j in range(gs.G)]
                    group_run[slot] = eng.prepare_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], outs_g, ws)
                group_run[slot](stream=torch.cuda.current_stream().cuda_stream)
                gs.commit()
                i += gs.G
                continue
            n = min(gs.G - gs.fill, first + count - i)
            idx = [j % NB for j in range(i, i + n)]
            eng.forward_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [gs.out() for _ in idx], ws)
            i += n
            if gs.full():
                gs.commit()

    def fence():
        if gs is not None:
            gs.flush()                     # exchange a partial group, wait for every collective
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(0, args.warmup)
    fence()
    eng.check_ids()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()
    run_steps(args.warmup, args.steps)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    ev_ms_timed = ev_ms
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel time for the roofline: HIP events on the launch stream around K forwards in STRICT order (one kernel at a
    # time, what rocprofv3's per-kernel duration measures), re-timed right after the timed region whenever that region
    # overlapped launches (fan-out) or held the all-gathers (N>1)
    region = "timed region"
    if dist_on or fan or lb > 1:
        region = "strict-order forward loop after the timed region"
        if fan:
            eng.set_many_streams(0)
        if lb > 1:
            eng.set_many_batches(1)
        torch.cuda.synchronize()
        ev0.record()
        idx = [i % NB for i in range(args.steps)]
        eng.forward_many([batches[j][0] for j in idx], [batches[j][1] for j in idx], [outs[j] for j in idx], ws)
        ev1.record()
        torch.cuda.synchronize()
        ev_ms = ev0.elapsed_time(ev1)
        if fan:
            eng.set_many_streams(fan)
        if lb > 1:
            eng.set_many_batches(lb)
    fwd_s = ev_ms * 1e-3 / args.steps          # avg forward duration (all kernels of one step)

    # output spot check against the oracle (outside the timed region)
    check = None
    if rank == 0 and not args.no_check:
        if dist_on:
            eng.forward(batches[0][0], batches[0][1], outs[0], ws)
            torch.cuda.synchronize()
        n = 4096
        got = outs[0][:n].cpu().numpy()
        ref = oracle_forward(args.workload, model, {k: v[:n] for k, v in feats[0].items()})[:, 0]
        check = float(np.abs(got - ref).max())
        if not check <= 1e-4:
            raise SystemExit("bench outputs differ from the oracle: max|err| = %g" % check)

    if rank == 0:
        value = B * world * args.steps / elapsed
        din_s = None
        legacy_din = os.environ.get("SPRK_DIN_LEGACY") == "1"
        if roof["bound"] == "hbm":
            achieved = roof["bytes_per_sample"] * B / fwd_s / 1e9
            rl = {"bound": "hbm", "kernel": roof["kernel"], "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                  "frac": achieved * 1e9 / HBM_PEAK, "algorithmic_bytes_per_sample": roof["bytes_per_sample"],
                  "avg_launch_us": fwd_s * 1e6, "timed_with": "HIP events, " + region}
            if fan:
                rl["timed_with"] += " (the timed region itself fans independent batches over %d streams: ms_per_step %.5f)" % (fan, elapsed * 1e3 / args.steps)
            if lb > 1:
                rl["kernel"] += " (one batch of %d rows per launch)" % B
                rl["timed_with"] += "; ONE batch per launch -- the north-star's 'at batch 65 536' figure.  The timed region scores %d batches per launch, see roofline_timed_region" % lb
        else:
            # DIN step = k_din_pool + k_tile_forward; time the attention kernel alone for its MFMA fraction
            pooled = torch.empty((B, eng.n_aux), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            ev0.record()
            for i in range(args.steps):
                eng.din_pool(batches[i % NB][0], pooled)
            ev1.record()
            torch.cuda.synchronize()
            din_s = ev0.elapsed_time(ev1) * 1e-3 / args.steps
            if legacy_din:
                # k_din_pool: the reference's K = 4D contraction on f32 MFMA
                achieved = roof["flops_per_sample"] * B / din_s / 1e12
                rl = {"bound": "mfma", "kernel": roof["kernel"], "achieved": achieved, "peak": MFMA_F32_PEAK / 1e12,
                      "unit": "TFLOP/s", "frac": achieved * 1e12 / MFMA_F32_PEAK,
                      "algorithmic_flops_per_sample": roof["flops_per_sample"]}
            else:
                # k_din_attn: after the K = 4D -> D fold and the move to split-f16 MFMA the matrix pipe is a small
                # share of the kernel; what bounds it is the history gather (+ the VALU work per gathered row),
                # so it is priced against HBM bandwidth on SURVEY 8(d)'s algorithmic bytes
                achieved = roof["bytes_per_sample"] * B / din_s / 1e9
                rl = {"bound": "hbm", "kernel": roof["kernel"], "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                      "frac": achieved * 1e9 / HBM_PEAK,
                      "matrix_flops_per_sample_issued": roof["executed_flops_per_sample"],
                      "reference_flops_per_sample": roof["flops_per_sample"],
                      "reference_equivalent_TFLOPs": roof["flops_per_sample"] * B / din_s / 1e12}
            rl.update({"algorithmic_bytes_per_sample": roof["bytes_per_sample"],
                       "avg_launch_us": din_s * 1e6, "step_us_all_kernels": fwd_s * 1e6,
                       "timed_with": "HIP events, %s-only loop after the timed region" % roof["kernel"]})
        # memory-side bytes per launch from the committed PMC passes (rocprofv3 --pmc runs are separate from the
        # timed run by design); only quoted for the batch size and kernel they were collected on
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                t = json.load(open(tpath)).get(args.workload)
                if t and t.get("batch") == B and t.get("kernel") == roof["kernel"]:
                    traffic = t["bytes_per_launch"]
                    rl["traffic_source"] = "profiles/traffic.json (PMC FETCH_SIZE x2 + WRITE_SIZE, round %s)" % t.get("round")
            except Exception:
                traffic = None
        rl["traffic"] = traffic
        line = {
            "metric": "ctr_samples_per_sec", "value": value, "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc), "batch_per_gpu": B, "global_batch": B * world,
                       "id_distribution": args.dist, "input_batches_cycled": NB,
                       "parallelism": ("rows sharded over %d GPU(s), tables replicated, all-gather of scores" % world)
                                      + ("" if world == 1 else " (one collective per %d steps, overlapped; %d collectives in the timed region)"
                                         % (gs.G, (args.steps + gs.G - 1) // gs.G)),
                       "oracle_check_max_abs_err": check,
                       "launch_overlap_streams": fan, "batches_per_launch": lb,
                       "arithmetic": "fp32 semantics; contractions whose operands are bounded at finalize (table rows x weights) run on "
                                     "v_mfma_f32_16x16x32_f16 with split operands hi + lo (22 significand bits) and f32 accumulation -- "
                                     "fp32-class error, tests/test_gpu_parity.py::test_deepfm_v2_split_f16_is_fp32_class; everything else "
                                     "on f32 MFMA / VALU (SPRK_V2_HALF=0 / SPRK_DIN_HALF=0 force f32 MFMA throughout)"},
            "roofline": rl,
        }
        if lb > 1 and not dist_on and roof["bound"] == "hbm":
            # the timed region's own launches: lb batches (own buffers, B rows each) per launch, strict stream order, so the
            # HIP events around the region bracket exactly ceil(K / lb) back-to-back launches of the multi-batch instantiation
            n_launch = (args.steps + lb - 1) // lb
            ach = roof["bytes_per_sample"] * B * args.steps / (ev_ms_timed * 1e-3) / 1e9
            line["roofline_timed_region"] = {
                "bound": "hbm", "kernel": roof["kernel"] + " (multi-batch instantiation, %d batches of %d rows per launch)" % (lb, B),
                "achieved": ach, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach * 1e9 / HBM_PEAK,
                "launches": n_launch, "avg_launch_us": ev_ms_timed * 1e3 / n_launch, "timed_with": "HIP events around the timed region"}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.workload, model, feats, args.cpu_seconds)
        # RCCL prints a version banner through C stdio, which (stdout being a pipe) would otherwise be flushed at exit,
        # AFTER this line: flush it first so the JSON is the last line of output
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

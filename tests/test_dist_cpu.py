"""N>1 path on CPU: two gloo ranks shard the rows, score their shard, all-gather the scores.
The per-shard forward is injected (here: the plan interpreter) because there is no GPU; what is
under test is the sharding + collective logic of sparrowrecsys_amd/dist.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparrowrecsys_amd.dist import shard_bounds


def test_shard_bounds_cover_rows_exactly():
    for n in (0, 1, 7, 8, 9, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparrowrecsys_amd import models as M
        from sparrowrecsys_amd.dist import RowShardedPredictor, all_gather_scores
        from tests.plan_interp import run_plan
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", "samples_256.npz"))
        samples = {k: z[k].astype(object)[:n_rows] for k in z.files}
        model = M.DeepFMv2(seed=16)
        plan, slots = model.build_plan()
        ids, dense = model.pack(samples)
        calls = []

        def forward(ids_t, dense_t):
            calls.append(int(ids_t.shape[0]))
            return torch.from_numpy(run_plan(plan, slots, ids_t.numpy(), dense_t.numpy(), np.float64).astype(np.float32))

        pred = RowShardedPredictor(forward)
        scores = pred.predict(torch.from_numpy(ids), torch.from_numpy(dense))
        eq = all_gather_scores(torch.full((4,), float(rank)))
        q.put((rank, scores.numpy(), calls, eq.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [256, 101])
def test_two_rank_row_sharding_matches_single_process(n_rows):
    from oracle import ctr_oracle as O
    from sparrowrecsys_amd import models as M
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "samples_256.npz"))
    samples = {k: z[k].astype(object)[:n_rows] for k in z.files}
    ref = O.deepfm_v2_forward(samples, M.DeepFMv2(seed=16).weights, dtype=np.float64)[:, 0]
    for rank, scores, calls, eq in results:
        assert scores.shape == (n_rows,)
        np.testing.assert_allclose(scores, ref, atol=1e-6)           # every rank ends with ALL scores, in row order
        lo, hi = shard_bounds(n_rows, rank, world)
        assert calls == [hi - lo]                                    # ... having scored only its own shard
        np.testing.assert_array_equal(eq, [0, 0, 0, 0, 1, 1, 1, 1])


def _group_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparrowrecsys_amd.dist import GroupedScoreGather
        B, G, steps = 8, 3, 8                        # 8 batches in groups of 3: 3 + 3 + partial 2
        got = []
        gs = GroupedScoreGather(B, G, "cpu", sink=lambda gi, view, nb: got.append((gi, nb, view.clone())))
        i = 0
        while i < steps:
            if i == 3:                               # second group through the whole-group form (cached views)
                for j, out in enumerate(gs.group_outs()):
                    out.copy_(torch.arange(B, dtype=torch.float32) + 1000 * rank + 100 * (i + j))
                assert gs.full()
                gs.commit()
                i += G
                continue
            out = gs.out()
            out.copy_(torch.arange(B, dtype=torch.float32) + 1000 * rank + 100 * i)     # "forward" of batch i
            i += 1
            if gs.full():
                gs.commit()
        gs.flush()
        q.put((rank, gs.collectives, [(gi, nb, v.numpy()) for gi, nb, v in got]))
    finally:
        dist.destroy_process_group()


def test_grouped_score_gather_two_ranks():
    """GroupedScoreGather: G batches per collective, partial last group, every rank sees every rank's
    scores of every batch in (group, rank, batch) order."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, collectives, got in results:
        assert collectives == 3
        assert [(gi, nb) for gi, nb, _ in got] == [(0, 3), (1, 3), (2, 2)]
        for gi, nb, v in got:
            assert v.shape == (world, nb, 8)
            for r in range(world):
                for j in range(nb):
                    np.testing.assert_array_equal(v[r, j], np.arange(8, dtype=np.float32) + 1000 * r + 100 * (3 * gi + j))


def _fd_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparrowrecsys_amd.dist import exchange_fds
        rd, wr = os.pipe()                                   # this rank keeps the read end, the peers get the write end
        peers = exchange_fds(wr, None)
        os.close(wr)
        for peer, fd in sorted(peers.items()):
            os.write(fd, b"%d->%d;" % (rank, peer))          # through the PEER's pipe, opened in this process by SCM_RIGHTS
            os.close(fd)
        dist.barrier()
        data = b""
        while data.count(b";") < world - 1:
            data += os.read(rd, 256)
        q.put((rank, sorted(peers), data))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_exchange_file_descriptors(world):
    """What sparrowrecsys_amd.dist.ShardedTable needs from the host side: every rank's descriptor (there: the shareable handle of
    its shard of a row-sharded table, include/sparrow_hip.h sprk_vtable_export) open in every other rank -- here pipe ends, so each
    rank can prove it holds the peers' descriptors by writing through them."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fd_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, peers, data in results:
        assert peers == [r for r in range(world) if r != rank]
        assert sorted(data.decode().strip(";").split(";")) == sorted("%d->%d" % (r, rank) for r in range(world) if r != rank)

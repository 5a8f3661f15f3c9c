"""The score all-gather as direct peer writes (``sprk_peer_*``: k_peer_gather.h) on what a 1-GPU box can
run: a world of one (the rank is its own peer), and TWO processes sharing the device -- each maps the other's receive buffer
through its IPC handle, stores its slice into it and waits for the other's arrival flag, exactly the cross-GPU protocol with
the xGMI hop replaced by the shared memory system (``-m gpu``).  Unmeasured on a multi-GPU node."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "gpu tests need a HIP device"
    return t


def test_peer_allgather_world_of_one(torch):
    from sparrowrecsys_amd.dist import PeerScoreComm
    comm = PeerScoreComm(slot_floats=65536)
    assert comm.world == 1 and comm.rank == 0 and comm.memory_kind in ("uncached", "fine-grained", "default")
    print("receive buffer memory kind:", comm.memory_kind)
    side = torch.cuda.Stream()
    for step, n in enumerate([65536, 4096, 1, 7, 65533, 65536]):       # unaligned tails, both parities, a slot reused twice
        local = torch.arange(n, dtype=torch.float32, device="cuda") + 1000.0 * step
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            recv = comm.all_gather(local, side.cuda_stream)
            got = recv[0, :n].clone()
        comm.check(side.cuda_stream)
        assert recv.shape == (1, 65536)
        assert torch.equal(got, local)
    odd = torch.arange(40, dtype=torch.float32, device="cuda")[1:]       # a view that is not 16-byte aligned: scalar stores
    assert torch.equal(comm.all_gather(odd)[0, :39].clone(), odd)
    comm.check()
    with pytest.raises(ValueError):
        comm.all_gather(torch.zeros(65540, device="cuda"))
    with pytest.raises(ValueError):
        comm.all_gather(torch.zeros(16, device="cuda", dtype=torch.float64))
    comm.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    import traceback
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SPRK_PEER_TIMEOUT_MS"] = "1500" if world <= 2 else "8000"    # eight processes time-share the one device
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparrowrecsys_amd.dist import PeerScoreComm, RowShardedPredictor
        torch.cuda.set_device(0)                                 # both ranks share the box's one device
        slot = 8192
        comm = PeerScoreComm(slot_floats=slot)
        ok = True
        for step in range(7):                                    # both parities several times over
            n = [8192, 8192, 100, 8191, 8192, 4, 8192][step]
            local = torch.arange(n, dtype=torch.float32, device="cuda") + 10000.0 * rank + 100000.0 * step
            recv = comm.all_gather(local)
            got = recv[:, :n].clone()
            comm.check()
            for r in range(world):
                want = torch.arange(n, dtype=torch.float32, device="cuda") + 10000.0 * r + 100000.0 * step
                ok = ok and bool(torch.equal(got[r], want))
            if step == 2 and rank == 1:
                torch.cuda.synchronize()
                import time
                time.sleep(0.3)                                  # a slow rank: the other one waits on the device, not the host
        # the row-sharded predictor over it: every rank ends with all scores of the global batch, in row order
        n_rows = 3001
        ids = torch.arange(n_rows * 2, dtype=torch.int32, device="cuda").reshape(n_rows, 2)
        dense = torch.arange(n_rows * 3, dtype=torch.float32, device="cuda").reshape(n_rows, 3)
        fwd = lambda i, d: (i[:, 0].float() * 0.5 + d[:, 2]).contiguous()
        comm2 = PeerScoreComm(slot_floats=(n_rows + world - 1) // world)
        scores = RowShardedPredictor(fwd, comm=comm2).predict(ids, dense)
        comm2.check()
        ok_pred = bool(torch.equal(scores, fwd(ids, dense)))
        dist.barrier()
        # a peer that never sends: the device-side wait gives up at the deadline and the check reports it
        timed_out = None
        if rank == 0:
            comm.all_gather(torch.zeros(16, device="cuda"))
            try:
                comm.check()
                timed_out = False
            except RuntimeError as e:
                timed_out = "did not arrive" in str(e)
        dist.barrier()
        q.put((rank, ok, ok_pred, timed_out, comm.memory_kind, None))
    except Exception:
        q.put((rank, False, False, None, "", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_peer_allgather_processes_share_the_device(torch, world):
    """world = 8: the shape of the 8-GPU node -- every rank maps seven peers' buffers, stores into all of them
    and polls seven flags -- with the xGMI hops replaced by the one device's memory system."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=480) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, ok, ok_pred, timed_out, kind, tb in sorted(results):
        assert tb is None, tb
        print("rank", rank, "receive buffer:", kind)
        assert ok, "rank %d: gathered slices differ" % rank
        assert ok_pred, "rank %d: row-sharded predictor over peer writes differs" % rank
        if rank == 0:
            assert timed_out is True
    for p in procs:
        assert p.exitcode == 0


def _worker_sink(rank, world, port, q):
    """GroupedScoreGather + PeerScoreComm + a sink, one rank's sink slow (the sink must never see a peer's
    exchange e + 2 land in what it reads)."""
    import sys
    import time
    import traceback
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SPRK_PEER_TIMEOUT_MS"] = "4000"
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparrowrecsys_amd.dist import GroupedScoreGather, PeerScoreComm
        torch.cuda.set_device(0)
        B, G, n_batches = 1024, 4, 4 * 9 + 3                     # nine full groups and a ragged last one
        comm = PeerScoreComm(slot_floats=B * G)
        seen = []

        def sink(gi, view, nb):
            if rank == 1 and gi in (1, 2, 5):
                time.sleep(0.25)                                 # this rank falls a group behind its peer
            v = view.clone()
            torch.cuda.synchronize()
            good = True
            for r in range(world):
                for b in range(nb):
                    want = float(100000 * r + 100 * (gi * G + b))
                    good = good and bool((v[r, b] == want).all())
            seen.append((gi, nb, good))

        gs = GroupedScoreGather(B, G, torch.device("cuda", 0), sink=sink, comm=comm)
        assert gs.comm_stream is not None
        for i in range(n_batches):
            gs.out().fill_(float(100000 * rank + 100 * i))
            if gs.full():
                gs.commit()
        gs.flush()
        comm.check(gs.comm_stream.cuda_stream)
        dist.barrier()
        q.put((rank, seen, None))
    except Exception:
        q.put((rank, [], traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_grouped_gather_with_sink_over_peer_writes_survives_a_slow_rank(torch):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sink, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, seen, tb in sorted(results):
        assert tb is None, tb
        assert [g for g, _, _ in seen] == list(range(10)), "rank %d: groups seen %r" % (rank, seen)
        assert seen[-1][1] == 3
        assert all(good for _, _, good in seen), "rank %d: a sink saw foreign data: %r" % (rank, seen)
    for p in procs:
        assert p.exitcode == 0

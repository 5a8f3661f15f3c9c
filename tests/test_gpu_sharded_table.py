"""BASELINE config 4 read literally -- "DeepFM emb_dim=64, ... 27 M-row synthetic table, ROW-SHARDED across 8 x MI355X" -- on what a
one-GPU box can run (``-m gpu``): ``sparrowrecsys_amd.dist.ShardedTable`` (include/sparrow_hip.h ``sprk_vtable_*``: every rank's shard of
the table mapped into ONE virtual range, the fused kernels gather ``table[id]`` unchanged, a row another GPU owns is loaded over the
link between the two) with a world of one, and with 2 and 4 PROCESSES sharing the device -- each allocates only its rows, maps the
peers' allocations through file descriptors passed over Unix sockets, and scores a batch whose ids hit every shard: the same bits
as one process with the whole table.  ``sprk_upload_external`` (a slot that reads caller-owned memory) against ``sprk_upload``."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

V_ITEM, V_USER, D = 50_021, 3_001, 64                      # an item table that does not divide evenly, emb_dim 64 (config 4's width)
FIELDS = [("movieId", "id", V_ITEM), ("userId", "id", V_USER), ("userGenre1", "genre", 19), ("movieGenre1", "genre", 19)]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "gpu tests need a HIP device"
    return t


def _rows(torch, lo, hi, seed):
    """Rows [lo, hi) of the reference item table: a function of the GLOBAL row index, so every process can make its own shard."""
    i = torch.arange(lo, hi, dtype=torch.float32, device="cuda").unsqueeze(1)
    j = torch.arange(D, dtype=torch.float32, device="cuda").unsqueeze(0)
    return torch.sin(i * (0.37 + seed) + j * 1.3) * 0.05 + torch.cos(i * 0.011 * (1 + seed) - j) * 0.02


def _model_and_batch(torch, item_fm, item_deep, B=4099, V_ITEM=V_ITEM):
    from sparrowrecsys_amd import models as M, synthetic as SY
    FIELDS = [("movieId", "id", V_ITEM)] + globals()["FIELDS"][1:]
    small = M.DeepFM(seed=5, emb_dim=D, fields=[(k, kind, min(v, 64)) for k, kind, v in FIELDS], pairs=None)
    w = dict(small.weights)
    rng = np.random.default_rng(11)
    w["emb/userId"] = (rng.standard_normal((V_USER, D)) * 0.05).astype(np.float32)
    w["deep_emb/userId"] = (rng.standard_normal((V_USER, D)) * 0.05).astype(np.float32)
    w["emb/movieId"], w["deep_emb/movieId"] = item_fm, item_deep
    n_fo = sum(v for _, _, v in FIELDS)
    hk = (rng.standard_normal((n_fo + len(small.pairs) + small.hidden[-1], 1)) * 0.05).astype(np.float32)
    w["head/kernel"] = hk
    model = M.DeepFM(weights=w, emb_dim=D, fields=FIELDS, pairs=None)
    feats = SY.synth_fields(B, FIELDS, seed=77)
    feats["movieId"][:8] = [0, 1, V_ITEM - 1, V_ITEM - 2, V_ITEM // 2, V_ITEM // 2 + 1, V_ITEM // 4, 3 * V_ITEM // 4]   # every shard's edges
    return model, feats


def _reference_scores(torch):
    fm, deep = _rows(torch, 0, V_ITEM, 0), _rows(torch, 0, V_ITEM, 1)
    model, feats = _model_and_batch(torch, fm, deep)
    assert model.engine.describe()["kernel"].startswith("k_deepfm_pairs")
    out = model.predict(feats)[:, 0]
    model.engine.close()
    return out, feats, (fm, deep)


def test_external_upload_and_world_of_one(torch):
    from oracle import ctr_oracle as O
    from sparrowrecsys_amd.dist import ShardedTable
    from sparrowrecsys_amd.plan import DeviceTable
    ref, feats, (fm, deep) = _reference_scores(torch)
    # (1) the same tables handed over WITHOUT sprk_upload's copy
    model, _ = _model_and_batch(torch, DeviceTable.from_rows(fm), DeviceTable.from_rows(deep))
    np.testing.assert_array_equal(model.predict(feats)[:, 0], ref)
    model.engine.close()
    # (2) as row-sharded tables of a world of one
    t_fm, t_deep = ShardedTable(V_ITEM, D), ShardedTable(V_ITEM, D)
    assert t_fm.world == 1 and t_fm.owned_rows() == (0, V_ITEM) and t_fm.shard_rows >= V_ITEM + 1
    t_fm.fill_local(lambda lo, hi: _rows(torch, lo, hi, 0))
    t_deep.fill_local(lambda lo, hi: _rows(torch, lo, hi, 1))
    model, _ = _model_and_batch(torch, t_fm.table(), t_deep.table())
    got = model.predict(feats)[:, 0]
    np.testing.assert_array_equal(got, ref)
    # and the oracle, on numpy copies of the same weights
    w = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in model.weights.items()}
    want = O.deepfm_forward(feats, w, dtype=np.float64, fields=FIELDS, pairs=model.pairs)[:, 0]
    assert np.abs(got - want).max() <= 3e-5 and want.std() > 0.01
    bad = dict(feats)
    bad["movieId"] = feats["movieId"].copy()
    bad["movieId"][5] = V_ITEM                              # outside the table: flagged, no wild read into the mapped range
    with pytest.raises(ValueError):
        model.predict(bad)
    model.engine.close()
    t_fm.close()
    t_deep.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, V_ITEM=V_ITEM, deny=False):
    import sys
    import traceback
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)       # a hang shows WHERE (and costs 100 s of GPU time, not the queue's timeout)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sparrowrecsys_amd.dist import ShardedTable
        torch.cuda.set_device(0)                            # the ranks share the box's one device: the peers' HBM is the same HBM
        if deny:                                            # sprk_vtable_import's "the owner is a device this one cannot reach" exit
            os.environ["SPRK_TEST_VTABLE_PEER_DENY"] = "1"
            try:
                ShardedTable(V_ITEM, D)
                q.put((rank, None, None, 0, None, "ShardedTable() did not raise"))
            except RuntimeError as e:
                q.put((rank, str(e), None, 0, None, None))
            faulthandler.cancel_dump_traceback_later()
            return
        t_fm, t_deep = ShardedTable(V_ITEM, D), ShardedTable(V_ITEM, D)
        lo, hi = t_fm.owned_rows()
        t_fm.fill_local(lambda a, b: _rows(torch, a, b, 0))            # ONLY this rank's rows
        t_deep.fill_local(lambda a, b: _rows(torch, a, b, 1))
        torch.cuda.synchronize()
        dist.barrier()                                      # every shard is filled before anybody gathers from it
        model, feats = _model_and_batch(torch, t_fm.table(), t_deep.table(), V_ITEM=V_ITEM)
        got = model.predict(feats)[:, 0]
        owners = np.minimum(np.asarray(feats["movieId"]) // t_fm.shard_rows, world - 1)
        dist.barrier()                                      # nobody unmaps while a peer still scores
        model.engine.close()
        q.put((rank, got, (lo, hi), int(t_fm.shard_rows), np.bincount(owners, minlength=world).tolist(), None))
        t_fm.close()
        t_deep.close()
        faulthandler.cancel_dump_traceback_later()
    except Exception:
        q.put((rank, None, None, 0, None, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_row_sharded_table_between_processes(torch, world):
    import torch.multiprocessing as mp
    ref, feats, _ = _reference_scores(torch)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=150) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    covered = []
    for rank, got, span, shard_rows, hits, tb in sorted(results, key=lambda r: r[0]):
        assert tb is None, tb
        np.testing.assert_array_equal(got, ref)             # every rank: the same bits as one process holding the whole table
        assert all(h > 0 for h in hits), "the batch must gather from every shard: %r" % hits
        covered.append(span)
    assert covered[0][0] == 0 and covered[-1][1] == V_ITEM and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    for p in procs:
        assert p.exitcode == 0


def _run_world(torch, world, V, timeout=150, deny=False):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, V, deny)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=timeout) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    return sorted(results, key=lambda r: r[0]), procs


def test_config4_27m_row_table_sharded_over_two_processes(torch):
    """[r5, VERDICT r04 "missing" 5] BASELINE config 4 AT ITS STATED SIZE under N > 1: the 27 M x 64 item tables (FM part and deep part,
    6.9 GB each) row-sharded over two processes sharing the device -- each allocates 13.5 M rows of either table (3.5 GB), maps the
    peer's through the descriptor exchange, and scores a batch whose ids hit both shards and their edges -- against ONE process that
    holds both whole tables: the same bits.  (What the one-GPU box cannot show -- the peers' rows arriving over xGMI -- stays
    unmeasured: DESIGN section 9.)"""
    from sparrowrecsys_amd.plan import DeviceTable
    V = 27_000_000
    fm, deep = _rows(torch, 0, V, 0), _rows(torch, 0, V, 1)
    model, feats = _model_and_batch(torch, DeviceTable.from_rows(fm), DeviceTable.from_rows(deep), V_ITEM=V)
    del fm, deep
    assert model.engine.describe()["kernel"].startswith("k_deepfm_pairs") and model.engine.table_bytes() > 13.8e9
    ref = model.predict(feats)[:, 0]
    model.engine.close()
    del model
    torch.cuda.empty_cache()
    results, procs = _run_world(torch, 2, V, timeout=400)
    for rank, got, span, shard_rows, hits, tb in results:
        assert tb is None, tb
        np.testing.assert_array_equal(got, ref)
        assert all(h > 0 for h in hits) and shard_rows >= V // 2
    assert results[0][2][0] == 0 and results[0][2][1] == results[1][2][0] and results[1][2][1] == V
    for p in procs:
        assert p.exitcode == 0
    assert ref.std() > 0.01


def test_import_of_a_shard_this_device_cannot_reach_is_an_error_not_a_fault(torch):
    """[r5, VERDICT r04 next-round 5] sprk_vtable_import asks whose memory a descriptor is (hipMemGetAllocationPropertiesFromHandle) and
    whether this device can access that device as a peer; a shard it cannot reach is refused with both device numbers in the message,
    the handle released -- instead of mapping it and faulting at the first gather.  One GPU cannot produce the condition, so the test
    hook SPRK_TEST_VTABLE_PEER_DENY=1 walks that exit between two processes."""
    results, procs = _run_world(torch, 2, 4099, deny=True)
    for rank, msg, _span, _rows_, _hits, tb in results:
        assert tb is None, tb
        assert "cannot access as a peer" in msg and "device 0" in msg, msg
    for p in procs:
        assert p.exitcode == 0


def test_sharded_table_close_refuses_under_an_open_engine_and_the_table_is_collectable(torch):
    """[r5, ADVICE r04] (1) close() while an engine built on the table is open raises (an engine gathering from an unmapped range
    faults); after the engine is closed it succeeds.  (2) The views hold no reference back to the table, so dropping the last
    reference runs __del__ -- the reserved range and the shard are freed without an explicit close()."""
    import gc
    import weakref
    from sparrowrecsys_amd.dist import ShardedTable
    t_fm, t_deep = ShardedTable(V_ITEM, D), ShardedTable(V_ITEM, D)
    t_fm.fill_local(lambda lo, hi: _rows(torch, lo, hi, 0))
    t_deep.fill_local(lambda lo, hi: _rows(torch, lo, hi, 1))
    model, feats = _model_and_batch(torch, t_fm.table(), t_deep.table())
    model.predict(feats)
    with pytest.raises(RuntimeError, match="still open"):
        t_fm.close()
    model.engine.close()
    t_fm.close()                                            # now fine (the model object itself may live on)
    assert t_fm.handle is None
    del model
    ref = weakref.ref(t_deep)
    shard_bytes = t_deep.shard_rows * t_deep.Dp * 4
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    del t_deep
    gc.collect()
    assert ref() is None, "a reference cycle keeps the ShardedTable alive"
    assert torch.cuda.mem_get_info()[0] - free0 >= shard_bytes // 2   # (the shard went back to the device)

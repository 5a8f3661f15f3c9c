"""Multi-GPU: one process per GPU, batch rows sharded across ranks, tables and MLP weights
replicated, ONE collective on the data path -- an all-gather of the per-rank score slice
(the reference has no distributed path at all).  ``torch.distributed`` with
backend "nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" for the CPU tests.

Two shapes of the same exchange:
  * ``RowShardedPredictor``  one global batch -> shard -> forward -> all-gather (latency path, what a
    single ``/getrecforyou`` request needs);
  * ``ScoreComm`` / ``PeerScoreComm``  the exchange behind the C ABI: RCCL's all-gather, or one step of direct peer
    writes over the xGMI mesh (every rank stores its slice into all peers' receive buffers);
  * ``GroupedScoreGather``   the ``model.predict(dataset)`` loop: a forward of one 65 536-row shard takes
    ~9 us, an RCCL all-gather of its 256 KiB score slice costs more than that in launch latency alone, so
    scores of ``group`` consecutive batches are written into one ring slot and exchanged by ONE larger
    collective on a second stream while the next group is being scored (xGMI is point-to-point: fewer,
    larger messages per link).
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple


class ScoreComm:
    """The path's one collective behind the C ABI (``sprk_comm_*``, include/sparrow_hip.h): an RCCL all-gather of float32
    score slices enqueued on a HIP stream, without torch on the data path.  The 128-byte RCCL id is created on rank 0 and
    handed to the other ranks through the torch.distributed process group (host side, once)."""

    def __init__(self, group=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _lib as L
        self.lib = L.load_library()
        self._L, self._C = L, C
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            L.check(self.lib.sprk_comm_unique_id(buf))
        if self.world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = C.create_string_buffer(box[0], 128)
        self.handle = C.c_void_p()
        torch.cuda.synchronize()
        L.check(self.lib.sprk_comm_create(buf, self.rank, self.world, C.byref(self.handle)))

    def all_gather(self, local, gathered, stream=None):
        """local [count] float32 -> gathered [world * count] float32 on every rank (async on ``stream``)."""
        import torch
        C = self._C
        if local.dtype != torch.float32 or gathered.dtype != torch.float32 or not local.is_cuda or not gathered.is_cuda:
            raise ValueError("ScoreComm.all_gather: float32 device tensors only")
        if not local.is_contiguous() or not gathered.is_contiguous() or gathered.numel() != self.world * local.numel():
            raise ValueError("ScoreComm.all_gather: gathered must be contiguous with world * local.numel() elements")
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        self._L.check(self.lib.sprk_comm_allgather_scores(self.handle, C.c_void_p(local.data_ptr()), C.c_void_p(gathered.data_ptr()),
                                                         local.numel(), C.c_void_p(stream)))
        return gathered

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sprk_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevView:
    """A device buffer owned by the C library, seen by torch without a copy (``torch.as_tensor(view, device=...)``)."""

    def __init__(self, ptr: int, shape, owner):
        self.owner = owner                                      # keeps the communicator (and its buffer) alive
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2, "strides": None}


class PeerScoreComm:
    """The exchange as direct peer writes over xGMI (``sprk_peer_*``, include/sparrow_hip.h): every rank
    stores its score slice into all peers' receive buffers in one step and waits for their arrival flags -- two kernels on
    the caller's stream, no RCCL, no ring.  The 64-byte IPC handles of the receive buffers are exchanged once through the
    torch.distributed process group (host side).  ``slot_floats``: the largest slice a rank will ever send."""

    def __init__(self, slot_floats: int, group=None, device=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _lib as L
        self.lib = L.load_library()
        self._L, self._C = L, C
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.slot = (int(slot_floats) + 3) & ~3
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        hd = C.create_string_buffer(64)
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.sprk_peer_create(self.rank, self.world, self.slot, hd, C.byref(self.handle)))
        handles = [bytes(hd.raw)]
        if self.world > 1:
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(hd.raw), group=group)
        with torch.cuda.device(self.device):
            L.check(self.lib.sprk_peer_connect(self.handle, C.create_string_buffer(b"".join(handles), 64 * self.world)))
        if self.world > 1:
            dist.barrier(group=group)                           # nobody stores into a buffer its owner has not mapped / zeroed yet
        self.memory_kind = self.lib.sprk_peer_memory_kind(self.handle).decode()
        self._views = {}
        self._hip = None

    def all_gather(self, local, stream=None):
        """local [count <= slot] float32 (device) -> [world, slot] float32 view of this rank's receive buffer: row r holds rank
        r's slice in [:count].  Asynchronous on ``stream`` (default: torch's current stream); the view stays valid for work on
        that stream until the exchange after the next one."""
        import torch
        C = self._C
        if local.dtype != torch.float32 or not local.is_cuda or not local.is_contiguous():
            raise ValueError("PeerScoreComm.all_gather: a contiguous float32 device tensor is required")
        if local.numel() > self.slot:
            raise ValueError("PeerScoreComm.all_gather: %d scores exceed the slot of %d" % (local.numel(), self.slot))
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        out = C.c_void_p()
        self._L.check(self.lib.sprk_peer_allgather_scores(self.handle, C.c_void_p(local.data_ptr()), local.numel(), C.byref(out), C.c_void_p(stream)))
        if out.value not in self._views:
            try:
                self._views[out.value] = (torch.as_tensor(_DevView(out.value, (self.world, self.slot), self), device=self.device), False)
            except Exception:                                   # a torch build without __cuda_array_interface__ import: one device copy
                if self._hip is None:
                    self._hip = C.CDLL("libamdhip64.so")
                    self._hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
                self._views[out.value] = (torch.empty((self.world, self.slot), dtype=torch.float32, device=self.device), True)
        view, copy = self._views[out.value]
        if copy:
            rc = self._hip.hipMemcpyAsync(C.c_void_p(view.data_ptr()), out, view.numel() * 4, 3, C.c_void_p(stream))   # 3 = device to device
            if rc != 0:
                raise RuntimeError("hipMemcpyAsync of the receive buffer failed (%d)" % rc)
        return view

    def check(self, stream=None):
        """Synchronises ``stream`` and raises if a peer's slice did not arrive within the deadline."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self._L.check(self.lib.sprk_peer_check(self.handle, self._C.c_void_p(stream)))

    def close(self):
        if getattr(self, "handle", None):
            self._views = {}
            self.lib.sprk_peer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def exchange_fds(my_fd: int, group=None) -> dict:
    """Every rank of a NODE hands one open file descriptor to every other rank: {peer rank: the peer's descriptor, open in this
    process}.  Descriptors cross process boundaries only as SCM_RIGHTS ancillary data of a Unix-domain socket, so: rank 0 makes a
    private directory and broadcasts its name through the process group, every rank listens on <dir>/<rank>, a barrier, every
    rank connects to every peer and sends (its rank, its descriptor), then accepts world - 1 connections.  Works on any backend
    (the descriptors never touch the collective); CPU test: tests/test_dist_cpu.py over gloo with pipe descriptors."""
    import shutil
    import socket
    import tempfile

    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return {}
    box = [tempfile.mkdtemp(prefix="sprk_fds_") if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    d = box[0]
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    got = {}
    try:
        srv.bind(os.path.join(d, str(rank)))
        srv.listen(world)
        dist.barrier(group=group)                              # everybody listens
        for peer in range(world):
            if peer == rank:
                continue
            c = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            c.connect(os.path.join(d, str(peer)))
            socket.send_fds(c, [("%d" % rank).encode()], [my_fd])
            c.close()
        for _ in range(world - 1):
            conn, _addr = srv.accept()
            msg, fds, _flags, _a = socket.recv_fds(conn, 64, 1)
            conn.close()
            if len(fds) != 1:
                raise RuntimeError("exchange_fds: a peer sent %d descriptors" % len(fds))
            got[int(msg.decode())] = fds[0]
        dist.barrier(group=group)                              # nobody removes the directory under a peer that still connects
    finally:
        srv.close()
        if rank == 0:
            shutil.rmtree(d, ignore_errors=True)
    if sorted(got) != [r for r in range(world) if r != rank]:
        raise RuntimeError("exchange_fds: descriptors from ranks %s, expected all of the %d peers" % (sorted(got), world - 1))
    return got


class _DevView:
    """A raw device pointer as something torch.as_tensor can wrap without copying (__cuda_array_interface__ v3)."""

    def __init__(self, ptr: int, shape, owner=None):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None}
        self._owner = owner


class ShardedTable:
    """An embedding table of ``rows`` x ``dim`` float32 ROW-SHARDED over the ranks of a node, which every rank reads as ONE device
    array (include/sparrow_hip.h sprk_vtable_*: HIP virtual memory maps the peers' shards into one range; a row another GPU owns is
    LOADED over the xGMI link between the two by the same fused kernel that gathers a replicated table -- no all-to-all of ids and
    rows in front of the forward).  BASELINE config 4 read literally ("27 M-row table, row-sharded across 8 x MI355X").

    ``local``  this rank's rows as a torch view ``[shard_rows, Dp]`` -- fill it (a checkpoint shard, an initialiser); global row g
               lives on rank ``g // shard_rows`` at local row ``g % shard_rows``;
    ``table()``  the whole table as a ``plan.DeviceTable`` (device layout ``[rows + 1, Dp]``, the all-zero row at index ``rows``) to
               hand to a model as its ``emb/<key>`` weight.
    Collective: every rank of the group constructs it with the same geometry.  The table must outlive the engines built on it."""

    def __init__(self, rows: int, dim: int, group=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        from . import _lib as L
        from .plan import pad4
        self.lib, self._L, self._C = L.load_library(), L, C
        on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if on else 0
        self.world = dist.get_world_size(group) if on else 1
        self.rows, self.dim, self.Dp = int(rows), int(dim), pad4(int(dim))
        self.handle = C.c_void_p()
        L.check(self.lib.sprk_vtable_create(self.rows + 1, self.Dp * 4, self.world, self.rank, C.byref(self.handle)))
        try:
            if self.world > 1:
                fd = C.c_int32(-1)
                L.check(self.lib.sprk_vtable_export(self.handle, C.byref(fd)))
                peers = exchange_fds(int(fd.value), group)
                for peer, pfd in sorted(peers.items()):
                    try:
                        L.check(self.lib.sprk_vtable_import(self.handle, peer, pfd))
                    finally:
                        os.close(pfd)                              # (the import holds its own reference)
            base, srows, mapped = C.c_void_p(), C.c_int64(), C.c_int32()
            L.check(self.lib.sprk_vtable_info(self.handle, C.byref(base), C.byref(srows), C.byref(mapped)))
            if mapped.value != self.world:
                raise RuntimeError("ShardedTable: %d of %d shards mapped" % (mapped.value, self.world))
        except Exception:
            self.close()
            raise
        self.base, self.shard_rows = int(base.value), int(srows.value)
        dev = torch.device("cuda", torch.cuda.current_device())
        # [r5, ADVICE r04] The views hold NO reference to this object: torch keeps the array-interface object alive on the C++ side, where
        # Python's collector cannot see it, and an owner stored there closed a cycle (table -> tensor -> view -> table) that kept
        # __del__ -- and with it the multi-GB shard, the reserved range and the exported descriptor -- from ever running.  What keeps
        # the mapping alive under an engine is DeviceTable.keepalive (table()); ``local`` / the tensors it hands out are only valid
        # while the table is.
        self.local = torch.as_tensor(_DevView(self.base + self.rank * self.shard_rows * self.Dp * 4, (self.shard_rows, self.Dp)), device=dev)
        self._full = torch.as_tensor(_DevView(self.base, (self.rows + 1, self.Dp)), device=dev)
        import weakref
        self._handed = weakref.WeakSet()                        # the DeviceTables handed to models (and through them to engines)

    def owned_rows(self) -> Tuple[int, int]:
        """Global rows [lo, hi) that live in this rank's shard (clipped to the table)."""
        lo = min(self.rank * self.shard_rows, self.rows)
        return lo, min(lo + self.shard_rows, self.rows)

    def fill_local(self, rows_lo_hi_to_values: Callable):
        """``local[: hi - lo, : dim] = f(lo, hi)`` for this rank's global row range -- f returns a [hi - lo, dim] CUDA tensor."""
        lo, hi = self.owned_rows()
        if hi > lo:
            self.local[:hi - lo, :self.dim] = rows_lo_hi_to_values(lo, hi)

    def table(self):
        from .plan import DeviceTable
        t = DeviceTable(self._full, self.rows, self.dim, keepalive=self)
        self._handed.add(t)
        return t

    def close(self, force: bool = False):
        """Unmaps every shard and frees this rank's.  Refuses (RuntimeError) while an engine built on a ``table()`` of this object is
        still open, unless ``force``: an engine gathering from an unmapped range faults.  Close the engines first."""
        if getattr(self, "handle", None):
            alive = sum(len(t.engines) for t in getattr(self, "_handed", ()))
            if alive and not force:
                raise RuntimeError("ShardedTable.close(): %d engine(s) built on this table are still open; close them first "
                                   "(or close(force=True))" % alive)
            self.local = self._full = None
            self.lib.sprk_vtable_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close(force=True)                              # (unreachable while a DeviceTable.keepalive points here)
        except Exception:
            pass


def shard_bounds(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) owned by ``rank``: contiguous, sizes differ by at most one, earlier ranks larger."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RowShardedPredictor:
    """Scores a global batch that every rank holds (or can slice): each rank runs ``forward`` on
    its row shard only, then the score slices are all-gathered so every rank ends with all scores
    (what the Jetty ranker needs to sort the candidates, RecForYouProcess.java:89-91)."""

    def __init__(self, forward: Callable, group=None, comm: Optional["ScoreComm"] = None):
        import torch.distributed as dist
        self.forward = forward
        self.group = group
        self.comm = comm                     # ScoreComm (RCCL) / PeerScoreComm (peer writes): the exchange through the C ABI
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._bufs = {}                      # (rows, dtype, device) -> (slot, gathered): no allocation per request

    def predict(self, ids, dense):
        """Scores of all ``n`` rows, in row order, on every rank.  The result is a FRESH tensor (the staging and receive
        buffers are reused across calls; the copy is 4 bytes per row)."""
        import torch
        import torch.distributed as dist
        n = int(ids.shape[0])
        lo, hi = shard_bounds(n, self.rank, self.world)
        local = self.forward(ids[lo:hi].contiguous(), dense[lo:hi].contiguous())
        if self.world == 1:
            return local
        per = (n + self.world - 1) // self.world                 # slot size (largest shard)
        key = (n, local.dtype, local.device)
        if key not in self._bufs:
            if len(self._bufs) > 8:
                self._bufs.clear()
            self._bufs[key] = (torch.zeros(per, dtype=local.dtype, device=local.device),
                               torch.empty(per * self.world, dtype=local.dtype, device=local.device))
        slot, gathered = self._bufs[key]
        slot[:hi - lo] = local
        if isinstance(self.comm, PeerScoreComm) and local.is_cuda:
            recv = self.comm.all_gather(slot)                    # [world, comm.slot] view of the receive buffer
            if n % self.world == 0 and recv.shape[1] == per:
                return recv.reshape(-1).clone()                  # peers overwrite the receive buffer two exchanges later
            return torch.cat([recv[r, :shard_bounds(n, r, self.world)[1] - shard_bounds(n, r, self.world)[0]] for r in range(self.world)])
        if self.comm is not None and local.is_cuda:
            self.comm.all_gather(slot, gathered)
        else:
            dist.all_gather_into_tensor(gathered, slot, group=self.group)
        if n % self.world == 0:
            return gathered.clone()                              # `gathered` is reused by the next predict() of the same size
        pieces = []
        for r in range(self.world):
            rlo, rhi = shard_bounds(n, r, self.world)
            pieces.append(gathered[r * per:r * per + (rhi - rlo)])
        return torch.cat(pieces)


def all_gather_scores(local, gathered=None, group=None):
    """Equal-sized score slices -> the concatenated vector on every rank (one RCCL all-gather)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty(local.numel() * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    return gathered


class GroupedScoreGather:
    """Double-buffered ring of score slots for a predict-over-batches loop.

    ``out(i)`` is the [B] tensor the forward of this rank's i-th batch must write.  After every
    ``group`` batches call ``commit()``: the [group*B] slot is all-gathered into ``gathered[slot]``
    ([world, group*B]) on a side stream (CUDA) while the compute stream goes on with the other slot;
    ``flush()`` exchanges a partial last group and waits for everything.  Completed groups are
    handed to ``sink(group_index, gathered_view, n_batches)`` if given (views are only valid until the
    slot is reused two groups later).  With a ``PeerScoreComm`` and NO sink the groups stay in the communicator's receive
    buffer and ``gathered`` is not filled."""

    def __init__(self, batch_rows: int, group: int, device, dtype=None, pg=None, sink: Optional[Callable] = None,
                 comm: Optional["ScoreComm"] = None):
        import torch
        import torch.distributed as dist
        if group < 1:
            raise ValueError("group must be >= 1")
        self.B, self.G, self.pg, self.sink = int(batch_rows), int(group), pg, sink
        self.comm = comm                      # ScoreComm: collectives through sprk_comm_allgather_scores (RCCL behind the C ABI)
        self.world = dist.get_world_size(pg) if dist.is_initialized() else 1
        dtype = dtype or torch.float32
        self.local = torch.zeros((2, self.G, self.B), dtype=dtype, device=device)
        self.gathered = torch.zeros((2, self.world, self.G * self.B), dtype=dtype, device=device)
        self.cuda = torch.device(device).type == "cuda"
        # SPRK_FORCE_COLLECTIVE=1: issue the collective even in a world of one (functional check of the RCCL path)
        self.force = dist.is_initialized() and os.environ.get("SPRK_FORCE_COLLECTIVE") == "1"
        self.comm_stream = torch.cuda.Stream(device=device) if (self.cuda and (self.world > 1 or self.force)) else None
        self.done = [None, None]              # CUDA event per slot: its collective has finished
        self.slot, self.fill, self.groups_done, self.collectives = 0, 0, 0, 0
        self._pending = [None, None]          # (group_index, n_batches) waiting for the sink
        self._views = [None, None]            # cached per-slot output views (group_outs)
        if isinstance(comm, PeerScoreComm) and comm.slot != self.G * self.B:
            raise ValueError("PeerScoreComm slot of %d floats != group * batch_rows = %d" % (comm.slot, self.G * self.B))

    def out(self):
        """Tensor for the NEXT batch's scores (call once per batch, before its forward is enqueued)."""
        import torch
        if self.fill == 0 and self.done[self.slot] is not None:
            # this slot's previous collective must have read it before it is overwritten
            torch.cuda.current_stream().wait_event(self.done[self.slot])
            self._drain(self.slot)
        t = self.local[self.slot, self.fill]
        self.fill += 1
        return t

    def full(self) -> bool:
        return self.fill == self.G

    def group_outs(self):
        """All G output tensors of the CURRENT (empty) slot at once -- the whole-group form of G calls of ``out()``,
        for loops that enqueue a group with one foreign call; the views are created once per slot and reused (a tensor
        view costs microseconds of host time, the forward it belongs to ~7 us of GPU time).  Follow with ``commit()``."""
        import torch
        if self.fill != 0:
            raise RuntimeError("group_outs() needs an empty slot (fill = %d)" % self.fill)
        if self.done[self.slot] is not None:
            torch.cuda.current_stream().wait_event(self.done[self.slot])
            self._drain(self.slot)
        if self._views[self.slot] is None:
            self._views[self.slot] = [self.local[self.slot, i] for i in range(self.G)]
        self.fill = self.G
        return self._views[self.slot]

    def _drain(self, slot):
        if self._pending[slot] is not None and self.sink is not None:
            gi, nb = self._pending[slot]
            if self.cuda and self.done[slot] is not None:
                self.done[slot].synchronize()
            self.sink(gi, self.gathered[slot].view(self.world, self.G, self.B)[:, :nb], nb)
        self._pending[slot] = None

    def commit(self):
        """Exchange the current slot (``fill`` batches of it) and switch to the other one."""
        import torch
        import torch.distributed as dist
        if self.fill == 0:
            return
        slot, nb = self.slot, self.fill
        src = self.local[slot].view(-1)
        dst = self.gathered[slot].view(-1)
        if self.world == 1 and not self.force:
            dst.copy_(src)
            if self.cuda:
                # the slot is reused two groups later: without an event here out()/group_outs() would never drain it
                # and the sink would only ever see the last two groups
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.done[slot] = ev
        elif self.comm_stream is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ready)
            with torch.cuda.stream(self.comm_stream):
                if isinstance(self.comm, PeerScoreComm):
                    recv = self.comm.all_gather(src, self.comm_stream.cuda_stream)   # two parities = the ring's two slots
                    if self.sink is not None:
                        # The receive buffer is only protected against peers for readers enqueued on the exchange stream
                        # BEFORE this rank launches its next exchange (k_peer_gather.h): a sink running on the host after
                        # commit() of the next group could see a peer's exchange e + 2 land in it.  So the group is copied
                        # out here, on the exchange stream, ahead of `done` -- the sink reads `gathered[slot]`, which only
                        # this rank writes.  Without a sink nobody reads the group and the copy is skipped.
                        self.gathered[slot].copy_(recv[:, :self.G * self.B])
                elif self.comm is not None:
                    self.comm.all_gather(src, dst, self.comm_stream.cuda_stream)
                else:
                    dist.all_gather_into_tensor(dst, src, group=self.pg)
                ev = torch.cuda.Event()
                ev.record(self.comm_stream)
            self.done[slot] = ev
        else:
            dist.all_gather_into_tensor(dst, src, group=self.pg)
        self.collectives += 1
        self._pending[slot] = (self.groups_done, nb)
        self.groups_done += 1
        self.slot, self.fill = 1 - slot, 0
        if not self.cuda:
            self._drain(slot)

    def flush(self):
        """Exchange a partial group, wait for every outstanding collective, deliver to the sink."""
        import torch
        self.commit()
        for slot in (self.slot, 1 - self.slot):       # older slot first
            if self.cuda and self.done[slot] is not None:
                torch.cuda.current_stream().wait_event(self.done[slot])
            self._drain(slot)
        if self.cuda:
            torch.cuda.current_stream().synchronize()

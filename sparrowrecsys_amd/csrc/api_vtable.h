// api_vtable.h -- C ABI: an embedding table ROW-SHARDED over the ranks of a node that every rank sees as ONE table (sprk_vtable_*), and
// sprk_upload_external (a plan slot that reads caller-owned device memory: no copy of a 6.9 GB table).
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
//
// BASELINE config 4 reads "DeepFM emb_dim=64, 138 k-movie x 27 M-row synthetic table, row-sharded across 8 x MI355X"; SURVEY.md 8(e)
// prices the textbook form -- all-to-all of ids, gather at the owners, all-to-all of rows, two collectives and two extra passes over the
// rows in front of every forward -- and defers it.  On this hardware the table does not have to be exchanged at all: xGMI is a
// point-to-point mesh a kernel can LOAD through.  HIP's virtual-memory API lets a process reserve one contiguous virtual range and
// map physical allocations of OTHER processes into it (hipMemCreate -> hipMemExportToShareableHandle as a POSIX fd ->
// hipMemImportFromShareableHandle -> hipMemMap -> hipMemSetAccess; scripts/ubench/vmm_probe.hip proved it on this stack).  So: rank r
// allocates rows [r S, (r + 1) S) of the table (S = rows per rank), every rank maps every shard at byte r S row_bytes of ITS range,
// and the fused kernels gather table[id] exactly as they do from a replicated table -- a row another GPU owns arrives over the link
// that connects the two, no collective, no staging pass, no kernel change.  What is exchanged once, at set-up, is world - 1 file
// descriptors per rank (SCM_RIGHTS over a Unix socket: sparrowrecsys_amd/dist.py ShardedTable).

struct sprk_vtable_s {
    int world = 1, rank = 0, device = 0;
    long long rows_total = 0, shard_rows = 0;
    size_t row_bytes = 0, shard_bytes = 0;
    void* base = nullptr;                                  // world * shard_bytes of reserved virtual addresses
    std::vector<hipMemGenericAllocationHandle_t> handle;  // [world]: this rank's own allocation + the imported ones
    std::vector<char> mapped;                              // [world]
    int export_fd = -1;
};

extern "C" {

int sprk_vtable_create(int64_t rows_total, int32_t row_bytes, int32_t world, int32_t rank, sprk_vtable* out) {
    if (!out) return fail(SPRK_EINVAL, "out is NULL");
    *out = nullptr;
    if (rows_total <= 0 || row_bytes <= 0 || (row_bytes & 15) || world < 1 || world > 64 || rank < 0 || rank >= world)
        return fail(SPRK_EINVAL, "bad table geometry: %lld rows of %d bytes, rank %d of %d", (long long)rows_total, row_bytes, rank, world);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
    size_t gran = 0;
    HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = 4096;
    // rows per rank: the same on every rank, and a whole number of allocation granules so that shard r + 1 starts where shard r ends
    size_t g = gran, b = (size_t)row_bytes;
    while (b) { const size_t t = g % b; g = b; b = t; }    // gcd(gran, row_bytes)
    const long long rows_mult = (long long)(gran / g);
    long long shard_rows = (rows_total + world - 1) / world;
    shard_rows = (shard_rows + rows_mult - 1) / rows_mult * rows_mult;
    sprk_vtable_s* v = new (std::nothrow) sprk_vtable_s();
    if (!v) return fail(SPRK_EHIP, "out of host memory");
    v->world = world; v->rank = rank; v->device = dev; v->rows_total = rows_total; v->shard_rows = shard_rows;
    v->row_bytes = (size_t)row_bytes; v->shard_bytes = (size_t)shard_rows * row_bytes;
    v->handle.assign(world, hipMemGenericAllocationHandle_t{});
    v->mapped.assign(world, 0);
    auto bail = [&](hipError_t e, const char* what) { const int rc = fail(SPRK_EHIP, "%s failed: %s", what, hipGetErrorString(e)); sprk_vtable_destroy(v); return rc; };
    hipError_t e = hipMemAddressReserve(&v->base, v->shard_bytes * world, 0, nullptr, 0);
    if (e != hipSuccess) { v->base = nullptr; return bail(e, "hipMemAddressReserve"); }
    e = hipMemCreate(&v->handle[rank], v->shard_bytes, &prop, 0);
    if (e != hipSuccess) return bail(e, "hipMemCreate (this rank's rows)");
    char* mine = (char*)v->base + (size_t)rank * v->shard_bytes;
    e = hipMemMap(mine, v->shard_bytes, 0, v->handle[rank], 0);
    if (e != hipSuccess) { (void)hipMemRelease(v->handle[rank]); return bail(e, "hipMemMap"); }
    v->mapped[rank] = 1;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(mine, v->shard_bytes, &acc, 1);
    if (e != hipSuccess) return bail(e, "hipMemSetAccess");
    e = hipMemset(mine, 0, v->shard_bytes);               // rows past rows_total (and the all-zero row V of the kernels' layout) read as zeros
    if (e != hipSuccess) return bail(e, "hipMemset");
    *out = v;
    return SPRK_OK;
}

int sprk_vtable_export(sprk_vtable v, int32_t* fd_out) {
    if (!v || !fd_out) return fail(SPRK_EINVAL, "NULL argument");
    if (v->export_fd < 0) {
        int fd = -1;
        HIP_TRY(hipMemExportToShareableHandle(&fd, v->handle[v->rank], hipMemHandleTypePosixFileDescriptor, 0));
        v->export_fd = fd;
    }
    *fd_out = v->export_fd;
    return SPRK_OK;
}

int sprk_vtable_import(sprk_vtable v, int32_t peer_rank, int32_t fd) {
    if (!v) return fail(SPRK_EINVAL, "table is NULL");
    if (peer_rank < 0 || peer_rank >= v->world || peer_rank == v->rank) return fail(SPRK_EINVAL, "peer rank %d (this is rank %d of %d)", peer_rank, v->rank, v->world);
    if (v->mapped[peer_rank]) return fail(SPRK_ESTATE, "rank %d's rows are mapped already", peer_rank);
    if (fd < 0) return fail(SPRK_EINVAL, "bad file descriptor");
    const char* form = "pointer";
    {
        // The osHandle argument of hipMemImportFromShareableHandle: CUDA's convention -- and this image's ROCm 7.2 runtime -- is the
        // descriptor's VALUE cast to a pointer; the ROCm 7.0 runtime that PyTorch 2.10 bundles (and that every process which imported
        // torch therefore runs on) reads the descriptor THROUGH the pointer, and given the value it dereferences address 26.
        // [r5, ADVICE r04] The SAFE form goes first on every runtime: a pointer's bits taken as a descriptor are an error return (EBADF),
        // never a wild read.  The value form is tried only after that error, and only on a runtime (>= 7.2) known to take it.
        int ver = 0;
        (void)hipRuntimeGetVersion(&ver);
        int fdv = fd;
        hipError_t e = hipMemImportFromShareableHandle(&v->handle[peer_rank], &fdv, hipMemHandleTypePosixFileDescriptor);
        if (e != hipSuccess && ver >= 70200000) {
            (void)hipGetLastError();
            form = "value";
            e = hipMemImportFromShareableHandle(&v->handle[peer_rank], (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor);
        }
        if (e != hipSuccess) return fail(SPRK_EHIP, "hipMemImportFromShareableHandle (rank %d's rows, HIP runtime %d, last form tried: descriptor by %s) failed: %s", peer_rank, ver, form, hipGetErrorString(e));
    }
    {
        // [r5] Whose memory is it?  A shard that lives on ANOTHER device is only usable if this device can load through the link: say so
        // here, with both device numbers, instead of faulting at the first gather.  (SPRK_TEST_VTABLE_PEER_DENY=1 makes the one-GPU test
        // suite walk this exit.)
        hipMemAllocationProp pp = {};
        int owner = v->device;
        if (hipMemGetAllocationPropertiesFromHandle(&pp, v->handle[peer_rank]) == hipSuccess && pp.location.type == hipMemLocationTypeDevice) owner = pp.location.id;
        else (void)hipGetLastError();
        int can = 1;
        if (owner != v->device && hipDeviceCanAccessPeer(&can, v->device, owner) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        const char* deny = getenv("SPRK_TEST_VTABLE_PEER_DENY");
        if (deny && deny[0] == '1') can = 0;
        if (!can) {
            (void)hipMemRelease(v->handle[peer_rank]);
            return fail(SPRK_EHIP, "rank %d's rows live on device %d, which device %d cannot access as a peer (no xGMI / PCIe peer path): "
                                   "the row-sharded table needs peer access between every pair of ranks", peer_rank, owner, v->device);
        }
    }
    char* at = (char*)v->base + (size_t)peer_rank * v->shard_bytes;
    hipError_t e = hipMemMap(at, v->shard_bytes, 0, v->handle[peer_rank], 0);
    if (e != hipSuccess) { (void)hipMemRelease(v->handle[peer_rank]); return fail(SPRK_EHIP, "hipMemMap (rank %d's rows) failed: %s", peer_rank, hipGetErrorString(e)); }
    v->mapped[peer_rank] = 1;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = v->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(at, v->shard_bytes, &acc, 1);
    if (e != hipSuccess) {
        (void)hipMemUnmap(at, v->shard_bytes); (void)hipMemRelease(v->handle[peer_rank]); v->mapped[peer_rank] = 0;
        return fail(SPRK_EHIP, "hipMemSetAccess (rank %d's rows for device %d; descriptor imported by %s) failed: %s", peer_rank, v->device, form, hipGetErrorString(e));
    }
    return SPRK_OK;
}

int sprk_vtable_info(sprk_vtable v, void** base, int64_t* shard_rows, int32_t* ranks_mapped) {
    if (!v) return fail(SPRK_EINVAL, "table is NULL");
    if (base) *base = v->base;
    if (shard_rows) *shard_rows = v->shard_rows;
    if (ranks_mapped) { int n = 0; for (char m : v->mapped) n += m ? 1 : 0; *ranks_mapped = n; }
    return SPRK_OK;
}

void sprk_vtable_destroy(sprk_vtable v) {
    if (!v) return;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < v->world; ++r) {
        if (v->base && v->mapped[r]) {
            (void)hipMemUnmap((char*)v->base + (size_t)r * v->shard_bytes, v->shard_bytes);
            (void)hipMemRelease(v->handle[r]);
        }
    }
    if (v->export_fd >= 0) (void)close(v->export_fd);
    if (v->base) (void)hipMemAddressFree(v->base, v->shard_bytes * v->world);
    delete v;
}

int sprk_upload_external(sprk_handle h, int32_t slot, const void* dev_ptr, size_t bytes) {
    if (!h || !dev_ptr || bytes == 0) return fail(SPRK_EINVAL, "bad upload arguments");
    if (slot < 0 || slot >= h->plan.n_slots) return fail(SPRK_EINVAL, "slot %d outside [0,%d)", slot, h->plan.n_slots);
    if (h->finalized) return fail(SPRK_ESTATE, "upload after finalize");
    if (((uintptr_t)dev_ptr) & 15) return fail(SPRK_EINVAL, "external table must be 16-byte aligned");
    hipPointerAttribute_t attr;
    const hipError_t e = hipPointerGetAttributes(&attr, dev_ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); }      // (a virtual-memory range may not be known to the pointer query on every runtime: accepted)
    else if (attr.type == hipMemoryTypeHost) return fail(SPRK_EINVAL, "external table must be device memory");
    if (h->slot_ptr[slot] && !h->slot_external[slot]) (void)hipFree(h->slot_ptr[slot]);
    h->slot_ptr[slot] = const_cast<void*>(dev_ptr);
    h->slot_bytes[slot] = bytes;
    h->slot_external[slot] = 1;
    return SPRK_OK;
}

}  // extern "C"

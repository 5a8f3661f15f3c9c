// api_comm.h -- C ABI: the score all-gather over RCCL (sprk_comm_*) and as direct peer writes (sprk_peer_*).
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.

// ---------------------------------------------------------------------------------------------
// Multi-GPU: the ONE collective of the path behind the C ABI -- an all-gather of the per-rank score
// slices over RCCL (xGMI), enqueued on the caller's HIP stream.  RCCL is bound at run time (dlopen), so libsparrow_hip.so has
// no link-time dependency on it and single-GPU users never load it.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclUid { char b[SPRK_COMM_ID_BYTES]; };           // ncclUniqueId: 128 opaque bytes, passed by value
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
    if (g_rccl.lib) return SPRK_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) return fail(SPRK_EHIP, "cannot load RCCL (librccl.so.1): %s", dlerror());
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather) return fail(SPRK_EHIP, "librccl.so lacks the nccl entry points");
    g_rccl = a;
    return SPRK_OK;
}
int rccl_fail(const char* what, int rc) {
    return fail(SPRK_EHIP, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error");
}
}  // namespace

struct sprk_comm_s {
    void* comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int sprk_comm_unique_id(uint8_t id[SPRK_COMM_ID_BYTES]) {
    if (!id) return fail(SPRK_EINVAL, "id is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    RcclUid uid;
    const int nrc = g_rccl.GetUniqueId(&uid);
    if (nrc) return rccl_fail("ncclGetUniqueId", nrc);
    memcpy(id, uid.b, SPRK_COMM_ID_BYTES);
    return SPRK_OK;
}

int sprk_comm_create(const uint8_t id[SPRK_COMM_ID_BYTES], int32_t rank, int32_t world, sprk_comm* out) {
    if (!id || !out) return fail(SPRK_EINVAL, "id/out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(SPRK_EINVAL, "bad rank/world %d/%d", rank, world);
    int rc = rccl_load();
    if (rc) return rc;
    sprk_comm_s* c = new (std::nothrow) sprk_comm_s();
    if (!c) return fail(SPRK_EHIP, "out of host memory");
    c->rank = rank; c->world = world;
    RcclUid uid;
    memcpy(uid.b, id, SPRK_COMM_ID_BYTES);
    const int nrc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (nrc) { delete c; return rccl_fail("ncclCommInitRank", nrc); }
    *out = c;
    return SPRK_OK;
}

int sprk_comm_allgather_scores(sprk_comm c, const float* local, float* gathered, size_t count, void* stream) {
    RoctxRange roctx_range_("sprk_comm_allgather_scores");
    if (!c || !c->comm) return fail(SPRK_EINVAL, "communicator is NULL");
    if (!local || !gathered) return fail(SPRK_EINVAL, "NULL buffer");
    if (count == 0) return SPRK_OK;
    const int nrc = g_rccl.AllGather(local, gathered, count, 7 /* ncclFloat32 */, c->comm, (hipStream_t)stream);
    return nrc ? rccl_fail("ncclAllGather", nrc) : SPRK_OK;
}

void sprk_comm_destroy(sprk_comm c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

// ---- the same exchange as direct peer writes (k_peer_gather.h) ----
}  // extern "C"

struct sprk_peer_s {
    int rank = 0, world = 1;
    size_t slot = 0;                      // floats per rank slot
    void* base = nullptr;                 // [2][world][slot] floats | [2][world] flags: one allocation, exported by IPC
    size_t flags_off = 0;
    const char* mem_kind = "";
    void* peer_base[PEER_MAX_WORLD] = {};
    unsigned* done = nullptr;             // [world] local workgroup counters
    int* err = nullptr;
    unsigned epoch = 0;
    bool connected = false;
    unsigned long long ticks = 200000000ull;   // 2 s of the 100 MHz wall clock
};

extern "C" {

int sprk_peer_create(int32_t rank, int32_t world, size_t slot_floats, uint8_t handle_out[SPRK_PEER_HANDLE_BYTES], sprk_peer* out) {
    static_assert(sizeof(hipIpcMemHandle_t) == SPRK_PEER_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    if (!out || !handle_out) return fail(SPRK_EINVAL, "NULL argument");
    *out = nullptr;
    if (world < 1 || world > PEER_MAX_WORLD || rank < 0 || rank >= world) return fail(SPRK_EINVAL, "bad rank/world %d/%d (world <= %d)", rank, world, PEER_MAX_WORLD);
    if (slot_floats == 0 || (slot_floats & 3)) return fail(SPRK_EINVAL, "slot_floats must be a positive multiple of 4 (16-byte stores)");
    sprk_peer_s* c = new sprk_peer_s;
    c->rank = rank; c->world = world; c->slot = slot_floats;
    c->flags_off = (2 * (size_t)world * slot_floats * sizeof(float) + 255) & ~(size_t)255;
    const size_t bytes = c->flags_off + 2 * (size_t)world * sizeof(unsigned);
    // peers store into this buffer over xGMI while kernels of this device poll and read it: fine-grained (uncached) memory where
    // the runtime can export it by IPC, plain device memory otherwise (same-device peers share the L2)
    const struct { unsigned flag; const char* name; } kinds[] = {
        {hipDeviceMallocUncached, "uncached"}, {hipDeviceMallocFinegrained, "fine-grained"}, {hipDeviceMallocDefault, "default"}};
    hipIpcMemHandle_t hd;
    const char* forced = getenv("SPRK_PEER_MEM");             // "default" | "fine-grained" | "uncached": pin the kind (experiments)
    for (const auto& k : kinds) {
        if (forced && strcmp(forced, k.name) != 0) continue;
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, k.flag) != hipSuccess || !p) { (void)hipGetLastError(); continue; }
        if (world == 1) memset(&hd, 0, sizeof(hd));               // a world of one exports nothing
        else if (hipIpcGetMemHandle(&hd, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); continue; }
        c->base = p; c->mem_kind = k.name;
        break;
    }
    if (!c->base) { delete c; return fail(SPRK_EHIP, "cannot allocate an IPC-exportable receive buffer of %zu bytes", bytes); }
    if (hipMemset(c->base, 0, bytes) != hipSuccess || hipMalloc((void**)&c->done, PEER_MAX_WORLD * sizeof(unsigned)) != hipSuccess ||
        hipMemset(c->done, 0, PEER_MAX_WORLD * sizeof(unsigned)) != hipSuccess || hipMalloc((void**)&c->err, sizeof(int)) != hipSuccess ||
        hipMemset(c->err, 0, sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(c->base); if (c->done) (void)hipFree(c->done); if (c->err) (void)hipFree(c->err);
        delete c;
        return fail(SPRK_EHIP, "peer buffer setup failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (const char* t = getenv("SPRK_PEER_TIMEOUT_MS")) { const long ms = atol(t); if (ms > 0) c->ticks = (unsigned long long)ms * 100000ull; }
    memcpy(handle_out, &hd, SPRK_PEER_HANDLE_BYTES);
    c->peer_base[rank] = c->base;
    *out = c;
    return SPRK_OK;
}

int sprk_peer_connect(sprk_peer c, const uint8_t* handles) {
    if (!c || !handles) return fail(SPRK_EINVAL, "NULL argument");
    if (c->connected) return fail(SPRK_ESTATE, "peer communicator is already connected");
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t hd;
        memcpy(&hd, handles + (size_t)p * SPRK_PEER_HANDLE_BYTES, sizeof(hd));
        void* q = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&q, hd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess || !q) { (void)hipGetLastError(); return fail(SPRK_EHIP, "hipIpcOpenMemHandle of rank %d's buffer: %s", p, hipGetErrorString(e)); }
        c->peer_base[p] = q;
    }
    c->connected = true;
    return SPRK_OK;
}

int sprk_peer_allgather_scores(sprk_peer c, const float* local, size_t count, const float** gathered, void* stream) {
    RoctxRange roctx_range_("sprk_peer_allgather_scores");
    if (!c || !local || !gathered) return fail(SPRK_EINVAL, "NULL argument");
    if (!c->connected) return fail(SPRK_ESTATE, "peer all-gather before sprk_peer_connect");
    if (count > c->slot) return fail(SPRK_EINVAL, "count %zu exceeds the slot of %zu floats", count, c->slot);
    const unsigned e = ++c->epoch;
    const int parity = (int)(e & 1);
    *gathered = (const float*)c->base + (size_t)parity * c->world * c->slot;
    PeerPut a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < c->world; ++p) {
        a.dst[p] = (float*)c->peer_base[p] + ((size_t)parity * c->world + c->rank) * c->slot;
        a.flag[p] = (unsigned*)((char*)c->peer_base[p] + c->flags_off) + parity * c->world + c->rank;
    }
    a.src = local; a.count = count; a.epoch = e; a.world = c->world; a.done = c->done;
    long long bpp = ((long long)count / 4 + 1023) / 1024;      // ~4 sixteen-byte stores per thread
    if (bpp < 1) bpp = 1;
    if (bpp > 32) bpp = 32;
    a.blocks_per_peer = (int)bpp;
    hipLaunchKernelGGL(k_peer_put, dim3((unsigned)(c->world * bpp)), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const unsigned*)((const char*)c->base + c->flags_off) + parity * c->world, c->world, e, c->ticks, c->err);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

int sprk_peer_check(sprk_peer c, void* stream) {
    if (!c) return fail(SPRK_EINVAL, "communicator is NULL");
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, c->err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        HIP_TRY(hipMemsetAsync(c->err, 0, sizeof(int), (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        return fail(SPRK_EHIP, "peer all-gather: a rank's slice did not arrive within the deadline (rank %d of %d, exchange %u)", c->rank, c->world, c->epoch);
    }
    return SPRK_OK;
}

const char* sprk_peer_memory_kind(sprk_peer c) { return c ? c->mem_kind : ""; }

void sprk_peer_destroy(sprk_peer c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank && c->peer_base[p]) (void)hipIpcCloseMemHandle(c->peer_base[p]);
    if (c->base) (void)hipFree(c->base);
    if (c->done) (void)hipFree(c->done);
    if (c->err) (void)hipFree(c->err);
    delete c;
}

}  // extern "C"

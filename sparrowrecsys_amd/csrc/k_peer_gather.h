// k_peer_gather.h -- the score all-gather as direct peer writes over xGMI; the reference has no
// multi-GPU path.  Included inside sparrow_hip.hip's anonymous namespace.
//
// The path's one exchange moves a few hundred KB per rank (65 536 scores = 256 KB): a ring collective spends most of its
// time in hops (7 dependent steps on 8 GPUs), while xGMI is a full point-to-point mesh -- every rank can store its slice
// straight into all 7 peers' receive buffers at once, one link each, and be done in ONE step:
//
//   rank r, exchange e (parity = e & 1):
//     k_peer_put    for every peer p (itself included): recv_p[parity][r][0 .. count) = local[0 .. count)   (16-byte stores),
//                   __threadfence_system(), and the last workgroup of each peer's share releases flag_p[parity][r] = e
//     k_peer_wait   one wave: lane p spins (system-scope acquire loads) until flag_r[parity][p] == e for every p, bounded by
//                   a wall-clock deadline (a dead peer raises the handle's error flag instead of hanging the GPU)
//   then recv_r[parity] = [world][slot] holds everybody's slice; kernels enqueued behind k_peer_wait read it.
//
// Every rank's receive buffer + flags are ONE fine-grained device allocation, exported with hipIpcGetMemHandle and opened by
// the peers (sprk_peer_*: include/sparrow_hip.h).  Two parities: rank r starts exchange e + 2 (which overwrites parity e's
// slots in its peers) only after its own k_peer_wait of exchange e + 1 has seen every peer's flag, i.e. after every peer has
// LAUNCHED exchange e + 1 -- which, in stream order, follows that peer's readers of exchange e.  Consumers must therefore
// read recv on the stream the exchange was enqueued on (or order themselves behind it) -- the usual stream contract.

#define PEER_MAX_WORLD 16

struct PeerPut {
    float* dst[PEER_MAX_WORLD];           // peer p's receive slot for THIS rank at this parity (peer-mapped device pointers)
    unsigned* flag[PEER_MAX_WORLD];       // peer p's arrival flag for this rank at this parity
    const float* src;
    unsigned long long count;             // floats
    unsigned epoch;
    int world, blocks_per_peer;
    unsigned* done;                       // [world] local counters: workgroups of peer p's share that have finished
};

static __global__ __launch_bounds__(256) void k_peer_put(const PeerPut A) {
    const int p = blockIdx.x / A.blocks_per_peer, j = blockIdx.x - p * A.blocks_per_peer;
    float* dst = A.dst[p];
    const unsigned long long n4 = A.count >> 2;
    const bool vec = !(((uintptr_t)A.src | (uintptr_t)dst) & 15);
    if (vec) {
        for (unsigned long long i = (unsigned long long)j * 256 + threadIdx.x; i < n4; i += (unsigned long long)A.blocks_per_peer * 256)
            __builtin_nontemporal_store(reinterpret_cast<const f32x4*>(A.src)[i], reinterpret_cast<f32x4*>(dst) + i);
        for (unsigned long long i = (n4 << 2) + (unsigned long long)j * 256 + threadIdx.x; i < A.count; i += (unsigned long long)A.blocks_per_peer * 256)
            dst[i] = A.src[i];
    } else {
        for (unsigned long long i = (unsigned long long)j * 256 + threadIdx.x; i < A.count; i += (unsigned long long)A.blocks_per_peer * 256)
            dst[i] = A.src[i];
    }
    __threadfence_system();                                       // this thread's stores have reached the peer
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned before = __hip_atomic_fetch_add(A.done + p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == (unsigned)A.blocks_per_peer - 1) {          // every workgroup of this peer's share has fenced
            __hip_atomic_store(A.done + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next exchange
            __hip_atomic_store(A.flag[p], A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// lane p < world waits for flags[p] == epoch; ticks = deadline in wall_clock64 ticks (100 MHz)
static __global__ __launch_bounds__(64) void k_peer_wait(const unsigned* __restrict__ flags, int world, unsigned epoch,
                                                  unsigned long long ticks, int* __restrict__ err) {
    const int p = threadIdx.x;
    if (p >= world) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
        if (wall_clock64() - t0 > ticks) { atomicOr(err, 2); break; }
        __builtin_amdgcn_s_sleep(8);
    }
}

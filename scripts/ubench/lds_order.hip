// lds_order.hip -- do a wave's LDS reads always complete in the order they were issued?  hipcc's `s_waitcnt lgkmcnt(N)` with N > 0 relies on it.
//
// Why (round 5): k_dien_fused<16,...> sporadically scored a whole 16-sample tile with what looks like a STALE bias vector (a sample-
// independent error of the recurrence's state), in builds whose loop mixes ordinary per-lane `ds_read_b128` (fragments, bias vectors) with
// reads of ONE address by all 64 lanes (the per-block un-scale scalars) under partial waits (`lgkmcnt(4)`); every counted wait replays fine
// if returns are in order (scripts/isa/isa_waitcnt_paths.py).  The test: a slow read (64 lanes on ONE bank, different rows: a 64-way conflict)
// is issued first into a register holding a sentinel, a fast read (all lanes one address) second; `s_waitcnt lgkmcnt(1)`; copy the first
// register.  The sentinel in the copy = the second read retired first.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/lds_order scripts/ubench/lds_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int STRIDE_BYTES, int MODE>
__global__ __launch_bounds__(1024, 4) void k_order(int iters, unsigned* bad) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 1024) lds[i] = 7.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&lds[0]);
    const unsigned slow = base + lane * STRIDE_BYTES;          // one bank, 64 rows
    const unsigned fast = base + 32;                           // one address for the wave
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f2 a = {-1.f, -1.f}, o;
        f4 u; f2 u2;
        if constexpr (MODE == 0)
            asm volatile("ds_read_b64 %0, %3\n\t"
                         "ds_read_b128 %1, %4\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         "v_mov_b64 %2, %0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(a), "=&v"(u), "=&v"(o) : "v"(slow), "v"(fast) : "memory");
        else
            asm volatile("ds_read_b64 %0, %4\n\t"
                         "ds_read_b128 %1, %5\n\t"
                         "ds_read_b64 %2, %5 offset:64\n\t"
                         "s_waitcnt lgkmcnt(2)\n\t"
                         "v_mov_b64 %3, %0\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(a), "=&v"(u), "=&v"(u2), "=&v"(o) : "v"(slow), "v"(fast) : "memory");
        if (o[0] != 7.f || o[1] != 7.f) ++nbad;
        if (u[0] != 7.f) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int STRIDE_BYTES, int MODE>
int run(int iters, unsigned* d) {
    CHECK(hipMemset(d, 0, 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_order<STRIDE_BYTES, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL((k_order<STRIDE_BYTES, MODE>), dim3(256), dim3(1024), 64 * 1024, 0, iters, d);
    CHECK(hipDeviceSynchronize());
    unsigned h;
    CHECK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("first read: stride %4d B per lane, then %s, lgkmcnt(%d): %u stale copies of %llu\n", STRIDE_BYTES,
           MODE == 0 ? "one same-address b128" : "same-address b128 + b64", MODE == 0 ? 1 : 2, h, 256ull * 1024 * iters);
    return 0;
}

int main() {
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 4));
    const int N = 20000;
    run<8, 0>(N, d); run<256, 0>(N, d); run<512, 0>(N, d); run<1024, 0>(N, d);
    run<8, 1>(N, d); run<256, 1>(N, d); run<512, 1>(N, d); run<1024, 1>(N, d);
    return 0;
}

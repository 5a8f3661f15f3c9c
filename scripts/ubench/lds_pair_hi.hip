// lds_pair_hi.hip -- the schedule marker of every flaky k_dien_* build: a uniform-address `ds_read_b64 v[P:P+1]` (two scalars), K more LDS reads behind
// it, `s_waitcnt lgkmcnt(K)`, and the HIGH register of the pair consumed through `v_pk_fma_f32 ... op_sel:[0,1,0]`.  Is the high register there
// when the counter says so?  The pair holds a sentinel before the read; LDS holds 7.0 / 9.0 at the two addresses; d = 1 * pair.hi + 0 must be 9.
// Measured (profiles/r05/experiments/r05_32/lds_pair_hi.txt): 0 wrong of 5.2e9 for lgkmcnt(4) / (3) / (2) / (0), high or low register: the marker is
// only a marker.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/lds_pair_hi scripts/ubench/lds_pair_hi.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int K, int MODE>
__global__ __launch_bounds__(1024, 4) void k_pair(int iters, unsigned* bad) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = (i == 1) ? 9.0f : 7.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&lds[0]);
    const unsigned uni = base, per = base + 1024 + lane * 16;
    const f2 ones = {1.f, 1.f}, zeros = {0.f, 0.f};
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f2 p = {-1.f, -1.f}, d;
        f4 x0, x1, x2, x3;
        if constexpr (MODE == 0)          // op_sel:[0,1,0]: both result lanes from the pair's HIGH register
            asm volatile("ds_read_b64 %0, %6\n\t"
                         "ds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:1024\n\tds_read_b128 %4, %7 offset:2048\n\tds_read_b128 %5, %7 offset:3072\n\t"
                         "s_waitcnt lgkmcnt(%10)\n\t"
                         "v_pk_fma_f32 %1, %8, %0, %9 op_sel:[0,1,0]\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(p), "=&v"(d), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(uni), "v"(per), "v"(ones), "v"(zeros), "n"(K) : "memory");
        else                              // the clean builds' form: the LOW register broadcast
            asm volatile("ds_read_b64 %0, %6\n\t"
                         "ds_read_b128 %2, %7\n\tds_read_b128 %3, %7 offset:1024\n\tds_read_b128 %4, %7 offset:2048\n\tds_read_b128 %5, %7 offset:3072\n\t"
                         "s_waitcnt lgkmcnt(%10)\n\t"
                         "v_pk_fma_f32 %1, %8, %0, %9 op_sel_hi:[1,0,1]\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "+v"(p), "=&v"(d), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(uni), "v"(per), "v"(ones), "v"(zeros), "n"(K) : "memory");
        const float want = MODE == 0 ? 9.f : 7.f;
        if (d[0] != want || d[1] != want) ++nbad;
        if (x0[0] + x1[0] + x2[0] + x3[0] != 28.f) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int K, int MODE>
int run(unsigned* d) {
    CHECK(hipMemset(d, 0, 4));
    hipLaunchKernelGGL((k_pair<K, MODE>), dim3(256), dim3(1024), 0, 0, 20000, d);
    CHECK(hipDeviceSynchronize());
    unsigned h;
    CHECK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("uniform ds_read_b64 + 4 ds_read_b128, lgkmcnt(%d), %s: %u wrong of %llu\n", K, MODE == 0 ? "op_sel:[0,1,0] (high register)" : "op_sel_hi:[1,0,1] (low register)", h, 256ull * 1024 * 20000);
    return 0;
}
int main() {
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 4));
    run<4, 0>(d); run<3, 0>(d); run<2, 0>(d); run<0, 0>(d);
    run<4, 1>(d); run<2, 1>(d);
    return 0;
}

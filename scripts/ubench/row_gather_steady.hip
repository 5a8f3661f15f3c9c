// Micro-benchmark (not product code): the STEADY-STATE floor of BASELINE config 2's gather -- ids -> 3 random 128-byte lines per sample (4 lanes
// x 16 B of the first half + one 4-byte scalar of the same line) -> one float per sample, NO scoring -- when ONE launch walks 64 batches of 65 536
// samples with persistent waves (what sprk_forward_many does at 64 batches per launch).  row_gather.hip prices the same gather per strict launch
// (5.9 us from the Infinity Cache); this one has no launch boundary, no staging and no ramp in it: what is left is the rate at which 256 CUs get
// random lines through their texture path, L2 and the fabric.  Sweep: waves per CU x tasks (of 16 samples) in flight per wave, table window.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/row_gather_steady.hip -o scripts/ubench/row_gather_steady && scripts/ubench/row_gather_steady
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every wave: tasks w, w + nwaves, ...; DEPTH tasks' gathers issued before the first is consumed
template <int DEPTH, int THREADS>
__global__ __launch_bounds__(THREADS) void k_steady(const char* __restrict__ tab, const unsigned* __restrict__ ids, float* __restrict__ out, int ntasks) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int wave_global = (blockIdx.x * THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * THREADS) >> 6;
    for (int t0 = wave_global; t0 < ntasks; t0 += nwaves * DEPTH) {
        unsigned id[DEPTH][3];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 + d * nwaves;
            const size_t m = (size_t)(t < ntasks ? t : t0) * 16 + r;
#pragma unroll
            for (int f = 0; f < 3; ++f) id[d][f] = ids[m * 3 + f];
        }
        f32x4 x[DEPTH][3];
        float sc[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int f = 0; f < 3; ++f) x[d][f] = *reinterpret_cast<const f32x4*>(tab + ((size_t)id[d][f] * 128u + 16u * q));
            const unsigned s = q == 1 ? id[d][1] : (q == 2 ? id[d][2] : id[d][0]);
            sc[d] = *reinterpret_cast<const float*>(tab + ((size_t)s * 128u + 64u));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 + d * nwaves;
            f32x4 s = x[d][0] + x[d][1] + x[d][2];
            float z = s.x + s.y + s.z + s.w + (q < 3 ? sc[d] : 0.f);
            z += __shfl_xor(z, 16);
            z += __shfl_xor(z, 32);
            if (q == 0 && t < ntasks) out[(size_t)t * 16 + r] = z;
        }
    }
}

static unsigned long long g_s = 0x9E3779B97F4A7C15ull;
static unsigned long long xr() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return g_s; }

template <int DEPTH, int THREADS>
static double run(const char* tab, const unsigned* ids, float* out, int ntasks, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_steady<DEPTH, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, tab, ids, out, ntasks);
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_steady<DEPTH, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, tab, ids, out, ntasks);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / n;
}

int main() {
    const int B = 65536, NB = 64, ntasks = NB * B / 16;
    const size_t bytes = 3200ull << 20, rows_big = bytes / 128;
    char* tab;
    CK(hipMalloc(&tab, bytes));
    CK(hipMemset(tab, 0, bytes));
    float* out;
    CK(hipMalloc(&out, (size_t)NB * B * sizeof(float)));
    struct Dist { const char* name; size_t rows; };
    const Dist dists[] = {{"rows inside a 26 MB window", (26ull << 20) / 128},
                          {"rows inside a 51 MB window (config 2's three big tables: 401 020 rows of 128 bytes)", 401020},
                          {"rows inside a 130 MB window (all of config 2's device tables)", (130ull << 20) / 128},
                          {"rows inside a 200 MB window (Infinity Cache)", (200ull << 20) / 128},
                          {"rows over the whole 3.2 GB table (HBM)", rows_big}};
    for (const Dist& d : dists) {
        std::vector<unsigned> h((size_t)NB * B * 3);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(xr() % d.rows);
        unsigned* ids;
        CK(hipMalloc(&ids, h.size() * 4));
        CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("%s: us per 65 536-sample step, one launch = 64 steps\n", d.name);
#define ROW(DEPTH, THREADS, BPC) { const double us = run<DEPTH, THREADS>(tab, ids, out, ntasks, 256 * BPC) / NB; \
        printf("  %2d waves per CU x %d task(s) in flight : %.2f us/step = %.2f TB/s of 128-B lines\n", THREADS / 64 * BPC, DEPTH, us, 196608.0 * 128 / us / 1e6); }
        ROW(1, 512, 1) ROW(2, 512, 1) ROW(4, 512, 1)
        ROW(1, 1024, 1) ROW(2, 1024, 1) ROW(4, 1024, 1)
        ROW(1, 1024, 2) ROW(2, 1024, 2)
        CK(hipFree(ids));
    }
    return 0;
}

// Micro-benchmark (not product code): what the lane -> address mapping of k_din_attn_cols / k_din_fused's row gather costs.
// 16 samples' 128-byte rows per step, 2 x global_load_dwordx4 per lane, 2 048 waves x 50 steps (BASELINE config 3's attention):
//   A  operand layout: lane (r = lane & 15, q = lane >> 4) loads bytes [32 q, 32 q + 32) of row r -- a quad of consecutive lanes
//      touches FOUR different rows, 16 bytes each (what the kernels do: the gathered registers ARE the MFMA B operand);
//   B  coalesced: lane L loads piece L & 7 of row L >> 3 (+ 8 rows for the second instruction) -- a quad covers 64 contiguous bytes;
//   C  as B, but through LDS-DMA (global_load_lds_dwordx4) and read back from LDS in the operand layout (two ds_read_b128).
// Same bytes, same rows, same prefetch depth (8 loads in flight per wave).
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/gather_pattern scripts/ubench/gather_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) float smem[];

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_gather(const char* __restrict__ tab, const int* __restrict__ ids, int T, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 8 + wave;
    const int r = lane & 15, q = lane >> 4;
    const int* idrow = ids + (size_t)task * 16 * T;          // [16 samples][T] of this task
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float* lds = smem + wave * (4 * 512);                     // C: four 2-KB slots per wave
    auto addr = [&](int t, int half) -> const char* {
        if (MODE == 0) { const int id = idrow[r * T + t]; return tab + (size_t)id * 128 + q * 32 + half * 16; }
        const int row = (lane >> 3) + 8 * half;               // B / C: rows 0..7, then 8..15
        const int id = idrow[row * T + t];
        return tab + (size_t)id * 128 + (lane & 7) * 16;
    };
    if (MODE < 2) {
        f32x4 v[4][2];
#pragma unroll
        for (int p = 0; p < 4; ++p) { v[p][0] = *(const f32x4*)addr(p, 0); v[p][1] = *(const f32x4*)addr(p, 1); }
        for (int t = 0; t < T; t += 4) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                acc += v[p][0] + v[p][1];
                const int tn = t + 4 + p < T ? t + 4 + p : T - 1;
                v[p][0] = *(const f32x4*)addr(tn, 0); v[p][1] = *(const f32x4*)addr(tn, 1);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) acc += v[p][0] + v[p][1];
    } else {
        auto dma = [&](int t, int slot) {
#pragma unroll
            for (int half = 0; half < 2; ++half)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr(t, half),
                                                 (__attribute__((address_space(3))) void*)(lds + slot * 512 + half * 256), 16, 0, 0);
        };
#pragma unroll
        for (int p = 0; p < 4; ++p) dma(p, p);
        for (int t = 0; t < T; t += 4) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // the oldest slot has landed when at most 3 slots (6 DMA instructions) are outstanding
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                const f32x4 a = *(const f32x4*)(lds + p * 512 + r * 32 + q * 8);        // row r, bytes [32 q, 32 q + 16)
                const f32x4 b = *(const f32x4*)(lds + p * 512 + r * 32 + q * 8 + 4);
                acc += a + b;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int tn = t + 4 + p < T ? t + 4 + p : T - 1;
                dma(tn, p);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    out[(size_t)task * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    const int V = 131263, T = 50, B = 32768, NT = B / 16;
    char* tab; hipMalloc(&tab, (size_t)V * 128); hipMemset(tab, 0, (size_t)V * 128);
    std::vector<int> h((size_t)B * T);
    std::mt19937 g(1);
    for (auto& x : h) x = g() % V;
    int* ids; hipMalloc(&ids, h.size() * 4); hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, (size_t)NT * 64 * 4);
    const double mb = (double)B * T * 128 / 1e6;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gather<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 4 * 2048);
    for (int rep = 0; rep < 2; ++rep) {
        float a = timeit([&] { hipLaunchKernelGGL(k_gather<0>, dim3(NT / 8), dim3(512), 0, 0, tab, ids, T, out); }, 50);
        float b = timeit([&] { hipLaunchKernelGGL(k_gather<1>, dim3(NT / 8), dim3(512), 0, 0, tab, ids, T, out); }, 50);
        float c = timeit([&] { hipLaunchKernelGGL(k_gather<2>, dim3(NT / 8), dim3(512), 8 * 4 * 2048, 0, tab, ids, T, out); }, 50);
        printf("%.1f MB of rows per launch: A operand layout %.2f us (%.2f TB/s)   B coalesced %.2f us (%.2f TB/s)   C LDS-DMA + operand-layout LDS reads %.2f us (%.2f TB/s)\n",
               mb, a * 1e3, mb / a / 1e3, b * 1e3, mb / b / 1e3, c * 1e3, mb / c / 1e3);
    }
    return 0;
}

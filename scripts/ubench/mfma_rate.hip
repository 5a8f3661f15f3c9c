// Micro-benchmark (not product code): sustained issue rate of v_mfma_f32_16x16x4_f32 on gfx950,
// as a function of independent accumulator chains per wave and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float a, float b) {
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, (float)c};
    const float av = a + threadIdx.x * 1e-6f, bv = b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = acc[0];
#pragma unroll
    for (int c = 1; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y + s.z + s.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((unsigned long long*)out)[1 << 20] = t1 - t0;
}

template <int CHAINS>
void run(int wgs_per_cu, int iters) {
    float* out;
    hipMalloc(&out, (1 << 23) + 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k_mfma<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ticks; hipMemcpy(&ticks, ((unsigned long long*)out) + (1 << 20), 8, hipMemcpyDeviceToHost);
    const double n_per_wave = (double)iters * 8 * CHAINS;
    const double per_simd = n_per_wave * wgs_per_cu;      // 4 waves per WG -> 1 wave per SIMD per WG
    printf("chains=%d waves/SIMD=%d iters=%d: %.1f us, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz), wave0 memtime ticks/MFMA %.1f, TF %.1f\n",
           CHAINS, wgs_per_cu, iters, ms * 1e3, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, (double)ticks / n_per_wave,
           per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    for (int iters : {20, 400, 4000}) {
        run<1>(1, iters); run<2>(1, iters); run<4>(1, iters);
        run<1>(2, iters); run<2>(2, iters); run<4>(2, iters);
        run<2>(4, iters); run<4>(4, iters);
    }
    return 0;
}

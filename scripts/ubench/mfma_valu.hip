// Micro-benchmark (not product code): how much VALU work hides in the shadow of v_mfma_f32_16x16x4_f32
// (a) inside one wave: 1 MFMA + NV independent v_fma_f32 per step;  (b) across the two waves of a SIMD:
// one wave issues only MFMAs, its SIMD partner only v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float a, float b) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 1};
    float v[NV > 0 ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x + i;
    const float av = a + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(b));
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc1, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(b));
        }
    }
    float s = acc0.x + acc1.y;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 0: all waves MFMA; mode 1: all waves VALU; mode 2: waves 0-3 MFMA, waves 4-7 VALU (512-thread WG: w and w+4 share a SIMD)
__global__ __launch_bounds__(512) void k_pair(float* out, int iters, float a, float b, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    const float av = a + threadIdx.x * 1e-6f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 1};
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    if (do_mfma) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc1, 0, 0, 0);
            }
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 32; ++u) {     // 128 v_fma per iteration = 16 MFMA-times at 2 cycles each?
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(av), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(av), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(av), "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(av), "v"(b));
            }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0.x + acc1.y + v0 + v1 + v2 + v3;
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int NV> void run_mix(float* out, int iters) {
    float ms = timeit([&] { hipLaunchKernelGGL(k_mix<NV>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    printf("one wave/SIMD, per MFMA %2d v_fma: %.1f ns per (MFMA + %d VALU) = %.1f cycles at 2.4 GHz\n", NV, ms * 1e6 / (iters * 16.0), NV,
           ms * 1e6 / (iters * 16.0) * 2.4);
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    const int iters = 4000;
    run_mix<0>(out, iters); run_mix<1>(out, iters); run_mix<2>(out, iters); run_mix<3>(out, iters); run_mix<4>(out, iters);
    run_mix<6>(out, iters); run_mix<8>(out, iters); run_mix<12>(out, iters);
    for (int mode = 0; mode < 3; ++mode) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_pair, dim3(256), dim3(512), 0, 0, out, iters, 1.0f, 0.5f, mode); });
        printf("pair mode %d (%s): %.1f us  (per iteration: 16 MFMA and/or 128 v_fma per wave) = %.1f cycles/iteration\n", mode,
               mode == 0 ? "both waves MFMA" : mode == 1 ? "both waves VALU" : "one MFMA wave + one VALU wave per SIMD", ms * 1e3, ms * 1e6 / iters * 2.4);
    }
    return 0;
}

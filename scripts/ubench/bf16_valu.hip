// Micro-benchmark (not product code): does v_mfma_f32_16x16x32_bf16 (matrix pipe) overlap with VALU work
// and with v_mfma_f32_16x16x4_f32 issued by the same wave?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NB, int NV, int NF>   // per step: NB bf16 MFMAs, NV v_fma, NF f32 MFMAs
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float a, float b) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 1}, {0, 0, 0, 2}, {0, 0, 0, 3}};
    f32x4 facc[2] = {{0, 0, 0, 0}, {0, 0, 0, 1}};
    float v[NV > 0 ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = threadIdx.x + i;
    bf16x8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b + threadIdx.x); }
    const float av = a + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i & 3], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(b));
#pragma unroll
            for (int i = 0; i < NF; ++i) facc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, facc[i & 1], 0, 0, 0);
        }
    }
    float s = acc[0].x + acc[1].y + acc[2].z + acc[3].w + facc[0].x + facc[1].y;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int NB, int NV, int NF> void run(float* out, int iters, int wgs) {
    float ms = timeit([&] { hipLaunchKernelGGL((k_mix<NB, NV, NF>), dim3(256 * wgs), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
    printf("waves/SIMD=%d  per step: %d bf16 MFMA + %2d v_fma + %d f32 MFMA: %.1f cycles/step at 2.4 GHz\n", wgs, NB, NV, NF, ms * 1e6 / (iters * 8.0) * 2.4 / wgs);
}
int main() {
    float* out; hipMalloc(&out, 1 << 24);
    const int it = 4000;
    run<1, 0, 0>(out, it, 1); run<2, 0, 0>(out, it, 1); run<4, 0, 0>(out, it, 1);
    run<0, 4, 0>(out, it, 1); run<0, 8, 0>(out, it, 1);
    run<1, 4, 0>(out, it, 1); run<2, 4, 0>(out, it, 1); run<2, 8, 0>(out, it, 1); run<4, 8, 0>(out, it, 1); run<4, 16, 0>(out, it, 1);
    run<0, 0, 1>(out, it, 1); run<2, 0, 1>(out, it, 1); run<4, 0, 1>(out, it, 1); run<4, 8, 1>(out, it, 1); run<2, 4, 2>(out, it, 1);
    run<4, 8, 1>(out, it, 2); run<2, 8, 1>(out, it, 2); run<4, 0, 0>(out, it, 2);
    return 0;
}

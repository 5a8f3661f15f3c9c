// Micro-benchmark (not product code): what one dependent kernel boundary costs on this stack as a function of the
// launch's shape -- kernarg size (by-value structs), workgroup size, dynamic LDS, launch bounds, grid -- measured as
// HIP-event time over N back-to-back launches on one stream.  MI355X_MICROARCH.md quotes 1.45 us between trivial
// 256-workgroup kernels; round 1 measured 3.3 us for the fused DeepFM_v2 kernel returning at entry.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/launch_floor.hip -o scripts/ubench/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

struct Small { int a[8]; };
struct Big { const void* p[3 * 64]; int n, m; };      // the round-1 V2JMany: 1.5 KB of pointers by value
struct Run { int a[24]; const float* t0; const float* t1; float f[8]; };

extern __shared__ float dyn_smem[];

__global__ __launch_bounds__(256) void k_empty256(float* out) { if (out == (float*)1) out[0] = 1.f; }
__global__ __launch_bounds__(512) void k_empty512(float* out) { if (out == (float*)1) out[0] = 1.f; }
__global__ __launch_bounds__(512, 2) void k_empty512_lb2(float* out) { if (out == (float*)1) out[0] = 1.f; }
__global__ __launch_bounds__(512, 2) void k_lds512(float* out) {
    if (out == (float*)1) { dyn_smem[threadIdx.x] = 1.f; out[0] = dyn_smem[0]; }
}
__global__ __launch_bounds__(512, 2) void k_small_arg(const Small s, float* out) { if (out == (float*)1) out[0] = (float)s.a[3]; }
__global__ __launch_bounds__(512, 2) void k_big_arg(const Run r, float* out, int B, int* err, const float* image, const Big m) {
    if (out == (float*)1) { dyn_smem[threadIdx.x] = (float)r.a[3]; out[0] = dyn_smem[0] + (float)m.n + (float)B + (err ? 1.f : 0.f) + (image ? 1.f : 0.f); }
}
__global__ __launch_bounds__(512, 2) void k_run_arg(const Run r, float* out, int B, int* err, const float* image) {
    if (out == (float*)1) { dyn_smem[threadIdx.x] = (float)r.a[3]; out[0] = dyn_smem[0] + (float)B + (err ? 1.f : 0.f) + (image ? 1.f : 0.f); }
}
// a kernel that touches memory like a real one: every thread stores one float (dirty lines at the boundary)
__global__ __launch_bounds__(256) void k_store256(float* out) { out[blockIdx.x * 256 + threadIdx.x] = 1.f; }
// a kernel whose every wave reads its kernarg-provided pointer table (s_load of a far kernarg offset)
__global__ __launch_bounds__(512, 2) void k_big_arg_used(const Run r, float* out, int B, int* err, const float* image, const Big m) {
    const int b = blockIdx.x & 63;
    if (m.p[b] == (const void*)1) out[0] = 1.f;
}

template <typename F>
static double time_launches(F launch, int n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) launch();
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms * 1e3 / n < best) best = ms * 1e3 / n;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main() {
    float* out; hipMalloc(&out, 1 << 24);
    int* err; hipMalloc(&err, 64);
    Small s; memset(&s, 0, sizeof(s));
    Big m; memset(&m, 0, sizeof(m));
    Run r; memset(&r, 0, sizeof(r));
    const int N = 4000;
    const size_t LDS = 100 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds512), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_big_arg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_run_arg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_big_arg_used), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    printf("launch floor, us per launch (best of 5 runs of %d back-to-back launches on the null stream)\n", N);
    for (int grid : {1, 64, 256, 512, 2048}) {
        printf("grid %4d:", grid);
        printf(" empty256 %.2f", time_launches([&] { hipLaunchKernelGGL(k_empty256, dim3(grid), dim3(256), 0, 0, out); }, N));
        printf(" | empty512 %.2f", time_launches([&] { hipLaunchKernelGGL(k_empty512, dim3(grid), dim3(512), 0, 0, out); }, N));
        printf(" | empty512_lb2 %.2f", time_launches([&] { hipLaunchKernelGGL(k_empty512_lb2, dim3(grid), dim3(512), 0, 0, out); }, N));
        printf(" | +100KB-LDS %.2f", time_launches([&] { hipLaunchKernelGGL(k_lds512, dim3(grid), dim3(512), LDS, 0, out); }, N));
        printf(" | small-arg %.2f", time_launches([&] { hipLaunchKernelGGL(k_small_arg, dim3(grid), dim3(512), 0, 0, s, out); }, N));
        printf(" | run-arg+LDS %.2f", time_launches([&] { hipLaunchKernelGGL(k_run_arg, dim3(grid), dim3(512), LDS, 0, r, out, 1, err, (const float*)out); }, N));
        printf(" | big-arg+LDS %.2f", time_launches([&] { hipLaunchKernelGGL(k_big_arg, dim3(grid), dim3(512), LDS, 0, r, out, 1, err, (const float*)out, m); }, N));
        printf(" | big-arg-used+LDS %.2f", time_launches([&] { hipLaunchKernelGGL(k_big_arg_used, dim3(grid), dim3(512), LDS, 0, r, out, 1, err, (const float*)out, m); }, N));
        printf(" | store256 %.2f\n", time_launches([&] { hipLaunchKernelGGL(k_store256, dim3(grid), dim3(256), 0, 0, out); }, N));
    }
    // the same on a created (non-null) stream, and with two streams alternating (independent launches)
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int variant = 0; variant < 3; ++variant) {
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty256, dim3(256), dim3(256), 0, s1, out);
            hipDeviceSynchronize();
            hipEventRecord(e0, s1);
            for (int i = 0; i < N; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_empty256, dim3(256), dim3(256), 0, s1, out);
                if (variant == 1) hipLaunchKernelGGL(k_big_arg, dim3(256), dim3(512), LDS, s1, r, out, 1, err, (const float*)out, m);
                if (variant == 2) hipLaunchKernelGGL(k_run_arg, dim3(256), dim3(512), LDS, s1, r, out, 1, err, (const float*)out);
            }
            hipEventRecord(e1, s1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("created stream, grid 256, %s: %.2f us per launch\n", variant == 0 ? "empty256" : variant == 1 ? "big-arg+LDS" : "run-arg+LDS", ms * 1e3 / N);
        }
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}

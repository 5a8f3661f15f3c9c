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

// dispatch skew: every wave stamps the 100 MHz wall clock at entry; host reports last - first entry
__global__ __launch_bounds__(1024) void k_stamp(unsigned long long* t, int regs_hint) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) t[w] = __builtin_amdgcn_s_memrealtime();
    if (regs_hint == 12345) dyn_smem[threadIdx.x] = 1.f;
}
// the same with a big register footprint (256 VGPRs: 2 waves per SIMD like the fused kernels)
__global__ __launch_bounds__(512, 2) void k_stamp_fat(unsigned long long* t, float* sink, int n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) t[w] = __builtin_amdgcn_s_memrealtime();
    if (n > 0) {                                   // never taken; keeps ~200 live registers in the descriptor
        float a[200];
#pragma unroll
        for (int i = 0; i < 200; ++i) a[i] = sink[i * 64 + threadIdx.x];
        for (int it = 0; it < n; ++it)
#pragma unroll
            for (int i = 0; i < 200; ++i) a[i] = a[i] * a[(i + 1) % 200] + 1.f;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 200; ++i) s += a[i];
        sink[threadIdx.x] = s;
    }
}
__global__ void k_spin(unsigned long long ticks, float* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (out == (float*)1) out[0] = 1.f;
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
    // device-side boundary: the queue is filled while a 3 ms spin kernel holds the GPU, so the host's launch rate is out of the picture
    {
        hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        for (int variant = 0; variant < 4; ++variant) {
            const int NQ = 1000;
            hipDeviceSynchronize();
            hipEventRecord(e0, s1);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, 300000ULL, out);     // 3 ms at 100 MHz
            hipEventRecord(e1, s1);
            for (int i = 0; i < NQ; ++i) {
                if (variant == 0) hipLaunchKernelGGL(k_empty256, dim3(256), dim3(256), 0, s1, out);
                if (variant == 1) hipLaunchKernelGGL(k_run_arg, dim3(256), dim3(512), LDS, s1, r, out, 1, err, (const float*)out);
                if (variant == 2) hipLaunchKernelGGL(k_big_arg, dim3(256), dim3(512), LDS, s1, r, out, 1, err, (const float*)out, m);
                if (variant == 3) hipLaunchKernelGGL(k_store256, dim3(256), dim3(256), 0, s1, out);
            }
            hipEventRecord(e2, s1);
            hipEventSynchronize(e2);
            float ms_spin, ms_q; hipEventElapsedTime(&ms_spin, e0, e1); hipEventElapsedTime(&ms_q, e1, e2);
            printf("queue pre-filled behind a %.2f ms spin, grid 256, %s: %.2f us per launch (device-side boundary)\n", ms_spin,
                   variant == 0 ? "empty256" : variant == 1 ? "run-arg+LDS 512thr" : variant == 2 ? "big-arg+LDS 512thr" : "store256", ms_q * 1e3 / NQ);
        }
    }
    // dispatch skew of ONE launch: first wave entry -> last wave entry (100 MHz clock, 10 ns resolution)
    {
        unsigned long long* t; hipMalloc(&t, 8 * 65536);
        unsigned long long* th = (unsigned long long*)malloc(8 * 65536);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_stamp), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
        struct Cfg { int grid, block; size_t lds; bool fat; const char* name; };
        const Cfg cfgs[] = {{256, 512, 0, false, "256 WG x 512 thr, no LDS"}, {256, 512, 112 * 1024, false, "256 WG x 512 thr, 112 KB LDS"},
                            {256, 512, 112 * 1024, true, "256 WG x 512 thr, 112 KB LDS, 256-VGPR kernel"},
                            {256, 256, 0, false, "256 WG x 256 thr"}, {256, 256, 112 * 1024, false, "256 WG x 256 thr, 112 KB LDS"},
                            {512, 256, 0, false, "512 WG x 256 thr"}, {512, 256, 64 * 1024, false, "512 WG x 256 thr, 64 KB LDS"},
                            {1024, 128, 0, false, "1024 WG x 128 thr"}, {2048, 64, 0, false, "2048 WG x 64 thr"},
                            {256, 1024, 0, false, "256 WG x 1024 thr"}, {256, 64, 0, false, "256 WG x 64 thr"}, {1024, 256, 0, false, "1024 WG x 256 thr"}};
        for (const Cfg& c : cfgs) {
            const int nw = c.grid * c.block / 64;
            double best = 1e30, sum = 0;
            for (int rep = 0; rep < 7; ++rep) {
                hipMemset(t, 0, 8 * 65536);
                hipDeviceSynchronize();
                if (c.fat) hipLaunchKernelGGL(k_stamp_fat, dim3(c.grid), dim3(c.block), c.lds, 0, t, out, 0);
                else hipLaunchKernelGGL(k_stamp, dim3(c.grid), dim3(c.block), c.lds, 0, t, 0);
                hipDeviceSynchronize();
                hipMemcpy(th, t, 8 * nw, hipMemcpyDeviceToHost);
                unsigned long long lo = ~0ULL, hi = 0;
                for (int i = 0; i < nw; ++i) { if (th[i] < lo) lo = th[i]; if (th[i] > hi) hi = th[i]; }
                const double us = (double)(hi - lo) * 0.01;
                if (rep > 0) { sum += us; if (us < best) best = us; }
            }
            printf("dispatch skew %-48s: %4d waves, first->last entry min %.2f us, mean %.2f us\n", c.name, nw, best, sum / 6);
        }
    }
    printf("done\n");
    return 0;
}

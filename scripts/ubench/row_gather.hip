// Micro-benchmark (not product code): the memory system's floor for what k_deepfm_v2_joint does per launch at BASELINE
// config 2 -- ids -> 3 random 128-byte row lines per sample (4 lanes x 16 B of the line's first half + one 4-byte
// scalar from the same line) -> one float per sample -- with NO scoring work, as a function of
//   * where the rows live: a 3.2 GB table (HBM + TLB reach), a 200 MB window (Infinity Cache), or 400 k rows that are
//     cache-resident but spread over the whole 3.2 GB (cache-resident DATA, HBM-sized TRANSLATION footprint:
//     isolates the TLB from DRAM);
//   * the launch shape: 2 048 waves x 2 tasks (the fused kernel's), 4 096 waves x 1 task.
// Strict stream order, one launch per 65 536-sample batch, 32 id batches cycled, HIP events.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/row_gather.hip -o scripts/ubench/row_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// TPW tasks of 16 samples per wave, all gathers of a wave issued before the first use
template <int TPW, int THREADS>
__global__ __launch_bounds__(THREADS) void k_gather(const char* __restrict__ tab, const unsigned* __restrict__ ids,
                                                    float* __restrict__ out, int B) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int wave_global = (blockIdx.x * THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * THREADS) >> 6;
    unsigned id[TPW][3];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int m = (wave_global + t * nwaves) * 16 + r;
#pragma unroll
        for (int f = 0; f < 3; ++f) id[t][f] = m < B ? ids[(size_t)m * 3 + f] : 0u;
    }
    f32x4 x[TPW][3];
    float sc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int f = 0; f < 3; ++f) x[t][f] = *reinterpret_cast<const f32x4*>(tab + ((size_t)id[t][f] * 128u + 16u * q));
        const unsigned s = q == 1 ? id[t][1] : (q == 2 ? id[t][2] : id[t][0]);
        sc[t] = *reinterpret_cast<const float*>(tab + ((size_t)s * 128u + 64u));
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        f32x4 s = x[t][0] + x[t][1] + x[t][2];
        float z = s.x + s.y + s.z + s.w + (q < 3 ? sc[t] : 0.f);
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        const int m = (wave_global + t * nwaves) * 16 + r;
        if (q == 0 && m < B) out[m] = z;
    }
}

// ---- decomposition of what k_deepfm_v2_joint1 adds on top of the gather (512 wg x 512 thr, one task per wave) ----
//   STAGE: every workgroup first copies `stage_kb` KB global -> LDS by LDS-DMA (its share per wave), then a barrier
//   READS: after its rows have landed every wave reads `reads_kb` KB from LDS (16 B per lane per read)
extern __shared__ float dyn_lds[];
template <bool STAGE>
__global__ __launch_bounds__(512, 4) void k_gather1(const char* __restrict__ tab, const unsigned* __restrict__ ids, float* __restrict__ out, int B,
                                                  const float* __restrict__ image, int stage_kb, int reads_kb) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 8 + wave;
    const int m = task * 16 + r;
    unsigned id[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) id[f] = m < B ? ids[(size_t)m * 3 + f] : 0u;
    if (STAGE) {
        for (int c = wave; c < stage_kb; c += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(image + c * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dyn_lds + c * 256), 16, 0, 0);
    }
    f32x4 x[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) x[f] = *reinterpret_cast<const f32x4*>(tab + ((size_t)id[f] * 128u + 16u * q));
    const unsigned s = q == 1 ? id[1] : (q == 2 ? id[2] : id[0]);
    const float sc = *reinterpret_cast<const float*>(tab + ((size_t)s * 128u + 64u));
    if (STAGE) {
        __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
        __builtin_amdgcn_s_barrier();
    }
    f32x4 acc = x[0] + x[1] + x[2];
    for (int k = 0; k < reads_kb; ++k) acc += *reinterpret_cast<const f32x4*>(dyn_lds + ((k * 64 + lane) * 4) % (stage_kb > 0 ? stage_kb * 256 : 256));
    float z = acc.x + acc.y + acc.z + acc.w + (q < 3 ? sc : 0.f);
    z += __shfl_xor(z, 16);
    z += __shfl_xor(z, 32);
    if (q == 0 && m < B) out[m] = z;
}

static unsigned long long rng_state = 88172645463325252ull;
static inline unsigned long long xr() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
    const int B = 65536, NB = 32;
    const size_t rows_big = argc > 1 ? strtoull(argv[1], 0, 10) : 3ull * 8388608ull;   // 3.2 GB of 128-byte rows
    const size_t bytes = rows_big * 128;
    char* tab;
    // [r6, VERDICT r05 item 4 (i)] argv[2] = "vmm": the table as ONE physical allocation mapped through the virtual-memory API (hipMemCreate /
    // hipMemMap) at the runtime's recommended granularity -- does a larger fragment buy translation reach over hipMalloc's pages?
    if (argc > 2 && !strcmp(argv[2], "vmm")) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gmin = 0, grec = 0;
        CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
        const size_t chunk = argc > 3 ? strtoull(argv[3], 0, 10) : 0;                 // 0: one handle for the whole table
        const size_t gran = grec > gmin ? grec : gmin;
        const size_t total = (bytes + gran - 1) / gran * gran;
        printf("VMM allocation: granularity min %zu, recommended %zu; %zu bytes in %s\n", gmin, grec, total, chunk ? "chunks" : "one handle");
        void* va = nullptr;
        CK(hipMemAddressReserve(&va, total, 1ull << 30, nullptr, 0));
        size_t step = chunk ? (chunk + gran - 1) / gran * gran : total;
        for (size_t off = 0; off < total; off += step) {
            const size_t n = off + step <= total ? step : total - off;
            hipMemGenericAllocationHandle_t hnd;
            CK(hipMemCreate(&hnd, n, &prop, 0));
            CK(hipMemMap((char*)va + off, n, 0, hnd, 0));
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        tab = (char*)va;
        printf("table at %p (1 GB-aligned reservation)\n", va);
    } else {
        CK(hipMalloc(&tab, bytes));
        printf("hipMalloc table at %p\n", (void*)tab);
    }
    CK(hipMemset(tab, 0, bytes));
    float* out;
    CK(hipMalloc(&out, B * sizeof(float) * NB));
    struct Dist { const char* name; int kind; };
    const Dist dists[] = {{"rows over the whole 3.2 GB table (HBM + full translation footprint)", 0},
                          {"rows inside one 200 MB window (Infinity Cache, 100 x 2 MB pages)", 1},
                          {"409 600 distinct rows (52 MB of lines) spread over the whole table (cache-resident data, full translation footprint)", 2},
                          {"rows inside one 1 GB window (HBM, 512 x 2 MB pages)", 3}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Dist& d : dists) {
        std::vector<unsigned> h((size_t)NB * B * 3);
        const size_t win = (200ull << 20) / 128, win1g = (1ull << 30) / 128, nhot = 409600;
        for (size_t i = 0; i < h.size(); ++i) {
            const unsigned long long u = xr();
            if (d.kind == 0) h[i] = (unsigned)(u % rows_big);
            else if (d.kind == 1) h[i] = (unsigned)(u % win);
            else if (d.kind == 3) h[i] = (unsigned)(u % (win1g < rows_big ? win1g : rows_big));
            else h[i] = (unsigned)((u % nhot) * (rows_big / nhot) + 17);
        }
        unsigned* ids;
        CK(hipMalloc(&ids, h.size() * 4));
        CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("%s\n", d.name);
        for (int shape = 0; shape < 3; ++shape) {
            auto launch = [&](int b) {
                const unsigned* ib = ids + (size_t)b * B * 3;
                float* ob = out + (size_t)b * B;
                if (shape == 0) hipLaunchKernelGGL((k_gather<2, 512>), dim3(256), dim3(512), 0, 0, tab, ib, ob, B);
                else if (shape == 1) hipLaunchKernelGGL((k_gather<1, 256>), dim3(1024), dim3(256), 0, 0, tab, ib, ob, B);
                else hipLaunchKernelGGL((k_gather<4, 256>), dim3(256), dim3(256), 0, 0, tab, ib, ob, B);
            };
            for (int i = 0; i < 200; ++i) launch(i % NB);
            CK(hipDeviceSynchronize());
            double best = 1e30, sum = 0;
            const int n = 2000, reps = 5;
            for (int rep = 0; rep < reps; ++rep) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < n; ++i) launch(i % NB);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1e3 / n;
                sum += us;
                if (us < best) best = us;
            }
            const char* sn = shape == 0 ? "256 wg x 512 thr, 2 tasks/wave" : shape == 1 ? "1024 wg x 256 thr, 1 task/wave " : "256 wg x 256 thr, 4 tasks/wave ";
            printf("  %s : %.2f us/launch (best %.2f)  = %.2f TB/s of 128-B lines, %.1f %% of 8 TB/s on 464 B/sample\n", sn, sum / reps, best,
                   196608.0 * 128 / (sum / reps) / 1e6, 464.0 * B / (sum / reps) / 8e6 * 100);
        }
        if (d.kind == 0 || d.kind == 1) {
            float* image;
            CK(hipMalloc(&image, 64 * 1024));
            CK(hipMemset(image, 0, 64 * 1024));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gather1<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gather1<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            struct V { const char* name; bool stage; int stage_kb, reads_kb; };
            const V vs[] = {{"512 wg x 512 thr, 1 task/wave, gather only          ", false, 0, 0},
                            {"  + 31 KB LDS-DMA per workgroup + barrier            ", true, 31, 0},
                            {"  + 31 KB staging + 13 KB of LDS reads per wave      ", true, 31, 13},
                            {"  + 31 KB staging + 26 KB of LDS reads per wave      ", true, 31, 26},
                            {"  + 16 KB staging + 13 KB of LDS reads per wave      ", true, 16, 13}};
            for (const V& v : vs) {
                auto launch = [&](int b) {
                    const unsigned* ib = ids + (size_t)b * B * 3;
                    float* ob = out + (size_t)b * B;
                    if (v.stage) hipLaunchKernelGGL((k_gather1<true>), dim3(512), dim3(512), 40 * 1024, 0, tab, ib, ob, B, image, v.stage_kb, v.reads_kb);
                    else hipLaunchKernelGGL((k_gather1<false>), dim3(512), dim3(512), 4096, 0, tab, ib, ob, B, image, v.stage_kb, v.reads_kb);
                };
                for (int i = 0; i < 200; ++i) launch(i % NB);
                CK(hipDeviceSynchronize());
                double sum = 0;
                const int n = 2000, reps = 5;
                for (int rep = 0; rep < reps; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    for (int i = 0; i < n; ++i) launch(i % NB);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    sum += ms * 1e3 / n;
                }
                printf("  %s : %.2f us/launch\n", v.name, sum / reps);
            }
            CK(hipFree(image));
        }
        CK(hipFree(ids));
    }
    return 0;
}

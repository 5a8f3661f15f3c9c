// Micro-benchmark (not product code): what one gfx950 SIMD issues per cycle, by instruction kind and by waves per SIMD, and how
// far the f16 matrix pipe overlaps VALU work from the SAME wave and from OTHER waves of the SIMD.  Round 4: k_din_attn_cols
// spends ~680 cycles per (16 samples, slot) per SIMD whether it runs 2 or 4 waves per SIMD, with the matrix pipe 27 % and the
// VALU 30 % busy -- this tells which instruction mix a SIMD can actually sustain.
//   build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/issue_rates scripts/ubench/issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { K_FMA, K_PKFMA, K_MIX, K_MIXLO, K_EXP, K_RCP, K_PERM16, K_CNDMASK, K_MFMA, K_LDS, K_SALU, K_PKMULH, K_PKFMAH, K_CVTPK, K_CNDS, K_PKADD, K_MAX, K_CVT32, K_FMAABS, K_CND64VCC, K_CMPCND32, K_CMPCND64, K_CMP32, K_NKINDS };
static const char* kind_name[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_mix_f32", "v_fma_mixlo_f16", "v_exp_f32", "v_rcp_f32",
                                  "v_permlane16_swap", "v_cndmask_b32", "v_mfma_16x16x32_f16", "ds_read_b128", "s_add_u32",
                                  "v_pk_mul_f16", "v_pk_fma_f16", "v_cvt_pkrtz_f16_f32", "v_cndmask_b32_e64 sgpr", "v_pk_add_f32", "v_max_f32", "v_cvt_f32_f16", "v_fma_f32 |abs|",
                                  "v_cndmask_b32_e64 vcc", "v_cmp_e32 + v_cndmask_e32 (pair)", "v_cmp_e64 sgpr + v_cndmask_e64 (pair)", "v_cmp_gt_f32_e32"};

extern __shared__ float smem[];

// 64 instructions of one kind per loop trip, eight independent dependency chains
template <int KIND>
__global__ __launch_bounds__(256) void k_kind(float* out, int iters, float a, float b) {
    float v[8];
    f32x2 p[8];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 1}, {0, 0, 0, 2}, {0, 0, 0, 3}};
    f32x4 l[8];
    unsigned sreg = 0;
    unsigned long long mask64 = __builtin_amdgcn_read_exec() ^ (unsigned long long)iters;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; p[i] = f32x2{v[i], v[i] + 1}; l[i] = f32x4{0, 0, 0, 0}; }
    f16x8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b + (threadIdx.x & 7)); }
    const float av = a + threadIdx.x * 1e-6f;
    const f32x2 ap = {av, av};
    const f32x2 bp = {b, b};
    if (KIND == K_LDS) { smem[threadIdx.x * 4] = av; __syncthreads(); }
    const float* lp = smem + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(b));
                if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(ap), "v"(bp));
                if (KIND == K_MIX) asm volatile("v_fma_mix_f32 %0, %1, %0, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(v[i]) : "v"(av), "v"(b));
                if (KIND == K_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]" : "+v"(v[i]) : "v"(av), "v"(b));
                if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if (KIND == K_PERM16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 1) & 7]));
                if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(av));
                if (KIND == K_MFMA) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i & 3], 0, 0, 0);
                if (KIND == K_LDS) asm volatile("ds_read_b128 %0, %1" : "=v"(l[i]) : "v"((unsigned)(size_t)lp) : "memory");
                if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
                if (KIND == K_PKMULH) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (KIND == K_PKFMAH) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "v"(b));
                if (KIND == K_CVTPK) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (KIND == K_CNDS) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(av), "s"(mask64));
                if (KIND == K_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ap));
                if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(av));
                if (KIND == K_CVT32) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
                if (KIND == K_FMAABS) asm volatile("v_fma_f32 %0, %1, |%0|, %2" : "+v"(v[i]) : "v"(av), "v"(b));
                if (KIND == K_CND64VCC) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(av));
                if (KIND == K_CMPCND32) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(av) : "vcc");
                if (KIND == K_CMPCND64) asm volatile("v_cmp_gt_f32_e64 %1, %2, %0\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(v[i]), "+s"(mask64) : "v"(av));
                if (KIND == K_CMP32) asm volatile("v_cmp_gt_f32_e32 vcc, %1, %0" : : "v"(v[i]), "v"(av) : "vcc");
            }
            if (KIND == K_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float s = acc[0].x + acc[1].y + acc[2].z + acc[3].w + (float)sreg;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + l[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the attention loop's mix, per step of a wave: NM f16 MFMAs in two dependent chains (as the kernel's two n-blocks), NV v_fma in
// eight chains; INTER: VALU interleaved between the MFMAs (1) or all after them (0); SPLIT: even waves of a SIMD issue only the
// MFMAs and odd waves only the VALU (needs >= 2 waves per SIMD) -- overlap across waves without any in-wave ordering
template <int NM, int NV, int INTER, int SPLIT>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float a, float b) {
    float v[8];
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 1}};
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    f16x8 ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b + (threadIdx.x & 7)); }
    const float av = a + threadIdx.x * 1e-6f;
    const bool do_m = !SPLIT || (blockIdx.x & 1) == 0;       // (workgroup-uniform; consecutive workgroups share a CU's SIMDs)
    const bool do_v = !SPLIT || (blockIdx.x & 1) == 1;
    for (int it = 0; it < iters; ++it) {
        constexpr int VPM = NM > 0 ? NV / NM : 0;
        if (INTER && NM > 0) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (do_m) acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[m & 1], 0, 0, 0);
                if (do_v) {
#pragma unroll
                    for (int i = 0; i < VPM; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m * VPM + i) & 7]) : "v"(av), "v"(b));
                }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int m = 0; m < NM; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[m & 1], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(av), "v"(b));
            }
        }
    }
    float s = acc[0].x + acc[1].y;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
static int g_cus = 256;
template <int KIND> void run_kind(float* out) {
    const int iters = KIND == K_MFMA ? 600 : 1500;
    for (int w : {1, 2, 4}) {
        float ms = timeit([&] { hipLaunchKernelGGL((k_kind<KIND>), dim3(g_cus * w), dim3(256), 4096, 0, out, iters, 1.0f, 0.5f); });
        printf("%-22s waves/SIMD=%d: %6.2f ns per instruction per SIMD (= %5.2f cycles at 2.4 GHz)\n", kind_name[KIND], w,
               ms * 1e6 / (iters * 64.0 * w), ms * 1e6 / (iters * 64.0 * w) * 2.4);
    }
}
template <int NM, int NV, int INTER, int SPLIT> void run_mix(float* out) {
    const int iters = 4000;
    for (int w : {1, 2, 4}) {
        if (SPLIT && w < 2) continue;
        float ms = timeit([&] { hipLaunchKernelGGL((k_mix<NM, NV, INTER, SPLIT>), dim3(g_cus * w), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        const double steps_per_simd = (double)iters * (SPLIT ? w / 2 : w);
        printf("mix %2d MFMA + %2d v_fma %s%s waves/SIMD=%d: %7.1f cycles per step per SIMD at 2.4 GHz (serial sum %d)\n", NM, NV,
               INTER ? "interleaved" : "MFMAs first", SPLIT ? " (MFMA waves | VALU waves)" : "", w, ms * 1e6 / steps_per_simd * 2.4, NM * 16 + NV * 4);
    }
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); g_cus = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", prop.name, g_cus, prop.clockRate / 1000);
    float* out; hipMalloc(&out, 1 << 24);
    run_kind<K_FMA>(out); run_kind<K_PKFMA>(out); run_kind<K_MIX>(out); run_kind<K_MIXLO>(out); run_kind<K_EXP>(out); run_kind<K_RCP>(out);
    run_kind<K_PERM16>(out); run_kind<K_CNDMASK>(out); run_kind<K_MFMA>(out); run_kind<K_LDS>(out); run_kind<K_SALU>(out);
    run_kind<K_PKMULH>(out); run_kind<K_PKFMAH>(out); run_kind<K_CVTPK>(out); run_kind<K_CNDS>(out); run_kind<K_PKADD>(out); run_kind<K_MAX>(out); run_kind<K_CVT32>(out); run_kind<K_FMAABS>(out);
    run_kind<K_CND64VCC>(out); run_kind<K_CMPCND32>(out); run_kind<K_CMPCND64>(out); run_kind<K_CMP32>(out);
    run_mix<12, 0, 0, 0>(out); run_mix<0, 60, 0, 0>(out);
    run_mix<12, 60, 0, 0>(out); run_mix<12, 60, 1, 0>(out); run_mix<12, 60, 0, 1>(out);
    run_mix<12, 36, 1, 0>(out); run_mix<12, 24, 1, 0>(out); run_mix<12, 12, 1, 0>(out);
    return 0;
}

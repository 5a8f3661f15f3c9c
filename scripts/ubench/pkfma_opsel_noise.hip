// pkfma_opsel_noise.hip -- [r6] the DIEN flaky tiles came down to ONE instruction (scripts/r06/: the failing build's assembly with only this
// instruction replaced is clean): `v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0]` returns src2.lo (the product lost) in its LOW half
// for lanes 48..63, about once in 1e4 executions, only with more than one wave per SIMD; dien_site_repro.hip (every wave running the site's own
// instructions) does not reproduce it.  Which neighbour does it need?  Here the waves of a SIMD play roles: wave 0 of each SIMD runs the packed fma
// in a loop and checks it; the other three run a NOISE loop of one instruction kind:
//   0 none (s_nop)  1 v_exp_f32 / v_rcp_f32  2 v_fma_mix_f32 op_sel_hi:[0,0,1]  3 v_cvt_pk_f16_f32  4 v_permlane16_swap / 32_swap  5 v_mfma_f32_16x16x32_f16
//   6 ds_read_b128  7 v_pk_mul_f32  8 v_cndmask_b32  9 global_load_dwordx4  10 all of them
// FORM 0 = op_sel:[0,1,0], 1 = no op_sel (pair swapped).
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/pkfma_opsel_noise scripts/ubench/pkfma_opsel_noise.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NOISE, int FORM>
__global__ __launch_bounds__(1024) void k_noise(int iters, const float* __restrict__ g, unsigned* bad, unsigned* info) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 1024) lds[i] = 1.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned laddr = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&lds[0]) + lane * 16;
    if (wave < 4) {                                               // the victim: one wave per SIMD
        unsigned nbad = 0, where = 0;
        for (int it = 0; it < iters; ++it) {
            float o4, o5;
            const float a0 = 3.0f + (it & 7), a1 = 5.0f, s0 = FORM == 0 ? 100.0f : 2.0f, s1 = FORM == 0 ? 2.0f : 100.0f, b0 = 0.5f, b1 = 0.25f;
            asm volatile(
                "v_mov_b32 v36, %2\n\tv_mov_b32 v37, %3\n\tv_mov_b32 v0, %4\n\tv_mov_b32 v1, %5\n\tv_mov_b32 v48, %6\n\tv_mov_b32 v49, %7\n\t"
                "s_nop 3\n\t"
                ".if %8 == 0\n\tv_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0]\n\t.else\n\tv_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel_hi:[1,0,1]\n\t.endif\n\t"
                "s_nop 3\n\t"
                "v_mov_b32 %0, v4\n\tv_mov_b32 %1, v5"
                : "=&v"(o4), "=&v"(o5) : "v"(a0), "v"(a1), "v"(s0), "v"(s1), "v"(b0), "v"(b1), "n"(FORM)
                : "v0", "v1", "v4", "v5", "v36", "v37", "v48", "v49");
            const float e4 = a0 * 2.0f + b0, e5 = a1 * 2.0f + b1;
            if (o4 != e4 || o5 != e5) { ++nbad; where |= (o4 != e4 ? 1u : 0u) | (o5 != e5 ? 2u : 0u) | (1u << (4 + (lane >> 4))); info[2] = __float_as_uint(o4 != e4 ? o4 : o5); }
        }
        if (nbad) { atomicAdd(bad, nbad); atomicOr(info, where); }
        return;
    }
    // noise waves: about the same run time as the victim's loop
    float x = 1.0f + lane * 0.001f, y = 0.5f, z = 0.f;
    const float* gp = g + (size_t)((blockIdx.x * 1024 + threadIdx.x) & 0xffff) * 4;
    for (int it = 0; it < iters; ++it) {
        if (NOISE == 0) asm volatile("s_nop 7\n\ts_nop 7");
        if (NOISE == 1 || NOISE == 10) asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(x), "+v"(y));
        if (NOISE == 2 || NOISE == 10) asm volatile("v_fma_mix_f32 %0, %1, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %0, %1, %1, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(z) : "v"(x), "v"(y));
        if (NOISE == 3 || NOISE == 10) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_cvt_pk_f16_f32 %0, %2, %1" : "+v"(z) : "v"(x), "v"(y));
        if (NOISE == 4 || NOISE == 10) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
        if (NOISE == 5 || NOISE == 10) asm volatile("v_mfma_f32_16x16x32_f16 v[60:63], v[64:67], v[64:67], 0\n\tv_mfma_f32_16x16x32_f16 v[60:63], v[64:67], v[64:67], v[60:63]" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        if (NOISE == 6 || NOISE == 10) asm volatile("ds_read_b128 v[68:71], %0\n\tds_read_b64 v[72:73], %0\n\ts_waitcnt lgkmcnt(0)" :: "v"(laddr) : "v68", "v69", "v70", "v71", "v72", "v73", "memory");
        if (NOISE == 7 || NOISE == 10) asm volatile("v_pk_mul_f32 v[74:75], v[74:75], v[74:75]\n\tv_pk_fma_f32 v[74:75], v[74:75], v[74:75], v[74:75] op_sel_hi:[1,0,1]" ::: "v74", "v75");
        if (NOISE == 8 || NOISE == 10) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(z) : "v"(x), "v"(y) : "vcc");
        if (NOISE == 9 || NOISE == 10) asm volatile("global_load_dwordx4 v[76:79], %0, off\n\ts_waitcnt vmcnt(0)" :: "v"(gp) : "v76", "v77", "v78", "v79", "memory");
    }
    if (x + y + z == 12345.f) info[3] = 1;
}

static int g_iters = 20000;
template <int NOISE, int FORM>
int run(const float* g, unsigned* d) {
    CHECK(hipMemset(d, 0, 32));
    hipLaunchKernelGGL((k_noise<NOISE, FORM>), dim3(256), dim3(1024), 0, 0, g_iters, g, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[4];
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[3], 4);
    static const char* names[] = {"s_nop", "v_exp / v_rcp", "v_fma_mix_f32", "v_cvt_pk_f16_f32", "permlane swaps", "v_mfma", "ds_read", "v_pk_mul / v_pk_fma", "v_cmp / v_cndmask", "global_load", "all"};
    printf("form %d, noise %-20s: %9u wrong lanes of %llu", FORM, names[NOISE], h[0], 64ull * 256 * 4 * g_iters);
    if (h[0]) printf("   (lo %d hi %d; lane quarters %x; a wrong value %g)", h[1] & 1, (h[1] >> 1) & 1, (h[1] >> 4) & 15, w);
    printf("\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    unsigned* d; float* g;
    CHECK(hipMalloc((void**)&d, 32));
    CHECK(hipMalloc((void**)&g, 65536 * 16 + 64));
    CHECK(hipMemset(g, 0, 65536 * 16 + 64));
    run<0, 0>(g, d); run<1, 0>(g, d); run<2, 0>(g, d); run<3, 0>(g, d); run<4, 0>(g, d); run<5, 0>(g, d); run<6, 0>(g, d); run<7, 0>(g, d); run<8, 0>(g, d); run<9, 0>(g, d); run<10, 0>(g, d);
    run<10, 1>(g, d);
    return 0;
}

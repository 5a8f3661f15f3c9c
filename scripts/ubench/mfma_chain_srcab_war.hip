// mfma_chain_srcab_war.hip -- mfma_srcab_war.hip asked whether a load may land in SrcA / SrcB of an INDEPENDENT MFMA issued just before it (no:
// safe).  This asks the same of the LAST MFMA of a DEPENDENT chain (SrcC = the previous MFMA's result), which is what hipcc emits in the failing
// k_dien_seq_mfma build (docs/open_issue_dien_tiles.md):
//     v_mfma D, Alo, Bh, 0 ; v_mfma D, Ahi, Bl, D ; v_mfma D, Ahi, Bh, D ; ds_read_b128 Ahi, <next block's fragment> ; ...
// If a dependent MFMA leaves the wave's issue stage before its SrcC is ready and picks its A / B operands up only when it starts, then with four
// waves per SIMD queueing chains on one matrix pipe the load (64+ cycles) can beat it.
// Per iteration: A2 := ones ; CHAIN-1 MFMAs D += A.B (32 each) ; D += A2.B ; GAP x s_nop 0 ; ds_read_b128 A2 <- LDS (zeros) ; wait ; check D == 32 CHAIN.
// A zeroed A2 read by the last MFMA gives 32 (CHAIN-1).  WHICH: 0 = the load lands in SrcA, 1 = SrcB.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_chain_srcab_war scripts/ubench/mfma_chain_srcab_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int CHAIN, int GAP, int WHICH, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_war(int iters, unsigned* bad, unsigned* seen) {
    __shared__ float zeros[64 * 4];
    for (int i = threadIdx.x; i < 64 * 4; i += WPB * 64) zeros[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&zeros[0]) + lane * 16;
    h8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
    unsigned nbad = 0;
    float worst = 0.f;
    for (int it = 0; it < iters; ++it) {
        float o0, o1, o2, o3;
        asm volatile(
            "v_mov_b32 v104, %4\n\tv_mov_b32 v105, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7\n\t"      // A2 (or B2) := ones
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], %8, %8, 0\n\t"
            ".rept %9\n\tv_mfma_f32_16x16x32_f16 v[100:103], %8, %8, v[100:103]\n\t.endr\n\t"
            ".if %10 == 0\n\tv_mfma_f32_16x16x32_f16 v[100:103], v[104:107], %8, v[100:103]\n\t.else\n\tv_mfma_f32_16x16x32_f16 v[100:103], %8, v[104:107], v[100:103]\n\t.endif\n\t"
            ".rept %11\n\ts_nop 0\n\t.endr\n\t"
            "ds_read_b128 v[104:107], %12\n\t"
            "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
            : "v"(((float*)&a)[0]), "v"(((float*)&a)[1]), "v"(((float*)&a)[2]), "v"(((float*)&a)[3]), "v"(a), "n"(CHAIN - 2), "n"(WHICH), "n"(GAP), "v"(addr)
            : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
        const float want = 32.f * CHAIN;
        if (o0 != want || o1 != want || o2 != want || o3 != want) { ++nbad; worst = o0 != want ? o0 : (o1 != want ? o1 : (o2 != want ? o2 : o3)); }
    }
    if (nbad) { atomicAdd(bad, nbad); atomicExch(seen, __float_as_uint(worst)); }
}

static int g_iters = 20000;
template <int CHAIN, int GAP, int WHICH, int WPB>
int run(int bpc, unsigned* d) {
    CHECK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL((k_war<CHAIN, GAP, WHICH, WPB>), dim3(256 * bpc), dim3(WPB * 64), 0, 0, g_iters, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[2];
    CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[1], 4);
    printf("load into Src%c of the last of %d chained MFMAs, %d wait states in front of the load, %2d waves/SIMD: %10u wrong lanes of %llu", WHICH ? 'B' : 'A', CHAIN, GAP,
           WPB * bpc / 4, h[0], 64ull * 256 * bpc * WPB * g_iters);
    if (h[0]) printf("   (a wrong value: %g, expected %g)", w, 32.f * CHAIN);
    printf("\n");
    return 0;
}
#define ROW(CHAIN, GAP, WHICH) run<CHAIN, GAP, WHICH, 4>(4, d); run<CHAIN, GAP, WHICH, 4>(2, d); run<CHAIN, GAP, WHICH, 4>(1, d); run<CHAIN, GAP, WHICH, 16>(1, d);

int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 8));
    ROW(2, 0, 0) ROW(3, 0, 0) ROW(4, 0, 0) ROW(6, 0, 0) ROW(3, 2, 0) ROW(3, 4, 0) ROW(3, 8, 0)
    ROW(2, 0, 1) ROW(3, 0, 1) ROW(4, 0, 1) ROW(6, 0, 1) ROW(3, 4, 1)
    return 0;
}

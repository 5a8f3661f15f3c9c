// mfma_to_valu_raw.hip -- how many wait states does a VALU instruction need behind the v_mfma_f32_16x16x32_f16 whose result it reads, and
// does the answer change (a) behind a dependent CHAIN of MFMAs, (b) when INDEPENDENT MFMAs or LDS reads are what fills the wait states,
// (c) with four waves per SIMD sharing the matrix pipe?  [r6; ADVICE r05: the hunt for k_dien_seq_mfma's flaky tiles covered VALU -> MFMA and
// the MFMA's WAR hazards (valu_to_mfma.hip, mfma_srcc_war.hip, mfma_srcab_war.hip), not MFMA -> VALU RAW.]
//
// hipcc 7.2 pads an MFMA -> VALU read of this opcode to EIGHT wait states, counting every instruction in between as one (LLVM's rule for a
// 4-pass XDL op on gfx950: passes + 3 + 1), e.g. in the failing DIEN build (docs/open_issue_dien_tiles.md):
//     v_mfma D, ..  (third of a dependent chain) ; 5 x ds_read ; s_waitcnt ; v_mfma E, .., 0 ; s_waitcnt ; v_pk_fma_f32 .., D[0:1], ..
// The test, per iteration and wave:  D := 1000 (stale marker) ; CHAIN dependent MFMAs into D (all-ones operands: +32 each) ; NLDS independent
// ds_read_b128 ; MID independent MFMAs ; GAP x s_nop 0 ; v_pk_fma_f32 O, D[0:1], 1.0, 0 ; check O == 32 * CHAIN.  Wait states between the last
// MFMA into D and its reader = NLDS + MID + GAP.  Registers are fixed (v[100:139]) so that the string can name halves of D.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_to_valu_raw scripts/ubench/mfma_to_valu_raw.hip
//   scripts/ubench/mfma_to_valu_raw [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// WPB waves per block, BPC blocks per CU (grid = 256 * BPC): WPB * BPC / 4 waves per SIMD
template <int CHAIN, int NLDS, int MID, int GAP, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_raw(int iters, unsigned* bad, unsigned* seen) {
    __shared__ float junk[64 * 4 * 4];
    for (int i = threadIdx.x; i < 64 * 4 * 4; i += WPB * 64) junk[i] = 3.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&junk[0]) + lane * 16;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
    const f2 one = {1.f, 1.f}, zero = {0.f, 0.f};
    const float stale = 1000.0f;
    unsigned nbad = 0;
    float worst = 0.f;
    for (int it = 0; it < iters; ++it) {
        float o0, o1;
        asm volatile(
            "v_mov_b32 v100, %4\n\tv_mov_b32 v101, %4\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %4\n\t"
            "s_nop 7\n\ts_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], %2, %3, 0\n\t"
            ".rept %5\n\tv_mfma_f32_16x16x32_f16 v[100:103], %2, %3, v[100:103]\n\t.endr\n\t"
            ".if %6 > 0\n\tds_read_b128 v[116:119], %7\n\t.endif\n\t"
            ".if %6 > 1\n\tds_read_b128 v[120:123], %7 offset:1024\n\t.endif\n\t"
            ".if %6 > 2\n\tds_read_b128 v[124:127], %7 offset:2048\n\t.endif\n\t"
            ".if %6 > 3\n\tds_read_b128 v[128:131], %7 offset:3072\n\t.endif\n\t"
            ".if %6 > 4\n\tds_read_b128 v[132:135], %7\n\t.endif\n\t"
            ".if %8 > 0\n\tv_mfma_f32_16x16x32_f16 v[104:107], %2, %3, 0\n\t.endif\n\t"
            ".if %8 > 1\n\tv_mfma_f32_16x16x32_f16 v[108:111], %2, %3, 0\n\t.endif\n\t"
            ".if %8 > 2\n\tv_mfma_f32_16x16x32_f16 v[112:115], %2, %3, 0\n\t.endif\n\t"
            ".rept %9\n\ts_nop 0\n\t.endr\n\t"
            "v_pk_fma_f32 v[136:137], v[100:101], %10, %11\n\t"
            "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_mov_b32 %0, v136\n\tv_mov_b32 %1, v137"
            : "=&v"(o0), "=&v"(o1)
            : "v"(a), "v"(b), "v"(stale), "n"(CHAIN - 1), "n"(NLDS), "v"(addr), "n"(MID), "n"(GAP), "v"(one), "v"(zero)
            : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
              "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131",
              "v132", "v133", "v134", "v135", "v136", "v137");
        const float want = 32.f * CHAIN;
        if (o0 != want || o1 != want) { ++nbad; worst = o0 != want ? o0 : o1; }
    }
    if (nbad) { atomicAdd(bad, nbad); atomicExch(seen, __float_as_uint(worst)); }
}

static int g_iters = 4000;
template <int CHAIN, int NLDS, int MID, int GAP, int WPB>
int run(int bpc, unsigned* d) {
    CHECK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL((k_raw<CHAIN, NLDS, MID, GAP, WPB>), dim3(256 * bpc), dim3(WPB * 64), 0, 0, g_iters, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[2];
    CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[1], 4);
    printf("chain %d | %d ds_read + %d MFMA + %2d s_nop = %2d wait states | %2d waves/SIMD: %10u wrong of %llu", CHAIN, NLDS, MID, GAP, NLDS + MID + GAP,
           WPB * bpc / 4, h[0], 256ull * bpc * WPB * g_iters);
    if (h[0]) printf("   (a wrong value: %g, expected %g)", w, 32.f * CHAIN);
    printf("\n");
    return 0;
}

// one row of the sweep at 4 waves per SIMD (4 blocks of 4 waves per CU, the DIEN kernel's shape) and at 1
#define ROW(CHAIN, NLDS, MID, GAP) run<CHAIN, NLDS, MID, GAP, 4>(4, d); run<CHAIN, NLDS, MID, GAP, 4>(1, d);
#define SWEEP(CHAIN, NLDS, MID) ROW(CHAIN, NLDS, MID, 0) ROW(CHAIN, NLDS, MID, 1) ROW(CHAIN, NLDS, MID, 2) ROW(CHAIN, NLDS, MID, 3) ROW(CHAIN, NLDS, MID, 4) \
    ROW(CHAIN, NLDS, MID, 5) ROW(CHAIN, NLDS, MID, 6) ROW(CHAIN, NLDS, MID, 7) ROW(CHAIN, NLDS, MID, 8) ROW(CHAIN, NLDS, MID, 10)

int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 8));
    printf("== one MFMA, s_nop only ==\n");            SWEEP(1, 0, 0)
    printf("== chain of three, s_nop only ==\n");      SWEEP(3, 0, 0)
    printf("== chain of three, one independent MFMA in between ==\n");   SWEEP(3, 0, 1)
    printf("== chain of three, three independent MFMAs in between ==\n"); SWEEP(3, 0, 3)
    printf("== chain of three, five ds_read + one MFMA in between (the failing DIEN site) ==\n"); SWEEP(3, 5, 1)
    printf("== one MFMA, five ds_read in between ==\n"); SWEEP(1, 5, 0)
    return 0;
}

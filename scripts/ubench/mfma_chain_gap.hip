// mfma_chain_gap.hip -- a DEPENDENT chain of three v_mfma_f32_16x16x32_f16 whose members are NOT back to back: GAP1 instructions between the
// first and the second, GAP2 between the second and the third -- the shape hipcc's scheduler produced at the failing site of k_dien_seq_mfma<16,32>
// (docs/open_issue_dien_tiles.md; build/r06 listing):
//     v_mfma E, E(=A lo), Bh, 0 ; s_waitcnt ; v_pk_fma_f32 .. ; v_mfma E, Ahi, Bl, E ; v_fma ; v_fma ; ds_read ; v_pk_mul ; v_mfma D, Ahi, Bh, E
// LLVM pads nothing between an MFMA and the next MFMA that takes its whole result as SrcC ("exactly the same vDst": the hardware interlocks /
// forwards); does that hold for every gap, at four waves per SIMD?  Result must be 96 = 3 x 32 (all-ones operands).  FILL: 0 = s_nop 0, 1 = independent
// v_fma_f32, 2 = v_pk_fma_f32.  FIRST_A_IS_D: the first MFMA's SrcA register is its destination (as in the listing).  LAST_MOVES: the third MFMA
// writes another register quad than its SrcC.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_chain_gap scripts/ubench/mfma_chain_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int GAP1, int GAP2, int FILL, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_gap(int iters, unsigned* bad, unsigned* seen) {
    const int lane = threadIdx.x & 63;
    h8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
    unsigned nbad = 0;
    float worst = 0.f;
    const float one = 1.0f + 0.f * lane;
    for (int it = 0; it < iters; ++it) {
        float o0, o1, o2, o3;
        asm volatile(
            "v_mov_b32 v104, %4\n\tv_mov_b32 v105, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7\n\t"      // E := A (ones): SrcA of the first MFMA and its destination
            "v_mov_b32 v110, %9\n\tv_mov_b32 v111, %9\n\tv_mov_b32 v112, %9\n\tv_mov_b32 v113, %9\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[104:107], v[104:107], %8, 0\n\t"
            ".rept %10\n\t.if %12 == 0\n\ts_nop 0\n\t.elseif %12 == 1\n\tv_fma_f32 v110, v111, v112, v110\n\t.else\n\tv_pk_fma_f32 v[110:111], v[112:113], v[112:113], v[110:111]\n\t.endif\n\t.endr\n\t"
            "v_mfma_f32_16x16x32_f16 v[104:107], %8, %8, v[104:107]\n\t"
            ".rept %11\n\t.if %12 == 0\n\ts_nop 0\n\t.elseif %12 == 1\n\tv_fma_f32 v110, v111, v112, v110\n\t.else\n\tv_pk_fma_f32 v[110:111], v[112:113], v[112:113], v[110:111]\n\t.endif\n\t.endr\n\t"
            "v_mfma_f32_16x16x32_f16 v[100:103], %8, %8, v[104:107]\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
            : "v"(((float*)&a)[0]), "v"(((float*)&a)[1]), "v"(((float*)&a)[2]), "v"(((float*)&a)[3]), "v"(a), "v"(one), "n"(GAP1), "n"(GAP2), "n"(FILL)
            : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113");
        if (o0 != 96.f || o1 != 96.f || o2 != 96.f || o3 != 96.f) { ++nbad; worst = o0 != 96.f ? o0 : (o1 != 96.f ? o1 : (o2 != 96.f ? o2 : o3)); }
    }
    if (nbad) { atomicAdd(bad, nbad); atomicExch(seen, __float_as_uint(worst)); }
}

static int g_iters = 20000;
template <int GAP1, int GAP2, int FILL, int WPB>
int run(int bpc, unsigned* d) {
    CHECK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL((k_gap<GAP1, GAP2, FILL, WPB>), dim3(256 * bpc), dim3(WPB * 64), 0, 0, g_iters, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[2];
    CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[1], 4);
    printf("gaps %d / %d (%s), %2d waves/SIMD: %10u wrong lanes of %llu", GAP1, GAP2, FILL == 0 ? "s_nop" : FILL == 1 ? "v_fma_f32" : "v_pk_fma_f32", WPB * bpc / 4, h[0],
           64ull * 256 * bpc * WPB * g_iters);
    if (h[0]) printf("   (a wrong value: %g, expected 96)", w);
    printf("\n");
    return 0;
}
#define ROW(G1, G2, FILL) run<G1, G2, FILL, 4>(4, d); run<G1, G2, FILL, 4>(1, d);
#define SWEEP(FILL) ROW(0, 0, FILL) ROW(1, 0, FILL) ROW(2, 0, FILL) ROW(3, 0, FILL) ROW(4, 0, FILL) ROW(0, 1, FILL) ROW(0, 2, FILL) ROW(0, 3, FILL) ROW(0, 4, FILL) \
    ROW(1, 1, FILL) ROW(2, 2, FILL) ROW(2, 4, FILL) ROW(3, 3, FILL) ROW(4, 4, FILL) ROW(5, 5, FILL) ROW(6, 6, FILL) ROW(8, 8, FILL)

int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 8));
    SWEEP(0) SWEEP(1) SWEEP(2)
    return 0;
}

// mfma_srcab_war.hip -- does a load that lands in the SrcA / SrcB registers of an MFMA issued just before it corrupt that MFMA when the matrix
// pipe is backed up?  hipcc places `ds_read_b128 vX` DIRECTLY behind `v_mfma ..., vX(A), ...` (operands are read at issue, it assumes); in the
// failing k_dien_fused build the AUGRU gate's bias load went into the registers the next block's first MFMA had just been given as SrcA, right
// behind a chain of three dependent MFMAs, four waves per SIMD (k_dien_fused.h; scripts/ubench/mfma_srcc_war.hip asked the same about SrcC: no).
// The test: PRE MFMAs (a dependent chain, or independent ones) ; c = mfma(a2, b, 0) ; s_nop GAP ; ds_read_b128 -> a2 (or b2) ; check c == 32.
// Measured (profiles/r05/experiments/r05_32/mfma_srcab_war.txt, 5.2e9 lane-results per line): with 4, 8 or 16 MFMAs in front -- one dependent chain or
// four accumulators, 0 to 7 wait states before the load -- NOT ONE wrong result for SrcA or SrcB: the operands are safe from a load issued behind
// the MFMA, the compiler's assumption holds.  The PRE = 0 lines DO fail (262 144 and 8e8 wrong), for another reason worth knowing: there the
// `v_mov`s that refresh a2 / b2 sit directly in front of the MFMA inside the asm statement, and an MFMA that reads a register a VALU instruction
// wrote less than two wait states earlier gets the OLD value (here: last trip's 1000.0f) -- the hazard hipcc pads with `s_nop 1` wherever it can
// see both instructions, and one more thing an asm statement has to carry itself.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_srcab_war scripts/ubench/mfma_srcab_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// WHICH: 0 = the load lands in SrcA, 1 = in SrcB.  CHAIN: the PRE fillers are one dependent chain (true) or four independent accumulators.
template <int PRE, int GAP, int WHICH, bool CHAIN>
__global__ __launch_bounds__(1024, 4) void k_war(int iters, unsigned* bad, unsigned* worst) {
    __shared__ float junk[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 1024) junk[i] = 1000.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&junk[0]) + lane * 16;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0, c, x;
        h8 a2 = a, b2 = b;
        asm volatile("" : "+v"(a2), "+v"(b2));                       // (own registers for the operands the load will overwrite)
        for (int p = 0; p < PRE; p += 4) {
            if constexpr (CHAIN)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d0) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n\t"
                             "v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_f16 %3, %4, %5, %3"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
        }
        if constexpr (WHICH == 0)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0\n\t"
                         "s_nop %4\n\t"
                         "ds_read_b128 %1, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                         : "=&v"(c), "+v"(a2) : "v"(b2), "v"(addr), "n"(GAP) : "memory");
        else
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %1, 0\n\t"
                         "s_nop %4\n\t"
                         "ds_read_b128 %1, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                         : "=&v"(c), "+v"(b2) : "v"(a2), "v"(addr), "n"(GAP) : "memory");
        if (c[0] != 32.f || c[1] != 32.f || c[2] != 32.f || c[3] != 32.f) { ++nbad; atomicMax(worst, __float_as_uint(c[0] < 0 ? -c[0] : c[0])); }
        if (d0[0] + d1[0] + d2[0] + d3[0] < 0.f) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int PRE, int GAP, int WHICH, bool CHAIN>
int run(int iters, unsigned* d) {
    CHECK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL((k_war<PRE, GAP, WHICH, CHAIN>), dim3(256), dim3(1024), 0, 0, iters, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[2];
    CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[1], 4);
    printf("load into Src%c, %2d filler MFMAs (%s), %2d wait states before the load: %10u wrong lane-results of %llu  (a wrong c[0]: %g)\n", WHICH ? 'B' : 'A', PRE,
           CHAIN ? "one dependent chain" : "four accumulators ", GAP, h[0], 256ull * 1024 * iters, h[0] ? w : 0.f);
    return 0;
}

int main() {
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 8));
    const int N = 20000;
    run<0, 0, 0, true>(N, d);  run<4, 0, 0, true>(N, d);  run<8, 0, 0, true>(N, d);  run<16, 0, 0, true>(N, d);
    run<4, 0, 0, false>(N, d); run<8, 0, 0, false>(N, d); run<16, 0, 0, false>(N, d);
    run<8, 3, 0, true>(N, d);  run<8, 7, 0, true>(N, d);  run<16, 7, 0, true>(N, d);
    run<0, 0, 1, true>(N, d);  run<4, 0, 1, true>(N, d);  run<8, 0, 1, true>(N, d);  run<16, 0, 1, true>(N, d);
    run<8, 0, 1, false>(N, d); run<16, 0, 1, false>(N, d);
    return 0;
}

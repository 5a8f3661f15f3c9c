// mfma_srcc_war.hip -- does a load that lands in the SrcC registers of a QUEUED matrix instruction corrupt that instruction?
//
// Why (round 5).  k_dien_fused<16,...> returned, for one or a few whole 16-sample tiles per launch and different tiles every run, scores off
// by ~1e-4 (the recurrence's state by ~1e-3: the size of a lost lo.hi term of a split-f16 product).  Everything the compiler can see was in
// order (scripts/isa/isa_waitcnt_check.py, isa_undef_reads.py, asm_hazards.py: nothing), the inline-asm statements padded with wait states
// changed nothing, but ONE `s_nop 1` between the groups of three MFMAs -- a statement the scheduler may not move LDS reads across -- made it go
// away, and the builds that fail are the ones whose register allocation (128 VGPRs, four waves per SIMD) recycles the accumulator of a chain
//     c1 = mfma(al, bh, 0);  c2 = mfma(ah, bl, c1);  c3 = mfma(ah, bh, c2)        (c1, c2, c3 in DIFFERENT registers)
// for the next `ds_read_b128` THREE wait states behind the instruction that reads it as SrcC.  hipcc protects the write-after-write against the
// chain's destination (s_nop up to passes + 3) but has no rule for a load's write-after-READ of SrcC: LDS latency is assumed to cover it.  A
// dependent MFMA cannot read SrcC before its predecessor has finished, and with four waves per SIMD it also queues behind the other waves'
// matrix work; the load's data does not wait.
//
// The test: every wave of a 16-wave workgroup (four per SIMD) runs, N times,
//     PRE independent MFMAs (matrix pipe busy) ; c1 = a.b ; c2 = a.b + c1 ; c3 = a.b + c2 ; s_nop GAP ; ds_read_b128 -> c2 (1000.0f)
// with a = b = 1.0 (f16): c3 must be 96 in every lane.  A c3 of 1032 means MFMA #3 read SrcC AFTER the load had landed.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/mfma_srcc_war scripts/ubench/mfma_srcc_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PRE, int GAP, bool INPLACE>
__global__ __launch_bounds__(1024, 4) void k_war(int iters, unsigned* bad, unsigned* worst) {
    __shared__ float junk[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 1024) junk[i] = 1000.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(size_t)((__attribute__((address_space(3))) float*)&junk[0]) + lane * 16;   // LDS byte offset
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f4 c1, c2, c3, d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
        if constexpr (PRE >= 4)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_f16 %3, %4, %5, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
        if constexpr (PRE >= 8)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_f16 %3, %4, %5, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));
        if constexpr (INPLACE) {
            // the accumulator stays in ONE register quadruple: nothing to recycle (the control)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, 0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"
                         "s_nop %5\n\t"
                         "ds_read_b128 %1, %4\n\t"
                         "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                         : "=&v"(c3), "=&v"(c2) : "v"(a), "v"(b), "v"(addr), "n"(GAP) : "memory");
        } else {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %3, %4, 0\n\t"
                         "v_mfma_f32_16x16x32_f16 %1, %3, %4, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %2, %3, %4, %1\n\t"
                         "s_nop %6\n\t"
                         "ds_read_b128 %1, %5\n\t"                       // lands in SrcC of the third instruction
                         "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                         : "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(a), "v"(b), "v"(addr), "n"(GAP) : "memory");
        }
        const bool wrong = c3[0] != 96.f || c3[1] != 96.f || c3[2] != 96.f || c3[3] != 96.f || c2[0] != 1000.f;
        if (wrong) { ++nbad; atomicMax(worst, __float_as_uint(c3[0])); }
        if (d0[0] + d1[0] + d2[0] + d3[0] < 0.f) ++nbad;                  // (keeps the filler alive)
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int PRE, int GAP, bool INPLACE>
int run(int iters, unsigned* d) {
    CHECK(hipMemset(d, 0, 8));
    hipLaunchKernelGGL((k_war<PRE, GAP, INPLACE>), dim3(256), dim3(1024), 0, 0, iters, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[2];
    CHECK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[1], 4);
    printf("%-9s filler MFMAs %d  wait states before the load %2d : %10u wrong lane-results of %llu  (a wrong c3[0]: %g)\n", INPLACE ? "in-place" : "recycled",
           PRE, GAP, h[0], 256ull * 1024 * iters, h[0] ? w : 0.f);
    return 0;
}

int main() {
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 8));
    const int N = 20000;
    run<0, 0, true>(N, d); run<8, 0, true>(N, d);
    run<0, 0, false>(N, d); run<0, 3, false>(N, d); run<0, 7, false>(N, d); run<0, 15, false>(N, d);
    run<4, 0, false>(N, d); run<4, 3, false>(N, d); run<4, 7, false>(N, d); run<4, 15, false>(N, d);
    run<8, 0, false>(N, d); run<8, 3, false>(N, d); run<8, 7, false>(N, d); run<8, 15, false>(N, d);
    return 0;
}

// valu_to_mfma.hip -- how many wait states does an MFMA need behind the VALU instruction that wrote one of its operand registers, per PRODUCER?
// hipcc pads two (`v_mov ; s_nop 1 ; v_mfma`); k_dien_*'s B operands are written by v_cvt_pk_f16_f32 (new on gfx950, two values per issue) and
// v_pk_mul_f32 right in front of the MFMAs (dyn_split.h).  mfma_srcab_war.hip showed what too few wait states do: the MFMA reads the OLD value.
// The test, all in ONE asm statement on fixed registers: v63 (last register of B = v[60:63]) := 0, long wait, PRODUCER writes v63 := two 1.0
// halfs, N wait states, v_mfma_f32_16x16x32_f16 c = a . B with a = 1: c = 32 if the new value was seen, 24 if the old one.
// Measured (profiles/r05/experiments/r05_32/valu_to_mfma.txt, 1.3e9 lane-results per line): with NO wait state 94 - 99 % of the MFMAs read the old
// value, whatever the producer (v_mov_b32, v_cvt_pk_f16_f32, v_pk_mul_f32, v_fma_mix_f32); with ONE wait state or more none does.  hipcc's two
// are enough for every producer the kernels use: not the cause of k_dien_*'s tiles either.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/valu_to_mfma scripts/ubench/valu_to_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)
#define PRE "v_mov_b32 v60, %[pat]\n\tv_mov_b32 v61, %[pat]\n\tv_mov_b32 v62, %[pat]\n\tv_mov_b32 v63, 0\n\ts_nop 7\n\ts_nop 7\n\t"
#define POST "v_mfma_f32_16x16x32_f16 %[c], %[a], v[60:63], 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
#define CASE(P, PROD, W, WAIT) \
    if (prod == P && wait == W) asm volatile(PRE PROD "\n\t" WAIT POST : [c] "=&v"(c) : [a] "v"(a), [pat] "v"(pat), [one] "v"(one), [pat2] "v"(pat2), [ones2] "v"(ones2) \
                                             : "v60", "v61", "v62", "v63", "memory");
#define PRODUCER(P, PROD) CASE(P, PROD, 0, "") CASE(P, PROD, 1, "s_nop 0\n\t") CASE(P, PROD, 2, "s_nop 1\n\t") CASE(P, PROD, 3, "s_nop 2\n\t") CASE(P, PROD, 4, "s_nop 3\n\t") CASE(P, PROD, 6, "s_nop 5\n\t")

__global__ __launch_bounds__(1024, 4) void k(int prod, int wait, int iters, unsigned* bad) {
    const unsigned pat = 0x3C003C00u;                                  // two halfs 1.0
    const float one = 1.0f;
    f2 pat2 = {__uint_as_float(pat), __uint_as_float(pat)}, ones2 = {1.f, 1.f};
    h8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        f4 c = {0, 0, 0, 0};
        PRODUCER(0, "v_mov_b32 v63, %[pat]")
        PRODUCER(1, "v_cvt_pk_f16_f32 v63, %[one], %[one]")
        PRODUCER(2, "v_pk_mul_f32 v[62:63], %[pat2], %[ones2]")
        PRODUCER(3, "v_fma_mix_f32 v63, %[pat], %[one], 0 op_sel_hi:[0,0,0]")     // (f32 x f32 + 0: the bits of pat)
        if (c[0] != 32.f || c[1] != 32.f || c[2] != 32.f || c[3] != 32.f) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 4));
    const char* names[] = {"v_mov_b32", "v_cvt_pk_f16_f32", "v_pk_mul_f32", "v_fma_mix_f32"};
    const int waits[] = {0, 1, 2, 3, 4, 6};
    for (int p = 0; p < 4; ++p)
        for (int w : waits) {
            CHECK(hipMemset(d, 0, 4));
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, p, w, 5000, d);
            CHECK(hipDeviceSynchronize());
            unsigned h;
            CHECK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
            printf("%-18s -> v_mfma_f32_16x16x32_f16 SrcB, %d wait states: %10u wrong of %llu\n", names[p], w, h, 256ull * 1024 * 5000);
        }
    return 0;
}

// pkfma_opsel_mfma.hip -- [r6] characterises what pkfma_opsel_noise.hip found: a packed-f32 VALU instruction that takes the HIGH half of a source for
// its LOW result (`v_pk_fma_f32 ... op_sel:[0,1,0]`) loses the product in lanes 48..63 (returns src2.lo) when ANOTHER wave of the same SIMD keeps the
// matrix pipe busy.  Victim instruction forms x kinds of MFMA in the three neighbour waves.  One victim wave per SIMD, values chosen so that every
// mis-selected operand gives a different number.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/pkfma_opsel_mfma scripts/ubench/pkfma_opsel_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// operands: v[36:37] = (3, 5)   v[0:1] = (100, 2)   v[48:49] = (0.5, 0.25);  f16 pair in v50 = (lo 1.0, hi 4.0)
#define VICTIMS(X) \
    X(0, "v_pk_fma_f32 op_sel:[0,1,0]",            "v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0]",               3 * 2 + 0.5f,    5 * 2 + 0.25f) \
    X(1, "v_pk_fma_f32 op_sel_hi:[1,0,1]",         "v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel_hi:[1,0,1]",            3 * 100 + 0.5f,  5 * 100 + 0.25f) \
    X(2, "v_pk_fma_f32 (no op_sel)",               "v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49]",                              3 * 100 + 0.5f,  5 * 2 + 0.25f) \
    X(3, "v_pk_fma_f32 op_sel:[1,0,0]",            "v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[1,0,0]",               5 * 100 + 0.5f,  5 * 2 + 0.25f) \
    X(4, "v_pk_fma_f32 op_sel:[0,0,1]",            "v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,0,1]",               3 * 100 + 0.25f, 5 * 2 + 0.25f) \
    X(5, "v_pk_mul_f32 op_sel:[0,1]",              "v_pk_mul_f32 v[4:5], v[36:37], v[0:1] op_sel:[0,1]",                           3 * 2.f,         5 * 2.f) \
    X(6, "v_pk_add_f32 op_sel:[0,1]",              "v_pk_add_f32 v[4:5], v[36:37], v[0:1] op_sel:[0,1]",                           3 + 2.f,         5 + 2.f) \
    X(7, "v_fma_mix_f32 op_sel:[0,0,1] (f16 hi)",  "v_fma_mix_f32 v4, v36, v1, -v50 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 v5, v37, v1, -v50 op_sel:[0,0,0] op_sel_hi:[0,0,1]", 3 * 2 - 4.f, 5 * 2 - 1.f) \
    X(8, "v_pk_fma_f32 op_sel_hi:[1,0,1] op_sel:[0,1,0]", "v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0] op_sel_hi:[1,0,1]", 3 * 2 + 0.5f, 5 * 100 + 0.25f) \
    X(9, "v_pk_fma_f32 op_sel_hi:[0,1,1]",         "v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel_hi:[0,1,1]",            3 * 100 + 0.5f,  3 * 2 + 0.25f) \
    X(10, "v_pk_fma_f32 SGPR src1 op_sel:[0,1,0]",  "s_mov_b32 s20, 0x42c80000\n\ts_mov_b32 s21, 2.0\n\ts_nop 3\n\tv_pk_fma_f32 v[4:5], v[36:37], s[20:21], v[48:49] op_sel:[0,1,0]", 3 * 2 + 0.5f, 5 * 2 + 0.25f) \
    X(11, "v_pk_mul_f32 op_sel:[1,0] (src0 hi)",    "v_pk_mul_f32 v[4:5], v[0:1], v[36:37] op_sel:[1,0]",                          2 * 3.f,         2 * 5.f) \
    X(12, "v_pk_mov_b32 op_sel:[0,1]",              "v_pk_mov_b32 v[4:5], v[36:37], v[0:1] op_sel:[0,1]",                          3.f,             2.f) \
    X(13, "v_pk_mov_b32 op_sel:[1,0] (hipcc's form)", "v_pk_mov_b32 v[4:5], v[36:37], v[0:1] op_sel:[1,0]",                        5.f,             100.f) \
    X(14, "v_pk_fma_f16 op_sel:[0,1,0] (halves of one register)", "v_mov_b32 v36, 0x40003c00\n\tv_mov_b32 v0, 0x44004200\n\tv_mov_b32 v48, 0x34003800\n\ts_nop 3\n\tv_pk_fma_f16 v4, v36, v0, v48 op_sel:[0,1,0]\n\tv_mov_b32 v5, v4", __builtin_bit_cast(float, 0x48204480u), __builtin_bit_cast(float, 0x48204480u))

template <int V> struct Victim;
#define X(N, NAME, ASM, E4, E5) template <> struct Victim<N> { static constexpr const char* name = NAME; static constexpr float e4 = E4, e5 = E5; \
    static __device__ __forceinline__ void run(float& o4, float& o5) { \
        asm volatile("v_mov_b32 v36, 3.0\n\tv_mov_b32 v37, 0x40a00000\n\tv_mov_b32 v0, 0x42c80000\n\tv_mov_b32 v1, 2.0\n\tv_mov_b32 v48, 0.5\n\tv_mov_b32 v49, 0x3e800000\n\tv_mov_b32 v50, 0x44003c00\n\t" \
                     "s_nop 3\n\t" ASM "\n\ts_nop 3\n\tv_mov_b32 %0, v4\n\tv_mov_b32 %1, v5" : "=&v"(o4), "=&v"(o5) :: "v0", "v1", "v4", "v5", "v36", "v37", "v48", "v49", "v50", "s20", "s21"); } };
VICTIMS(X)
#undef X

// NOISE: 0 none, 1 v_mfma_f32_16x16x32_f16 chain of 2, 2 v_mfma_f32_16x16x4_f32, 3 v_mfma_f32_32x32x16_f16, 4 ONE 16x16x32 f16 per 64 wait states, 5 v_mfma_f32_16x16x16_f16 (the gfx90a op)
template <int V, int NOISE>
__global__ __launch_bounds__(1024) void k_v(int iters, unsigned* bad, unsigned* info) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        unsigned nbad = 0, where = 0;
        for (int it = 0; it < iters; ++it) {
            float o4, o5;
            Victim<V>::run(o4, o5);
            if (o4 != Victim<V>::e4 || o5 != Victim<V>::e5) {
                ++nbad; where |= (o4 != Victim<V>::e4 ? 1u : 0u) | (o5 != Victim<V>::e5 ? 2u : 0u) | (1u << (4 + (lane >> 4)));
                info[2] = __float_as_uint(o4 != Victim<V>::e4 ? o4 : o5);
            }
        }
        if (nbad) { atomicAdd(bad, nbad); atomicOr(info, where); }
        return;
    }
    for (int it = 0; it < iters; ++it) {
        if (NOISE == 0) asm volatile("s_nop 7\n\ts_nop 7");
        if (NOISE == 1) asm volatile("v_mfma_f32_16x16x32_f16 v[60:63], v[64:67], v[64:67], 0\n\tv_mfma_f32_16x16x32_f16 v[60:63], v[64:67], v[64:67], v[60:63]" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        if (NOISE == 2) asm volatile("v_mfma_f32_16x16x4_f32 v[60:63], v64, v65, 0\n\tv_mfma_f32_16x16x4_f32 v[60:63], v64, v65, v[60:63]" ::: "v60", "v61", "v62", "v63", "v64", "v65");
        if (NOISE == 3) asm volatile("v_mfma_f32_32x32x16_f16 v[60:75], v[76:79], v[76:79], 0" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79");
        if (NOISE == 4) asm volatile("v_mfma_f32_16x16x32_f16 v[60:63], v[64:67], v[64:67], 0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        if (NOISE == 5) asm volatile("v_mfma_f32_16x16x16_f16 v[60:63], v[64:65], v[64:65], 0\n\tv_mfma_f32_16x16x16_f16 v[60:63], v[64:65], v[64:65], v[60:63]" ::: "v60", "v61", "v62", "v63", "v64", "v65");
        if (NOISE == 6) asm volatile("v_mfma_f32_16x16x32_bf16 v[60:63], v[64:67], v[64:67], 0\n\tv_mfma_f32_16x16x32_bf16 v[60:63], v[64:67], v[64:67], v[60:63]" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        if (NOISE == 7) asm volatile("v_mfma_f32_16x16x32_f16 v[60:63], v[64:67], v[64:67], 0\n\tv_mfma_f32_16x16x32_f16 v[68:71], v[64:67], v[64:67], 0" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
        if (NOISE == 8) asm volatile("v_mfma_f32_16x16x32_f16 v[60:63], v[64:67], v[64:67], 0\n\ts_nop 7" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
        if (NOISE == 9) asm volatile("v_mfma_i32_16x16x64_i8 v[60:63], v[64:67], v[64:67], 0\n\tv_mfma_i32_16x16x64_i8 v[60:63], v[64:67], v[64:67], v[60:63]" ::: "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
    }
}

static int g_iters = 40000;
template <int V, int NOISE>
int run(unsigned* d) {
    CHECK(hipMemset(d, 0, 32));
    hipLaunchKernelGGL((k_v<V, NOISE>), dim3(256), dim3(1024), 0, 0, g_iters, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[4];
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[3], 4);
    static const char* nn[] = {"no MFMA", "16x16x32 f16 x2", "16x16x4 f32 x2", "32x32x16 f16", "one 16x16x32 f16 / 64 states", "16x16x16 f16 x2", "16x16x32 bf16 x2", "16x16x32 f16, two independent", "one 16x16x32 f16 / 8 states", "16x16x64 i8 x2"};
    printf("%-46s | neighbours: %-28s: %9u wrong lanes of %llu", Victim<V>::name, nn[NOISE], h[0], 64ull * 256 * 4 * g_iters);
    if (h[0]) printf("   (lo %d hi %d; lane quarters %x; a wrong value %g; expected %g / %g)", h[1] & 1, (h[1] >> 1) & 1, (h[1] >> 4) & 15, w, Victim<V>::e4, Victim<V>::e5);
    printf("\n");
    return 0;
}
#define ALLN(V) run<V, 0>(d); run<V, 1>(d); run<V, 2>(d); run<V, 3>(d); run<V, 4>(d); run<V, 5>(d);
#define NEWN(V) run<V, 6>(d); run<V, 7>(d); run<V, 8>(d); run<V, 9>(d);
int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 32));
    const bool second = argc > 2;                                 // (second sweep: the SGPR form, src0-hi multiply, more kinds of MFMA)
    if (!second) { ALLN(0) ALLN(1) ALLN(2) ALLN(3) ALLN(4) ALLN(5) ALLN(6) ALLN(7) ALLN(8) ALLN(9) }
    else { ALLN(10) ALLN(11) NEWN(0) NEWN(10) NEWN(1) NEWN(5) ALLN(12) ALLN(13) ALLN(14) }
    return 0;
}

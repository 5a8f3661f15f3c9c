// dien_site_repro.hip -- the failing site of k_dien_seq_mfma<16,32> (docs/open_issue_dien_tiles.md, [r6] the dump of scripts/r06/dien_dump_run.py: in
// a bad tile the low half of `v_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0]` comes out as the BIAS in lanes 48..63 -- v36 of the
// MFMA chain's result read stale in the wave's last quarter, every input correct a few instructions later) as a stand-alone kernel: the same
// instructions on the same registers, all-ones operands so that every result is known:
//   chain (block 7): v[36:39] = Alo.Bh ; s_waitcnt ; += Ahi.Bl ; += Ahi.Bh  (= 96) ; five ds_read (one into Ahi) ; s_waitcnt ; first MFMA of the
//   next chain (block 6) ; s_waitcnt ; the packed fma on v[36:37] ; the next chain's second MFMA ; two v_fma_f32 on v38, v39.
// GAP extra `s_nop 0` in front of the packed fma; FORM 0 = the packed fma with op_sel:[0,1,0] (the failing build), 1 = two v_fma_f32 (what the
// clean builds have), 2 = packed, un pair swapped so that no op_sel is needed.  JIT: waves decorrelated by a lane-id dependent number of v_exp_f32
// in front of every iteration (the real kernel's waves sit in different phases of a step).
//   hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/dien_site_repro scripts/ubench/dien_site_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int GAP, int FORM, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_site(int iters, int jit, unsigned* bad, unsigned* info) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    // [0, 1024) words: fragments of ones (halfs 0x3C00); [1024, 1040): bias vectors (0.5, 0.25, 0.125, 0.0625 per q); [1040..]: scalars 1.0f
    for (int i = threadIdx.x; i < 4096; i += WPB * 64) lds[i] = i < 1024 ? 0x3C003C00u : (i < 1056 ? __float_as_uint(0.5f / (1 << (i & 3))) : __float_as_uint(1.0f));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned*)&lds[0]);
    const unsigned a_frag = base + lane * 16, a_vec = base + 4096 + (lane >> 4) * 16, a_uni = base + 4224;
    unsigned nbad = 0, where = 0;
    const unsigned ones = 0x3C003C00u;
    float sink = 1.0f + lane;
    const int spin = jit ? ((blockIdx.x * WPB + wave) * 7 + 3) % jit : 0;
    for (int it = 0; it < iters; ++it) {
        for (int s = 0; s < spin + (it & 3) * (jit ? 1 : 0); ++s) sink = __builtin_amdgcn_exp2f(sink) * 0.25f;
        float o4, o5, o12, o13;
        asm volatile(
            "v_mov_b32 v28, %4\n\tv_mov_b32 v29, %4\n\tv_mov_b32 v30, %4\n\tv_mov_b32 v31, %4\n\t"
            "v_mov_b32 v32, %4\n\tv_mov_b32 v33, %4\n\tv_mov_b32 v34, %4\n\tv_mov_b32 v35, %4\n\t"
            "v_mov_b32 v36, %4\n\tv_mov_b32 v37, %4\n\tv_mov_b32 v38, %4\n\tv_mov_b32 v39, %4\n\t"
            "v_mov_b32 v44, %4\n\tv_mov_b32 v45, %4\n\tv_mov_b32 v46, %4\n\tv_mov_b32 v47, %4\n\t"
            "ds_read_b128 v[40:43], %5\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_16x16x32_f16 v[36:39], v[36:39], v[28:31], 0\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f32_16x16x32_f16 v[36:39], v[40:43], v[32:35], v[36:39]\n\t"
            "v_mfma_f32_16x16x32_f16 v[36:39], v[40:43], v[28:31], v[36:39]\n\t"
            "ds_read_b128 v[40:43], %5 offset:1024\n\t"
            ".if %9 == 2\n\tds_read_b64 v[0:1], %7 offset:8\n\t.else\n\tds_read_b64 v[0:1], %7\n\t.endif\n\t"
            "ds_read_b128 v[48:51], %6\n\t"
            "ds_read_b128 v[52:55], %6 offset:64\n\t"
            "ds_read_b32 v106, %7 offset:16\n\t"
            "s_waitcnt lgkmcnt(5)\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[44:47], v[28:31], 0\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            ".rept %8\n\ts_nop 0\n\t.endr\n\t"
            ".if %9 == 0\n\tv_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel:[0,1,0]\n\t"
            ".elseif %9 == 1\n\tv_fma_f32 v4, v36, v1, v48\n\tv_fma_f32 v5, v37, v1, v49\n\t"
            ".else\n\tv_pk_fma_f32 v[4:5], v[36:37], v[0:1], v[48:49] op_sel_hi:[1,0,1]\n\t.endif\n\t"
            "v_mfma_f32_16x16x32_f16 v[44:47], v[40:43], v[32:35], v[44:47]\n\t"
            "v_fma_f32 v12, v38, v1, v50\n\t"
            "v_fma_f32 v13, v39, v1, v51\n\t"
            "s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
            "v_mov_b32 %0, v4\n\tv_mov_b32 %1, v5\n\tv_mov_b32 %2, v12\n\tv_mov_b32 %3, v13"
            : "=&v"(o4), "=&v"(o5), "=&v"(o12), "=&v"(o13)
            : "v"(ones), "v"(a_frag), "v"(a_vec), "v"(a_uni), "n"(GAP), "n"(FORM)
            : "memory", "v0", "v1", "v4", "v5", "v12", "v13", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",
              "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v106");
        const float e4 = 96.5f, e5 = 96.25f, e12 = 96.125f, e13 = 96.0625f;
        if (o4 != e4 || o5 != e5 || o12 != e12 || o13 != e13) {
            ++nbad;
            where |= (o4 != e4 ? 1u : 0u) | (o5 != e5 ? 2u : 0u) | (o12 != e12 ? 4u : 0u) | (o13 != e13 ? 8u : 0u) | (1u << (4 + (lane >> 4)));
            info[2] = __float_as_uint(o4 != e4 ? o4 : (o5 != e5 ? o5 : (o12 != e12 ? o12 : o13)));
        }
    }
    if (sink == 12345.f) info[3] = 1;
    if (nbad) { atomicAdd(bad, nbad); atomicOr(info, where); }
}

static int g_iters = 20000;
template <int GAP, int FORM, int WPB>
int run(int bpc, int jit, unsigned* d) {
    CHECK(hipMemset(d, 0, 32));
    hipLaunchKernelGGL((k_site<GAP, FORM, WPB>), dim3(256 * bpc), dim3(WPB * 64), 0, 0, g_iters, jit, d, d + 1);
    CHECK(hipDeviceSynchronize());
    unsigned h[4];
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    float w;
    memcpy(&w, &h[3], 4);
    printf("form %d (%s), %d extra states, %2d waves/SIMD, jitter %2d: %9u wrong lanes of %llu", FORM, FORM == 0 ? "v_pk_fma op_sel:[0,1,0]" : FORM == 1 ? "two v_fma_f32" : "v_pk_fma, pair lo",
           GAP, WPB * bpc / 4, jit, h[0], 64ull * 256 * bpc * WPB * g_iters);
    if (h[0]) printf("   (outputs wrong: v4 %d v5 %d v12 %d v13 %d; lane quarters %x; a wrong value %g)", h[1] & 1, (h[1] >> 1) & 1, (h[1] >> 2) & 1, (h[1] >> 3) & 1, (h[1] >> 4) & 15, w);
    printf("\n");
    return 0;
}
#define ROW(GAP, FORM) run<GAP, FORM, 4>(4, 0, d); run<GAP, FORM, 4>(4, 5, d); run<GAP, FORM, 4>(4, 23, d); run<GAP, FORM, 4>(1, 0, d); run<GAP, FORM, 4>(1, 23, d);

int main(int argc, char** argv) {
    if (argc > 1) g_iters = atoi(argv[1]);
    unsigned* d;
    CHECK(hipMalloc((void**)&d, 32));
    ROW(0, 0) ROW(1, 0) ROW(2, 0) ROW(4, 0) ROW(8, 0)
    ROW(0, 1) ROW(4, 1)
    ROW(0, 2) ROW(4, 2)
    return 0;
}

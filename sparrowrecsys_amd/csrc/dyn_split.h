// dyn_split.h -- per-sample (per MFMA column) dynamic power-of-two scaling + hi/lo f16 split of a hidden activation
// vector that sits in registers in the MFMA C/D layout, so that the NEXT Dense layer can run on
// v_mfma_f32_16x16x32_f16 with fp32-class accuracy although the activations' range is data dependent.
// Included inside sparrow_hip.hip's anonymous namespace (after k_din_attn.h: uses rows4_*).
//
// D[n][col] = sum_k A[n][k] B[k][col]: every column (= sample) may carry its own scale s_col, the result is unscaled per
// column afterwards.  s_col = 2^(14 - exponent(max_k |h[k][col]|)) puts the sample's largest activation in [2^14, 2^15);
// hi = f16(h s), lo = f16(h s - hi) keep 22 significand bits relative to that maximum (absolute error <= max|h| 2^-22
// per element: below fp32 rounding of the layer's own sums).  Weights are split the same way with a static scale.
//
// K-block layout: lane (r = column, q) holds h[16c + 4q + j] for chunk c; a K = 32 block b takes chunks 2b, 2b+1, i.e. the
// lane's 8 values are k_local = {4q..4q+3} U {16+4q..16+4q+3} -- a permutation of the block's 32 k's that the packed A
// fragments (k_dyn_pack_w) follow, so no data moves.

// [r6] A scalar factor that must never be addressed as the HIGH dword of a register pair.  What gfx950 does (found as the cause of the DIEN flaky
// tiles, docs/open_issue_dien_tiles.md; scripts/ubench/pkfma_opsel_mfma.hip reproduces it in 40 lines): a packed-f32 VALU instruction whose LOW result
// takes the HIGH dword of a VGPR src1 -- `v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1,0]` -- reads that dword as ZERO in lanes 48..63
// (the fma returns src2.lo) while ANOTHER wave of the same SIMD issues 16x16 MFMAs with 128-bit operands back to back (v_mfma_f32_16x16x32_f16 / _bf16,
// v_mfma_i32_16x16x64_i8); about 6 % of the executions next to a saturated matrix pipe, never with an SGPR pair, never for src0 / src2, never with
// op_sel_hi.  hipcc emits the form by itself: two neighbouring un-scale scalars become ONE ds_read_b64, SLP packs `acc.x * un + bias.x, acc.y * un +
// bias.y` into a v_pk_fma_f32, and the operand folder points src1 at the pair's high half.  A value that went through this (empty, non-volatile)
// statement is a 32-bit register of its own; a pair built from it has it as the LOW dword (`op_sel_hi:[1,0,1]`, measured clean).  The guard that
// does not depend on the compiler: scripts/isa/isa_pk_opsel.py over every unit (tests/test_isa_checks_cpu.py) and over the built library
// (__graft_entry__.build()).
__device__ __forceinline__ float lone_scalar(float x) {
    asm("" : "+v"(x));
    return x;
}

// max over the four 16-lane rows of a wave, result in every lane
__device__ __forceinline__ float rows4_max(float v) {
#if defined(SPRK_NO_ASM) || defined(SPRK_NO_ASM_ROWS4)
    { const float t = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(t, __shfl_xor(t, 32)); }
#endif
#ifdef SPRK_ASM_PAD
    { float a = v, b = v;
      asm volatile("s_nop 7\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
      a = fmaxf(a, b); b = a;
      asm volatile("s_nop 7\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
      return fmaxf(a, b); }
#endif
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b);
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// (x0..x3 | y0..y3) * scale -> packed hi halfs and lo halfs: hi = f16(x s), lo = f16(x s - hi), round to nearest even.
// [r4] Rounds 2-3 wrote each half with v_fma_mixlo/hi_f16 (two per value).  scripts/ubench/issue_rates.hip measured what that
// instruction costs a gfx950 SIMD: 8.7 cycles per wave64 issue against 3.1 for v_fma_f32, 4.3 for v_cvt_pk_f16_f32 (new on gfx950:
// TWO values per issue) and 4.6 for v_fma_mix_f32 -- 17.4 cycles per value.  Now: v_pk_mul_f32 + v_cvt_pk_f16_f32 for the hi pair,
// one v_fma_mix_f32 per value for the remainder x s - hi (exact in f32: hi holds the leading 11 bits of x s), v_cvt_pk_f16_f32 for
// the lo pair: five issues per two values, 11.9 cycles per value, THE SAME BITS (both forms round x s and the exact remainder to
// nearest even).  Only the remainder is an asm statement: it reads a register the compiler's own v_cvt_pk wrote and feeds the
// compiler's own v_cvt_pk -- VALU to VALU both ways, nothing for the hazard recognizer to miss (the old form's MFMA-operand
// writes inside asm statements needed a hand-placed s_nop).
typedef _Float16 dyn_h2 __attribute__((ext_vector_type(2)));
typedef float dyn_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dyn_split2(float x0, float x1, float scale, dyn_h2& hi, dyn_h2& lo) {
    const dyn_f2 t = dyn_f2{x0, x1} * scale;
    hi = __builtin_convertvector(t, dyn_h2);
    const unsigned hb = __builtin_bit_cast(unsigned, hi);
#if defined(SPRK_NO_ASM) || defined(SPRK_NO_ASM_SPLIT)
    { lo = __builtin_convertvector(dyn_f2{fmaf(x0, scale, -(float)hi[0]), fmaf(x1, scale, -(float)hi[1])}, dyn_h2); return; }
#endif
#ifdef SPRK_ASM_PAD
    { float r0, r1;
      asm volatile("s_nop 3\n\tv_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\ts_nop 1" : "=v"(r0) : "v"(x0), "v"(scale), "v"(hb));
      asm volatile("s_nop 3\n\tv_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\ts_nop 1" : "=v"(r1) : "v"(x1), "v"(scale), "v"(hb));
      lo = __builtin_convertvector(dyn_f2{r0, r1}, dyn_h2); return; }
#endif
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(scale), "v"(hb));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(scale), "v"(hb));
    lo = __builtin_convertvector(dyn_f2{r0, r1}, dyn_h2);
}
__device__ __forceinline__ void dyn_split8(f32x4 x, f32x4 y, float scale, din_f16x8& hi, din_f16x8& lo) {
    dyn_h2 h[4], l[4];
    dyn_split2(x[0], x[1], scale, h[0], l[0]);
    dyn_split2(x[2], x[3], scale, h[1], l[1]);
    dyn_split2(y[0], y[1], scale, h[2], l[2]);
    dyn_split2(y[2], y[3], scale, h[3], l[3]);
    hi = din_f16x8{h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
    lo = din_f16x8{l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
}

// scale = 2^(14 - exponent(m)) (1 for m == 0), inv = 1 / (scale * w_scale); m >= 0 finite
__device__ __forceinline__ void dyn_scale(float m, float inv_w_scale, float& scale, float& inv) {
    const int e = (int)((__float_as_uint(m) >> 23) & 0xffu);           // biased exponent, 0 for zero / subnormal
    int se = e == 0 ? 0 : 141 - e;                                      // 14 - (e - 127)
    se = se > 100 ? 100 : se;                                           // tiny activations: no need to blow them up to 2^14
    scale = __builtin_amdgcn_ldexpf(1.0f, se);
    inv = __builtin_amdgcn_ldexpf(inv_w_scale, -se);
}

// One-time (finalize) kernel: W^T [N][ld] (K columns) * w_scale -> A fragments of v_mfma_f32_16x16x32_f16 in the
// K-block layout above.  out: for (n block nb, K block b): [hi: 16 rows x 4 q x 8 halfs][lo: same] = 2 x 1 KB.
static __global__ __launch_bounds__(256) void k_dyn_pack_w(const float* __restrict__ W, int ld, int N, int K, float w_scale,
                                                    _Float16* __restrict__ out, int kvalid) {
    const int KB = K / 32;
    const int total = (N / 16) * KB * 2 * 512;                           // halfs
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        // (lane order inside a 1-KB fragment: lane l = 16 q + r at 16 l bytes -- a ds_read_b128 over 64 consecutive 16-byte pieces is
        // conflict-free; rounds 2-4 kept sample-major order (r * 4 + q), whose 64-byte stride inside a 16-lane group is a 2-way bank conflict)
        const int e = i & 7, rr = (i >> 3) & 15, qq = (i >> 7) & 3, hl = (i >> 9) & 1, blk = i >> 10;
        const int nb = blk / KB, b = blk - nb * KB;
        const int n = nb * 16 + rr;
        const int kl = e < 4 ? 4 * qq + e : 16 + 4 * qq + (e - 4);
        const float x = 32 * b + kl < kvalid ? W[(size_t)n * ld + 32 * b + kl] * w_scale : 0.f;   // (columns beyond kvalid: padding of the K block)
        const _Float16 hi = (_Float16)x;
        out[i] = hl ? (_Float16)(x - (float)hi) : hi;
    }
}

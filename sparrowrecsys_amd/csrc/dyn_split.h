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

// max over the four 16-lane rows of a wave, result in every lane
__device__ __forceinline__ float rows4_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b);
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// (x0..x3 | y0..y3) * scale -> packed hi halfs and lo halfs, 2 VALU per value (v_fma_mixlo/hi_f16: f16(x*scale) and
// f16(x*scale - hi) with the f16 source taken straight from the packed register)
__device__ __forceinline__ void dyn_split8(f32x4 x, f32x4 y, float scale, din_f16x8& hi, din_f16x8& lo) {
    // (mixlo writes the low half of its destination and keeps the high half, mixhi the other way round: the even element
    // of a pair goes first as a plain output -- whatever sits in the high half is replaced by the odd element next -- which
    // spares the eight v_mov 0 an initialised read-modify-write operand costs)
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? x[e] : y[e - 4];
        if (e & 1) {
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[e >> 1]) : "v"(v), "v"(scale));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l[e >> 1]) : "v"(v), "v"(scale), "v"(h[e >> 1]));
        } else {
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[e >> 1]) : "v"(v), "v"(scale));
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(l[e >> 1]) : "v"(v), "v"(scale), "v"(h[e >> 1]));
        }
    }
    // HAZARD GUARD.  The eight dwords above were written by VALU instructions INSIDE asm statements, which hipcc's hazard
    // recognizer does not see: it pads nothing between such a write and an MFMA that reads the register as its B operand
    // (a VALU write of a VGPR needs 2 wait states before an MFMA reads it; for a compiler-visible producer hipcc inserts
    // them).  Whether the pad happened to be there depended on how the scheduler interleaved the surrounding code: round 2's
    // k_deepfm_pairs_many scheduled an MFMA straight behind the last v_fma_mixhi and read a half-updated operand (scores
    // off by 7e-6 in one instantiation, exact in its twin) -- the same class as round 1's unexplained k_din_attn multi-batch
    // failure.  One statement that owns all eight dwords and carries the wait states closes it for every user.
    asm volatile("s_nop 1" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    hi = __builtin_bit_cast(din_f16x8, u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(din_f16x8, u32x4{l[0], l[1], l[2], l[3]});
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
__global__ __launch_bounds__(256) void k_dyn_pack_w(const float* __restrict__ W, int ld, int N, int K, float w_scale,
                                                    _Float16* __restrict__ out, int kvalid) {
    const int KB = K / 32;
    const int total = (N / 16) * KB * 2 * 512;                           // halfs
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 7, qq = (i >> 3) & 3, rr = (i >> 5) & 15, hl = (i >> 9) & 1, blk = i >> 10;
        const int nb = blk / KB, b = blk - nb * KB;
        const int n = nb * 16 + rr;
        const int kl = e < 4 ? 4 * qq + e : 16 + 4 * qq + (e - 4);
        const float x = 32 * b + kl < kvalid ? W[(size_t)n * ld + 32 * b + kl] * w_scale : 0.f;   // (columns beyond kvalid: padding of the K block)
        const _Float16 hi = (_Float16)x;
        out[i] = hl ? (_Float16)(x - (float)hi) : hi;
    }
}

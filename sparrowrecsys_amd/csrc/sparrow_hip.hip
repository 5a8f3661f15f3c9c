// sparrow_hip.hip -- MI355X (gfx950 / CDNA4) CTR-ranking forward engine behind include/sparrow_hip.h.
//
// Two hand-written kernels carry the hot path (DESIGN.md has the full data-layout story):
//
//   k_tile_forward  one 256-thread workgroup per tile of 64 samples.  Phase 1 gathers every sparse
//                   slot's embedding row (16-B lanes, 64-B..256-B rows), the first-order weights and
//                   the numeric columns of the tile into an LDS activation buffer; phase 2 runs the
//                   model's small op list over LDS (fp32 MFMA 16x16x4 Dense layers with the weights
//                   as the A operand so every epilogue store is a 16-B ds_write, FM sum-of-squares,
//                   pairwise dots); phase 3 applies the output layer + sigmoid and stores one score
//                   per sample.  Activations never leave the CU; HBM traffic is the algorithmic
//                   minimum (ids + rows + numerics in, scores out).
//   k_din_pool      DIN's activation unit + weighted sum pooling: the T history rows of a few samples
//                   are gathered ONCE into LDS, the [h-c, h, c, h*c] -> Dense(hidden) contraction runs
//                   on fp32 MFMA with the B operand built on the fly from LDS, PReLU/Dense(1)/sigmoid
//                   finish in registers + two cross-lane adds, and the pooled vector is reduced from
//                   the same LDS rows.
//
// Reference constructs each piece replaces are cited in include/sparrow_hip.h.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <thread>
#include <vector>

#include "sparrow_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(SPRK_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device-side plan (slots resolved to pointers); lives in device memory, read through scalar loads
// ---------------------------------------------------------------------------------------------
#define SEG_ROWS_ACC 100   // internal (created by the first-Dense fold at finalize): += folded row into buffer `buf`
struct DevSeg {
    int kind, field, field2, row_stride, count, dst, vocab, buf;   // buf: destination buffer (ROWS_ACC only; others: 0)
    const float* table;
};
struct DevOp {
    int kind, src_buf, src_off, K, dst_buf, dst_off, N, ldw, act, groups, group_stride, acc_init;   // acc_init: Dense accumulates onto dst
    const float* W;
    const float* bias;
    const float* alpha;
};
struct DevTap {
    int buf, off, len, pad_;
    float scale, bias;
    const float* w;
};
struct DevDin {
    int enabled, T, hist_col, cand_col, row_stride, vocab, hidden, pad_;
    float b2;
    float pad2_;
    const float* table;
    const float* W;      // [hidden][4*row_stride]
    const float* bias;   // [hidden]
    const float* alpha;  // [T][hidden]
    const float* w2;     // [hidden]
};
struct DevPlan {
    int F, ND, NA, n_segs, n_ops, n_taps, n_pairs, n_bufs;
    int buf_stride[SPRK_MAX_BUFS];
    int buf_base[SPRK_MAX_BUFS];   // float offset of each buffer inside dynamic LDS
    int ids_base;                  // float offset of the tile's ids block [64][n_idc] inside dynamic LDS
    int n_idc;                     // ids columns the segments read, staged compactly (segs[].field / field2 index THIS list)
    int idc[SPRK_MAX_SEGS];
    int n_acc;                     // ROWS_ACC segments (first-Dense fold); they are the LAST n_acc entries of segs[]
    float head_bias;
    float pad2_;
    int pair_a[SPRK_MAX_PAIRS];
    int pair_b[SPRK_MAX_PAIRS];
    DevSeg segs[SPRK_MAX_SEGS];
    DevOp ops[SPRK_MAX_OPS];
    DevTap taps[SPRK_MAX_TAPS];
    DevDin din;
};

// ---------------------------------------------------------------------------------------------
// FingerprintCat64 chain of tf.feature_column.crossed_column (WideNDeep.py:72-73)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }
__host__ __device__ __forceinline__ uint64_t fingerprint_cat64(uint64_t fp1, uint64_t fp2) {
    const uint64_t kMul = 0xc6a4a7935bd1e995ULL;
    uint64_t result = fp1 ^ kMul;
    result ^= shift_mix(fp2 * kMul) * kMul;
    result *= kMul;
    result = shift_mix(result) * kMul;
    result = shift_mix(result);
    return result;
}
__host__ __device__ __forceinline__ uint64_t cross_bucket(int a, int b, uint64_t buckets) {
    uint64_t h = 0xDECAFCAFFEULL;
    h = fingerprint_cat64(h, (uint64_t)(int64_t)a);
    h = fingerprint_cat64(h, (uint64_t)(int64_t)b);
    return h % buckets;
}

__device__ __forceinline__ float sigmoidf_acc(float z) {
    // 1/(1+exp(-z)); expf overflow -> inf -> 0, no NaN for finite z
    return 1.0f / (1.0f + expf(-z));
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ f32x4 mfma4(f32x4 a, f32x4 b, f32x4 c) {
    // four K-steps of v_mfma_f32_16x16x4_f32; lane (r = lane&15, q = lane>>4) feeds k = 4q+s at step s
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// ---------------------------------------------------------------------------------------------
// Dense layer on one wave: output block rows n = nb*16.., sample sub-tiles mi0..mi0+MI-1
//   D[n][m] = sum_k Wt[n][k] * X[m][k]   (A operand = W^T rows from global/L1, B operand = LDS rows)
// C layout of 16x16x4: lane holds D[row = 4q + j][col = r], j = 0..3 -> four consecutive output
// features of one sample -> one 16-B LDS store.
// ---------------------------------------------------------------------------------------------
template <int MI>
__device__ __forceinline__ void dense_unit(const DevOp& op, const float* __restrict__ src, int sstride,
                                           float* __restrict__ dst, int dstride, int nb, int mi0, int lane) {
    const int r = lane & 15, q = lane >> 4;
    const int K = op.K;
    const float* wrow = op.W + (size_t)(nb * 16 + r) * op.ldw + 4 * q;
    const float* xrow = src + (mi0 * 16 + r) * sstride + op.src_off + 4 * q;
    f32x4 acc[MI];
    const int n = nb * 16 + 4 * q;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        // folded first layer: the gather phase left sum_g (W_g^T E_g[id]) of the folded embedding columns here
        if (op.acc_init) acc[i] = ld4(dst + ((mi0 + i) * 16 + r) * dstride + op.dst_off + n);
    }
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a = (4 * q < K) ? ld4(wrow) : zero;
    for (int k = 0; k < K; k += 16) {
        const bool ok = (k + 4 * q) < K;
        const bool okn = (k + 16 + 4 * q) < K;
        const f32x4 an = okn ? ld4(wrow + k + 16) : zero;   // prefetch next W fragment
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const f32x4 b = ok ? ld4(xrow + i * 16 * sstride + k) : zero;
            acc[i] = mfma4(a, b, acc[i]);
        }
        a = an;
    }
    f32x4 bias = ld4(op.bias + n);
    f32x4 alpha = zero;
    if (op.act == SPRK_ACT_PRELU) alpha = ld4(op.alpha + n);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        f32x4 v = acc[i] + bias;
        if (op.act == SPRK_ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (op.act == SPRK_ACT_PRELU) {
            v.x = fmaxf(v.x, 0.f) + alpha.x * fminf(v.x, 0.f);
            v.y = fmaxf(v.y, 0.f) + alpha.y * fminf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f) + alpha.z * fminf(v.z, 0.f);
            v.w = fmaxf(v.w, 0.f) + alpha.w * fminf(v.w, 0.f);
        }
        st4(dst + ((mi0 + i) * 16 + r) * dstride + op.dst_off + n, v);
    }
}

__device__ __forceinline__ void run_dense(const DevOp& op, const float* src, int sstride, float* dst,
                                          int dstride, int wave, int lane) {
    const int NB = op.N >> 4;
    if (NB >= 3) {
        for (int nb = wave; nb < NB; nb += 4) dense_unit<4>(op, src, sstride, dst, dstride, nb, 0, lane);
    } else if (NB == 2) {
        dense_unit<2>(op, src, sstride, dst, dstride, wave >> 1, (wave & 1) * 2, lane);
    } else {
        dense_unit<1>(op, src, sstride, dst, dstride, 0, wave, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// k_tile_forward
// ---------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) float smem[];

__global__ __launch_bounds__(256) void k_tile_forward(const DevPlan* __restrict__ P,
                                                      const int* __restrict__ ids,
                                                      const float* __restrict__ dense,
                                                      const float* __restrict__ aux,
                                                      float* __restrict__ out, int B,
                                                      int* __restrict__ err) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = P->F, ND = P->ND, NA = P->NA;
    float* buf0 = smem + P->buf_base[0];
    const int stride0 = P->buf_stride[0];
    const int ntiles = (B + SPRK_TILE_M - 1) / SPRK_TILE_M;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * SPRK_TILE_M;
        const int mvalid = min(SPRK_TILE_M, B - m0);

        // ---------------- phase 1: gather the tile into LDS ----------------
        // The tile's ids block first (one coalesced pass, every later id read is an LDS read), then the row
        // gathers with several independent loads in flight per thread: at one tile per workgroup the kernel's
        // duration is this phase's chain of memory latencies, so what counts is how few round trips it takes.
        int* ids_s = reinterpret_cast<int*>(smem + P->ids_base);
        const int FC = P->n_idc;                                 // only the columns some segment reads (DIN's history ids stay out)
        {
            const int total = mvalid * FC;
            const int* src = ids + (size_t)m0 * F;
#pragma unroll 4
            for (int i = tid; i < total; i += 256) {
                const int m = i / FC, j = i - m * FC;
                ids_s[i] = src[m * F + P->idc[j]];
            }
        }
        __syncthreads();
        const int n_acc = P->n_acc;
        const int n_segs = P->n_segs - n_acc;
        if (n_acc > 0) {
            // folded embedding columns (first-Dense fold): dst[m][:] = sum over the folded columns g of F_g[id_g][:],
            // one thread per (sample, 16-byte piece), fixed summation order, all of a piece's loads in flight together
            const DevSeg* ag = &P->segs[n_segs];
            const int nvec = ag[0].count;
            const int total = SPRK_TILE_M * nvec;
            float* bufd = smem + P->buf_base[ag[0].buf];
            const int strided = P->buf_stride[ag[0].buf];
            for (int base = tid; base < total; base += 1024) {        // 4 pieces per thread per trip
                f32x4 acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int g0 = 0; g0 < n_acc; g0 += 4) {               // up to 4 pieces x 4 columns = 16 loads in flight
                    f32x4 v[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = base + u * 256;
                        const int m = idx / nvec;
                        const int c = idx - m * nvec;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            v[u][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                            if (g0 + g < n_acc && idx < total && m < mvalid) {
                                const DevSeg& a = ag[g0 + g];
                                const int id = ids_s[m * FC + a.field];
                                if ((unsigned)id < (unsigned)a.vocab) v[u][g] = ld4(a.table + (size_t)id * a.row_stride + 4 * c);
                                else if (id != -1) atomicOr(err, 1);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[u] += v[u][g];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = base + u * 256;
                    if (idx < total) {
                        const int m = idx / nvec;
                        const int c = idx - m * nvec;
                        st4(bufd + m * strided + ag[0].dst + 4 * c, acc[u]);
                    }
                }
            }
        }
        for (int s = 0; s < n_segs; ++s) {
            const DevSeg& sg = P->segs[s];
            const int kind = sg.kind;
            if (kind == SPRK_SEG_ROWS || kind == SPRK_SEG_CROSS_ROWS) {
                const int nvec = sg.count;
                const int total = SPRK_TILE_M * nvec;
                for (int base = tid; base < total; base += 1024) {
                    f32x4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = base + u * 256;
                        const int m = idx / nvec;
                        const int c = idx - m * nvec;
                        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (idx < total && m < mvalid) {
                            const int* idrow = ids_s + m * FC;
                            long long row;
                            if (kind == SPRK_SEG_ROWS) {
                                const int id = idrow[sg.field];
                                row = id;
                                if ((unsigned)id >= (unsigned)sg.vocab) {
                                    row = -1;
                                    if (id != -1) atomicOr(err, 1);
                                }
                            } else {
                                const int a = idrow[sg.field], b = idrow[sg.field2];
                                row = (long long)cross_bucket(a, b, (uint64_t)sg.vocab);
                            }
                            if (row >= 0) v[u] = ld4(sg.table + (size_t)row * sg.row_stride + 4 * c);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int idx = base + u * 256;
                        if (idx < total) {
                            const int m = idx / nvec;
                            const int c = idx - m * nvec;
                            st4(buf0 + m * stride0 + sg.dst + 4 * c, v[u]);
                        }
                    }
                }
            } else if (kind == SPRK_SEG_SCALAR || kind == SPRK_SEG_CROSS_SCALAR) {
                if (tid < SPRK_TILE_M) {
                    const int m = tid;
                    float v = 0.f;
                    if (m < mvalid) {
                        const int* idrow = ids_s + m * FC;
                        if (kind == SPRK_SEG_SCALAR) {
                            const int id = idrow[sg.field];
                            if ((unsigned)id < (unsigned)sg.vocab) v = sg.table[id];
                            else if (id != -1) atomicOr(err, 1);
                        } else {
                            const int a = idrow[sg.field], b = idrow[sg.field2];
                            v = sg.table[cross_bucket(a, b, (uint64_t)sg.vocab)];
                        }
                    }
                    buf0[m * stride0 + sg.dst] = v;
                }
            } else if (kind == SPRK_SEG_DENSE || kind == SPRK_SEG_AUX) {
                const int cnt = sg.count;
                const int total = SPRK_TILE_M * cnt;
                const float* base = (kind == SPRK_SEG_DENSE) ? dense : aux;
                const int rw = (kind == SPRK_SEG_DENSE) ? ND : NA;
                // (eight independent loads in flight per thread: this copy is pure latency otherwise)
                for (int b8 = tid; b8 < total; b8 += 2048) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = b8 + u * 256;
                        const int m = idx / cnt;
                        const int j = idx - m * cnt;
                        v[u] = 0.f;
                        if (idx < total && m < mvalid) v[u] = base[(size_t)(m0 + m) * rw + sg.field + j];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = b8 + u * 256;
                        if (idx < total) {
                            const int m = idx / cnt;
                            const int j = idx - m * cnt;
                            buf0[m * stride0 + sg.dst + j] = v[u];
                        }
                    }
                }
            } else {  // SPRK_SEG_ZERO
                const int cnt = sg.count;
                const int total = SPRK_TILE_M * cnt;
                for (int idx = tid; idx < total; idx += 256) {
                    const int m = idx / cnt;
                    const int j = idx - m * cnt;
                    buf0[m * stride0 + sg.dst + j] = 0.f;
                }
            }
        }
        __syncthreads();

        // ---------------- phase 2: op list over LDS ----------------
        const int n_ops = P->n_ops;
        for (int o = 0; o < n_ops; ++o) {
            const DevOp& op = P->ops[o];
            const float* src = smem + P->buf_base[op.src_buf];
            const int sstride = P->buf_stride[op.src_buf];
            float* dst = smem + P->buf_base[op.dst_buf];
            const int dstride = P->buf_stride[op.dst_buf];
            if (op.kind == SPRK_OP_DENSE) {
                run_dense(op, src, sstride, dst, dstride, wave, lane);
            } else if (op.kind == SPRK_OP_FM_SUMSQ) {
                const int K = op.K;
                const int total = SPRK_TILE_M * K;
                for (int idx = tid; idx < total; idx += 256) {
                    const int m = idx / K;
                    const int j = idx - m * K;
                    const float* p = src + m * sstride + op.src_off + j;
                    float s = 0.f, sq = 0.f;
                    for (int g = 0; g < op.groups; ++g) {
                        const float v = p[g * op.group_stride];
                        s += v;
                        sq += v * v;
                    }
                    dst[m * dstride + op.dst_off + j] = s * s - sq;
                }
            } else {  // SPRK_OP_PAIR_DOT
                const int np = P->n_pairs;
                const int total = SPRK_TILE_M * np;
                for (int idx = tid; idx < total; idx += 256) {
                    const int m = idx / np;
                    const int p = idx - m * np;
                    const float* xa = src + m * sstride + P->pair_a[p];
                    const float* xb = src + m * sstride + P->pair_b[p];
                    float s = 0.f;
                    for (int d = 0; d < op.K; d += 4) {
                        const f32x4 va = ld4(xa + d), vb = ld4(xb + d);
                        s += va.x * vb.x; s += va.y * vb.y; s += va.z * vb.z; s += va.w * vb.w;
                    }
                    dst[m * dstride + op.dst_off + p] = s;
                }
            }
            __syncthreads();
        }

        // ---------------- phase 3: output layer + sigmoid (4 lanes per sample) ----------------
        {
            const int m = tid >> 2, part = tid & 3;
            float z = 0.f;
            const int n_taps = P->n_taps;
            for (int t = 0; t < n_taps; ++t) {
                const DevTap& tp = P->taps[t];
                const float* x = smem + P->buf_base[tp.buf] + m * P->buf_stride[tp.buf] + tp.off;
                float s = 0.f;
                if (tp.w) {
                    for (int j = part; j < tp.len; j += 4) s += tp.w[j] * x[j];
                } else {
                    for (int j = part; j < tp.len; j += 4) s += x[j];
                }
                if (part == 0) s += tp.bias;
                z += tp.scale * s;
            }
            z += __shfl_xor(z, 1);
            z += __shfl_xor(z, 2);
            if (part == 0 && m < mvalid) out[m0 + m] = sigmoidf_acc(z + P->head_bias);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// k_din_pool: DIN activation unit + weighted sum pooling (DIN.py:132-158)
//   LDS: Hs[rows][hs] history rows, Cs[MS][hs] candidate rows, Ws[rows] attention weights
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_din_pool(const DevPlan* __restrict__ P, const int* __restrict__ ids,
                                                  float* __restrict__ pooled, float* __restrict__ att,
                                                  int B, int MS, int* __restrict__ err) {
    const DevDin& dn = P->din;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = dn.T, F = P->F, Dp = dn.row_stride, nvec = Dp >> 2, hs = Dp + 4;
    const int hidden = dn.hidden;
    const int rows_cap = MS * T;
    float* Hs = smem;
    float* Cs = Hs + (size_t)rows_cap * hs;
    float* Ws = Cs + (size_t)MS * hs;
    const int r = lane & 15, q = lane >> 4;
    const int K = 4 * Dp;
    const int nchunks = (B + MS - 1) / MS;
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int s0 = chunk * MS;
        const int ms = min(MS, B - s0);
        const int nrows = ms * T;

        // ---- phase 1: gather history + candidate rows into LDS (each row read from HBM once) ----
        for (int idx = tid; idx < nrows * nvec; idx += 256) {
            const int row = idx / nvec;
            const int c = idx - row * nvec;
            const int m = row / T;
            const int t = row - m * T;
            const int id = ids[(size_t)(s0 + m) * F + dn.hist_col + t];
            f32x4 v = zero;
            if ((unsigned)id < (unsigned)dn.vocab) v = ld4(dn.table + (size_t)id * Dp + 4 * c);
            else atomicOr(err, 1);
            st4(Hs + row * hs + 4 * c, v);
        }
        for (int idx = tid; idx < ms * nvec; idx += 256) {
            const int m = idx / nvec;
            const int c = idx - m * nvec;
            const int id = ids[(size_t)(s0 + m) * F + dn.cand_col];
            f32x4 v = zero;
            if ((unsigned)id < (unsigned)dn.vocab) v = ld4(dn.table + (size_t)id * Dp + 4 * c);
            else atomicOr(err, 1);
            st4(Cs + m * hs + 4 * c, v);
        }
        __syncthreads();

        // ---- phase 2: attention logits on fp32 MFMA, 16 (sample, slot) rows per step ----
        const int ngroups = (nrows + 15) >> 4;
        for (int g = wave; g < ngroups; g += 4) {
            const int row = g * 16 + r;
            const bool valid = row < nrows;
            const int rowc = valid ? row : 0;
            const int m = rowc / T;
            const int t = rowc - m * T;
            const float* hrow = Hs + rowc * hs;
            const float* crow = Cs + m * hs;
            float sum = 0.f;
            for (int nb0 = 0; nb0 < (hidden >> 4); nb0 += 2) {
                const bool two = (nb0 + 1) < (hidden >> 4);
                f32x4 acc0 = zero, acc1 = zero;
                const float* w0 = dn.W + (size_t)(nb0 * 16 + r) * K + 4 * q;
                const float* w1 = w0 + (size_t)16 * K;
                for (int k = 0; k < K; k += 16) {
                    const int kk = k + 4 * q;
                    const bool ok = kk < K;
                    f32x4 b = zero, a0 = zero, a1 = zero;
                    if (ok) {
                        const int blk = kk / Dp;
                        const int d = kk - blk * Dp;
                        const f32x4 hv = ld4(hrow + d), cv = ld4(crow + d);
                        b = (blk == 0) ? (hv - cv) : (blk == 1) ? hv : (blk == 2) ? cv : (hv * cv);
                        if (!valid) b = zero;
                        a0 = ld4(w0 + k);
                        if (two) a1 = ld4(w1 + k);
                    }
                    acc0 = mfma4(a0, b, acc0);
                    if (two) acc1 = mfma4(a1, b, acc1);
                }
                // epilogue: + bias, PReLU(alpha[t][n]), dot with att1 kernel
                {
                    const int n = nb0 * 16 + 4 * q;
                    const f32x4 bias = ld4(dn.bias + n), al = ld4(dn.alpha + (size_t)t * hidden + n), w2 = ld4(dn.w2 + n);
                    f32x4 u = acc0 + bias;
                    sum += w2.x * (fmaxf(u.x, 0.f) + al.x * fminf(u.x, 0.f));
                    sum += w2.y * (fmaxf(u.y, 0.f) + al.y * fminf(u.y, 0.f));
                    sum += w2.z * (fmaxf(u.z, 0.f) + al.z * fminf(u.z, 0.f));
                    sum += w2.w * (fmaxf(u.w, 0.f) + al.w * fminf(u.w, 0.f));
                }
                if (two) {
                    const int n = (nb0 + 1) * 16 + 4 * q;
                    const f32x4 bias = ld4(dn.bias + n), al = ld4(dn.alpha + (size_t)t * hidden + n), w2 = ld4(dn.w2 + n);
                    f32x4 u = acc1 + bias;
                    sum += w2.x * (fmaxf(u.x, 0.f) + al.x * fminf(u.x, 0.f));
                    sum += w2.y * (fmaxf(u.y, 0.f) + al.y * fminf(u.y, 0.f));
                    sum += w2.z * (fmaxf(u.z, 0.f) + al.z * fminf(u.z, 0.f));
                    sum += w2.w * (fmaxf(u.w, 0.f) + al.w * fminf(u.w, 0.f));
                }
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            if (q == 0 && valid) {
                const float wgt = sigmoidf_acc(sum + dn.b2);
                Ws[row] = wgt;
                if (att) att[(size_t)(s0 + m) * T + t] = wgt;
            }
        }
        __syncthreads();

        // ---- phase 3: pooled[m] = sum_t w[m,t] * h[m,t,:]  (t ascending, as the reference sums) ----
        for (int idx = tid; idx < ms * nvec; idx += 256) {
            const int m = idx / nvec;
            const int c = idx - m * nvec;
            f32x4 acc = zero;
            const float* hp = Hs + (size_t)m * T * hs + 4 * c;
            const float* wp = Ws + m * T;
            for (int t = 0; t < T; ++t) {
                const float w = wp[t];
                acc += w * ld4(hp + t * hs);
            }
            st4(pooled + (size_t)(s0 + m) * Dp + 4 * c, acc);
        }
        __syncthreads();
    }
}

// One-time (finalize) kernel of the first-Dense fold: F[v][n] = sum_j Wt[n][col0 + j] * table[v][j]
// (Wt = the layer's W^T [N][ldw], col0 = the embedding column's offset inside the layer's input slice).
__global__ __launch_bounds__(256) void k_fold_dense_rows(const float* __restrict__ table, long long vocab, int row_stride,
                                                         int width, const float* __restrict__ Wt, int ldw, int col0,
                                                         int N, float* __restrict__ F) {
    const long long total = vocab * N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / N;
        const int n = (int)(i - v * N);
        const float* e = table + v * row_stride;
        const float* w = Wt + (size_t)n * ldw + col0;
        float acc = 0.f;
        for (int j = 0; j < width; ++j) acc = fmaf(w[j], e[j], acc);
        F[i] = acc;
    }
}
// copy of a W^T with the columns [c0, c1) zeroed (folded columns that stay inside the layer's K range)
__global__ __launch_bounds__(256) void k_zero_columns(float* __restrict__ Wt, int N, int ldw, int c0, int c1) {
    const int w = c1 - c0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * w; i += gridDim.x * 256) Wt[(size_t)(i / w) * ldw + c0 + i % w] = 0.f;
}

#include "k_chain_v2.h"
#include "k_chain_v2j.h"
#include "k_chain_v2j1.h"
#include "k_rows_chain.h"
#include "k_din_attn.h"
#include "dyn_split.h"
#include "k_din_tail.h"
#include "k_chain_v1.h"
#include "k_mlp_chain.h"
#include "k_mlp_rows.h"
#include "k_emb_rank.h"
#include "k_dien_seq.h"
#include "k_peer_gather.h"
#include "k_csv_pack.h"

// ---------------------------------------------------------------------------------------------
// stand-alone operators
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embedding_gather(const float* __restrict__ table, int V, int nvec,
                                                          int row_stride, const int* __restrict__ ids, int B,
                                                          float* __restrict__ out) {
    const long long total = (long long)B * nvec;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int b = (int)(idx / nvec);
        const int c = (int)(idx - (long long)b * nvec);
        const int id = ids[b];
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((unsigned)id < (unsigned)V) v = ld4(table + (size_t)id * row_stride + 4 * c);
        st4(out + (size_t)b * nvec * 4 + 4 * c, v);
    }
}

__global__ __launch_bounds__(256) void k_cross_hash(const int* __restrict__ a, const int* __restrict__ b, int B,
                                                    unsigned long long buckets, long long* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256)
        out[i] = (long long)cross_bucket(a[i], b[i], buckets);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct sprk_engine {
    sprk_plan plan;
    std::vector<void*> slot_ptr;
    std::vector<size_t> slot_bytes;
    DevPlan* dev_plan = nullptr;
    int* dev_err = nullptr;
    bool finalized = false;
    int device = 0;
    int num_cus = 256;
    int buf_stride[SPRK_MAX_BUFS] = {0, 0, 0};
    int buf_base[SPRK_MAX_BUFS] = {0, 0, 0};
    size_t tile_lds_bytes = 0;
    int ids_base = 0;              // float offset of the tile's ids block inside the tile kernel's LDS
    std::vector<int> idc;          // ids columns read by the gather segments (compact staging order)
    int tile_grid_cap = 0;
    std::vector<void*> fold_bufs;  // first-Dense fold: folded tables + the W^T copy (device)
    // sprk_forward_many fan-out: independent batches alternate over helper streams (hardware queues), so that one
    // kernel's dispatch / drain (3.3 us even for an empty kernel in a dependent launch chain) overlaps its neighbours
    int many_streams = 0;
    hipStream_t many_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t many_fork = nullptr, many_join[4] = {nullptr, nullptr, nullptr, nullptr};
    // register-chained pairwise-dot DeepFM (k_deepfm_pairs); -1 = the tile interpreter
    int v1_variant = -1;
    bool v1_one = false;                  // one-batch launches use k_deepfm_pairs1 (one task per wave, four waves per SIMD)
    V1Run v1_run;
    std::vector<void*> v1_bufs;
    // register-chained DenseFeatures -> Dense -> Dense -> Dense(1) graphs (k_mlp_chain); -1 = the tile interpreter
    int mlp_variant = -1;
    MlpChainRun mlp_run;
    float* mlp_image = nullptr;
    // ... with every embedding column folded through the first layer, genre tables in LDS (k_mlp_rows); -1 = not used
    int mlp_rows_nbig = -1;
    MlpRowsRun mlp_rows_run;
    float* mlp_rows_image = nullptr;
    float* mlp_rows_small = nullptr;
    size_t mlp_rows_lds = 0;
    std::vector<void*> mlp_rows_bufs;
    // register-chained DIN tail (k_din_tail); -1 = the tile interpreter runs the tail
    int din_tail_variant = -1;
    DinTailRun din_tail_run;
    float* din_tail_image = nullptr;
    // DIN launch geometry
    int din_ms = 0;
    size_t din_lds_bytes = 0;
    int din_grid_cap = 0;
    // wave-per-sample attention kernel (k_din_attn); -1 = the generic k_din_pool
    int din_variant = -1;
    DienRun dien_run{};
    DinRun din_run;
    float* din_w12 = nullptr;      // (W1+W2)^T, W4^T fragments and the per-id c-term table (device)
    float* din_w4 = nullptr;
    float* din_vc = nullptr;
    float* din_tsplit = nullptr;   // HALF: the movie table pre-split into f16 hi/lo pairs
    size_t din_attn_lds = 0;
    int din_attn_grid_cap = 0;
    int din_wpb = 4;               // waves per k_din_attn workgroup
    bool din_attn_many = true;     // forward_many: one attention launch per group of batches (SPRK_DIN_ATTN_MB=0: per batch)
    // register-chained fast path (k_deepfm_v2_chain); -1 = use the tile interpreter
    int v2_variant = -1;
    V2Args v2;
    V2Run v2run;
    size_t v2_lds_bytes = 0;
    int v2_grid_cap = 0;
    float* v2_image = nullptr;     // pre-packed LDS weight image (device)
    float* v2_fo_all = nullptr;    // concatenated first-order weight blocks (device)
    float* v2_folded = nullptr;    // projected tables of all fields, back to back (device)
    size_t v2_fo_floats = 0;
    // ... with the small-vocabulary fields folded into one joint table (k_deepfm_v2_joint); -1 = not used
    int v2j_variant = -1;
    float* v2j1_image = nullptr;          // k_deepfm_v2_joint1 (one task per wave): its LDS image; NULL = shape not available
    size_t v2j1_lds_bytes = 0;
    int many_batches = 1;                 // sprk_forward_many: batches scored per launch (sprk_set_many_batches)
    // "one row per id" chain (k_rows_chain): DeepFM_v2 with projections wider than 16 (the reference's Dense(64)) and NeuralCF
    int rows_variant = -1;
    bool rows_from_v2 = false;            // set by match_v2_chain: h->v2 holds the parsed DeepFM_v2 plan, tables still to build
    int rows_g_emb = 0;
    RowsRun rows_run;
    float* rows_tab = nullptr;            // big fields' rows {P | Q}
    float* rows_scal = nullptr;           // big fields' per-id scalars
    float* rows_small = nullptr;          // small fields' LDS rows (device image)
    float* rows_image = nullptr;          // weight image
    size_t rows_lds_bytes = 0;
    int n_acc_folded = 0;                 // embedding columns folded into the first Dense layer (fold_first_dense)
    size_t derived_bytes = 0;             // device memory of tables DERIVED at finalize (folded rows, split halfs, per-id terms)
    int v2_xflags = 0;                    // SPRK_V2_XFLAGS experiment switches, read ONCE at finalize (never on the launch path)
    bool v2_xflags_set = false;
    V2JRun v2j_run;
    float* v2j_tab = nullptr;      // small fields' LDS rows (device image)
    size_t v2j_lds_bytes = 0;
    float* v2j_big = nullptr;      // HALF: split-half rows of the big fields (device)
};

namespace {

int lds_stride(int width) {
    // floats per sample row in LDS: multiple of 4 with (stride/4) odd, so the 16-B slots of the 16
    // rows a ds_read_b128 lane group touches spread over the 256-B bank row
    int s = (width + 3) & ~3;
    if (((s >> 2) & 1) == 0) s += 4;
    return s;
}

int check_slot(const sprk_plan& p, int slot, bool allow_none, const char* what) {
    if (slot == -1 && allow_none) return 0;
    if (slot < 0 || slot >= p.n_slots) return fail(SPRK_EINVAL, "%s: slot %d outside [0,%d)", what, slot, p.n_slots);
    return 0;
}

int validate_plan(const sprk_plan& p) {
    if (p.abi_version != SPRK_ABI_VERSION) return fail(SPRK_EINVAL, "plan abi_version %d != %d", p.abi_version, SPRK_ABI_VERSION);
    if (p.n_id_cols < 0 || p.n_dense < 0 || p.n_aux < 0) return fail(SPRK_EINVAL, "negative column count");
    if (p.n_bufs < 1 || p.n_bufs > SPRK_MAX_BUFS) return fail(SPRK_EINVAL, "n_bufs %d outside [1,%d]", p.n_bufs, SPRK_MAX_BUFS);
    if (p.n_segs < 0 || p.n_segs > SPRK_MAX_SEGS) return fail(SPRK_EINVAL, "n_segs %d too large", p.n_segs);
    if (p.n_ops < 0 || p.n_ops > SPRK_MAX_OPS) return fail(SPRK_EINVAL, "n_ops %d too large", p.n_ops);
    if (p.n_taps < 0 || p.n_taps > SPRK_MAX_TAPS) return fail(SPRK_EINVAL, "n_taps %d too large", p.n_taps);
    if (p.n_pairs < 0 || p.n_pairs > SPRK_MAX_PAIRS) return fail(SPRK_EINVAL, "n_pairs %d too large", p.n_pairs);
    if (p.n_slots < 0 || p.n_slots > 4096) return fail(SPRK_EINVAL, "n_slots %d out of range", p.n_slots);
    for (int b = 0; b < p.n_bufs; ++b)
        if (p.buf_width[b] <= 0 || (p.buf_width[b] & 3)) return fail(SPRK_EINVAL, "buf_width[%d]=%d must be a positive multiple of 4", b, p.buf_width[b]);
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& s = p.segs[i];
        int width = 0;
        switch (s.kind) {
            case SPRK_SEG_ROWS:
            case SPRK_SEG_CROSS_ROWS:
                if (check_slot(p, s.slot, false, "segment table")) return SPRK_EINVAL;
                if (s.row_stride <= 0 || (s.row_stride & 3) || s.count <= 0 || 4 * s.count > s.row_stride || (s.dst & 3))
                    return fail(SPRK_EINVAL, "segment %d: bad row geometry (row_stride %d, count %d, dst %d)", i, s.row_stride, s.count, s.dst);
                width = 4 * s.count;
                break;
            case SPRK_SEG_SCALAR:
            case SPRK_SEG_CROSS_SCALAR:
                if (check_slot(p, s.slot, false, "segment table")) return SPRK_EINVAL;
                width = 1;
                break;
            case SPRK_SEG_DENSE:
                if (s.count <= 0 || s.field < 0 || s.field + s.count > p.n_dense) return fail(SPRK_EINVAL, "segment %d: dense columns out of range", i);
                width = s.count;
                break;
            case SPRK_SEG_AUX:
                if (s.count <= 0 || s.field < 0 || s.field + s.count > p.n_aux) return fail(SPRK_EINVAL, "segment %d: aux columns out of range", i);
                width = s.count;
                break;
            case SPRK_SEG_ZERO:
                if (s.count <= 0) return fail(SPRK_EINVAL, "segment %d: empty zero fill", i);
                width = s.count;
                break;
            default:
                return fail(SPRK_EINVAL, "segment %d: unknown kind %d", i, s.kind);
        }
        if (s.kind == SPRK_SEG_ROWS || s.kind == SPRK_SEG_SCALAR) {
            if (s.field < 0 || s.field >= p.n_id_cols) return fail(SPRK_EINVAL, "segment %d: ids column %d out of range", i, s.field);
            if (s.vocab <= 0) return fail(SPRK_EINVAL, "segment %d: vocab must be positive", i);
        }
        if (s.kind == SPRK_SEG_CROSS_ROWS || s.kind == SPRK_SEG_CROSS_SCALAR) {
            if (s.field < 0 || s.field >= p.n_id_cols || s.field2 < 0 || s.field2 >= p.n_id_cols)
                return fail(SPRK_EINVAL, "segment %d: cross ids columns out of range", i);
            if (s.vocab <= 0) return fail(SPRK_EINVAL, "segment %d: bucket count must be positive", i);
        }
        if (s.dst < 0 || s.dst + width > p.buf_width[0]) return fail(SPRK_EINVAL, "segment %d: writes [%d,%d) outside buffer 0 (width %d)", i, s.dst, s.dst + width, p.buf_width[0]);
    }
    for (int i = 0; i < p.n_ops; ++i) {
        const sprk_op& o = p.ops[i];
        if (o.src_buf < 0 || o.src_buf >= p.n_bufs || o.dst_buf < 0 || o.dst_buf >= p.n_bufs) return fail(SPRK_EINVAL, "op %d: buffer index out of range", i);
        if (o.kind == SPRK_OP_DENSE) {
            if (o.src_buf == o.dst_buf) return fail(SPRK_EINVAL, "op %d: Dense must not run in place", i);
            if (o.K <= 0 || (o.K & 3) || o.N <= 0 || (o.N & 15) || o.ldw < o.K || (o.ldw & 3)) return fail(SPRK_EINVAL, "op %d: bad Dense geometry K=%d N=%d ldw=%d", i, o.K, o.N, o.ldw);
            if ((o.src_off & 3) || (o.dst_off & 3)) return fail(SPRK_EINVAL, "op %d: offsets must be multiples of 4", i);
            if (o.src_off < 0 || o.src_off + o.K > p.buf_width[o.src_buf] || o.dst_off < 0 || o.dst_off + o.N > p.buf_width[o.dst_buf]) return fail(SPRK_EINVAL, "op %d: Dense slice outside its buffer", i);
            if (check_slot(p, o.w_slot, false, "Dense kernel") || check_slot(p, o.b_slot, false, "Dense bias")) return SPRK_EINVAL;
            if (o.act == SPRK_ACT_PRELU && check_slot(p, o.alpha_slot, false, "PReLU alpha")) return SPRK_EINVAL;
            if (o.act < 0 || o.act > SPRK_ACT_PRELU) return fail(SPRK_EINVAL, "op %d: unknown activation", i);
        } else if (o.kind == SPRK_OP_FM_SUMSQ) {
            if (o.K <= 0 || o.groups <= 0 || o.group_stride < o.K) return fail(SPRK_EINVAL, "op %d: bad FM geometry", i);
            if (o.src_off < 0 || o.src_off + (o.groups - 1) * o.group_stride + o.K > p.buf_width[o.src_buf] || o.dst_off < 0 || o.dst_off + o.K > p.buf_width[o.dst_buf]) return fail(SPRK_EINVAL, "op %d: FM slice outside its buffer", i);
            if (o.src_buf == o.dst_buf && o.dst_off < o.src_off + (o.groups - 1) * o.group_stride + o.K && o.dst_off + o.K > o.src_off) return fail(SPRK_EINVAL, "op %d: FM output overlaps its input", i);
        } else if (o.kind == SPRK_OP_PAIR_DOT) {
            if (o.K <= 0 || (o.K & 3) || p.n_pairs <= 0) return fail(SPRK_EINVAL, "op %d: bad pair-dot geometry", i);
            for (int j = 0; j < p.n_pairs; ++j)
                if (p.pair_a[j] < 0 || (p.pair_a[j] & 3) || p.pair_a[j] + o.K > p.buf_width[o.src_buf] || p.pair_b[j] < 0 || (p.pair_b[j] & 3) || p.pair_b[j] + o.K > p.buf_width[o.src_buf]) return fail(SPRK_EINVAL, "op %d: pair %d outside its buffer", i, j);
            if (o.dst_off < 0 || o.dst_off + p.n_pairs > p.buf_width[o.dst_buf]) return fail(SPRK_EINVAL, "op %d: pair-dot output outside its buffer", i);
        } else {
            return fail(SPRK_EINVAL, "op %d: unknown kind %d", i, o.kind);
        }
    }
    for (int i = 0; i < p.n_taps; ++i) {
        const sprk_tap& t = p.taps[i];
        if (t.buf < 0 || t.buf >= p.n_bufs || t.off < 0 || t.len <= 0 || t.off + t.len > p.buf_width[t.buf]) return fail(SPRK_EINVAL, "tap %d outside its buffer", i);
        if (check_slot(p, t.w_slot, true, "tap weights")) return SPRK_EINVAL;
    }
    if (p.din.enabled == 2) {
        const sprk_din& d = p.din;
        if (d.T <= 0 || d.T > 256) return fail(SPRK_EINVAL, "DIEN history length %d outside [1,256]", d.T);
        if (d.hist_col < 0 || d.hist_col + d.T > p.n_id_cols || d.cand_col < 0 || d.cand_col >= p.n_id_cols) return fail(SPRK_EINVAL, "DIEN ids columns out of range");
        if (d.row_stride <= 0 || (d.row_stride & 3) || d.vocab <= 0) return fail(SPRK_EINVAL, "DIEN bad table geometry");
        if ((d.emb_dim != 10 && d.emb_dim != 16) || d.emb_dim > d.row_stride) return fail(SPRK_EINVAL, "DIEN emb_dim %d: instantiated for 10 and 16", d.emb_dim);
        if (d.hidden != 32) return fail(SPRK_EINVAL, "DIEN attention width must be 32 (DIEN.py:184)");
        if (p.n_aux != d.row_stride) return fail(SPRK_EINVAL, "DIEN: n_aux (%d) must equal row_stride (%d)", p.n_aux, d.row_stride);
        if (check_slot(p, d.table_slot, false, "DIEN table") || check_slot(p, d.seq_slot, false, "DIEN sequence weights")) return SPRK_EINVAL;
    } else if (p.din.enabled) {
        const sprk_din& d = p.din;
        if (p.din.enabled != 1) return fail(SPRK_EINVAL, "din.enabled must be 0, 1 (DIN) or 2 (DIEN)");
        if (d.T <= 0 || d.T > 256) return fail(SPRK_EINVAL, "DIN history length %d outside [1,256]", d.T);
        if (d.hist_col < 0 || d.hist_col + d.T > p.n_id_cols || d.cand_col < 0 || d.cand_col >= p.n_id_cols) return fail(SPRK_EINVAL, "DIN ids columns out of range");
        if (d.row_stride <= 0 || (d.row_stride & 3) || d.vocab <= 0) return fail(SPRK_EINVAL, "DIN bad table geometry");
        if (d.hidden <= 0 || (d.hidden & 15)) return fail(SPRK_EINVAL, "DIN hidden width must be a multiple of 16");
        if (p.n_aux != d.row_stride) return fail(SPRK_EINVAL, "DIN: n_aux (%d) must equal row_stride (%d)", p.n_aux, d.row_stride);
        if (check_slot(p, d.table_slot, false, "DIN table") || check_slot(p, d.w_slot, false, "DIN att0 kernel") || check_slot(p, d.b_slot, false, "DIN att0 bias") || check_slot(p, d.alpha_slot, false, "DIN alpha") || check_slot(p, d.w2_slot, false, "DIN att1 kernel")) return SPRK_EINVAL;
    } else if (p.n_aux != 0) {
        return fail(SPRK_EINVAL, "n_aux %d without a DIN stage", p.n_aux);
    }
    return SPRK_OK;
}


// ---- fast-path dispatch table for k_deepfm_v2_chain<G_EMB, DV, KPC, H0C, H1C, WAVES, FOLD, TRACE, REG> ----
constexpr int V2_WAVES = 8;
typedef void (*V2LaunchFn)(const V2Run&, const int*, const float*, float*, int, int*, const float*, int, size_t, hipStream_t);
struct V2Variant {
    int g_emb, dv, kpc, h0c, h1c;     // dv is ignored for FOLD variants (they gather KP-wide projected rows)
    bool fold;
    bool reg;                         // register-resident weights, 2 waves per SIMD (one 8-wave workgroup per CU)
    const void* fn;
    const void* fn_trace;             // TRACE instantiation (diagnostics), or NULL
    size_t lds_bytes;
    V2LaunchFn launch, launch_trace;
    void (*pack)(const V2Args&, float*);
};
template <int G_EMB, int DV, int KPC, int H0C, int H1C, bool FOLD, bool TRACE, bool REG>
void v2_launch(const V2Run& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image,
               int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_deepfm_v2_chain<G_EMB, DV, KPC, H0C, H1C, V2_WAVES, FOLD, TRACE, REG>), dim3(grid), dim3(V2_WAVES * 64), lds, st,
                       a, ids, dense, out, B, err, image);
}
template <int G_EMB, int DV, int KPC, int H0C, int H1C, bool FOLD>
void v2_pack(const V2Args& a, float* image) {
    hipLaunchKernelGGL((k_v2_pack_image<G_EMB, DV, KPC, H0C, H1C, FOLD>), dim3(1), dim3(256), 0, 0, a, image);
}
#define V2_LDS(G_EMB, DV, KPC, H0C, H1C, FOLD) \
    (sizeof(float) * (V2Lds<G_EMB, DV, KPC, H0C, H1C, FOLD>::total_pad + V2_WAVES * V2Lds<G_EMB, DV, KPC, H0C, H1C, FOLD>::stage_floats))
#define V2_KFN(G_EMB, DV, KPC, H0C, H1C, FOLD, TRACE, REG) \
    reinterpret_cast<const void*>(&k_deepfm_v2_chain<G_EMB, DV, KPC, H0C, H1C, V2_WAVES, FOLD, TRACE, REG>)
#define V2_VARIANT(G_EMB, DV, KPC, H0C, H1C, FOLD, REG)                                                                \
    {G_EMB, DV, KPC, H0C, H1C, FOLD, REG, V2_KFN(G_EMB, DV, KPC, H0C, H1C, FOLD, false, REG), nullptr,                 \
     V2_LDS(G_EMB, DV, KPC, H0C, H1C, FOLD), &v2_launch<G_EMB, DV, KPC, H0C, H1C, FOLD, false, REG>, nullptr,          \
     &v2_pack<G_EMB, DV, KPC, H0C, H1C, FOLD>}
#define V2_VARIANT_TRACED(G_EMB, DV, KPC, H0C, H1C, FOLD, REG)                                                         \
    {G_EMB, DV, KPC, H0C, H1C, FOLD, REG, V2_KFN(G_EMB, DV, KPC, H0C, H1C, FOLD, false, REG),                          \
     V2_KFN(G_EMB, DV, KPC, H0C, H1C, FOLD, true, REG), V2_LDS(G_EMB, DV, KPC, H0C, H1C, FOLD),                        \
     &v2_launch<G_EMB, DV, KPC, H0C, H1C, FOLD, false, REG>, &v2_launch<G_EMB, DV, KPC, H0C, H1C, FOLD, true, REG>,    \
     &v2_pack<G_EMB, DV, KPC, H0C, H1C, FOLD>}
const V2Variant kV2Variants[] = {
    V2_VARIANT_TRACED(6, 4, 1, 2, 1, true, true),    // BASELINE config 2: 6 fields, projection 16 (folded into the tables), deep 32-16
    V2_VARIANT(6, 4, 1, 2, 1, false, false),         // ... with the projections computed per sample (D=16), weights in LDS
    V2_VARIANT(4, 4, 1, 2, 1, true, true),           // 4 fields, projection 16 (config-4 shape gathers 128-B projected rows instead of 256-B)
    V2_VARIANT(4, 4, 1, 2, 1, false, false),         // 4 fields, D=16
    V2_VARIANT(5, 4, 1, 2, 1, true, true),           // other field counts (folded only)
    V2_VARIANT(3, 4, 1, 2, 1, true, true),
    V2_VARIANT(2, 4, 1, 2, 1, true, true),
};

// ---- dispatch table for k_deepfm_v2_joint<G_BIG, NJF, KPC, H0C, H1C, WAVES> ----
typedef void (*V2JLaunchFn)(const V2JRun&, const int*, const float*, float*, int, int*, const float*, int, size_t, hipStream_t);
typedef void (*V2JLaunchManyFn)(const V2JRun&, const V2JMany&, int, int*, const float*, int, size_t, hipStream_t);
template <int G_BIG, int NJF, bool HALF>
void v2j_launch(const V2JRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image,
                int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_deepfm_v2_joint<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), dim3(grid), dim3(V2_WAVES * 64), lds, st,
                       a, ids, dense, out, B, err, image);
}
template <int G_BIG, int NJF, bool HALF>
void v2j_launch_many(const V2JRun& a, const V2JMany& m, int B, int* err, const float* image, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_deepfm_v2_joint_many<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), dim3(grid), dim3(V2_WAVES * 64), lds, st,
                       a, m, B, err, image);
}
struct V2JVariant {
    int g_big, njf, kpc;
    bool half;                            // big fields on split-f16 MFMA
    const void* fn;
    const void* fn_many;
    V2JLaunchFn launch;
    V2JLaunchManyFn launch_many;
};
#define V2J_VARIANT(G_BIG, NJF, HALF) \
    {G_BIG, NJF, 1, HALF, reinterpret_cast<const void*>(&k_deepfm_v2_joint<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), \
     reinterpret_cast<const void*>(&k_deepfm_v2_joint_many<G_BIG, NJF, 1, 2, 1, V2_WAVES, HALF>), &v2j_launch<G_BIG, NJF, HALF>, \
     &v2j_launch_many<G_BIG, NJF, HALF>}
#define V2J_BOTH(G_BIG, NJF) V2J_VARIANT(G_BIG, NJF, true), V2J_VARIANT(G_BIG, NJF, false)
const V2JVariant kV2JVariants[] = {
    V2J_BOTH(3, 3),    // BASELINE config 2: movieId, userId, userRatedMovie1 + a joint table of the three genre fields
    V2J_BOTH(2, 2),    // the reference's own four fields (movieId, userId + two genres), config-4 shape
    V2J_BOTH(3, 2), V2J_BOTH(3, 1), V2J_BOTH(2, 3), V2J_BOTH(2, 1), V2J_BOTH(1, 3), V2J_BOTH(1, 2), V2J_BOTH(1, 1),
};

// k_deepfm_v2_joint1<G_BIG, NJF>: the one-task-per-wave shape of the split-f16 joint kernel (k_chain_v2j1.h)
typedef void (*V2J1LaunchFn)(const V2JRun&, const int*, const float*, float*, int, int*, const float*, int, size_t, hipStream_t);
template <int G_BIG, int NJF>
void v2j1_launch(const V2JRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image, int grid,
                 size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_deepfm_v2_joint1<G_BIG, NJF>), dim3(grid), dim3(V2J1_WAVES * 64), lds, st, a, ids, dense, out, B, err, image);
}
struct V2J1Variant { int g_big, njf; const void* fn; V2J1LaunchFn launch; int image_floats; };
#define V2J1_VARIANT(G_BIG, NJF) \
    {G_BIG, NJF, reinterpret_cast<const void*>(&k_deepfm_v2_joint1<G_BIG, NJF>), &v2j1_launch<G_BIG, NJF>, V2J1Lds<G_BIG>::total_pad}
const V2J1Variant kV2J1Variants[] = {
    V2J1_VARIANT(3, 3), V2J1_VARIANT(2, 2), V2J1_VARIANT(3, 2), V2J1_VARIANT(3, 1), V2J1_VARIANT(2, 3), V2J1_VARIANT(2, 1),
    V2J1_VARIANT(1, 3), V2J1_VARIANT(1, 2), V2J1_VARIANT(1, 1),
};

// Recognise the plan models.DeepFMv2 emits (DeepFM_v2.py graph) and fill the fused kernel's arguments.
bool match_v2_chain(sprk_engine* h) {
    const sprk_plan& p = h->plan;
    if (p.model_kind != SPRK_MODEL_DEEPFM_V2 || p.din.enabled || p.n_bufs != 2) return false;
    V2Args a;
    memset(&a, 0, sizeof(a));
    int g_emb = 0;
    while (g_emb < p.n_segs && p.segs[g_emb].kind == SPRK_SEG_ROWS) ++g_emb;
    if (g_emb < 1 || g_emb > V2_MAX_FIELDS) return false;
    if (p.n_id_cols > 8 || p.n_dense > 8 || p.n_dense < 1) return false;   // one 16-B/lane load stages a task's ids+numerics
    const int Dp = p.segs[0].row_stride;
    bool raw_over_4g = false;                 // a raw table beyond 32-bit byte offsets: fine when folded (the kernel never reads it)
    for (int g = 0; g < g_emb; ++g) {
        const sprk_seg& s = p.segs[g];
        if (s.row_stride != Dp || s.count * 4 != Dp || s.dst != g * Dp) return false;
        // the fused kernel needs the all-zero row at index vocab and (unfolded) 32-bit element offsets
        const size_t need = ((size_t)s.vocab + 1) * Dp * sizeof(float);
        if (h->slot_bytes[s.slot] < need) return false;
        if (need >= ((size_t)1 << 32)) raw_over_4g = true;
        a.emb_col[g] = s.field; a.emb_vocab[g] = s.vocab; a.table[g] = (const float*)h->slot_ptr[s.slot];
    }
    int si = g_emb;
    if (si >= p.n_segs || p.segs[si].kind != SPRK_SEG_DENSE || p.segs[si].field != 0 || p.segs[si].count > 8) return false;
    const int n_num = p.segs[si].count, num_off = p.segs[si].dst;
    ++si;
    if (si < p.n_segs && p.segs[si].kind == SPRK_SEG_ZERO) ++si;
    const int n_fo = p.n_segs - si;
    if (n_fo < 1 || n_fo > V2_MAX_FIELDS) return false;
    const int scal_off = p.segs[si].dst;
    for (int i = 0; i < n_fo; ++i) {
        const sprk_seg& s = p.segs[si + i];
        if (s.kind != SPRK_SEG_SCALAR || s.dst != scal_off + i) return false;
        if (h->slot_bytes[s.slot] < ((size_t)s.vocab + 1) * sizeof(float)) return false;
        a.fo_col[i] = s.field; a.fo_vocab[i] = s.vocab; a.w1[i] = (const float*)h->slot_ptr[s.slot];
    }
    if (p.n_ops != g_emb + 4 || p.n_taps != 4) return false;
    const int Kp = p.ops[0].N, G = g_emb + 1;
    for (int g = 0; g < g_emb; ++g) {
        const sprk_op& o = p.ops[g];
        if (o.kind != SPRK_OP_DENSE || o.act != SPRK_ACT_NONE || o.src_buf != 0 || o.src_off != g * Dp || o.K != Dp ||
            o.dst_buf != 1 || o.dst_off != g * Kp || o.N != Kp || o.ldw != Dp) return false;
        a.Wp[g] = (const float*)h->slot_ptr[o.w_slot]; a.bp[g] = (const float*)h->slot_ptr[o.b_slot];
    }
    {
        const sprk_op& o = p.ops[g_emb];
        if (o.kind != SPRK_OP_DENSE || o.act != SPRK_ACT_NONE || o.src_buf != 0 || o.src_off != num_off || o.K > 8 ||
            o.K < n_num || o.dst_buf != 1 || o.dst_off != g_emb * Kp || o.N != Kp) return false;
        a.Wp[g_emb] = (const float*)h->slot_ptr[o.w_slot]; a.bp[g_emb] = (const float*)h->slot_ptr[o.b_slot];
        a.ldp_num = o.ldw;
    }
    a.ldp_emb = Dp;
    const sprk_op& fm = p.ops[g_emb + 1];
    if (fm.kind != SPRK_OP_FM_SUMSQ || fm.src_buf != 1 || fm.src_off != 0 || fm.groups != G || fm.group_stride != Kp ||
        fm.K > Kp || fm.dst_buf != 0) return false;
    const sprk_op& d0 = p.ops[g_emb + 2];
    if (d0.kind != SPRK_OP_DENSE || d0.act != SPRK_ACT_RELU || d0.src_buf != 1 || d0.src_off != 0 || d0.K != G * Kp ||
        d0.ldw != G * Kp || d0.dst_buf != 0 || d0.dst_off != 0) return false;
    const sprk_op& d1 = p.ops[g_emb + 3];
    if (d1.kind != SPRK_OP_DENSE || d1.act != SPRK_ACT_RELU || d1.src_buf != 0 || d1.src_off != 0 || d1.K != d0.N ||
        d1.ldw != d0.N || d1.dst_buf != 1 || d1.dst_off != 0) return false;
    a.W0 = (const float*)h->slot_ptr[d0.w_slot]; a.b0 = (const float*)h->slot_ptr[d0.b_slot];
    a.W1 = (const float*)h->slot_ptr[d1.w_slot]; a.b1 = (const float*)h->slot_ptr[d1.b_slot];
    const sprk_tap &t0 = p.taps[0], &t1 = p.taps[1], &t2 = p.taps[2], &t3 = p.taps[3];
    if (t0.buf != 0 || t0.off != scal_off || t0.len != n_fo || t0.w_slot != -1) return false;
    if (t1.buf != 0 || t1.off != num_off || t1.len != n_num || t1.w_slot < 0 || t1.scale != t0.scale) return false;
    if (t2.buf != 0 || t2.off != fm.dst_off || t2.len != fm.K || t2.w_slot < 0 || t2.scale != 1.0f || t2.bias != 0.0f) return false;
    if (t3.buf != 1 || t3.off != 0 || t3.len > d1.N || t3.w_slot < 0 || t3.scale != 1.0f || t3.bias != 0.0f) return false;
    a.fo_num_w = (const float*)h->slot_ptr[t1.w_slot];
    a.hfm = (const float*)h->slot_ptr[t2.w_slot]; a.n_hfm = t2.len;
    a.hdeep = (const float*)h->slot_ptr[t3.w_slot]; a.n_hdeep = t3.len;
    a.h0w = t0.scale; a.fo_bias = t0.bias + t1.bias; a.head_bias = p.head_bias;
    a.F = p.n_id_cols; a.ND = p.n_dense; a.n_num = n_num; a.n_fo = n_fo;
    const int dv = Dp / 4, kpc = Kp / 16, h0c = d0.N / 16, h1c = d1.N / 16;
    // fold the per-field projections into the tables when that never widens a gathered row
    const char* fmode = getenv("SPRK_V2_FOLD");              // A/B switch: "0" = compute projections per sample
    size_t total_rows = 0;
    for (int g = 0; g < g_emb; ++g) total_rows += (size_t)a.emb_vocab[g] + 1;
    // (32-bit byte offsets into ONE buffer of folded rows: needs < 4 GiB)
    const bool want_fold = Kp <= Dp && Kp + 16 <= 64 && total_rows * (size_t)(Kp + 16) * 4 < ((size_t)1 << 32) &&
                           !(fmode && fmode[0] == '0');
    const bool want_reg = want_fold;                         // folded tables <=> register-resident scoring stage
    if (raw_over_4g && !want_fold) return false;             // e.g. BASELINE config 4's 27 M x 64 table (6.9 GB): folded rows only
    // the fused kernel reads ONE id per field for both the embedding row and the first-order
    // weight: the two field lists must be the same set of ids columns
    if (n_fo != g_emb) return false;
    {
        const char* rowsw = getenv("SPRK_V2_ROWS");             // A/B switch: "1" = k_rows_chain even where k_deepfm_v2_joint fits
        if (rowsw && rowsw[0] == '1' && !raw_over_4g) {
            h->v2 = a;
            h->rows_g_emb = g_emb;
            h->rows_from_v2 = true;
            return false;
        }
    }
    for (size_t v = 0; v < sizeof(kV2Variants) / sizeof(kV2Variants[0]); ++v) {
        const V2Variant& vv = kV2Variants[v];
        if (vv.fold != want_fold || vv.reg != want_reg) continue;
        if (vv.g_emb == g_emb && (vv.fold || vv.dv == dv) && vv.kpc == kpc && vv.h0c == h0c && vv.h1c == h1c) {
            V2Run run;
            memset(&run, 0, sizeof(run));
            size_t fo_floats = 0;
            for (int g = 0; g < g_emb; ++g) {
                int hit = -1;
                for (int i = 0; i < n_fo; ++i)
                    if (a.fo_col[i] == a.emb_col[g] && a.fo_vocab[i] == a.emb_vocab[g]) hit = i;
                if (hit < 0) return false;
                run.col[g] = a.emb_col[g]; run.vocab[g] = a.emb_vocab[g];
                run.table[g] = a.table[g];
                run.fo_off[g] = (unsigned)fo_floats;
                fo_floats += (size_t)a.emb_vocab[g] + 1;
            }
            if (fo_floats >= ((size_t)1 << 31)) return false;
            // w1 pointers in embedding-group order
            const float* w1g[V2_MAX_FIELDS];
            for (int g = 0; g < g_emb; ++g) {
                for (int i = 0; i < n_fo; ++i)
                    if (a.fo_col[i] == a.emb_col[g] && a.fo_vocab[i] == a.emb_vocab[g]) w1g[g] = a.w1[i];
            }
            for (int g = 0; g < g_emb; ++g) a.w1[g] = w1g[g];
            run.F = a.F; run.ND = a.ND; run.n_num = a.n_num;
            run.h0w = a.h0w; run.fo_bias = a.fo_bias; run.head_bias = a.head_bias;
            h->v2run = run;
            h->v2 = a;
            h->v2_fo_floats = fo_floats;
            h->v2_variant = (int)v;
            h->v2_lds_bytes = vv.lds_bytes;
            return true;
        }
    }
    // no k_deepfm_v2_chain instantiation (e.g. the reference's Dense(64) projections): the parsed plan goes to k_rows_chain
    if (kpc >= 1 && kpc <= 4 && h0c >= 1 && h1c >= 1 && !raw_over_4g) {
        h->v2 = a;
        h->rows_g_emb = g_emb;
        h->rows_from_v2 = true;
    }
    return false;
}

// Split the fields of a folded DeepFM_v2 engine into big ones (gathered per field) and a joint group of
// small-vocabulary ones (one gather per sample), build the joint table.  Leaves v2j_variant = -1 when the
// model has no small field or no instantiation fits.
int wide_dynamic_range(const float* rows, long long nrows, int row_floats, int ncols, float mx, bool* wide);
int setup_v2_joint(sprk_engine* h) {
    const V2Variant& vv = kV2Variants[h->v2_variant];
    const char* jm = getenv("SPRK_V2_JOINT");                 // A/B switch: "0" = per-field gathers only
    if (!vv.fold || !vv.reg || vv.kpc != 1 || vv.h0c != 2 || vv.h1c != 1 || (jm && jm[0] == '0')) return SPRK_OK;
    const int KP = 16, H0 = 32, G = vv.g_emb;
    int big[V2_MAX_FIELDS], nbig = 0, jf[V2_MAX_FIELDS], njf = 0;
    for (int g = 0; g < G; ++g) {
        const long long v1 = (long long)h->v2run.vocab[g] + 1;
        if (v1 <= 32 && njf < V2J_MAX_JF) jf[njf++] = g;      // small enough to live in LDS
        else big[nbig++] = g;
    }
    if (njf < 1 || nbig < 1 || nbig > 3) return SPRK_OK;
    // HALF: scales from max|P| over the big fields' folded rows and max|W0|; refused for non-finite weights
    const char* hm = getenv("SPRK_V2_HALF");                  // A/B switch: "0" = big fields on f32 MFMA
    bool half = !(hm && hm[0] == '0');
    float p_scale = 1.f, w_scale = 1.f;
    if (half) {
        unsigned* d_max = nullptr;
        HIP_TRY(hipMalloc((void**)&d_max, 2 * sizeof(unsigned)));
        HIP_TRY(hipMemset(d_max, 0, 2 * sizeof(unsigned)));
        for (int b = 0; b < nbig; ++b) {
            const long long rows = (long long)h->v2run.vocab[big[b]] + 1;
            long long blocks = (rows * KP + 255) / 256;
            if (blocks > 8192) blocks = 8192;
            hipLaunchKernelGGL(k_v2_absmax, dim3((unsigned)blocks), dim3(256), 0, 0,
                               h->v2_folded + (size_t)h->v2run.rowbase[big[b]] * (KP + 16), rows, KP + 16, KP, d_max);
        }
        hipLaunchKernelGGL(k_v2_absmax, dim3(8), dim3(256), 0, 0, h->v2.W0, (long long)H0, (G + 1) * KP, (G + 1) * KP, d_max + 1);
        HIP_TRY(hipGetLastError());
        unsigned bits[2];
        HIP_TRY(hipMemcpy(bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
        (void)hipFree(d_max);
        float mx[2];
        memcpy(mx, bits, sizeof(mx));
        for (int i = 0; i < 2; ++i) {
            if (!(mx[i] < 3.0e38f)) { half = false; break; }     // NaN / Inf in the weights: keep the f32 path
            int e = 0;
            if (mx[i] > 0.f) { (void)frexpf(mx[i], &e); e = 15 - e; }   // mx * 2^e in [2^14, 2^15)
            if (e > 60) e = 60;
            if (e < -60) e = -60;
            (i == 0 ? p_scale : w_scale) = ldexpf(1.f, e);
        }
        // an outlier row next to ordinary ones: the ordinary rows' lo halves would be subnormal -> keep the f32 variant
        for (int b = 0; half && b < nbig; ++b) {
            bool wide = false;
            if (int rcw = wide_dynamic_range(h->v2_folded + (size_t)h->v2run.rowbase[big[b]] * (KP + 16),
                                             (long long)h->v2run.vocab[big[b]] + 1, KP + 16, KP, mx[0], &wide)) return rcw;
            if (wide) half = false;
        }
        if (half) {
            bool wide = false;
            if (int rcw = wide_dynamic_range(h->v2.W0, (long long)H0, (G + 1) * KP, (G + 1) * KP, mx[1], &wide)) return rcw;
            if (wide) half = false;
        }
    }
    int variant = -1;
    for (size_t v = 0; v < sizeof(kV2JVariants) / sizeof(kV2JVariants[0]); ++v)
        if (kV2JVariants[v].g_big == nbig && kV2JVariants[v].njf == njf && kV2JVariants[v].half == half) variant = (int)v;
    if (variant < 0) return SPRK_OK;
    V2JRun& r = h->v2j_run;
    memset(&r, 0, sizeof(r));
    r.F = h->v2run.F; r.ND = h->v2run.ND; r.n_num = h->v2run.n_num;
    r.h0w = h->v2run.h0w; r.fo_bias = h->v2run.fo_bias; r.head_bias = h->v2run.head_bias;
    for (int b = 0; b < nbig; ++b) {
        r.big_col[b] = h->v2run.col[big[b]]; r.big_vocab[b] = h->v2run.vocab[big[b]];
        r.big_rowbase[b] = h->v2run.rowbase[big[b]]; r.big_grp[b] = big[b];
    }
    size_t small_floats = 0;
    for (int f = 0; f < njf; ++f) {
        r.j_col[f] = h->v2run.col[jf[f]]; r.j_vocab[f] = h->v2run.vocab[jf[f]];
        r.s_off[f] = (int)small_floats;
        small_floats += ((size_t)r.j_vocab[f] + 1) * V2J_SS;
    }
    small_floats = (small_floats + 255) & ~(size_t)255;          // whole 1-KB LDS-DMA pieces
    r.wf_off = (int)small_floats;                                // HALF: the numerics' fold through deep0, [H0][8]
    if (half) small_floats += ((size_t)H0 * 8 + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void**)&h->v2j_tab, small_floats * sizeof(float)));
    HIP_TRY(hipMemset(h->v2j_tab, 0, small_floats * sizeof(float)));
    for (int f = 0; f < njf; ++f) {
        const int rows = r.j_vocab[f] + 1;
        hipLaunchKernelGGL(k_v2_fold_small, dim3((rows + 3) / 4), dim3(256), 0, 0,
                           h->v2_folded + (size_t)h->v2run.rowbase[jf[f]] * (KP + 16), KP, H0, jf[f], h->v2.W0, (G + 1) * KP,
                           h->v2.b0, f == 0 ? 1 : 0, h->v2j_tab + r.s_off[f], rows);
    }
    if (half)
        hipLaunchKernelGGL(k_v2j_fold_num, dim3(8), dim3(256), 0, 0, h->v2.W0, (G + 1) * KP, G * KP, h->v2.Wp[G], h->v2.ldp_num,
                           h->v2.bp[G], h->v2.n_num, KP, H0, h->v2j_tab + r.wf_off, h->v2j_tab + r.s_off[0], r.j_vocab[0] + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    r.small_floats = (int)small_floats;
    r.tab0 = h->v2_folded;
    r.small = h->v2j_tab;
    r.w_scale = w_scale; r.unscale_h = 1.f / (p_scale * w_scale); r.unscale_s = 1.f / p_scale;
    if (half) {
        size_t big_rows = 0;
        for (int b = 0; b < nbig; ++b) big_rows += (size_t)r.big_vocab[b] + 1;
        if (big_rows * (KP + 16) * sizeof(float) >= ((size_t)1 << 32)) return fail(SPRK_EINVAL, "split rows exceed 32-bit offsets");
        HIP_TRY(hipMalloc((void**)&h->v2j_big, big_rows * (KP + 16) * sizeof(float)));
        h->derived_bytes += big_rows * (KP + 16) * sizeof(float);
        size_t base = 0;
        for (int b = 0; b < nbig; ++b) {
            const long long rows = (long long)r.big_vocab[b] + 1;
            long long nb = (rows * 8 + 255) / 256;
            if (nb > 65536) nb = 65536;
            hipLaunchKernelGGL(k_v2_split_rows, dim3((unsigned)nb), dim3(256), 0, 0,
                               h->v2_folded + (size_t)r.big_rowbase[b] * (KP + 16), h->v2j_big + base * (KP + 16), rows, p_scale);
            r.big_rowbase[b] = (unsigned)base;
            base += (size_t)rows;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        r.tab0 = h->v2j_big;
    }
    h->v2j_lds_bytes = vv.lds_bytes + small_floats * sizeof(float);
    HIP_TRY(hipFuncSetAttribute(kV2JVariants[variant].fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2j_lds_bytes));
    HIP_TRY(hipFuncSetAttribute(kV2JVariants[variant].fn_many, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2j_lds_bytes));
    h->v2j_variant = variant;
    // the one-task-per-wave shape for strict one-batch launches (k_chain_v2j1.h); SPRK_V2J_ONE=0: looped kernel only
    const char* one = getenv("SPRK_V2J_ONE");
    if (half && !(one && one[0] == '0')) {
        for (size_t v = 0; v < sizeof(kV2J1Variants) / sizeof(kV2J1Variants[0]); ++v) {
            const V2J1Variant& ov = kV2J1Variants[v];
            if (ov.g_big != nbig || ov.njf != njf) continue;
            HIP_TRY(hipMalloc((void**)&h->v2j1_image, (size_t)ov.image_floats * sizeof(float)));
            hipLaunchKernelGGL(k_v2j1_pack_image, dim3(1), dim3(256), 0, 0, h->v2, r, nbig, G + 1, h->v2j1_image);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            h->v2j1_lds_bytes = ((size_t)ov.image_floats + small_floats + (size_t)V2J1_WAVES * 256) * sizeof(float);
            HIP_TRY(hipFuncSetAttribute(ov.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->v2j1_lds_bytes));
        }
    }
    return SPRK_OK;
}

// ---- dispatch table for k_rows_chain<KPC, H0C, H1C, G_BIG, NJF, HASNUM> ----
constexpr int RC_WAVES = 8;
typedef void (*RowsLaunchFn)(const RowsRun&, const int*, const float*, float*, int, int*, const float*, int, size_t, hipStream_t);
typedef void (*RowsLaunchManyFn)(const RowsRun&, const RowsMany&, int, int*, const float*, int, size_t, hipStream_t);
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM>
void rows_launch(const RowsRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image, int grid,
                 size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_rows_chain<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES>), dim3(grid), dim3(RC_WAVES * 64), lds, st, a, ids, dense,
                       out, B, err, image);
}
template <int KPC, int H0C, int H1C, int G_BIG, int NJF, bool HASNUM>
void rows_launch_many(const RowsRun& a, const RowsMany& m, int B, int* err, const float* image, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_rows_chain_many<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES>), dim3(grid), dim3(RC_WAVES * 64), lds, st, a, m, B,
                       err, image);
}
struct RowsVariant {
    int kpc, h0c, h1c, g_big, njf;
    bool hasnum;
    const void* fn;
    const void* fn_many;
    RowsLaunchFn launch;
    RowsLaunchManyFn launch_many;
    int image_floats, ss, rb;
};
#define ROWS_VARIANT(KPC, H0C, H1C, G_BIG, NJF, HASNUM)                                                                          \
    {KPC, H0C, H1C, G_BIG, NJF, HASNUM, reinterpret_cast<const void*>(&k_rows_chain<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES>),   \
     reinterpret_cast<const void*>(&k_rows_chain_many<KPC, H0C, H1C, G_BIG, NJF, HASNUM, RC_WAVES>),                               \
     &rows_launch<KPC, H0C, H1C, G_BIG, NJF, HASNUM>, &rows_launch_many<KPC, H0C, H1C, G_BIG, NJF, HASNUM>,                        \
     RowsLds<KPC, H0C, H1C, HASNUM>::total_pad, RowsLds<KPC, H0C, H1C, HASNUM>::SS, RowsLds<KPC, H0C, H1C, HASNUM>::RB}
const RowsVariant kRowsVariants[] = {
    ROWS_VARIANT(4, 2, 1, 2, 2, true),     // DeepFM_v2.py as written: Dense(64) projections, deep 32-16, movieId + userId + two genre fields
    ROWS_VARIANT(4, 2, 1, 2, 1, true), ROWS_VARIANT(4, 2, 1, 1, 1, true), ROWS_VARIANT(4, 2, 1, 3, 3, true), ROWS_VARIANT(4, 2, 1, 3, 1, true),
    ROWS_VARIANT(2, 2, 1, 2, 2, true),     // projection width 32
    ROWS_VARIANT(2, 2, 1, 3, 3, true),
    ROWS_VARIANT(1, 2, 1, 3, 3, true),     // BASELINE config 2's shape on this kernel (A/B against k_deepfm_v2_joint: SPRK_V2_ROWS=1)
    ROWS_VARIANT(0, 1, 1, 2, 0, false),    // NeuralCF.py:45-53: two embedding columns -> Dense(10) -> Dense(10) -> Dense(1)
};
int find_rows_variant(int kpc, int h0c, int h1c, int g_big, int njf, bool hasnum) {
    for (size_t v = 0; v < sizeof(kRowsVariants) / sizeof(kRowsVariants[0]); ++v) {
        const RowsVariant& r = kRowsVariants[v];
        if (r.kpc == kpc && r.h0c == h0c && r.h1c == h1c && r.g_big == g_big && r.njf == njf && r.hasnum == hasnum) return (int)v;
    }
    return -1;
}
// device -> host copy of a small float matrix
int pull(std::vector<float>& dst, const float* src, size_t n) {
    dst.resize(n);
    HIP_TRY(hipMemcpy(dst.data(), src, n * sizeof(float), hipMemcpyDeviceToHost));
    return SPRK_OK;
}
int rows_finish(sprk_engine* h, const RowsVariant& rv, const std::vector<float>& image, size_t small_floats) {
    HIP_TRY(hipMalloc((void**)&h->rows_image, (size_t)rv.image_floats * sizeof(float)));
    HIP_TRY(hipMemcpy(h->rows_image, image.data(), (size_t)rv.image_floats * sizeof(float), hipMemcpyHostToDevice));
    h->rows_lds_bytes = ((size_t)rv.image_floats + RC_WAVES * 256 + small_floats) * sizeof(float);
    if (h->rows_lds_bytes > 160 * 1024) return fail(SPRK_EINVAL, "rows chain needs %zu bytes of LDS", h->rows_lds_bytes);
    HIP_TRY(hipFuncSetAttribute(rv.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->rows_lds_bytes));
    HIP_TRY(hipFuncSetAttribute(rv.fn_many, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->rows_lds_bytes));
    HIP_TRY(hipDeviceSynchronize());
    return SPRK_OK;
}

// DeepFM_v2 plans whose projection width has no k_deepfm_v2_joint instantiation (the reference's own Dense(64)): every field
// becomes a table of rows {P | W0^T P} (+ scalars), see k_rows_chain.h.  h->v2 holds the parsed plan (match_v2_chain).
int setup_rows_v2(sprk_engine* h) {
    const V2Args& a = h->v2;
    const sprk_plan& p = h->plan;
    const int G = h->rows_g_emb;
    const sprk_op &d0 = p.ops[G + 2], &d1 = p.ops[G + 3];
    const int KP = p.ops[0].N, H0 = d0.N, H1 = d1.N, Dp = a.ldp_emb;
    int big[V2_MAX_FIELDS], nbig = 0, sm[V2_MAX_FIELDS], nsm = 0;
    for (int g = 0; g < G; ++g) {
        if ((long long)a.emb_vocab[g] + 1 <= 32 && nsm < RC_MAX_SMALL) sm[nsm++] = g;
        else big[nbig++] = g;
    }
    if (nbig < 1 || nbig > RC_MAX_BIG) return SPRK_OK;
    const int variant = find_rows_variant(KP / 16, H0 / 16, H1 / 16, nbig, nsm, true);
    if (variant < 0) return SPRK_OK;
    const RowsVariant& rv = kRowsVariants[variant];
    // first-order weights in embedding-group order (one ids column feeds both)
    const float* w1g[V2_MAX_FIELDS];
    for (int g = 0; g < G; ++g) {
        w1g[g] = nullptr;
        for (int i = 0; i < a.n_fo; ++i)
            if (a.fo_col[i] == a.emb_col[g] && a.fo_vocab[i] == a.emb_vocab[g]) w1g[g] = a.w1[i];
        if (!w1g[g]) return SPRK_OK;
    }
    RowsRun& r = h->rows_run;
    memset(&r, 0, sizeof(r));
    r.F = a.F; r.ND = a.ND; r.n_num = a.n_num;
    size_t big_rows = 0;
    for (int b = 0; b < nbig; ++b) {
        r.big_col[b] = a.emb_col[big[b]]; r.big_vocab[b] = a.emb_vocab[big[b]];
        r.big_rowbase[b] = (unsigned)big_rows; r.big_scal[b] = (unsigned)big_rows;
        big_rows += (size_t)a.emb_vocab[big[b]] + 1;
    }
    if (big_rows >= ((size_t)1 << 31)) return SPRK_OK;
    size_t small_floats = 0;
    for (int f = 0; f < nsm; ++f) {
        r.s_col[f] = a.emb_col[sm[f]]; r.s_vocab[f] = a.emb_vocab[sm[f]]; r.s_off[f] = (int)small_floats;
        small_floats += ((size_t)a.emb_vocab[sm[f]] + 1) * rv.ss;
    }
    small_floats = (small_floats + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void**)&h->rows_tab, big_rows * rv.rb + 64));
    HIP_TRY(hipMemset(h->rows_tab, 0, big_rows * rv.rb + 64));
    HIP_TRY(hipMalloc((void**)&h->rows_scal, big_rows * sizeof(float) + 16));
    h->derived_bytes += big_rows * rv.rb + big_rows * sizeof(float);
    if (small_floats) {
        HIP_TRY(hipMalloc((void**)&h->rows_small, small_floats * sizeof(float)));
        HIP_TRY(hipMemset(h->rows_small, 0, small_floats * sizeof(float)));
    }
    auto build = [&](int g, float* out, int out_stride, float* scal_out) {
        const long long rows = (long long)a.emb_vocab[g] + 1;
        long long blocks = (rows + 3) / 4;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_rows_build, dim3((unsigned)blocks), dim3(256), 0, 0, a.table[g], Dp, rows, a.Wp[g], a.ldp_emb, a.bp[g], KP,
                           a.W0, d0.ldw, g * KP, H0, KP, (const float*)nullptr, w1g[g], a.hfm, a.n_hfm, a.h0w, out, out_stride, scal_out,
                           scal_out ? 0 : 1);
    };
    for (int b = 0; b < nbig; ++b)
        build(big[b], h->rows_tab + (size_t)r.big_rowbase[b] * (rv.rb / 4), rv.rb / 4, h->rows_scal + r.big_scal[b]);
    for (int f = 0; f < nsm; ++f) build(sm[f], h->rows_small + r.s_off[f], rv.ss, nullptr);
    HIP_TRY(hipGetLastError());
    // weight image (host): Wn, bn, M = W0[:, num block] Wn, c0 = b0 + W0[:, num block] bn, W1, b1, hfm, hd, fn
    std::vector<float> Wn, bn, W0, b0, W1, b1, hfm, hd, fnw;
    int rc;
    if ((rc = pull(Wn, a.Wp[G], (size_t)KP * a.ldp_num)) || (rc = pull(bn, a.bp[G], KP)) || (rc = pull(W0, a.W0, (size_t)H0 * d0.ldw)) ||
        (rc = pull(b0, a.b0, H0)) || (rc = pull(W1, a.W1, (size_t)H1 * d1.ldw)) || (rc = pull(b1, a.b1, H1)) ||
        (rc = pull(hfm, a.hfm, a.n_hfm)) || (rc = pull(hd, a.hdeep, a.n_hdeep)) || (rc = pull(fnw, a.fo_num_w, a.n_num))) return rc;
    std::vector<float> img(rv.image_floats, 0.f);
    const int SN = 12, S1 = H0 + 4;
    int off = 0;
    const int off_wn = off; off += KP * SN;
    const int off_bn = off; off += KP;
    const int off_m = off; off += H0 * SN;
    const int off_c0 = off; off += H0;
    const int off_w1 = off; off += H1 * S1;
    const int off_b1 = off; off += H1;
    const int off_hfm = off; off += KP;
    const int off_hd = off; off += H1;
    const int off_fn = off; off += 8;
    if (off > rv.image_floats) return fail(SPRK_EINVAL, "rows image layout mismatch");
    for (int n = 0; n < KP; ++n) {
        for (int k = 0; k < a.n_num && k < 8; ++k) img[off_wn + n * SN + k] = Wn[(size_t)n * a.ldp_num + k];
        img[off_bn + n] = bn[n];
    }
    for (int m = 0; m < H0; ++m) {
        const float* w = &W0[(size_t)m * d0.ldw + (size_t)G * KP];
        for (int k = 0; k < a.n_num && k < 8; ++k) {
            double acc = 0.0;
            for (int n = 0; n < KP; ++n) acc += (double)w[n] * (double)Wn[(size_t)n * a.ldp_num + k];
            img[off_m + m * SN + k] = (float)acc;
        }
        double c = b0[m];
        for (int n = 0; n < KP; ++n) c += (double)w[n] * (double)bn[n];
        img[off_c0 + m] = (float)c;
    }
    for (int n = 0; n < H1; ++n) {
        for (int k = 0; k < H0; ++k) img[off_w1 + n * S1 + k] = W1[(size_t)n * d1.ldw + k];
        img[off_b1 + n] = b1[n];
    }
    for (int n = 0; n < a.n_hfm && n < KP; ++n) img[off_hfm + n] = hfm[n];
    for (int n = 0; n < a.n_hdeep && n < H1; ++n) img[off_hd + n] = hd[n];
    for (int k = 0; k < a.n_num && k < 8; ++k) img[off_fn + k] = a.h0w * fnw[k];
    r.rows = h->rows_tab; r.scal = h->rows_scal; r.small = h->rows_small; r.small_floats = (int)small_floats;
    r.bias = a.head_bias + a.h0w * a.fo_bias;
    if ((rc = rows_finish(h, rv, img, small_floats))) return rc;
    h->rows_variant = variant;
    return SPRK_OK;
}

// NeuralCF.py:45-53 (neural_cf_model_1): concat(item row, user row) -> Dense(relu) -> Dense(relu) -> Dense(1, sigmoid).  The first
// Dense is linear in each row, so each field becomes a table of its 16 (padded) pre-activations: Q_f[id] = W0[:, f]^T E_f[id].
int setup_rows_ncf(sprk_engine* h) {
    const char* sw = getenv("SPRK_NCF_CHAIN");                // A/B switch: "0" = tile interpreter
    if (sw && sw[0] == '0') return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (p.model_kind != SPRK_MODEL_NEURALCF || p.din.enabled || p.n_segs != 2 || p.n_ops != 2 || p.n_taps != 1 || p.n_dense != 0) return SPRK_OK;
    if (p.n_id_cols > 8) return SPRK_OK;
    const sprk_op &o0 = p.ops[0], &o1 = p.ops[1];
    const sprk_tap& tp = p.taps[0];
    if (o0.kind != SPRK_OP_DENSE || o1.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_RELU || o1.act != SPRK_ACT_RELU) return SPRK_OK;
    if (o0.src_buf != 0 || o0.dst_off != 0 || o1.src_buf != o0.dst_buf || o1.src_off != 0 || o1.K != o0.N || o1.dst_off != 0) return SPRK_OK;
    if (tp.buf != o1.dst_buf || tp.off != 0 || tp.len > o1.N || tp.w_slot < 0 || tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
    const int H0 = o0.N, H1 = o1.N;
    const int variant = find_rows_variant(0, H0 / 16, H1 / 16, 2, 0, false);
    if (variant < 0) return SPRK_OK;
    const RowsVariant& rv = kRowsVariants[variant];
    RowsRun& r = h->rows_run;
    memset(&r, 0, sizeof(r));
    r.F = p.n_id_cols; r.ND = 0; r.n_num = 0;
    size_t rows_total = 0;
    for (int b = 0; b < 2; ++b) {
        const sprk_seg& sg = p.segs[b];
        if (sg.kind != SPRK_SEG_ROWS || sg.dst < o0.src_off || sg.dst + 4 * sg.count > o0.src_off + o0.K || 4 * sg.count > 64) return SPRK_OK;
        if (h->slot_bytes[sg.slot] < ((size_t)sg.vocab + 1) * sg.row_stride * sizeof(float)) return SPRK_OK;   // needs the zero row at index vocab
        r.big_col[b] = sg.field; r.big_vocab[b] = sg.vocab; r.big_rowbase[b] = (unsigned)rows_total;
        rows_total += (size_t)sg.vocab + 1;
    }
    if (rows_total >= ((size_t)1 << 31)) return SPRK_OK;
    HIP_TRY(hipMalloc((void**)&h->rows_tab, rows_total * rv.rb + 64));
    HIP_TRY(hipMemset(h->rows_tab, 0, rows_total * rv.rb + 64));
    h->derived_bytes += rows_total * rv.rb;
    const float* W0 = (const float*)h->slot_ptr[o0.w_slot];
    for (int b = 0; b < 2; ++b) {
        const sprk_seg& sg = p.segs[b];
        const long long rows = (long long)sg.vocab + 1;
        long long blocks = (rows + 3) / 4;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_rows_build, dim3((unsigned)blocks), dim3(256), 0, 0, (const float*)h->slot_ptr[sg.slot], sg.row_stride, rows,
                           (const float*)nullptr, 0, (const float*)nullptr, 0, W0, o0.ldw, sg.dst - o0.src_off, H0, 4 * sg.count,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0.f,
                           h->rows_tab + (size_t)r.big_rowbase[b] * (rv.rb / 4), rv.rb / 4, (float*)nullptr, 0);
    }
    HIP_TRY(hipGetLastError());
    std::vector<float> b0, W1, b1, hd;
    int rc;
    if ((rc = pull(b0, (const float*)h->slot_ptr[o0.b_slot], H0)) || (rc = pull(W1, (const float*)h->slot_ptr[o1.w_slot], (size_t)H1 * o1.ldw)) ||
        (rc = pull(b1, (const float*)h->slot_ptr[o1.b_slot], H1)) || (rc = pull(hd, (const float*)h->slot_ptr[tp.w_slot], tp.len))) return rc;
    std::vector<float> img(rv.image_floats, 0.f);
    const int S1 = H0 + 4;
    const int off_c0 = 0, off_w1 = off_c0 + H0, off_b1 = off_w1 + H1 * S1, off_hfm = off_b1 + H1, off_hd = off_hfm + 0;
    for (int m = 0; m < H0; ++m) img[off_c0 + m] = b0[m];
    for (int n = 0; n < H1; ++n) {
        for (int k = 0; k < H0; ++k) img[off_w1 + n * S1 + k] = W1[(size_t)n * o1.ldw + k];
        img[off_b1 + n] = b1[n];
    }
    for (int n = 0; n < tp.len; ++n) img[off_hd + n] = hd[n];
    r.rows = h->rows_tab; r.scal = nullptr; r.small = nullptr; r.small_floats = 0;
    r.bias = p.head_bias;
    if ((rc = rows_finish(h, rv, img, 0))) return rc;
    h->rows_variant = variant;
    return SPRK_OK;
}

// ---- dispatch table for k_din_attn<KC, HC> ----
typedef void (*DinLaunchFn)(const DinRun&, const int*, float*, float*, int, int*, int, size_t, hipStream_t);
template <int KC, int HC, int NP, bool HALF, int WPB>
void din_launch(const DinRun& a, const int* ids, float* pooled, float* att, int B, int* err, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_din_attn<KC, HC, NP, HALF, WPB, false>), dim3(grid), dim3(WPB * 64), lds, st, a, ids, pooled, att, B, err, DinAttnOne{});
}
typedef void (*DinLaunchManyFn)(const DinRun&, const DinAttnMany&, int, int*, int, size_t, hipStream_t);
template <int KC, int HC, int NP, bool HALF, int WPB>
void din_launch_many(const DinRun& a, const DinAttnMany& m, int B, int* err, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_din_attn<KC, HC, NP, HALF, WPB, true>), dim3(grid), dim3(WPB * 64), lds, st, a, (const int*)nullptr, (float*)nullptr, (float*)nullptr, B,
                       err, m);
}
struct DinVariant {
    int kc, hc, np;                   // np: gather passes compiled in (each covers 64 / (row_stride/4) history slots)
    bool half;                        // K = D contraction on split-f16 MFMA
    int wpb;                          // waves per workgroup: 12 = one workgroup per CU at 3 waves per SIMD, 4 = two at 2 (round 1)
    const void* fn;
    size_t lds_bytes;
    DinLaunchFn launch;
    const void* fn_many;              // several batches per launch (12-wave forms only; NULL otherwise)
    DinLaunchManyFn launch_many;
};
#define DIN_MANY_12(KC, HC, NP, HALF) reinterpret_cast<const void*>(&k_din_attn<KC, HC, NP, HALF, 12, true>), &din_launch_many<KC, HC, NP, HALF, 12>
#define DIN_VARIANT1(KC, HC, NP, HALF, WPB) {KC, HC, NP, HALF, WPB, reinterpret_cast<const void*>(&k_din_attn<KC, HC, NP, HALF, WPB, false>), DinLds<KC, HC, WPB>::bytes, &din_launch<KC, HC, NP, HALF, WPB>, nullptr, nullptr}
#define DIN_VARIANT12(KC, HC, NP) {KC, HC, NP, true, 12, reinterpret_cast<const void*>(&k_din_attn<KC, HC, NP, true, 12, false>), DinLds<KC, HC, 12>::bytes, &din_launch<KC, HC, NP, true, 12>, DIN_MANY_12(KC, HC, NP, true)}
#define DIN_VARIANT16(KC, HC, NP) {KC, HC, NP, true, 16, reinterpret_cast<const void*>(&k_din_attn<KC, HC, NP, true, 16, false>), DinLds<KC, HC, 16>::bytes, &din_launch<KC, HC, NP, true, 16>, \
                                   reinterpret_cast<const void*>(&k_din_attn<KC, HC, NP, true, 16, true>), &din_launch_many<KC, HC, NP, true, 16>}
#define DIN_VARIANT(KC, HC, NP) DIN_VARIANT1(KC, HC, NP, true, 4), DIN_VARIANT1(KC, HC, NP, false, 4)
const DinVariant kDinVariants[] = {     // first match wins: smallest sufficient pass count first; 12-wave forms before their 4-wave twins
    DIN_VARIANT12(2, 2, 2), DIN_VARIANT(2, 2, 2), DIN_VARIANT12(2, 2, 4), DIN_VARIANT(2, 2, 4),
    DIN_VARIANT16(2, 2, 7),             // (chosen only with SPRK_DIN_WPB=16: T <= 56, 4 waves per SIMD, no row prefetch)
    DIN_VARIANT12(2, 2, 7),    // BASELINE config 3: emb_dim 32, 50 history slots, attention hidden 32
    DIN_VARIANT(2, 2, 7),
    DIN_VARIANT12(2, 2, 8), DIN_VARIANT(2, 2, 8),
    DIN_VARIANT(1, 2, 1),               // the reference's own DIN.py: emb_dim 10 (rows padded to 12), 5 slots, hidden 32
    DIN_VARIANT(1, 2, 4),
};

// First-Dense fold for plans the tile interpreter runs.  A Dense layer is linear in its input, so the share of
// an embedding column is a table of its own: F_g[id] = W_g^T E_g[id] (N floats per id).  When a ROWS segment feeds
// nothing but the plan's first Dense op, the fold replaces "gather E_g[id] into the input slice, multiply by W_g
// on the matrix pipe" by "gather F_g[id] and add it to the layer's accumulator": the layer's K shrinks to the
// columns that really are per-sample data (numerics, the DIN pooled vector, crossed columns), at the price of
// N instead of D floats per gathered row.  DIN tail (DIN.py:161-166): K 168 -> 40; EmbeddingMLP: 108 -> 8.
// Same fp32 arithmetic, other association.  SPRK_TILE_FOLD=0 switches it off (A/B, tests).
int fold_first_dense(sprk_engine* h, DevPlan* dp) {
    const char* fm = getenv("SPRK_TILE_FOLD");
    if (fm && fm[0] == '0') return SPRK_OK;
    if (dp->n_ops < 1) return SPRK_OK;
    DevOp& op = dp->ops[0];
    if (op.kind != SPRK_OP_DENSE || op.src_buf != 0 || op.dst_buf == 0 || op.N > 512) return SPRK_OK;
    const int lo0 = op.src_off, hi0 = op.src_off + op.K;
    auto used_elsewhere = [&](int a, int b) {                 // is the GATHERED content of buffer 0's [a,b) read by anything but ops[0]?
        std::vector<char> live(b - a, 1);                     // columns still holding gathered data (later ops may overwrite buffer 0)
        auto reads = [&](int s0, int s1) {
            for (int c = (s0 > a ? s0 : a); c < (s1 < b ? s1 : b); ++c) if (live[c - a]) return true;
            return false;
        };
        auto writes = [&](int s0, int s1) { for (int c = (s0 > a ? s0 : a); c < (s1 < b ? s1 : b); ++c) live[c - a] = 0; };
        for (int i = 1; i < dp->n_ops; ++i) {
            const DevOp& o = dp->ops[i];
            if (o.src_buf == 0) {
                if (o.kind == SPRK_OP_PAIR_DOT) {
                    for (int p = 0; p < dp->n_pairs; ++p)
                        if (reads(dp->pair_a[p], dp->pair_a[p] + o.K) || reads(dp->pair_b[p], dp->pair_b[p] + o.K)) return true;
                } else if (o.kind == SPRK_OP_FM_SUMSQ) {
                    if (reads(o.src_off, o.src_off + (o.groups - 1) * o.group_stride + o.K)) return true;
                } else if (reads(o.src_off, o.src_off + o.K)) {
                    return true;
                }
            }
            if (o.dst_buf == 0) {
                const int w = o.kind == SPRK_OP_DENSE ? o.N : o.kind == SPRK_OP_PAIR_DOT ? dp->n_pairs : o.K;
                writes(o.dst_off, o.dst_off + w);
            }
        }
        for (int t = 0; t < dp->n_taps; ++t)
            if (dp->taps[t].buf == 0 && reads(dp->taps[t].off, dp->taps[t].off + dp->taps[t].len)) return true;
        return false;
    };
    std::vector<int> fold;
    size_t bytes = 0;
    for (int i = 0; i < dp->n_segs; ++i) {
        const DevSeg& sg = dp->segs[i];
        if (sg.kind != SPRK_SEG_ROWS) continue;
        const int a = sg.dst, b = sg.dst + 4 * sg.count;
        if (a < lo0 || b > hi0 || used_elsewhere(a, b)) continue;
        const size_t need = (size_t)sg.vocab * op.N * sizeof(float);
        if (need > ((size_t)2 << 30) || bytes + need > ((size_t)8 << 30)) continue;   // keep huge tables as plain row gathers
        if (fold.size() == 8) break;                          // the gather keeps at most 8 folded columns in flight per piece
        bytes += need;
        fold.push_back(i);
    }
    if (fold.empty()) return SPRK_OK;
    // new K range: hull of the columns that stay (everything in [lo0,hi0) not covered by a folded segment)
    std::vector<char> keep(hi0 - lo0, 1);
    for (int i : fold)
        for (int c = dp->segs[i].dst; c < dp->segs[i].dst + 4 * dp->segs[i].count; ++c) keep[c - lo0] = 0;
    int lo = hi0, hi = lo0;
    for (int c = lo0; c < hi0; ++c)
        if (keep[c - lo0]) { if (c < lo) lo = c; if (c + 1 > hi) hi = c + 1; }
    if (lo >= hi) { lo = lo0; hi = lo0; }
    lo &= ~3;
    hi = (hi + 3) & ~3;
    if (hi > hi0) hi = hi0;
    // W^T copy with the folded columns inside the hull zeroed; F tables
    const size_t wbytes = (size_t)op.N * op.ldw * sizeof(float);
    float* wcopy = nullptr;
    HIP_TRY(hipMalloc((void**)&wcopy, wbytes + 16));
    h->fold_bufs.push_back(wcopy);
    HIP_TRY(hipMemcpy(wcopy, op.W, wbytes, hipMemcpyDeviceToDevice));
    bool first = true;
    for (int i : fold) {
        DevSeg& sg = dp->segs[i];
        float* F = nullptr;
        HIP_TRY(hipMalloc((void**)&F, (size_t)sg.vocab * op.N * sizeof(float) + 16));
        h->derived_bytes += (size_t)sg.vocab * op.N * sizeof(float);
        h->fold_bufs.push_back(F);
        long long blocks = ((long long)sg.vocab * op.N + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_fold_dense_rows, dim3((unsigned)blocks), dim3(256), 0, 0, sg.table, (long long)sg.vocab, sg.row_stride,
                           4 * sg.count, op.W, op.ldw, sg.dst - lo0, op.N, F);
        HIP_TRY(hipGetLastError());
        const int c0 = sg.dst - lo0, c1 = c0 + 4 * sg.count;
        hipLaunchKernelGGL(k_zero_columns, dim3(16), dim3(256), 0, 0, wcopy, op.N, op.ldw, c0, c1);
        HIP_TRY(hipGetLastError());
        sg.kind = SEG_ROWS_ACC; sg.table = F; sg.row_stride = op.N; sg.count = op.N / 4; sg.dst = op.dst_off;
        sg.buf = op.dst_buf; sg.field2 = first ? 0 : 1;
        first = false;
    }
    HIP_TRY(hipDeviceSynchronize());
    // folded columns go to the end of the segment list (the kernel handles them as one group)
    {
        std::vector<DevSeg> plain, acc;
        for (int i = 0; i < dp->n_segs; ++i) (dp->segs[i].kind == SEG_ROWS_ACC ? acc : plain).push_back(dp->segs[i]);
        int k = 0;
        for (const DevSeg& g : plain) dp->segs[k++] = g;
        for (const DevSeg& g : acc) dp->segs[k++] = g;
        dp->n_acc = (int)acc.size();
        h->n_acc_folded = dp->n_acc;
    }
    op.W = wcopy + (lo - lo0);
    op.src_off = lo;
    op.K = hi - lo;
    op.acc_init = 1;
    return SPRK_OK;
}

// Dynamic-range guard for a STATIC split-f16 scale (one power of two per table from max |x|): true when more than 1 in
// 1024 of the non-zero entries lie over 2^20 below the maximum -- their lo halves would be f16 subnormals and the entries
// would carry fewer than ~20 significand bits (an outlier row next to ordinary ones).  The caller then keeps the f32 MFMA
// variant of the same kernel.  SPRK_HALF_RANGE_GUARD=0 switches the check off (for the test that shows why it is there).
int wide_dynamic_range(const float* rows, long long nrows, int row_floats, int ncols, float mx, bool* wide) {
    *wide = false;
    const char* g = getenv("SPRK_HALF_RANGE_GUARD");
    if ((g && g[0] == '0') || !(mx > 0.f) || nrows <= 0) return SPRK_OK;
    unsigned long long* d_cnt = nullptr;
    HIP_TRY(hipMalloc((void**)&d_cnt, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d_cnt, 0, 2 * sizeof(unsigned long long)));
    long long blocks = (nrows * ncols + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_v2_count_small, dim3((unsigned)blocks), dim3(256), 0, 0, rows, nrows, row_floats, ncols, ldexpf(mx, -20), d_cnt);
    HIP_TRY(hipGetLastError());
    unsigned long long cnt[2] = {0, 0};
    HIP_TRY(hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
    (void)hipFree(d_cnt);
    *wide = cnt[0] * 1024ull > cnt[1];
    return SPRK_OK;
}

// A Dense layer's W^T [N][ld] (K columns) as split-f16 A fragments for the per-sample dynamic-scale path (dyn_split.h):
// static power-of-two scale putting max |W| in [2^14, 2^15).  *frag stays NULL when switched off (SPRK_DYN_F16=0), when
// the shape does not tile (N % 16, K % 32) or the weights are not finite.
int make_dyn_fragments(sprk_engine* h, const float* W, int ld, int N, int K, float** frag, float* w_scale_out) {
    *frag = nullptr;
    const char* dsw = getenv("SPRK_DYN_F16");                // A/B switch: "0" = f32 MFMA
    if ((dsw && dsw[0] == '0') || (N & 15) || (K & 31)) return SPRK_OK;
    unsigned* d_max = nullptr;
    HIP_TRY(hipMalloc((void**)&d_max, sizeof(unsigned)));
    HIP_TRY(hipMemset(d_max, 0, sizeof(unsigned)));
    hipLaunchKernelGGL(k_v2_absmax, dim3(8), dim3(256), 0, 0, W, (long long)N, ld, K, d_max);
    unsigned bits = 0;
    HIP_TRY(hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
    (void)hipFree(d_max);
    float mx;
    memcpy(&mx, &bits, sizeof(mx));
    if (!(mx < 3.0e38f)) return SPRK_OK;
    bool wide = false;
    if (int rcw = wide_dynamic_range(W, (long long)N, ld, K, mx, &wide)) return rcw;
    if (wide) return SPRK_OK;
    int e = 0;
    float w_scale = 1.f;
    if (mx > 0.f) { (void)frexpf(mx, &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; w_scale = ldexpf(1.f, e); }
    const size_t frag_floats = (size_t)(N / 16) * (K / 32) * 512;
    float* f = nullptr;
    HIP_TRY(hipMalloc((void**)&f, frag_floats * sizeof(float)));
    h->fold_bufs.push_back(f);
    hipLaunchKernelGGL(k_dyn_pack_w, dim3(32), dim3(256), 0, 0, W, ld, N, K, w_scale, reinterpret_cast<_Float16*>(f));
    HIP_TRY(hipGetLastError());
    *frag = f;
    *w_scale_out = w_scale;
    return SPRK_OK;
}

// ---- dispatch table for k_deepfm_pairs<NF, NV, H0C, H1C, WAVES, DYN, SEP> ----
constexpr int V1_WAVES = 8;
constexpr int V1_ONE_MAX_TASKS = 16384;       // one-task-per-wave shape (k_deepfm_pairs1) up to B = 262 144
typedef void (*V1LaunchFn)(const V1Run&, const int*, const float*, float*, int, int*, int, hipStream_t);
typedef void (*V1LaunchManyFn)(const V1Run&, const V1Many&, int, int*, int, hipStream_t);
template <int NF, int NV, bool SEP>
void v1_launch(const V1Run& a, const int* ids, const float* dense, float* out, int B, int* err, int grid, hipStream_t st) {
    const size_t lds = V1Lds<4, 4, (NV + 3) / 4>::bytes;
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, true, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, ids, dense, out, B, err);
    else
        hipLaunchKernelGGL((k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, false, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, ids, dense, out, B, err);
}
// one task per wave (narrow rows, split-f16 form only): grid = ceil(tasks / waves), no cap
template <int NF, int NV, bool SEP>
void v1_launch_one(const V1Run& a, const int* ids, const float* dense, float* out, int B, int* err, int grid, hipStream_t st) {
    if constexpr (NV <= 4) {
        const size_t lds = V1Lds<4, 4, 1>::bytes;
        hipLaunchKernelGGL((k_deepfm_pairs1<NF, NV, 4, 4, V1_WAVES, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, ids, dense, out, B, err);
    }
}
template <int NF, int NV, bool SEP>
void v1_launch_many(const V1Run& a, const V1Many& m, int B, int* err, int grid, hipStream_t st) {
    const size_t lds = V1Lds<4, 4, (NV + 3) / 4>::bytes;
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, true, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, m, B, err);
    else
        hipLaunchKernelGGL((k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, false, SEP>), dim3(grid), dim3(V1_WAVES * 64), lds, st, a, m, B, err);
}
template <int NF, int NV, bool SEP>
int v1_prepare(const V1Run& r, float* img) {
    constexpr int PC = (NV + 3) / 4;
    hipLaunchKernelGGL((k_v1_pack_image<4, 4, PC>), dim3(1), dim3(256), 0, 0, r, img);
    HIP_TRY(hipGetLastError());
    const size_t lds = V1Lds<4, 4, PC>::bytes;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, true, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs<NF, NV, 4, 4, V1_WAVES, false, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, true, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs_many<NF, NV, 4, 4, V1_WAVES, false, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if constexpr (NV <= 4)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_deepfm_pairs1<NF, NV, 4, 4, V1_WAVES, SEP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return SPRK_OK;
}
struct V1Variant { int nf, nv; bool sep; V1LaunchFn launch; V1LaunchFn launch_one; V1LaunchManyFn launch_many; int (*prepare)(const V1Run&, float*); size_t lds_bytes; };
#define V1_VARIANT(NF, NV, SEP) {NF, NV, SEP, &v1_launch<NF, NV, SEP>, &v1_launch_one<NF, NV, SEP>, &v1_launch_many<NF, NV, SEP>, &v1_prepare<NF, NV, SEP>, V1Lds<4, 4, (NV + 3) / 4>::bytes}
#define V1_BOTH(NF, NV) V1_VARIANT(NF, NV, true), V1_VARIANT(NF, NV, false)
const V1Variant kV1Variants[] = {
    V1_BOTH(6, 4),    // BASELINE config 2: 6 fields, emb_dim 16, deep 64-64 (sep = the deep part's own movieId / userId tables, DeepFM.py:106)
    V1_BOTH(4, 3),    // the reference's own DeepFM.py: 4 fields, emb_dim 10 (rows padded to 12)
    V1_BOTH(4, 4),
    V1_BOTH(4, 16),   // BASELINE config 4: emb_dim 64 -- 256-byte rows gathered whole (four pieces per lane)
};

// Recognise the plan models.DeepFM emits (DeepFM.py graph: pair dots + first order + 2-layer deep part) and set up
// k_deepfm_pairs for it.  Leaves v1_variant = -1 (tile interpreter) for any other shape.
int setup_deepfm_pairs(sprk_engine* h) {
    const char* sw = getenv("SPRK_V1_CHAIN");                 // A/B switch: "0" = tile interpreter
    if (sw && sw[0] == '0') return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (p.model_kind != SPRK_MODEL_DEEPFM || p.din.enabled || p.n_ops != 3 || p.n_taps != 3 || p.n_pairs < 1) return SPRK_OK;
    const sprk_op &od = p.ops[0], &o0 = p.ops[1], &o1 = p.ops[2];
    if (od.kind != SPRK_OP_PAIR_DOT || od.src_buf != 0 || od.dst_buf != 0) return SPRK_OK;
    if (o0.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_RELU || o0.src_buf != 0 || o0.dst_buf != 1 || o0.dst_off != 0 || o0.N != 64) return SPRK_OK;
    if (o1.kind != SPRK_OP_DENSE || o1.act != SPRK_ACT_RELU || o1.src_buf != 1 || o1.src_off != 0 || o1.K != o0.N || o1.dst_off != 0 || o1.N != 64) return SPRK_OK;
    V1Run r;
    memset(&r, 0, sizeof(r));
    int row_dst[V1_MAX_FIELDS], nf = 0, Dp = 0, num_dst = -1, scal_dst[V1_MAX_FIELDS], ns = 0, scal_col[V1_MAX_FIELDS], scal_vocab[V1_MAX_FIELDS];
    const float* scal_tab[V1_MAX_FIELDS];
    // the deep part's OWN tables (models.DeepFM without share_deep_tables; DeepFM.py:106): a ROWS segment that lands inside deep0's
    // input slice while ANOTHER ROWS segment of the same ids column lands outside it (the FM part's table of that key)
    const int ds0 = o0.src_off, ds1 = o0.src_off + o0.K;
    int dsep_col[V1_MAX_DEEP], dsep_vocab[V1_MAX_DEEP], dsep_dst[V1_MAX_DEEP], n_dsep = 0;
    const float* dsep_tab[V1_MAX_DEEP];
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& sg = p.segs[i];
        if (sg.kind == SPRK_SEG_ROWS) {
            if (nf == 0 && n_dsep == 0) Dp = sg.row_stride;
            if (sg.row_stride != Dp || sg.count * 4 != Dp || Dp > 64) return SPRK_OK;
            if (h->slot_bytes[sg.slot] < ((size_t)sg.vocab + 1) * Dp * sizeof(float)) return SPRK_OK;   // needs the zero row at index vocab
            bool twin_outside = false;
            for (int j = 0; j < p.n_segs; ++j)
                if (j != i && p.segs[j].kind == SPRK_SEG_ROWS && p.segs[j].field == sg.field &&
                    !(p.segs[j].dst >= ds0 && p.segs[j].dst + p.segs[j].row_stride <= ds1)) twin_outside = true;
            if (twin_outside && sg.dst >= ds0 && sg.dst + Dp <= ds1) {
                if (n_dsep == V1_MAX_DEEP) return SPRK_OK;
                dsep_col[n_dsep] = sg.field; dsep_vocab[n_dsep] = sg.vocab; dsep_tab[n_dsep] = (const float*)h->slot_ptr[sg.slot];
                dsep_dst[n_dsep++] = sg.dst;
                continue;
            }
            if (nf == V1_MAX_FIELDS) return SPRK_OK;
            r.col[nf] = sg.field; r.vocab[nf] = sg.vocab; r.table[nf] = (const float*)h->slot_ptr[sg.slot];
            row_dst[nf++] = sg.dst;
        } else if (sg.kind == SPRK_SEG_SCALAR) {
            if (ns == V1_MAX_FIELDS) return SPRK_OK;
            if (h->slot_bytes[sg.slot] < ((size_t)sg.vocab + 1) * sizeof(float)) return SPRK_OK;
            scal_col[ns] = sg.field; scal_vocab[ns] = sg.vocab; scal_tab[ns] = (const float*)h->slot_ptr[sg.slot]; scal_dst[ns++] = sg.dst;
        } else if (sg.kind == SPRK_SEG_DENSE) {
            if (num_dst >= 0 || sg.field != 0 || sg.count > 8) return SPRK_OK;
            num_dst = sg.dst; r.n_num = sg.count;
        } else if (sg.kind != SPRK_SEG_ZERO) {
            return SPRK_OK;
        }
    }
    if (nf < 2 || ns != nf || num_dst < 0 || r.n_num < 1) return SPRK_OK;
    for (int f = 0; f < nf; ++f) {                            // first-order table of the same ids column
        int hit = -1;
        for (int i = 0; i < ns; ++i) if (scal_col[i] == r.col[f] && scal_vocab[i] == r.vocab[f]) hit = i;
        if (hit < 0) return SPRK_OK;
        r.w1[f] = scal_tab[hit];
    }
    int smin = scal_dst[0];
    for (int i = 1; i < ns; ++i) if (scal_dst[i] < smin) smin = scal_dst[i];
    // taps: first order (all ones), pair dots (weights), deep output (weights)
    const sprk_tap *tf = nullptr, *tpair = nullptr, *tdeep = nullptr;
    for (int t = 0; t < 3; ++t) {
        const sprk_tap& tp = p.taps[t];
        if (tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
        if (tp.buf == 0 && tp.off == smin && tp.len == ns && tp.w_slot < 0) tf = &tp;
        else if (tp.buf == 0 && tp.off == od.dst_off && tp.len == p.n_pairs && tp.w_slot >= 0) tpair = &tp;
        else if (tp.buf == o1.dst_buf && tp.off == 0 && tp.len <= o1.N && tp.w_slot >= 0) tdeep = &tp;
    }
    if (!tf || !tpair || !tdeep) return SPRK_OK;
    for (int i = 0; i < ns; ++i) if (scal_dst[i] < smin || scal_dst[i] >= smin + ns) return SPRK_OK;
    // pairs -> (field a, field b) x head weight
    std::vector<float> pwh(p.n_pairs);
    HIP_TRY(hipMemcpy(pwh.data(), h->slot_ptr[tpair->w_slot], p.n_pairs * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < p.n_pairs; ++i) {
        int a = -1, b = -1;
        for (int f = 0; f < nf; ++f) { if (row_dst[f] == p.pair_a[i]) a = f; if (row_dst[f] == p.pair_b[i]) b = f; }
        if (a < 0 || b < 0 || a == b || od.K != Dp) return SPRK_OK;
        if (a > b) { const int t = a; a = b; b = t; }
        r.pw[a * V1_MAX_FIELDS + b] += pwh[i];
    }
    // deep part: the embedding columns inside deep0's input slice (at most V1_MAX_DEEP).  Tied tables: they are FM fields, which
    // become fields 0.. of the kernel.  Own tables (n_dsep > 0): every deep column must be one of them, and the FM fields with the
    // same ids columns become fields 0.. (the kernel looks deep row d up with field d's id).
    const int s0 = ds0, s1 = ds1;
    if (num_dst < s0 || num_dst + r.n_num > s1) return SPRK_OK;
    int order[V1_MAX_FIELDS], no = 0, deep_off[V1_MAX_DEEP] = {0, 0};
    const bool sep = n_dsep > 0;
    for (int f = 0; f < nf; ++f) {
        if (row_dst[f] >= s0 && row_dst[f] + Dp <= s1) {
            if (sep || r.n_deep == V1_MAX_DEEP) return SPRK_OK;
            deep_off[r.n_deep++] = row_dst[f] - s0;
            order[no++] = f;
        } else if (row_dst[f] < s1 && row_dst[f] + Dp > s0) {
            return SPRK_OK;
        }
    }
    if (sep) {
        for (int d = 0; d < n_dsep; ++d) {
            int twin = -1;
            for (int f = 0; f < nf; ++f) if (r.col[f] == dsep_col[d] && r.vocab[f] == dsep_vocab[d]) twin = f;
            if (twin < 0) return SPRK_OK;
            for (int i = 0; i < no; ++i) if (order[i] == twin) return SPRK_OK;
            deep_off[r.n_deep++] = dsep_dst[d] - s0;
            order[no++] = twin;
        }
    }
    for (int f = 0; f < nf; ++f) {
        bool deep = false;
        for (int i = 0; i < r.n_deep; ++i) deep |= order[i] == f;
        if (!deep) order[no++] = f;
    }
    {
        V1Run t = r;
        int inv[V1_MAX_FIELDS];
        for (int i = 0; i < nf; ++i) {
            const int f = order[i];
            inv[f] = i;
            t.col[i] = r.col[f]; t.vocab[i] = r.vocab[f]; t.table[i] = r.table[f]; t.w1[i] = r.w1[f];
        }
        memset(t.pw, 0, sizeof(t.pw));
        for (int a = 0; a < nf; ++a)
            for (int b = a + 1; b < nf; ++b) {
                const float w = r.pw[a * V1_MAX_FIELDS + b];
                if (w == 0.f) continue;
                int x = inv[a], y = inv[b];
                if (x > y) { const int tt = x; x = y; y = tt; }
                t.pw[x * V1_MAX_FIELDS + y] += w;
            }
        r = t;
    }
    r.sep = sep ? 1 : 0;
    for (int d = 0; d < n_dsep; ++d) r.table[nf + d] = dsep_tab[d];
    const float* const* deep_tables = sep ? &r.table[nf] : &r.table[0];   // tables deep0's embedding block reads
    int variant = -1;
    for (size_t v = 0; v < sizeof(kV1Variants) / sizeof(kV1Variants[0]); ++v)
        if (kV1Variants[v].nf == nf && kV1Variants[v].nv == Dp / 4 && kV1Variants[v].sep == sep) variant = (int)v;
    if (variant < 0) return SPRK_OK;
    const int H0 = o0.N, H1 = o1.N;
    const float* W0 = (const float*)h->slot_ptr[o0.w_slot];
    const int PC = (Dp / 4 + 3) / 4;                          // 16-float chunks per embedding row
    const int KW = 16 * (V1_MAX_DEEP * PC + 1);
    float* w0p = nullptr;
    HIP_TRY(hipMalloc((void**)&w0p, (size_t)H0 * KW * sizeof(float) + 16));
    h->v1_bufs.push_back(w0p);
    hipLaunchKernelGGL(k_v1_pack_w0, dim3(1), dim3(256), 0, 0, W0, o0.ldw, r.n_deep, deep_off[0], deep_off[1], Dp, num_dst - s0, r.n_num,
                       H0, PC, w0p);
    HIP_TRY(hipGetLastError());
    float* hd = nullptr;
    HIP_TRY(hipMalloc((void**)&hd, (size_t)H1 * sizeof(float) + 16));
    h->v1_bufs.push_back(hd);
    HIP_TRY(hipMemset(hd, 0, (size_t)H1 * sizeof(float)));
    HIP_TRY(hipMemcpy(hd, h->slot_ptr[tdeep->w_slot], (size_t)tdeep->len * sizeof(float), hipMemcpyDeviceToDevice));
    HIP_TRY(hipDeviceSynchronize());
    r.F = p.n_id_cols; r.ND = p.n_dense; r.nf = nf; r.row_floats = Dp;
    r.w0 = w0p; r.b0 = (const float*)h->slot_ptr[o0.b_slot];
    r.W1 = (const float*)h->slot_ptr[o1.w_slot]; r.ld1 = o1.ldw; r.b1 = (const float*)h->slot_ptr[o1.b_slot];
    r.hdeep = hd; r.head_bias = p.head_bias;
    r.w1frag = nullptr; r.inv_w1_scale = 0.f; r.w0frag = nullptr; r.inv_w0_scale = 0.f;
    {
        // DYN: deep1's kernel and the embedding columns of deep0's (the first 32 of the packed 48) as split-f16 fragments
        float w_scale = 0.f, w0_scale = 0.f;
        float *frag = nullptr, *frag0 = nullptr;
        int rc2 = make_dyn_fragments(h, r.W1, r.ld1, H1, H0, &frag, &w_scale);
        if (rc2) return rc2;
        if (frag && (rc2 = make_dyn_fragments(h, w0p, KW, H0, 32 * PC, &frag0, &w0_scale))) return rc2;
        if (frag && frag0) { r.w1frag = frag; r.inv_w1_scale = 1.0f / w_scale; r.w0frag = frag0; r.inv_w0_scale = 1.0f / w0_scale; }
    }
    {
        float* img = nullptr;
        HIP_TRY(hipMalloc((void**)&img, kV1Variants[variant].lds_bytes));
        h->v1_bufs.push_back(img);
        { const int rc3 = kV1Variants[variant].prepare(r, img); if (rc3) return rc3; }
        HIP_TRY(hipDeviceSynchronize());
        r.image = img;
    }
    r.e_scale = 0.f; r.e_inv = 0.f;
    {
        // static scale for deep0's embedding block: max |E| over the deep fields' tables, unless a table has outlier rows
        const char* es = getenv("SPRK_V1_STATIC_SCALE");        // A/B switch: "0" = per-sample scale
        if (r.w0frag && !(es && es[0] == '0')) {
            unsigned* d_max = nullptr;
            HIP_TRY(hipMalloc((void**)&d_max, sizeof(unsigned)));
            HIP_TRY(hipMemset(d_max, 0, sizeof(unsigned)));
            bool wide = false;
            for (int f = 0; f < r.n_deep; ++f) {
                const long long rows = (long long)r.vocab[f] + 1;
                long long blocks = (rows * Dp + 255) / 256;
                if (blocks > 8192) blocks = 8192;
                hipLaunchKernelGGL(k_v2_absmax, dim3((unsigned)blocks), dim3(256), 0, 0, deep_tables[f], rows, Dp, Dp, d_max);
            }
            HIP_TRY(hipGetLastError());
            unsigned bits = 0;
            HIP_TRY(hipMemcpy(&bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
            (void)hipFree(d_max);
            float mx;
            memcpy(&mx, &bits, sizeof(mx));
            for (int f = 0; f < r.n_deep && !wide && mx > 0.f && mx < 3.0e38f; ++f)
                if (int rcw = wide_dynamic_range(deep_tables[f], (long long)r.vocab[f] + 1, Dp, Dp, mx, &wide)) return rcw;
            if (mx > 0.f && mx < 3.0e38f && !wide) {
                int e = 0;
                (void)frexpf(mx, &e);
                e = 15 - e;
                if (e > 60) e = 60;
                if (e < -60) e = -60;
                r.e_scale = ldexpf(1.f, e);
                r.e_inv = r.inv_w0_scale / r.e_scale;
            }
        }
    }
    r.tab = nullptr;
    const char* rt = getenv("SPRK_V1_ROWTAB");                // A/B switch: "0" = gather from the uploaded tables
    if (PC == 1 && !(rt && rt[0] == '0')) {
        // own deep tables: rows of <= 12 floats ride in their field's line (float 20..), wider ones get rows of their own
        const bool pack = sep && Dp <= 12;
        size_t rows = 0;
        for (int f = 0; f < nf; ++f) rows += (size_t)r.vocab[f] + 1;
        if (sep && !pack) for (int d = 0; d < r.n_deep; ++d) rows += (size_t)r.vocab[d] + 1;
        if (rows * 128 < ((size_t)1 << 32)) {                     // 32-bit byte offsets
            float* tab = nullptr;
            HIP_TRY(hipMalloc((void**)&tab, rows * 128));
            h->v1_bufs.push_back(tab);
            h->derived_bytes += rows * 128;
            size_t base = 0;
            for (int f = 0; f < nf + ((sep && !pack) ? r.n_deep : 0); ++f) {
                const bool deep_row = f >= nf;
                const long long n = (long long)r.vocab[deep_row ? f - nf : f] + 1;
                long long nb = (n * 32 + 255) / 256;
                if (nb > 65536) nb = 65536;
                hipLaunchKernelGGL(k_v1_build_rows, dim3((unsigned)nb), dim3(256), 0, 0, r.table[f], Dp, deep_row ? (const float*)nullptr : r.w1[f], n,
                                   tab + base * 32, (pack && f < r.n_deep) ? r.table[nf + f] : (const float*)nullptr);
                r.rowbase[f] = (unsigned)base;
                base += (size_t)n;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            r.tab = tab;
            r.pack = pack ? 1 : 0;
        }
    }
    {
        const char* one = getenv("SPRK_V1_ONE");               // A/B switch: "0" = looped kernel for one-batch launches too
        h->v1_one = r.tab && r.w1frag && PC == 1 && !(one && one[0] == '0');
    }
    h->v1_run = r;
    h->v1_variant = variant;
    return SPRK_OK;
}

// ---- k_mlp_chain<N0C, N1C, WAVES> ----
constexpr int MC_WAVES = 8;
// Recognise what the first-Dense fold left of an EmbeddingMLP / Wide&Deep plan (EmbeddingMLP.py:72-77, WideNDeep.py:99-107):
// folded columns, unfolded embedding columns + numerics feeding Dense(128) -> Dense(128) -> weighted tap (+ the wide cross).
int setup_mlp_chain(sprk_engine* h, DevPlan* dp) {
    const char* sw = getenv("SPRK_MLP_CHAIN");                // A/B switch: "0" = tile interpreter
    if (sw && sw[0] == '0') return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (p.din.enabled || dp->n_ops != 2 || dp->n_taps < 1 || dp->n_taps > 2 || dp->n_acc > MC_MAX_ACC) return SPRK_OK;
    if (p.model_kind != SPRK_MODEL_EMBEDDING_MLP && p.model_kind != SPRK_MODEL_WIDE_DEEP) return SPRK_OK;
    const DevOp &o0 = dp->ops[0], &o1 = dp->ops[1];
    if (o0.kind != SPRK_OP_DENSE || o1.kind != SPRK_OP_DENSE || o0.act != o1.act || (o0.act != SPRK_ACT_RELU && o0.act != SPRK_ACT_PRELU)) return SPRK_OK;
    if (o0.src_buf != 0 || o0.dst_buf == 0 || o0.dst_off != 0 || o1.src_buf != o0.dst_buf || o1.src_off != 0 || o1.K != o0.N ||
        o1.dst_off != 0 || o0.N != 128 || o1.N != 128) return SPRK_OK;
    MlpChainRun r;
    memset(&r, 0, sizeof(r));
    int col_off[MC_MAX_CHUNKS], col_w[MC_MAX_CHUNKS];
    const int lo = o0.src_off, hi = o0.src_off + o0.K;
    const int n_plain = dp->n_segs - dp->n_acc;
    const DevSeg* cross = nullptr;
    int num_dst = -1;
    for (int i = 0; i < n_plain; ++i) {
        const DevSeg& sg = dp->segs[i];
        if (sg.kind == SPRK_SEG_ROWS) {
            if (sg.dst < lo || sg.dst + 4 * sg.count > hi) return SPRK_OK;
            for (int j = 0; j < 4 * sg.count; j += 16) {
                if (r.n_chunks == MC_MAX_CHUNKS) return SPRK_OK;
                const int c = r.n_chunks++;
                r.ch_col[c] = h->idc[sg.field]; r.ch_vocab[c] = sg.vocab; r.ch_off[c] = j; r.ch_stride[c] = sg.row_stride;
                r.ch_width[c] = 4 * sg.count - j < 16 ? 4 * sg.count - j : 16; r.ch_tab[c] = sg.table;
                col_off[c] = sg.dst + j - lo; col_w[c] = r.ch_width[c];
            }
        } else if (sg.kind == SPRK_SEG_DENSE) {
            if (num_dst >= 0 || sg.field != 0 || sg.count > 8 || sg.dst < lo || sg.dst + sg.count > hi) return SPRK_OK;
            num_dst = sg.dst; r.n_num = sg.count;
        } else if (sg.kind == SPRK_SEG_CROSS_ROWS || sg.kind == SPRK_SEG_CROSS_SCALAR) {
            if (cross || (sg.dst < hi && sg.dst + (sg.kind == SPRK_SEG_CROSS_ROWS ? 4 * sg.count : 1) > lo)) return SPRK_OK;
            cross = &sg;
        } else if (sg.kind != SPRK_SEG_ZERO) {
            return SPRK_OK;
        }
    }
    // (an embedding table needs its zero row at index vocab for missing ids: models.pad_table provides it; ROWS segments
    //  validated at finalize only hold vocab rows' worth of bytes when a caller built the plan by hand)
    for (int i = 0; i < p.n_segs; ++i)
        if (p.segs[i].kind == SPRK_SEG_ROWS && h->slot_bytes[p.segs[i].slot] < ((size_t)p.segs[i].vocab + 1) * p.segs[i].row_stride * sizeof(float)) return SPRK_OK;
    if (num_dst >= 0) {
        if (r.n_chunks == MC_MAX_CHUNKS) return SPRK_OK;
        const int c = r.n_chunks++;
        r.ch_col[c] = -1; col_off[c] = num_dst - lo; col_w[c] = r.n_num;
    }
    if (r.n_chunks < 1) return SPRK_OK;
    const DevTap *tdeep = nullptr, *twide = nullptr;
    for (int t = 0; t < dp->n_taps; ++t) {
        const DevTap& tp = dp->taps[t];
        if (tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
        if (tp.buf == o1.dst_buf && tp.off == 0 && tp.len <= o1.N && tp.w && !tdeep) tdeep = &tp;
        else if (cross && tp.buf == 0 && tp.off == cross->dst && !twide) twide = &tp;
        else return SPRK_OK;
    }
    if (!tdeep || (cross != nullptr) != (twide != nullptr)) return SPRK_OK;
    if (cross) {
        if (cross->kind == SPRK_SEG_CROSS_ROWS) {
            if (twide->len != 4 * cross->count || !twide->w || twide->len > 32) return SPRK_OK;
            r.wide_kind = 1; r.wide_dim = twide->len; r.wide_stride = cross->row_stride; r.wide_w = twide->w;
        } else {
            if (twide->len != 1 || twide->w) return SPRK_OK;
            r.wide_kind = 2;
        }
        r.wide_a = h->idc[cross->field]; r.wide_b = h->idc[cross->field2]; r.wide_buckets = cross->vocab; r.wide_tab = cross->table;
    }
    r.n_acc = dp->n_acc;
    for (int g = 0; g < dp->n_acc; ++g) {
        const DevSeg& sg = dp->segs[n_plain + g];
        r.acc_col[g] = h->idc[sg.field]; r.acc_vocab[g] = sg.vocab; r.acc_tab[g] = sg.table;
    }
    r.F = p.n_id_cols; r.ND = p.n_dense; r.head_bias = dp->head_bias;
    typedef MlpChainLds<8, 8> LD;
    HIP_TRY(hipMalloc((void**)&h->mlp_image, LD::bytes));
    int *d_off = nullptr, *d_w = nullptr;
    HIP_TRY(hipMalloc((void**)&d_off, sizeof(col_off)));
    HIP_TRY(hipMalloc((void**)&d_w, sizeof(col_w)));
    HIP_TRY(hipMemcpy(d_off, col_off, sizeof(col_off), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_w, col_w, sizeof(col_w), hipMemcpyHostToDevice));
    float* w1frag = nullptr;
    {
        float w_scale = 0.f;
        const int rc2 = make_dyn_fragments(h, o1.W, o1.ldw, o1.N, o1.K, &w1frag, &w_scale);
        if (rc2) return rc2;
        r.inv_w1_scale = w1frag ? 1.0f / w_scale : 0.f;
    }
    hipLaunchKernelGGL((k_mlp_chain_pack<8, 8>), dim3(1), dim3(256), 0, 0, o0.W, o0.ldw, r.n_chunks, d_off, d_w, o0.bias,
                       o0.act == SPRK_ACT_PRELU ? o0.alpha : nullptr, o1.W, o1.ldw, o1.bias, o1.act == SPRK_ACT_PRELU ? o1.alpha : nullptr,
                       tdeep->w, tdeep->len, w1frag, h->mlp_image);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(d_off); (void)hipFree(d_w);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_chain<8, 8, MC_WAVES, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LD::bytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_chain<8, 8, MC_WAVES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LD::bytes));
    h->mlp_run = r;
    h->mlp_variant = 0;
    return SPRK_OK;
}

// ---- k_mlp_rows<8, 8, NBIG, WAVES, DYN> ----
constexpr int MR_WAVES = 8;
template <int NBIG>
void mlp_rows_launch(const MlpRowsRun& a, const int* ids, const float* dense, float* out, int B, int* err, const float* image, int grid,
                     size_t lds, hipStream_t st) {
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_mlp_rows<8, 8, NBIG, MR_WAVES, true>), dim3(grid), dim3(MR_WAVES * 64), lds, st, a, ids, dense, out, B, err, image);
    else
        hipLaunchKernelGGL((k_mlp_rows<8, 8, NBIG, MR_WAVES, false>), dim3(grid), dim3(MR_WAVES * 64), lds, st, a, ids, dense, out, B, err, image);
}
template <int NBIG>
int mlp_rows_attr(size_t lds) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_rows<8, 8, NBIG, MR_WAVES, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_rows<8, 8, NBIG, MR_WAVES, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return SPRK_OK;
}
// Recognise an EmbeddingMLP / Wide&Deep plan (EmbeddingMLP.py:72-77, WideNDeep.py:99-107) with ReLU layers of 128 and fold
// EVERY embedding column through the first Dense layer (see k_mlp_rows.h).  Leaves mlp_rows_nbig = -1 for any other shape.
int setup_mlp_rows(sprk_engine* h) {
    const char* sw = getenv("SPRK_MLP_ROWS");                 // A/B switch: "0" = round-1 k_mlp_chain
    if (sw && sw[0] == '0') return SPRK_OK;
    sw = getenv("SPRK_MLP_CHAIN");                            // "0" = tile interpreter (switches both chain kernels off)
    if (sw && sw[0] == '0') return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (p.din.enabled || p.n_ops != 2 || p.n_taps < 1 || p.n_taps > 2) return SPRK_OK;
    if (p.model_kind != SPRK_MODEL_EMBEDDING_MLP && p.model_kind != SPRK_MODEL_WIDE_DEEP) return SPRK_OK;
    if (p.n_id_cols > 12 || p.n_dense > 8 || p.n_dense < 1) return SPRK_OK;
    const sprk_op &o0 = p.ops[0], &o1 = p.ops[1];
    if (o0.kind != SPRK_OP_DENSE || o1.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_RELU || o1.act != SPRK_ACT_RELU) return SPRK_OK;
    if (o0.src_buf != 0 || o0.dst_buf == 0 || o0.dst_off != 0 || o1.src_buf != o0.dst_buf || o1.src_off != 0 || o1.K != o0.N ||
        o1.dst_off != 0 || o0.N != 128 || o1.N != 128) return SPRK_OK;
    typedef MlpRowsLds<8, 8> LD;
    const int lo = o0.src_off, hi = o0.src_off + o0.K, N0 = 128;
    MlpRowsRun r;
    memset(&r, 0, sizeof(r));
    const sprk_seg* big_seg[MR_MAX_BIG];
    const sprk_seg* small_seg[MR_MAX_SMALL];
    const sprk_seg* cross = nullptr;
    int num_dst = -1;
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& sg = p.segs[i];
        if (sg.kind == SPRK_SEG_ROWS) {
            if (sg.dst < lo || sg.dst + 4 * sg.count > hi) return SPRK_OK;
            if ((long long)sg.vocab <= 31 && r.n_small < MR_MAX_SMALL) small_seg[r.n_small++] = &sg;
            else if (r.n_big < MR_MAX_BIG) big_seg[r.n_big++] = &sg;
            else return SPRK_OK;
            if ((size_t)sg.vocab * N0 * sizeof(float) > ((size_t)8 << 30)) return SPRK_OK;
        } else if (sg.kind == SPRK_SEG_DENSE) {
            if (num_dst >= 0 || sg.field != 0 || sg.count > 8 || sg.dst < lo || sg.dst + sg.count > hi) return SPRK_OK;
            num_dst = sg.dst; r.n_num = sg.count;
        } else if (sg.kind == SPRK_SEG_CROSS_ROWS || sg.kind == SPRK_SEG_CROSS_SCALAR) {
            if (cross || (sg.dst < hi && sg.dst + (sg.kind == SPRK_SEG_CROSS_ROWS ? 4 * sg.count : 1) > lo)) return SPRK_OK;
            cross = &sg;
        } else if (sg.kind != SPRK_SEG_ZERO) {
            return SPRK_OK;
        }
    }
    if (r.n_big < 1 || r.n_big > 2 || num_dst < 0 || r.n_num < 1) return SPRK_OK;   // (three big columns spill: 3 x 8 float4 in flight)
    const sprk_tap *tdeep = nullptr, *twide = nullptr;
    for (int t = 0; t < p.n_taps; ++t) {
        const sprk_tap& tp = p.taps[t];
        if (tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
        if (tp.buf == o1.dst_buf && tp.off == 0 && tp.len <= o1.N && tp.w_slot >= 0 && !tdeep) tdeep = &tp;
        else if (cross && tp.buf == 0 && tp.off == cross->dst && !twide) twide = &tp;
        else return SPRK_OK;
    }
    if (!tdeep || (cross != nullptr) != (twide != nullptr)) return SPRK_OK;
    if (cross) {
        if (cross->kind == SPRK_SEG_CROSS_ROWS) {
            if (twide->len != 4 * cross->count || twide->w_slot < 0 || twide->len > 32) return SPRK_OK;
            r.wide_kind = 1; r.wide_dim = twide->len; r.wide_stride = cross->row_stride; r.wide_w = (const float*)h->slot_ptr[twide->w_slot];
        } else {
            if (twide->len != 1 || twide->w_slot >= 0) return SPRK_OK;
            r.wide_kind = 2;
        }
        r.wide_a = cross->field; r.wide_b = cross->field2; r.wide_buckets = cross->vocab; r.wide_tab = (const float*)h->slot_ptr[cross->slot];
    }
    // LDS: fixed image + small tables (+ one shared zero row) + a staging slot per wave
    size_t small_floats = 0;
    for (int f = 0; f < r.n_small; ++f) { r.s_off[f] = (int)small_floats; small_floats += (size_t)small_seg[f]->vocab * N0; }
    r.zero_off = (int)small_floats;
    small_floats += N0;
    small_floats = (small_floats + 255) & ~(size_t)255;
    const size_t lds = ((size_t)LD::total_pad + small_floats + (size_t)MR_WAVES * MR_STAGE) * sizeof(float);
    if (lds > 160 * 1024) return SPRK_OK;
    const float* W0 = (const float*)h->slot_ptr[o0.w_slot];
    HIP_TRY(hipMalloc((void**)&h->mlp_rows_small, small_floats * sizeof(float)));
    HIP_TRY(hipMemset(h->mlp_rows_small, 0, small_floats * sizeof(float)));
    auto fold = [&](const sprk_seg& sg, float* F) {
        long long blocks = ((long long)sg.vocab * N0 + 255) / 256;
        if (blocks > 65536) blocks = 65536;
        hipLaunchKernelGGL(k_fold_dense_rows, dim3((unsigned)blocks), dim3(256), 0, 0, (const float*)h->slot_ptr[sg.slot], (long long)sg.vocab,
                           sg.row_stride, 4 * sg.count, W0, o0.ldw, sg.dst - lo, N0, F);
    };
    {
        // small columns: fold into a scratch buffer, then into the XOR-swizzled LDS layout (k_mlp_rows.h); s_off must keep the
        // low 7 bits of a row's float offset free for the swizzle
        float* tmp = nullptr;
        HIP_TRY(hipMalloc((void**)&tmp, (size_t)32 * N0 * sizeof(float)));
        for (int f = 0; f < r.n_small; ++f) {
            r.s_col[f] = small_seg[f]->field; r.s_vocab[f] = small_seg[f]->vocab;
            if (r.s_off[f] & 127) { (void)hipFree(tmp); return fail(SPRK_EINVAL, "small-table offset not a multiple of 128 floats"); }
            fold(*small_seg[f], tmp);
            hipLaunchKernelGGL(k_mlp_rows_swizzle, dim3(4), dim3(256), 0, 0, tmp, h->mlp_rows_small + r.s_off[f], small_seg[f]->vocab);
        }
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(tmp);
    }
    for (int b = 0; b < r.n_big; ++b) {
        const sprk_seg& sg = *big_seg[b];
        float* F = nullptr;
        const size_t bytes = ((size_t)sg.vocab + 1) * N0 * sizeof(float);
        HIP_TRY(hipMalloc((void**)&F, bytes));
        h->mlp_rows_bufs.push_back(F);
        h->derived_bytes += bytes;
        HIP_TRY(hipMemset(F + (size_t)sg.vocab * N0, 0, N0 * sizeof(float)));        // the "no id" row
        fold(sg, F);
        r.big_col[b] = sg.field; r.big_vocab[b] = sg.vocab; r.big_tab[b] = F;
    }
    HIP_TRY(hipGetLastError());
    float* w1frag = nullptr;
    {
        float w_scale = 0.f;
        const int rc2 = make_dyn_fragments(h, (const float*)h->slot_ptr[o1.w_slot], o1.ldw, o1.N, o1.K, &w1frag, &w_scale);
        if (rc2) return rc2;
        r.inv_w1_scale = w1frag ? 1.0f / w_scale : 0.f;
    }
    HIP_TRY(hipMalloc((void**)&h->mlp_rows_image, (size_t)LD::total_pad * sizeof(float)));
    hipLaunchKernelGGL((k_mlp_rows_pack<8, 8>), dim3(1), dim3(256), 0, 0, W0, o0.ldw, num_dst - lo, r.n_num, (const float*)h->slot_ptr[o0.b_slot],
                       (const float*)h->slot_ptr[o1.w_slot], o1.ldw, (const float*)h->slot_ptr[o1.b_slot],
                       (const float*)h->slot_ptr[tdeep->w_slot], tdeep->len, w1frag, h->mlp_rows_image);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    r.F = p.n_id_cols; r.ND = p.n_dense; r.head_bias = p.head_bias;
    r.small = h->mlp_rows_small; r.small_floats = (int)small_floats;
    int rc;
    if (r.n_big == 1) rc = mlp_rows_attr<1>(lds);
    else rc = mlp_rows_attr<2>(lds);
    if (rc) return rc;
    h->mlp_rows_run = r;
    h->mlp_rows_lds = lds;
    h->mlp_rows_nbig = r.n_big;
    return SPRK_OK;
}

// ---- dispatch table for k_din_tail<N0C, N1C, KPC, WAVES> ----
constexpr int DT_WAVES = 8;
typedef void (*DinTailLaunchFn)(const DinTailRun&, const int*, const float*, const float*, float*, int, int*, const float*, int, hipStream_t);
typedef void (*DinTailLaunchManyFn)(const DinTailRun&, const DinTailMany&, int, int*, const float*, int, hipStream_t);
typedef void (*DinTailPackFn)(const float*, int, int, int, int, int, const float*, const float*, const float*, int, const float*,
                              const float*, const float*, int, const float*, float*);
template <int N0C, int N1C, int KPC>
void din_tail_launch(const DinTailRun& a, const int* ids, const float* dense, const float* aux, float* out, int B, int* err,
                     const float* image, int grid, hipStream_t st) {
    const size_t lds = DinTailLds<N0C, N1C, KPC>::bytes;
    static const DinTailMany none{};
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, true, false>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, ids, dense, aux, out, B, err, image, none);
    else
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, false, false>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, ids, dense, aux, out, B, err, image, none);
}
template <int N0C, int N1C, int KPC>
void din_tail_launch_many(const DinTailRun& a, const DinTailMany& m, int B, int* err, const float* image, int grid, hipStream_t st) {
    const size_t lds = DinTailLds<N0C, N1C, KPC>::bytes;
    if (a.inv_w1_scale != 0.f)
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, true, true>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, err, image, m);
    else
        hipLaunchKernelGGL((k_din_tail<N0C, N1C, KPC, DT_WAVES, false, true>), dim3(grid), dim3(DT_WAVES * 64), lds, st,
                           a, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, err, image, m);
}
template <int N0C, int N1C, int KPC>
void din_tail_pack(const float* W0, int ldw0, int p_off, int Dp, int n_off, int n_num, const float* b0, const float* a0,
                   const float* W1, int ldw1, const float* b1, const float* a1, const float* hw, int n_hw, const float* w1frag,
                   float* img) {
    hipLaunchKernelGGL((k_din_tail_pack<N0C, N1C, KPC>), dim3(1), dim3(256), 0, 0, W0, ldw0, p_off, Dp, n_off, n_num, b0, a0, W1, ldw1,
                       b1, a1, hw, n_hw, w1frag, img);
}
struct DinTailVariant {
    int n0c, n1c, kpc;
    const void* fn[4];                // [DYN][MB] instantiations
    size_t lds_bytes;
    DinTailLaunchFn launch;
    DinTailLaunchManyFn launch_many;
    DinTailPackFn pack;
};
#define DIN_TAIL_VARIANT(N0C, N1C, KPC) {N0C, N1C, KPC, {reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, false, false>), \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, false, true>),               \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, true, false>),               \
                                         reinterpret_cast<const void*>(&k_din_tail<N0C, N1C, KPC, DT_WAVES, true, true>)},               \
                                         DinTailLds<N0C, N1C, KPC>::bytes, &din_tail_launch<N0C, N1C, KPC>, &din_tail_launch_many<N0C, N1C, KPC>, \
                                         &din_tail_pack<N0C, N1C, KPC>}
const DinTailVariant kDinTailVariants[] = {
    DIN_TAIL_VARIANT(8, 4, 2),        // DIN.py:161-167 widths 128 / 64, emb_dim 17..32 (BASELINE config 3)
    DIN_TAIL_VARIANT(8, 4, 1),        // ... emb_dim <= 16 (the reference's own emb_dim 10)
    DIN_TAIL_VARIANT(4, 2, 2), DIN_TAIL_VARIANT(4, 2, 1),     // half-width tails (64 / 32)
};

// Recognise the DIN tail the first-Dense fold left behind (every embedding column folded, fc0 reading only the
// pooled history + numerics, two PReLU Dense layers, one weighted tap) and set up k_din_tail for it.
int setup_din_tail(sprk_engine* h, DevPlan* dp) {
    const char* sw = getenv("SPRK_DIN_TAIL");                // A/B switch: "0" = tile interpreter
    if (sw && sw[0] == '0') return SPRK_OK;
    const sprk_plan& p = h->plan;
    if (!p.din.enabled || (p.model_kind != SPRK_MODEL_DIN && p.model_kind != SPRK_MODEL_DIEN) || dp->n_ops != 2 || dp->n_taps != 1) return SPRK_OK;
    if (dp->n_acc < 1 || dp->n_acc > DT_MAX_COLS) return SPRK_OK;
    const DevOp &o0 = dp->ops[0], &o1 = dp->ops[1];
    if (o0.kind != SPRK_OP_DENSE || o1.kind != SPRK_OP_DENSE || o0.act != SPRK_ACT_PRELU || o1.act != SPRK_ACT_PRELU) return SPRK_OK;
    if (!o0.acc_init || o0.src_buf != 0 || o0.dst_off != 0 || o1.src_buf != o0.dst_buf || o1.src_off != 0 || o1.K != o0.N ||
        o1.dst_off != 0) return SPRK_OK;
    const DevTap& tp = dp->taps[0];
    if (tp.buf != o1.dst_buf || tp.off != 0 || tp.len > o1.N || !tp.w || tp.scale != 1.0f || tp.bias != 0.0f) return SPRK_OK;
    int aux_dst = -1, num_dst = -1, n_num = 0, Dp = 0;
    const int n_plain = dp->n_segs - dp->n_acc;
    for (int i = 0; i < n_plain; ++i) {
        const DevSeg& sg = dp->segs[i];
        if (sg.kind == SPRK_SEG_AUX && aux_dst < 0 && sg.field == 0) { aux_dst = sg.dst; Dp = sg.count; }
        else if (sg.kind == SPRK_SEG_DENSE && num_dst < 0 && sg.field == 0) { num_dst = sg.dst; n_num = sg.count; }
        else if (sg.kind != SPRK_SEG_ZERO) return SPRK_OK;     // an unfolded gather remains: leave it to the interpreter
    }
    if (aux_dst < 0 || num_dst < 0 || Dp != p.n_aux || n_num < 1 || n_num > 8) return SPRK_OK;
    const int p_off = aux_dst - o0.src_off, n_off = num_dst - o0.src_off;
    if (p_off < 0 || p_off + Dp > o0.K || n_off < 0 || n_off + n_num > o0.K) return SPRK_OK;
    const int n0c = o0.N / 16, n1c = o1.N / 16, kpc = (Dp + 15) / 16;
    int variant = -1;
    for (size_t v = 0; v < sizeof(kDinTailVariants) / sizeof(kDinTailVariants[0]); ++v)
        if (kDinTailVariants[v].n0c == n0c && kDinTailVariants[v].n1c == n1c && kDinTailVariants[v].kpc == kpc) variant = (int)v;
    if (variant < 0) return SPRK_OK;
    const DinTailVariant& tv = kDinTailVariants[variant];
    DinTailRun& r = h->din_tail_run;
    memset(&r, 0, sizeof(r));
    r.F = p.n_id_cols; r.ND = p.n_dense; r.NA = p.n_aux; r.n_cols = dp->n_acc; r.n_num = n_num; r.head_bias = dp->head_bias;
    for (int g = 0; g < dp->n_acc; ++g) {
        const DevSeg& sg = dp->segs[n_plain + g];
        r.col[g] = h->idc[sg.field]; r.vocab[g] = sg.vocab; r.Ftab[g] = sg.table;
    }
    HIP_TRY(hipMalloc((void**)&h->din_tail_image, tv.lds_bytes));
    // DYN: fc1's weights split into f16 hi / lo fragments with a static power-of-two scale
    float* w1frag = nullptr;
    {
        float w_scale = 0.f;
        const int rc2 = make_dyn_fragments(h, o1.W, o1.ldw, o1.N, o1.K, &w1frag, &w_scale);
        if (rc2) return rc2;
        if (w1frag) r.inv_w1_scale = 1.0f / w_scale;
    }
    tv.pack(o0.W, o0.ldw, p_off, Dp, n_off, n_num, o0.bias, o0.alpha, o1.W, o1.ldw, o1.bias, o1.alpha, tp.w, tp.len, w1frag, h->din_tail_image);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < 4; ++i) HIP_TRY(hipFuncSetAttribute(tv.fn[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)tv.lds_bytes));
    h->din_tail_variant = variant;
    return SPRK_OK;
}

int need_bytes(const sprk_engine* h, int slot, size_t bytes, const char* what) {
    if (!h->slot_ptr[slot]) return fail(SPRK_ESTATE, "%s: slot %d was never uploaded", what, slot);
    if (h->slot_bytes[slot] < bytes) return fail(SPRK_EINVAL, "%s: slot %d holds %zu bytes, needs %zu", what, slot, h->slot_bytes[slot], bytes);
    return SPRK_OK;
}

}  // namespace

extern "C" {

const char* sprk_last_error(void) { return g_err.c_str(); }

int sprk_runtime_info(int32_t info[4]) {
    if (!info) return fail(SPRK_EINVAL, "info is NULL");
    info[0] = SPRK_ABI_VERSION;
    info[1] = info[2] = info[3] = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    info[1] = n;
    if (n > 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            info[2] = prop.multiProcessorCount;
            info[3] = strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
        }
    }
    return SPRK_OK;
}

int sprk_create(const sprk_plan* plan, sprk_handle* out) {
    if (!plan || !out) return fail(SPRK_EINVAL, "plan/out is NULL");
    *out = nullptr;
    int rc = validate_plan(*plan);
    if (rc) return rc;
    sprk_engine* h = new (std::nothrow) sprk_engine();
    if (!h) return fail(SPRK_EHIP, "out of host memory");
    h->plan = *plan;
    h->slot_ptr.assign(plan->n_slots, nullptr);
    h->slot_bytes.assign(plan->n_slots, 0);
    int off = 0;
    for (int b = 0; b < plan->n_bufs; ++b) {
        h->buf_stride[b] = lds_stride(plan->buf_width[b]);
        h->buf_base[b] = off;
        off += SPRK_TILE_M * h->buf_stride[b];
    }
    // the tile's ids block [64][columns some gather segment reads]
    auto use_col = [&](int c) {
        for (int x : h->idc) if (x == c) return;
        h->idc.push_back(c);
    };
    for (int i = 0; i < plan->n_segs; ++i) {
        const sprk_seg& sg = plan->segs[i];
        if (sg.kind == SPRK_SEG_ROWS || sg.kind == SPRK_SEG_SCALAR || sg.kind == SPRK_SEG_CROSS_ROWS || sg.kind == SPRK_SEG_CROSS_SCALAR) use_col(sg.field);
        if (sg.kind == SPRK_SEG_CROSS_ROWS || sg.kind == SPRK_SEG_CROSS_SCALAR) use_col(sg.field2);
    }
    h->ids_base = off;
    off += (SPRK_TILE_M * (int)h->idc.size() + 3) & ~3;
    h->tile_lds_bytes = (size_t)off * sizeof(float);
    if (h->tile_lds_bytes > 160 * 1024) {
        size_t need = h->tile_lds_bytes;
        delete h;
        return fail(SPRK_EINVAL, "plan needs %zu bytes of LDS per tile (> 160 KiB)", need);
    }
    *out = h;
    return SPRK_OK;
}

int sprk_upload(sprk_handle h, int32_t slot, const void* src, size_t bytes) {
    if (!h || !src || bytes == 0) return fail(SPRK_EINVAL, "bad upload arguments");
    if (slot < 0 || slot >= h->plan.n_slots) return fail(SPRK_EINVAL, "slot %d outside [0,%d)", slot, h->plan.n_slots);
    if (h->finalized) return fail(SPRK_ESTATE, "upload after finalize");
    if (h->slot_ptr[slot]) { (void)hipFree(h->slot_ptr[slot]); h->slot_ptr[slot] = nullptr; }
    // 16 spare bytes so a float4 tail read of a [len]-float vector never leaves the allocation
    HIP_TRY(hipMalloc(&h->slot_ptr[slot], bytes + 16));
    HIP_TRY(hipMemset(h->slot_ptr[slot], 0, bytes + 16));
    HIP_TRY(hipMemcpy(h->slot_ptr[slot], src, bytes, hipMemcpyDefault));
    h->slot_bytes[slot] = bytes;
    return SPRK_OK;
}

int sprk_finalize(sprk_handle h) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (h->finalized) return SPRK_OK;
    const sprk_plan& p = h->plan;
    DevPlan* dp = new (std::nothrow) DevPlan();
    if (!dp) return fail(SPRK_EHIP, "out of host memory");
    memset(dp, 0, sizeof(DevPlan));
    struct Guard { DevPlan* p; ~Guard() { delete p; } } guard{dp};
    dp->F = p.n_id_cols; dp->ND = p.n_dense; dp->NA = p.n_aux;
    dp->n_segs = p.n_segs; dp->n_ops = p.n_ops; dp->n_taps = p.n_taps; dp->n_pairs = p.n_pairs; dp->n_bufs = p.n_bufs;
    dp->head_bias = p.head_bias;
    for (int b = 0; b < SPRK_MAX_BUFS; ++b) { dp->buf_stride[b] = h->buf_stride[b]; dp->buf_base[b] = h->buf_base[b]; }
    dp->ids_base = h->ids_base;
    dp->n_idc = (int)h->idc.size();
    for (size_t i = 0; i < h->idc.size(); ++i) dp->idc[i] = h->idc[i];
    auto compact = [&](int c) { for (size_t i = 0; i < h->idc.size(); ++i) if (h->idc[i] == c) return (int)i; return 0; };
    for (int i = 0; i < p.n_pairs; ++i) { dp->pair_a[i] = p.pair_a[i]; dp->pair_b[i] = p.pair_b[i]; }
    int rc;
    for (int i = 0; i < p.n_segs; ++i) {
        const sprk_seg& s = p.segs[i];
        DevSeg& d = dp->segs[i];
        d.kind = s.kind; d.field = s.field; d.field2 = s.field2; d.row_stride = s.row_stride; d.count = s.count; d.dst = s.dst; d.vocab = s.vocab;
        if (s.kind == SPRK_SEG_ROWS || s.kind == SPRK_SEG_SCALAR || s.kind == SPRK_SEG_CROSS_ROWS || s.kind == SPRK_SEG_CROSS_SCALAR) d.field = compact(s.field);
        if (s.kind == SPRK_SEG_CROSS_ROWS || s.kind == SPRK_SEG_CROSS_SCALAR) d.field2 = compact(s.field2);
        d.table = nullptr;
        if (s.kind == SPRK_SEG_ROWS || s.kind == SPRK_SEG_CROSS_ROWS) {
            if ((rc = need_bytes(h, s.slot, (size_t)s.vocab * s.row_stride * 4, "embedding table"))) return rc;
            d.table = (const float*)h->slot_ptr[s.slot];
        } else if (s.kind == SPRK_SEG_SCALAR || s.kind == SPRK_SEG_CROSS_SCALAR) {
            if ((rc = need_bytes(h, s.slot, (size_t)s.vocab * 4, "first-order table"))) return rc;
            d.table = (const float*)h->slot_ptr[s.slot];
        }
    }
    for (int i = 0; i < p.n_ops; ++i) {
        const sprk_op& o = p.ops[i];
        DevOp& d = dp->ops[i];
        d.kind = o.kind; d.src_buf = o.src_buf; d.src_off = o.src_off; d.K = o.K; d.dst_buf = o.dst_buf; d.dst_off = o.dst_off;
        d.N = o.N; d.ldw = o.ldw; d.act = o.act; d.groups = o.groups; d.group_stride = o.group_stride;
        if (o.kind == SPRK_OP_DENSE) {
            if ((rc = need_bytes(h, o.w_slot, (size_t)o.N * o.ldw * 4, "Dense kernel"))) return rc;
            if ((rc = need_bytes(h, o.b_slot, (size_t)o.N * 4, "Dense bias"))) return rc;
            d.W = (const float*)h->slot_ptr[o.w_slot];
            d.bias = (const float*)h->slot_ptr[o.b_slot];
            if (o.act == SPRK_ACT_PRELU) {
                if ((rc = need_bytes(h, o.alpha_slot, (size_t)o.N * 4, "PReLU alpha"))) return rc;
                d.alpha = (const float*)h->slot_ptr[o.alpha_slot];
            }
        }
    }
    for (int i = 0; i < p.n_taps; ++i) {
        const sprk_tap& t = p.taps[i];
        DevTap& d = dp->taps[i];
        d.buf = t.buf; d.off = t.off; d.len = t.len; d.scale = t.scale; d.bias = t.bias; d.w = nullptr;
        if (t.w_slot >= 0) {
            if ((rc = need_bytes(h, t.w_slot, (size_t)t.len * 4, "tap weights"))) return rc;
            d.w = (const float*)h->slot_ptr[t.w_slot];
        }
    }
    HIP_TRY(hipGetDevice(&h->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (p.din.enabled == 2) {
        const sprk_din& s = p.din;
        DevDin& d = dp->din;
        d.enabled = 1; d.T = s.T; d.hist_col = s.hist_col; d.cand_col = s.cand_col; d.row_stride = s.row_stride; d.vocab = s.vocab; d.hidden = s.hidden;
        const size_t img = s.emb_dim == 10 ? DienLayout<10, 32>::total_pad : DienLayout<16, 32>::total_pad;
        if ((rc = need_bytes(h, s.table_slot, (size_t)s.vocab * s.row_stride * 4, "DIEN table"))) return rc;
        if ((rc = need_bytes(h, s.seq_slot, img * 4, "DIEN sequence weights"))) return rc;
        d.table = (const float*)h->slot_ptr[s.table_slot];
        h->dien_run.T = s.T; h->dien_run.F = p.n_id_cols; h->dien_run.hist_col = s.hist_col; h->dien_run.cand_col = s.cand_col;
        h->dien_run.Dp = s.row_stride; h->dien_run.vocab = s.vocab; h->dien_run.NA = p.n_aux;
        h->dien_run.table = d.table;
        h->dien_run.image = (const float*)h->slot_ptr[s.seq_slot];
    } else if (p.din.enabled) {
        const sprk_din& s = p.din;
        DevDin& d = dp->din;
        d.enabled = 1; d.T = s.T; d.hist_col = s.hist_col; d.cand_col = s.cand_col; d.row_stride = s.row_stride; d.vocab = s.vocab; d.hidden = s.hidden; d.b2 = s.b2;
        if ((rc = need_bytes(h, s.table_slot, (size_t)s.vocab * s.row_stride * 4, "DIN table"))) return rc;
        if ((rc = need_bytes(h, s.w_slot, (size_t)s.hidden * 4 * s.row_stride * 4, "DIN att0 kernel"))) return rc;
        if ((rc = need_bytes(h, s.b_slot, (size_t)s.hidden * 4, "DIN att0 bias"))) return rc;
        if ((rc = need_bytes(h, s.alpha_slot, (size_t)s.T * s.hidden * 4, "DIN alpha"))) return rc;
        if ((rc = need_bytes(h, s.w2_slot, (size_t)s.hidden * 4, "DIN att1 kernel"))) return rc;
        d.table = (const float*)h->slot_ptr[s.table_slot];
        d.W = (const float*)h->slot_ptr[s.w_slot];
        d.bias = (const float*)h->slot_ptr[s.b_slot];
        d.alpha = (const float*)h->slot_ptr[s.alpha_slot];
        d.w2 = (const float*)h->slot_ptr[s.w2_slot];
        // samples per workgroup pass: about 256 (sample, slot) rows in LDS
        int ms = 256 / s.T;
        if (ms < 1) ms = 1;
        if (ms > 64) ms = 64;
        h->din_ms = ms;
        const int hs = s.row_stride + 4;
        h->din_lds_bytes = ((size_t)ms * s.T * hs + (size_t)ms * hs + (size_t)ms * s.T) * sizeof(float);
        if (h->din_lds_bytes > 160 * 1024) return fail(SPRK_EINVAL, "DIN stage needs %zu bytes of LDS", h->din_lds_bytes);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_din_pool), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->din_lds_bytes));
        int per_cu = (int)(160 * 1024 / h->din_lds_bytes);
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        h->din_grid_cap = h->num_cus * per_cu;
        // wave-per-sample kernel when the shape has an instantiation (T <= 64 rows fit the wave's LDS tile)
        const char* legacy = getenv("SPRK_DIN_LEGACY");          // A/B switch: "1" = generic k_din_pool
        const int kc = (s.row_stride + 15) / 16, hc = s.hidden / 16;
        const size_t vc_bytes = (size_t)s.vocab * s.hidden * sizeof(float);
        if (!(legacy && legacy[0] == '1') && s.T <= 64 && vc_bytes < ((size_t)4 << 30) &&
            (size_t)s.vocab * s.row_stride * sizeof(float) < ((size_t)4 << 30)) {   // 32-bit element offsets
            const char* hm = getenv("SPRK_DIN_HALF");            // A/B switch: "0" = f32 MFMA
            bool want_half = !(hm && hm[0] == '0');
            const char* wm = getenv("SPRK_DIN_WPB");             // A/B switch: at most this many waves per workgroup ("4" = round 1's)
            const int max_wpb = wm ? atoi(wm) : 12;
            for (size_t v = 0; v < sizeof(kDinVariants) / sizeof(kDinVariants[0]); ++v) {
                const DinVariant& dv = kDinVariants[v];
                if (dv.half != want_half) continue;
                if (dv.wpb > 4 && dv.wpb > max_wpb) continue;
                if (dv.wpb == 16 && s.T > 56) continue;
                if (dv.kc != kc || dv.hc != hc || dv.np * (64 / (dv.half ? kc * 4 : s.row_stride / 4)) < s.T) continue;
                const int KP = kc * 16;
                if (!h->din_w12) {
                    HIP_TRY(hipMalloc((void**)&h->din_w12, (size_t)s.hidden * KP * sizeof(float)));
                    HIP_TRY(hipMalloc((void**)&h->din_w4, (size_t)s.hidden * KP * sizeof(float)));
                    HIP_TRY(hipMalloc((void**)&h->din_vc, vc_bytes));
                    h->derived_bytes += vc_bytes;
                }
                hipLaunchKernelGGL(k_din_prep_w, dim3(8), dim3(256), 0, 0, d.W, s.hidden, s.row_stride, KP, 1.0f, h->din_w12, h->din_w4);
                HIP_TRY(hipGetLastError());
                float h_scale = 1.f, a_scale = 1.f;
                if (dv.half) {
                    // power-of-two scales from max|E|, max|W12|, max|W4|: |A_b| <= max|W12| + max|W4| max|E|
                    unsigned* d_max = nullptr;
                    HIP_TRY(hipMalloc((void**)&d_max, 3 * sizeof(unsigned)));
                    HIP_TRY(hipMemset(d_max, 0, 3 * sizeof(unsigned)));
                    long long nb_ = ((long long)s.vocab * s.row_stride + 255) / 256;
                    if (nb_ > 8192) nb_ = 8192;
                    hipLaunchKernelGGL(k_v2_absmax, dim3((unsigned)nb_), dim3(256), 0, 0, d.table, (long long)s.vocab, s.row_stride, s.row_stride, d_max);
                    hipLaunchKernelGGL(k_v2_absmax, dim3(4), dim3(256), 0, 0, h->din_w12, (long long)s.hidden, KP, KP, d_max + 1);
                    hipLaunchKernelGGL(k_v2_absmax, dim3(4), dim3(256), 0, 0, h->din_w4, (long long)s.hidden, KP, KP, d_max + 2);
                    HIP_TRY(hipGetLastError());
                    unsigned bits[3];
                    HIP_TRY(hipMemcpy(bits, d_max, sizeof(bits), hipMemcpyDeviceToHost));
                    (void)hipFree(d_max);
                    float mx[3];
                    memcpy(mx, bits, sizeof(mx));
                    if (!(mx[0] < 3.0e38f) || !(mx[1] < 3.0e38f) || !(mx[2] < 3.0e38f)) { want_half = false; v = (size_t)-1; continue; }   // NaN / Inf weights: rescan for the f32 kernel
                    {
                        bool wide = false;                      // outlier rows: the ordinary rows would lose their lo halves
                        if (int rcw = wide_dynamic_range(d.table, (long long)s.vocab, s.row_stride, s.row_stride, mx[0], &wide)) return rcw;
                        if (wide) { want_half = false; v = (size_t)-1; continue; }
                    }
                    const float bound_a = mx[1] + mx[2] * mx[0];
                    int e = 0;
                    if (mx[0] > 0.f) { (void)frexpf(mx[0], &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; h_scale = ldexpf(1.f, e); }
                    if (bound_a > 0.f) { (void)frexpf(bound_a, &e); e = 15 - e; if (e > 60) e = 60; if (e < -60) e = -60; a_scale = ldexpf(1.f, e); }
                    hipLaunchKernelGGL(k_din_prep_w, dim3(8), dim3(256), 0, 0, d.W, s.hidden, s.row_stride, KP, a_scale, h->din_w12, h->din_w4);
                    HIP_TRY(hipGetLastError());
                    if ((size_t)s.vocab * KP * sizeof(float) >= ((size_t)4 << 30)) { want_half = false; v = (size_t)-1; continue; }
                    if (!h->din_tsplit) { HIP_TRY(hipMalloc((void**)&h->din_tsplit, (size_t)s.vocab * KP * sizeof(float) + 16)); h->derived_bytes += (size_t)s.vocab * KP * sizeof(float); }
                    long long sb = ((long long)s.vocab * KP + 255) / 256;
                    if (sb > 65536) sb = 65536;
                    hipLaunchKernelGGL(k_din_split_table, dim3((unsigned)sb), dim3(256), 0, 0, d.table, (long long)s.vocab, s.row_stride, KP,
                                       h_scale, reinterpret_cast<_Float16*>(h->din_tsplit));
                    HIP_TRY(hipGetLastError());
                }
                long long blocks = ((long long)s.vocab * s.hidden + 255) / 256;
                if (blocks > 65536) blocks = 65536;
                hipLaunchKernelGGL(k_din_prep_vc, dim3((unsigned)blocks), dim3(256), 0, 0, d.W, d.bias, d.table, s.hidden,
                                   s.row_stride, (long long)s.vocab, h->din_vc);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipDeviceSynchronize());
                DinRun& r = h->din_run;
                r.T = s.T; r.F = p.n_id_cols; r.hist_col = s.hist_col; r.cand_col = s.cand_col; r.Dp = s.row_stride; r.vocab = s.vocab;
                r.h_scale = h_scale; r.acc_scale = a_scale * h_scale; r.unscale = 1.0f / (a_scale * h_scale);
                r.tsplit = h->din_tsplit; r.inv_h_scale = 1.0f / h_scale;
                r.b2 = s.b2; r.table = d.table; r.w12 = h->din_w12; r.w4 = h->din_w4; r.vc = h->din_vc; r.alpha = d.alpha; r.w2 = d.w2;
                HIP_TRY(hipFuncSetAttribute(dv.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dv.lds_bytes));
                if (dv.fn_many) HIP_TRY(hipFuncSetAttribute(dv.fn_many, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dv.lds_bytes));
                int wgs = (int)(160 * 1024 / dv.lds_bytes);
                if (wgs > 2) wgs = 2;                                // launch bounds: 2 waves per SIMD
                if (dv.wpb >= 12) wgs = 1;                           // ... or one 12- / 16-wave workgroup: 3 / 4 waves per SIMD
                if (wgs < 1) wgs = 1;
                h->din_attn_grid_cap = h->num_cus * wgs;
                h->din_wpb = dv.wpb;
                { const char* am = getenv("SPRK_DIN_ATTN_MB"); h->din_attn_many = !(am && am[0] == '0'); }
                h->din_attn_lds = dv.lds_bytes;
                h->din_variant = (int)v;
                break;
            }
        }
    }
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_forward), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->tile_lds_bytes));
    {
        int per_cu = (int)(160 * 1024 / (h->tile_lds_bytes ? h->tile_lds_bytes : 1));
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        h->tile_grid_cap = h->num_cus * per_cu;
    }
    {
        const char* force = getenv("SPRK_FORCE_INTERPRETER");
        if (!(force && force[0] == '1') && match_v2_chain(h)) {
            const V2Variant& vv = kV2Variants[h->v2_variant];
            HIP_TRY(hipFuncSetAttribute(vv.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vv.lds_bytes));
            int per_cu = (int)(160 * 1024 / vv.lds_bytes);
            if (vv.fn_trace) HIP_TRY(hipFuncSetAttribute(vv.fn_trace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)vv.lds_bytes));
            const int by_regs = (vv.reg ? 2 : 4) * 4 / V2_WAVES;     // workgroups/CU the launch bounds allow
            if (per_cu > by_regs) per_cu = by_regs;
            const char* wg = getenv("SPRK_V2_WGS_PER_CU");           // tuning knob (1..by_regs)
            if (wg && wg[0] >= '1' && wg[0] <= '9' && (wg[0] - '0') < per_cu) per_cu = wg[0] - '0';
            if (per_cu < 1) per_cu = 1;
            h->v2_grid_cap = h->num_cus * per_cu;
            { const char* gc = getenv("SPRK_V2_GRID_CAP"); if (gc && atoi(gc) > 0 && atoi(gc) < h->v2_grid_cap) h->v2_grid_cap = atoi(gc); }
            // first-order weight blocks back to back, so one gather instruction can serve several fields
            HIP_TRY(hipMalloc((void**)&h->v2_fo_all, h->v2_fo_floats * sizeof(float)));
            for (int g = 0; g < vv.g_emb; ++g)
                HIP_TRY(hipMemcpy(h->v2_fo_all + h->v2run.fo_off[g], h->v2.w1[g], ((size_t)h->v2run.vocab[g] + 1) * sizeof(float), hipMemcpyDeviceToDevice));
            h->v2run.fo_all = h->v2_fo_all;
            if (vv.fold) {
                const int KP = vv.kpc * 16;
                size_t rows_total = 0;
                for (int g = 0; g < vv.g_emb; ++g) { h->v2run.rowbase[g] = (unsigned)rows_total; rows_total += (size_t)h->v2run.vocab[g] + 1; }
                HIP_TRY(hipMalloc((void**)&h->v2_folded, rows_total * (KP + 16) * sizeof(float)));
                h->derived_bytes += rows_total * (KP + 16) * sizeof(float);
                for (int g = 0; g < vv.g_emb; ++g) {
                    const long long rows = (long long)h->v2run.vocab[g] + 1;
                    long long blocks = (rows + 3) / 4;
                    if (blocks > 65536) blocks = 65536;
                    hipLaunchKernelGGL(k_v2_fold, dim3((unsigned)blocks), dim3(256), 0, 0, h->v2.table[g], h->v2.ldp_emb,
                                       h->v2.Wp[g], h->v2.ldp_emb, h->v2.bp[g], h->v2.w1[g], h->v2.hfm, h->v2.n_hfm, h->v2.h0w,
                                       h->v2_folded + (size_t)h->v2run.rowbase[g] * (KP + 16), KP, rows);
                    HIP_TRY(hipGetLastError());
                }
                h->v2run.tab0 = h->v2_folded;
                HIP_TRY(hipDeviceSynchronize());
                if ((rc = setup_v2_joint(h))) return rc;
            }
            HIP_TRY(hipMalloc((void**)&h->v2_image, vv.lds_bytes));
            HIP_TRY(hipMemset(h->v2_image, 0, vv.lds_bytes));
            vv.pack(h->v2, h->v2_image);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    {
        const char* force = getenv("SPRK_FORCE_INTERPRETER");
        if (!(force && force[0] == '1') && h->v2_variant < 0) {
            if (h->rows_from_v2 && (rc = setup_rows_v2(h))) return rc;
            if (h->rows_variant < 0 && (rc = setup_rows_ncf(h))) return rc;
        }
    }
    const bool rows_on = h->rows_variant >= 0;
    if (!rows_on && h->v2_variant < 0 && (rc = setup_deepfm_pairs(h))) return rc;
    if (!rows_on && h->v2_variant < 0 && h->v1_variant < 0) {
        const char* force = getenv("SPRK_FORCE_INTERPRETER");
        if (!(force && force[0] == '1') && (rc = setup_mlp_rows(h))) return rc;
    }
    const bool mrows_on = h->mlp_rows_nbig >= 0;
    if (!mrows_on && !rows_on && h->v2_variant < 0 && h->v1_variant < 0 && (rc = fold_first_dense(h, dp))) return rc;
    if (!mrows_on && !rows_on && h->v2_variant < 0 && (rc = setup_din_tail(h, dp))) return rc;
    if (!mrows_on && !rows_on && h->v2_variant < 0 && h->v1_variant < 0 && h->din_tail_variant < 0 && (rc = setup_mlp_chain(h, dp))) return rc;
    HIP_TRY(hipMalloc((void**)&h->dev_plan, sizeof(DevPlan)));
    HIP_TRY(hipMemcpy(h->dev_plan, dp, sizeof(DevPlan), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void**)&h->dev_err, sizeof(int)));
    HIP_TRY(hipMemset(h->dev_err, 0, sizeof(int)));
    {
        // helper streams for sprk_forward_many's fan-out (sprk_set_many_streams; SPRK_MANY_STREAMS presets it)
        {
            HIP_TRY(hipEventCreateWithFlags(&h->many_fork, hipEventDisableTiming));
            for (int i = 0; i < 4; ++i) {
                HIP_TRY(hipStreamCreateWithFlags(&h->many_stream[i], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&h->many_join[i], hipEventDisableTiming));
            }
            const char* ms = getenv("SPRK_MANY_STREAMS");      // 0 / 1 = strict stream order (default), 2..4 = fan out
            int n = ms ? atoi(ms) : 0;
            h->many_streams = n < 2 ? 0 : (n > 4 ? 4 : n);
        }
    }
    { const char* xf = getenv("SPRK_V2_XFLAGS"); if (xf) { h->v2_xflags = atoi(xf); h->v2_xflags_set = true; } }
    h->finalized = true;
    return SPRK_OK;
}

size_t sprk_workspace_bytes(sprk_handle h, int32_t B) {
    if (!h || B <= 0 || !h->plan.din.enabled) return 0;
    return (size_t)B * h->plan.n_aux * sizeof(float);
}

static int launch_din(sprk_handle h, const int32_t* ids, float* pooled, float* att, int32_t B, hipStream_t st) {
    if (h->plan.din.enabled == 2) {                               // DIEN: GRU -> attention gate -> AUGRU, one lane per sample
        if (att) return fail(SPRK_EINVAL, "DIEN stage has no attention output");
        int grid = (B + 63) / 64;
        if (grid > h->num_cus * 8) grid = h->num_cus * 8;
        if (h->plan.din.emb_dim == 10)
            hipLaunchKernelGGL((k_dien_seq<10, 32>), dim3(grid), dim3(64), 0, st, h->dien_run, ids, pooled, B, h->dev_err);
        else
            hipLaunchKernelGGL((k_dien_seq<16, 32>), dim3(grid), dim3(64), 0, st, h->dien_run, ids, pooled, B, h->dev_err);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->din_variant >= 0) {
        int grid = (B + h->din_wpb - 1) / h->din_wpb;
        if (grid > h->din_attn_grid_cap) grid = h->din_attn_grid_cap;
        kDinVariants[h->din_variant].launch(h->din_run, ids, pooled, att, B, h->dev_err, grid, h->din_attn_lds, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    const int nchunks = (B + h->din_ms - 1) / h->din_ms;
    const int grid = nchunks < h->din_grid_cap ? nchunks : h->din_grid_cap;
    hipLaunchKernelGGL(k_din_pool, dim3(grid), dim3(256), h->din_lds_bytes, st, h->dev_plan, ids, pooled, att, B, h->din_ms, h->dev_err);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

int sprk_din_pool(sprk_handle h, const int32_t* ids, float* pooled, float* att, int32_t B, void* stream) {
    if (!h || !ids || !pooled) return fail(SPRK_EINVAL, "NULL argument");
    if (!h->finalized) return fail(SPRK_ESTATE, "din_pool before finalize");
    if (!h->plan.din.enabled) return fail(SPRK_EKIND, "handle has no DIN stage");
    if (B <= 0) return B == 0 ? SPRK_OK : fail(SPRK_EINVAL, "negative batch");
    return launch_din(h, ids, pooled, att, B, (hipStream_t)stream);
}

int sprk_forward(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B,
                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "forward before finalize");
    if (B < 0) return fail(SPRK_EINVAL, "negative batch");
    if (B == 0) return SPRK_OK;
    if (!out) return fail(SPRK_EINVAL, "out is NULL");
    if (h->plan.n_id_cols > 0 && !ids) return fail(SPRK_EINVAL, "ids is NULL");
    if (h->plan.n_dense > 0 && !dense) return fail(SPRK_EINVAL, "dense is NULL");
    hipStream_t st = (hipStream_t)stream;
    const float* aux = nullptr;
    if (h->plan.din.enabled) {
        const size_t need = sprk_workspace_bytes(h, B);
        if (!workspace || workspace_bytes < need) return fail(SPRK_EINVAL, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
        int rc = launch_din(h, ids, (float*)workspace, nullptr, B, st);
        if (rc) return rc;
        aux = (const float*)workspace;
    }
    if (h->v2_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + V2_WAVES - 1) / V2_WAVES;
        if (grid > h->v2_grid_cap) grid = h->v2_grid_cap;
        V2Run run = h->v2run;
        run.flags = (((uintptr_t)ids | (uintptr_t)dense) & 15) ? 1 : 0;    // unaligned inputs: element-wise staging
        run.flags |= h->v2_xflags;                                          // experiment switches (cached at finalize)
        const V2Variant& vv = kV2Variants[h->v2_variant];
        if (h->v2j_variant >= 0 && !run.trace && !(run.flags & ~1)) {
            V2JRun jr = h->v2j_run;
            jr.flags = run.flags;
            if (h->v2j1_image && ntasks <= V2J1_MAX_TASKS) {
                // one strict launch of one batch: one task per wave, four waves per SIMD (k_chain_v2j1.h)
                for (size_t v = 0; v < sizeof(kV2J1Variants) / sizeof(kV2J1Variants[0]); ++v)
                    if (kV2J1Variants[v].g_big == kV2JVariants[h->v2j_variant].g_big && kV2J1Variants[v].njf == kV2JVariants[h->v2j_variant].njf) {
                        kV2J1Variants[v].launch(jr, ids, dense, out, B, h->dev_err, h->v2j1_image, (ntasks + V2J1_WAVES - 1) / V2J1_WAVES,
                                                h->v2j1_lds_bytes, st);
                        HIP_TRY(hipGetLastError());
                        return SPRK_OK;
                    }
            }
            kV2JVariants[h->v2j_variant].launch(jr, ids, dense, out, B, h->dev_err, h->v2_image, grid, h->v2j_lds_bytes, st);
            HIP_TRY(hipGetLastError());
            return SPRK_OK;
        }
        (run.trace ? vv.launch_trace : vv.launch)(run, ids, dense, out, B, h->dev_err, h->v2_image, grid, h->v2_lds_bytes, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->rows_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + RC_WAVES - 1) / RC_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (2 waves per SIMD)
        RowsRun rr = h->rows_run;
        rr.flags = (((uintptr_t)ids | (uintptr_t)dense) & 15) ? 1 : 0;
        kRowsVariants[h->rows_variant].launch(rr, ids, dense, out, B, h->dev_err, h->rows_image, grid, h->rows_lds_bytes, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->v1_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + V1_WAVES - 1) / V1_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (2 waves per SIMD; 3 per SIMD measured slower at B = 65 536)
        if (h->v1_one && ntasks <= V1_ONE_MAX_TASKS)
            kV1Variants[h->v1_variant].launch_one(h->v1_run, ids, dense, out, B, h->dev_err, (ntasks + V1_WAVES - 1) / V1_WAVES, st);
        else
            kV1Variants[h->v1_variant].launch(h->v1_run, ids, dense, out, B, h->dev_err, grid, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->mlp_rows_nbig >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + MR_WAVES - 1) / MR_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (the LDS holds weights + genre tables)
        MlpRowsRun rr = h->mlp_rows_run;
        rr.flags = (((uintptr_t)ids | (uintptr_t)dense) & 15) ? 1 : 0;
        if (h->mlp_rows_nbig == 1) mlp_rows_launch<1>(rr, ids, dense, out, B, h->dev_err, h->mlp_rows_image, grid, h->mlp_rows_lds, st);
        else mlp_rows_launch<2>(rr, ids, dense, out, B, h->dev_err, h->mlp_rows_image, grid, h->mlp_rows_lds, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->mlp_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + MC_WAVES - 1) / MC_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (120 KB of LDS weights)
        const size_t lds = MlpChainLds<8, 8>::bytes;
        if (h->mlp_run.inv_w1_scale != 0.f)
            hipLaunchKernelGGL((k_mlp_chain<8, 8, MC_WAVES, true>), dim3(grid), dim3(MC_WAVES * 64), lds, st, h->mlp_run, ids, dense, out, B,
                               h->dev_err, h->mlp_image);
        else
            hipLaunchKernelGGL((k_mlp_chain<8, 8, MC_WAVES, false>), dim3(grid), dim3(MC_WAVES * 64), lds, st, h->mlp_run, ids, dense, out, B,
                               h->dev_err, h->mlp_image);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    if (h->din_tail_variant >= 0) {
        const int ntasks = (B + 15) / 16;
        int grid = (ntasks + DT_WAVES - 1) / DT_WAVES;
        if (grid > h->num_cus) grid = h->num_cus;                  // one 8-wave workgroup per CU (2 waves per SIMD)
        kDinTailVariants[h->din_tail_variant].launch(h->din_tail_run, ids, dense, aux, out, B, h->dev_err, h->din_tail_image, grid, st);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    const int ntiles = (B + SPRK_TILE_M - 1) / SPRK_TILE_M;
    const int grid = ntiles < h->tile_grid_cap ? ntiles : h->tile_grid_cap;
    hipLaunchKernelGGL(k_tile_forward, dim3(grid), dim3(256), h->tile_lds_bytes, st, h->dev_plan, ids, dense, aux, out, B, h->dev_err);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

int sprk_forward_many(sprk_handle h, int32_t n_batches, const int32_t* const* ids, const float* const* dense,
                      float* const* out, int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (n_batches < 0) return fail(SPRK_EINVAL, "negative batch count");
    if (n_batches > 0 && !out) return fail(SPRK_EINVAL, "out is NULL");
    int S = (h->finalized && n_batches > 1) ? h->many_streams : 0;
    // a model with a workspace (DIN: attention kernel -> pooled vectors -> tail kernel) needs one workspace slice per
    // stream; with a single slice its forwards stay in strict order
    const size_t ws_need = (sprk_workspace_bytes(h, B) + 255) & ~(size_t)255;
    if (S >= 2 && ws_need > 0) {
        while (S >= 2 && (!workspace || workspace_bytes < (size_t)S * ws_need)) --S;
        if (S < 2) S = 0;
    }
    // several batches per launch (sprk_set_many_batches): the fused DeepFM_v2 kernel takes up to V2J_MB batches' buffers
    // and walks their tasks as one grid; everything else (other models, unaligned buffers, tracing) goes batch by batch
    if (h->finalized && h->many_batches > 1 && n_batches > 1 && h->v2_variant >= 0 && h->v2j_variant >= 0 && !h->v2run.trace &&
        !h->v2_xflags_set && B > 0 && ids && dense) {
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i)
            ok = ids[i] && dense[i] && out[i] && !(((uintptr_t)ids[i] | (uintptr_t)dense[i]) & 15);
        if (ok) {
            const int ntpb = (B + 15) / 16;
            const V2JVariant& jv = kV2JVariants[h->v2j_variant];
            V2JRun jr = h->v2j_run;
            jr.flags = 0;
            for (int32_t i0 = 0; i0 < n_batches; i0 += h->many_batches) {
                V2JMany m;
                memset(&m, 0, sizeof(m));
                m.n = n_batches - i0 < h->many_batches ? n_batches - i0 : h->many_batches;
                m.ntpb = ntpb;
                for (int j = 0; j < m.n; ++j) { m.ids[j] = ids[i0 + j]; m.dense[j] = dense[i0 + j]; m.out[j] = out[i0 + j]; }
                const long long ntasks = (long long)m.n * ntpb;
                long long grid = (ntasks + V2_WAVES - 1) / V2_WAVES;
                if (grid > h->v2_grid_cap) grid = h->v2_grid_cap;
                jv.launch_many(jr, m, B, h->dev_err, h->v2_image, (int)grid, h->v2j_lds_bytes, (hipStream_t)stream);
                HIP_TRY(hipGetLastError());
            }
            return SPRK_OK;
        }
    }
    // k_rows_chain: up to RC_MB batches per launch
    if (h->finalized && h->many_batches > 1 && n_batches > 1 && h->rows_variant >= 0 && B > 0 && ids && (dense || h->plan.n_dense == 0)) {
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i)
            ok = ids[i] && out[i] && (h->plan.n_dense == 0 || dense[i]) && !(((uintptr_t)ids[i] | (uintptr_t)(h->plan.n_dense ? dense[i] : nullptr)) & 15);
        if (ok) {
            const int per = h->many_batches < RC_MB ? h->many_batches : RC_MB;
            const int ntpb = (B + 15) / 16;
            RowsRun rr = h->rows_run;
            rr.flags = 0;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per) {
                RowsMany m;
                memset(&m, 0, sizeof(m));
                m.n = n_batches - i0 < per ? n_batches - i0 : per;
                m.ntpb = ntpb;
                for (int j = 0; j < m.n; ++j) { m.ids[j] = ids[i0 + j]; m.dense[j] = h->plan.n_dense ? dense[i0 + j] : nullptr; m.out[j] = out[i0 + j]; }
                long long grid = ((long long)m.n * ntpb + RC_WAVES - 1) / RC_WAVES;
                if (grid > h->num_cus) grid = h->num_cus;
                kRowsVariants[h->rows_variant].launch_many(rr, m, B, h->dev_err, h->rows_image, (int)grid, h->rows_lds_bytes, (hipStream_t)stream);
                HIP_TRY(hipGetLastError());
            }
            return SPRK_OK;
        }
    }
    // the pairwise-dot DeepFM kernel: up to V1_MB batches per launch
    if (h->finalized && h->many_batches > 1 && n_batches > 1 && h->v2_variant < 0 && h->v1_variant >= 0 && B > 0 && ids && dense) {
        bool ok = true;
        for (int32_t i = 0; i < n_batches && ok; ++i) ok = ids[i] && dense[i] && out[i];
        if (ok) {
            const int per = h->many_batches < V1_MB ? h->many_batches : V1_MB;
            const int ntpb = (B + 15) / 16;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per) {
                V1Many m;
                memset(&m, 0, sizeof(m));
                m.n = n_batches - i0 < per ? n_batches - i0 : per;
                m.ntpb = ntpb;
                for (int j = 0; j < m.n; ++j) { m.ids[j] = ids[i0 + j]; m.dense[j] = dense[i0 + j]; m.out[j] = out[i0 + j]; }
                long long grid = ((long long)m.n * ntpb + V1_WAVES - 1) / V1_WAVES;
                if (grid > h->num_cus) grid = h->num_cus;
                kV1Variants[h->v1_variant].launch_many(h->v1_run, m, B, h->dev_err, (int)grid, (hipStream_t)stream);
                HIP_TRY(hipGetLastError());
            }
            return SPRK_OK;
        }
    }
    // DIN (k_din_attn -> pooled vectors -> k_din_tail): the attention launches of a group of batches, then ONE tail launch for
    // the group; a workspace slice per batch of the group.  Groups alternate over the helper streams when there are slices for that.
    if (h->finalized && h->many_batches > 1 && n_batches > 1 && h->plan.din.enabled == 1 && h->din_variant >= 0 &&
        h->din_tail_variant >= 0 && B > 0 && ids && dense && workspace && ws_need > 0) {
        int per = h->many_batches < DIN_MB ? h->many_batches : DIN_MB;
        if ((size_t)per * ws_need > workspace_bytes) per = (int)(workspace_bytes / ws_need);
        bool ok = per >= 2;
        for (int32_t i = 0; i < n_batches && ok; ++i) ok = ids[i] && dense[i] && out[i];
        if (ok) {
            int SG = S >= 2 ? S : 1;                                   // streams the groups alternate over
            while (SG > 1 && (size_t)SG * per * ws_need > workspace_bytes) --SG;
            if (SG >= 2) {
                HIP_TRY(hipEventRecord(h->many_fork, (hipStream_t)stream));
                for (int s = 0; s < SG; ++s) HIP_TRY(hipStreamWaitEvent(h->many_stream[s], h->many_fork, 0));
            }
            const DinVariant& av = kDinVariants[h->din_variant];
            const DinTailVariant& tv = kDinTailVariants[h->din_tail_variant];
            const int ntpb = (B + 15) / 16;
            int g = 0;
            for (int32_t i0 = 0; i0 < n_batches; i0 += per, ++g) {
                const int n = n_batches - i0 < per ? n_batches - i0 : per;
                hipStream_t st = SG >= 2 ? h->many_stream[g % SG] : (hipStream_t)stream;
                char* wbase = (char*)workspace + (size_t)(g % SG) * per * ws_need;
                DinTailMany tm;
                memset(&tm, 0, sizeof(tm));
                tm.n = n; tm.ntpb = ntpb;
                for (int j = 0; j < n; ++j) {
                    float* pooled = (float*)(wbase + (size_t)j * ws_need);
                    tm.ids[j] = ids[i0 + j]; tm.dense[j] = dense[i0 + j]; tm.aux[j] = pooled; tm.out[j] = out[i0 + j];
                }
                // ONE attention launch for the group (k_din_attn<..., MB = true>: no launch boundary and no partial last round of waves
                // between the batches; SPRK_DIN_ATTN_MB=0: one launch per batch), then one tail launch
                if (av.launch_many && h->din_attn_many && n <= DIN_ATTN_MB) {
                    DinAttnMany am;
                    memset(&am, 0, sizeof(am));
                    am.n = n;
                    for (int j = 0; j < n; ++j) { am.ids[j] = tm.ids[j]; am.pooled[j] = const_cast<float*>(tm.aux[j]); }
                    long long ag = ((long long)n * B + h->din_wpb - 1) / h->din_wpb;
                    if (ag > h->din_attn_grid_cap) ag = h->din_attn_grid_cap;
                    av.launch_many(h->din_run, am, B, h->dev_err, (int)ag, h->din_attn_lds, st);
                } else {
                    int ag = (B + h->din_wpb - 1) / h->din_wpb;
                    if (ag > h->din_attn_grid_cap) ag = h->din_attn_grid_cap;
                    for (int j = 0; j < n; ++j)
                        av.launch(h->din_run, tm.ids[j], const_cast<float*>(tm.aux[j]), nullptr, B, h->dev_err, ag, h->din_attn_lds, st);
                }
                long long tg = ((long long)n * ntpb + DT_WAVES - 1) / DT_WAVES;
                if (tg > h->num_cus) tg = h->num_cus;
                tv.launch_many(h->din_tail_run, tm, B, h->dev_err, h->din_tail_image, (int)tg, st);
                HIP_TRY(hipGetLastError());
            }
            if (SG >= 2) {
                for (int s = 0; s < SG; ++s) {
                    HIP_TRY(hipEventRecord(h->many_join[s], h->many_stream[s]));
                    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->many_join[s], 0));
                }
            }
            return SPRK_OK;
        }
    }
    if (S >= 2) {
        HIP_TRY(hipEventRecord(h->many_fork, (hipStream_t)stream));
        for (int s = 0; s < S; ++s) HIP_TRY(hipStreamWaitEvent(h->many_stream[s], h->many_fork, 0));
    }
    for (int32_t i = 0; i < n_batches; ++i) {
        void* wsi = (S >= 2 && ws_need > 0) ? (void*)((char*)workspace + (size_t)(i % S) * ws_need) : workspace;
        const int rc = sprk_forward(h, ids ? ids[i] : nullptr, dense ? dense[i] : nullptr, out[i], B, wsi,
                                    (S >= 2 && ws_need > 0) ? ws_need : workspace_bytes, S >= 2 ? (void*)h->many_stream[i % S] : stream);
        if (rc) return rc;
    }
    if (S >= 2) {
        for (int s = 0; s < S; ++s) {
            HIP_TRY(hipEventRecord(h->many_join[s], h->many_stream[s]));
            HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, h->many_join[s], 0));
        }
    }
    return SPRK_OK;
}

#define SPRK_FORWARD_KIND(name, kind)                                                                      \
    int name(sprk_handle h, const int32_t* ids, const float* dense, float* out, int32_t B, void* ws,      \
             size_t ws_bytes, void* stream) {                                                              \
        if (!h) return fail(SPRK_EINVAL, "handle is NULL");                                                \
        if (h->plan.model_kind != kind) return fail(SPRK_EKIND, #name ": handle holds model kind %d", h->plan.model_kind); \
        return sprk_forward(h, ids, dense, out, B, ws, ws_bytes, stream);                                  \
    }
SPRK_FORWARD_KIND(sprk_forward_embedding_mlp, SPRK_MODEL_EMBEDDING_MLP)
SPRK_FORWARD_KIND(sprk_forward_widedeep, SPRK_MODEL_WIDE_DEEP)
SPRK_FORWARD_KIND(sprk_forward_neuralcf, SPRK_MODEL_NEURALCF)
SPRK_FORWARD_KIND(sprk_forward_deepfm, SPRK_MODEL_DEEPFM)
SPRK_FORWARD_KIND(sprk_forward_deepfm_v2, SPRK_MODEL_DEEPFM_V2)
SPRK_FORWARD_KIND(sprk_forward_din, SPRK_MODEL_DIN)
SPRK_FORWARD_KIND(sprk_forward_dien, SPRK_MODEL_DIEN)

int sprk_set_many_streams(sprk_handle h, int32_t n) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "set_many_streams before finalize");
    if (n < 0 || n > 4) return fail(SPRK_EINVAL, "stream count %d outside [0,4]", n);
    h->many_streams = n < 2 ? 0 : n;
    return SPRK_OK;
}

int sprk_set_many_batches(sprk_handle h, int32_t n) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "set_many_batches before finalize");
    if (n < 1 || n > V2J_MB) return fail(SPRK_EINVAL, "batches per launch %d outside [1,%d]", n, V2J_MB);   // (DIN caps at DIN_MB)
    h->many_batches = n;
    return SPRK_OK;
}

int sprk_describe(sprk_handle h, char* buf, size_t buf_bytes) {
    if (!h || !buf || buf_bytes == 0) return fail(SPRK_EINVAL, "describe: NULL argument");
    if (!h->finalized) return fail(SPRK_ESTATE, "describe before finalize");
    char kern[160];
    if (h->v2_variant >= 0 && h->v2j_variant >= 0) {
        const V2JVariant& jv = kV2JVariants[h->v2j_variant];
        snprintf(kern, sizeof(kern), "k_deepfm_v2_joint<G_BIG=%d,NJF=%d,KPC=%d,%s>", jv.g_big, jv.njf, jv.kpc, jv.half ? "split-f16" : "f32");
    } else if (h->v2_variant >= 0) {
        const V2Variant& vv = kV2Variants[h->v2_variant];
        snprintf(kern, sizeof(kern), "k_deepfm_v2_chain<G=%d,KPC=%d,%s>", vv.g_emb, vv.kpc, vv.fold ? "folded" : "unfolded");
    } else if (h->v1_variant >= 0) {
        snprintf(kern, sizeof(kern), "k_deepfm_pairs<NF=%d,NV=%d>", kV1Variants[h->v1_variant].nf, kV1Variants[h->v1_variant].nv);
    } else if (h->rows_variant >= 0) {
        const RowsVariant& rv = kRowsVariants[h->rows_variant];
        snprintf(kern, sizeof(kern), "k_rows_chain<KPC=%d,H0C=%d,H1C=%d,G_BIG=%d,NJF=%d>", rv.kpc, rv.h0c, rv.h1c, rv.g_big, rv.njf);
    } else if (h->mlp_rows_nbig >= 0) {
        snprintf(kern, sizeof(kern), "k_mlp_rows<8,8,NBIG=%d,NSMALL=%d>", h->mlp_rows_nbig, h->mlp_rows_run.n_small);
    } else if (h->mlp_variant >= 0) {
        snprintf(kern, sizeof(kern), "k_mlp_chain<8,8>");
    } else if (h->din_tail_variant >= 0) {
        const DinTailVariant& tv = kDinTailVariants[h->din_tail_variant];
        snprintf(kern, sizeof(kern), "k_din_tail<%d,%d,%d>", tv.n0c, tv.n1c, tv.kpc);
    } else {
        snprintf(kern, sizeof(kern), "k_tile_forward");
    }
    const char* stage = "";
    if (h->plan.din.enabled == 2) stage = "k_dien_seq";
    else if (h->plan.din.enabled == 1) stage = h->din_variant >= 0 ? "k_din_attn" : "k_din_pool";
    size_t uploaded = 0;
    for (size_t b : h->slot_bytes) uploaded += b;
    const int n = snprintf(buf, buf_bytes, "kernel=%s;stage=%s;stage_waves_per_workgroup=%d;fused=%d;uploaded_bytes=%zu;derived_bytes=%zu;first_dense_fold=%d", kern, stage,
                           h->din_variant >= 0 ? h->din_wpb : 0, strcmp(kern, "k_tile_forward") != 0 ? 1 : 0, uploaded, h->derived_bytes, h->n_acc_folded);
    if (n < 0 || (size_t)n >= buf_bytes) return fail(SPRK_EINVAL, "describe: buffer of %zu bytes is too small", buf_bytes);
    return SPRK_OK;
}

int sprk_check_ids(sprk_handle h, void* stream) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "check_ids before finalize");
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, h->dev_err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        HIP_TRY(hipMemsetAsync(h->dev_err, 0, sizeof(int), (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        return fail(SPRK_ERANGE, "an id was outside its table (TF would raise InvalidArgumentError: assert_less_than_num_buckets)");
    }
    return SPRK_OK;
}

int sprk_debug_set_trace(sprk_handle h, void* dev_buf, size_t bytes) {
    if (!h) return fail(SPRK_EINVAL, "handle is NULL");
    if (!h->finalized) return fail(SPRK_ESTATE, "set_trace before finalize");
    if (h->v2_variant < 0 || !kV2Variants[h->v2_variant].launch_trace) return fail(SPRK_EKIND, "handle does not run a traceable kernel");
    if (dev_buf && bytes < (size_t)h->v2_grid_cap * V2_WAVES * 16 * sizeof(unsigned long long))
        return fail(SPRK_EINVAL, "trace buffer too small: %zu bytes for %d waves", bytes, h->v2_grid_cap * V2_WAVES);
    h->v2run.trace = (unsigned long long*)dev_buf;
    return SPRK_OK;
}

void sprk_destroy(sprk_handle h) {
    if (!h) return;
    for (void* p : h->slot_ptr)
        if (p) (void)hipFree(p);
    if (h->dev_plan) (void)hipFree(h->dev_plan);
    if (h->v2_image) (void)hipFree(h->v2_image);
    if (h->v2_fo_all) (void)hipFree(h->v2_fo_all);
    if (h->v2_folded) (void)hipFree(h->v2_folded);
    if (h->v2j_tab) (void)hipFree(h->v2j_tab);
    for (void* p : h->fold_bufs) if (p) (void)hipFree(p);
    for (int i = 0; i < 4; ++i) { if (h->many_stream[i]) (void)hipStreamDestroy(h->many_stream[i]); if (h->many_join[i]) (void)hipEventDestroy(h->many_join[i]); }
    if (h->many_fork) (void)hipEventDestroy(h->many_fork);
    if (h->din_tail_image) (void)hipFree(h->din_tail_image);
    if (h->mlp_image) (void)hipFree(h->mlp_image);
    if (h->mlp_rows_image) (void)hipFree(h->mlp_rows_image);
    if (h->mlp_rows_small) (void)hipFree(h->mlp_rows_small);
    for (void* q : h->mlp_rows_bufs) if (q) (void)hipFree(q);
    for (void* p : h->v1_bufs) if (p) (void)hipFree(p);
    if (h->v2j_big) (void)hipFree(h->v2j_big);
    if (h->v2j1_image) (void)hipFree(h->v2j1_image);
    if (h->rows_tab) (void)hipFree(h->rows_tab);
    if (h->rows_scal) (void)hipFree(h->rows_scal);
    if (h->rows_small) (void)hipFree(h->rows_small);
    if (h->rows_image) (void)hipFree(h->rows_image);
    if (h->din_w12) (void)hipFree(h->din_w12);
    if (h->din_w4) (void)hipFree(h->din_w4);
    if (h->din_vc) (void)hipFree(h->din_vc);
    if (h->din_tsplit) (void)hipFree(h->din_tsplit);
    if (h->dev_err) (void)hipFree(h->dev_err);
    delete h;
}

int sprk_embedding_gather(const float* table, int32_t V, int32_t D, int32_t row_stride, const int32_t* ids,
                          int32_t B, float* out, void* stream) {
    if (!table || !ids || !out) return fail(SPRK_EINVAL, "NULL argument");
    if (V <= 0 || D <= 0 || (D & 3) || row_stride < D || (row_stride & 3)) return fail(SPRK_EINVAL, "bad gather geometry V=%d D=%d row_stride=%d", V, D, row_stride);
    if (B < 0) return fail(SPRK_EINVAL, "negative batch");
    if (B == 0) return SPRK_OK;
    const long long total = (long long)B * (D / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_embedding_gather, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, V, D / 4, row_stride, ids, B, out);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

// ---- host ingest: CSV text -> packed ids / dense (schema.py's read_samples_csv + pack_ids + pack_dense in one pass) ----
namespace {
const char* const kGenreVocab[19] = {"Film-Noir", "Action", "Adventure", "Horror", "Romance", "War", "Comedy", "Western",
                                     "Documentary", "Sci-Fi", "Drama", "Thriller", "Crime", "Fantasy", "Animation", "IMAX",
                                     "Mystery", "Children", "Musical"};   // DeepFM.py:64-66
struct CsvField { const char* p; size_t n; };
// splits one line (no trailing newline) into fields; supports "quoted, fields" with "" escapes (unescaped into `scratch`)
void split_csv_line(const char* p, const char* end, std::vector<CsvField>& out, std::string& scratch) {
    out.clear();
    scratch.clear();
    scratch.reserve((size_t)(end - p) + 1);                     // pointers into scratch stay valid
    while (true) {
        if (p < end && *p == '"') {
            const size_t start = scratch.size();
            ++p;
            while (p < end) {
                if (*p == '"') {
                    if (p + 1 < end && p[1] == '"') { scratch.push_back('"'); p += 2; continue; }
                    ++p;
                    break;
                }
                scratch.push_back(*p++);
            }
            out.push_back(CsvField{scratch.data() + start, scratch.size() - start});
            while (p < end && *p != ',') ++p;
        } else {
            const char* q = p;
            while (q < end && *q != ',') ++q;
            out.push_back(CsvField{p, (size_t)(q - p)});
            p = q;
        }
        if (p >= end) break;
        ++p;                                                    // the comma
        if (p == end) { out.push_back(CsvField{p, 0}); break; }
    }
}
bool parse_number(const CsvField& f, double* v) {
    char buf[64];
    if (f.n == 0 || f.n >= sizeof(buf)) return false;
    memcpy(buf, f.p, f.n);
    buf[f.n] = 0;
    char* e = nullptr;
    *v = strtod(buf, &e);
    return e != buf && *e == 0;
}
}  // namespace

int sprk_emb_rank(const float* item_emb, const uint8_t* item_has, int32_t n_items, int32_t D, int32_t item_stride,
                  const float* query_emb, const uint8_t* query_has, int32_t n_queries, int32_t query_stride,
                  const int32_t* cand, int32_t C, double* scores, int32_t* order, void* stream) {
    if (!item_emb || !query_emb || !cand || !scores) return fail(SPRK_EINVAL, "emb_rank: NULL table / queries / candidates / scores");
    if (n_items < 0 || n_queries < 0 || C < 0 || D < 1 || D > 1024 || item_stride < D || query_stride < D)
        return fail(SPRK_EINVAL, "emb_rank: bad sizes (need 1 <= D <= 1024, strides >= D)");
    if (order && C > ER_MAX_SORT) return fail(SPRK_EINVAL, "emb_rank: ranking supports at most 4096 candidates per query");
    if (n_queries == 0 || C == 0) return SPRK_OK;
    if (order && C <= 1024 && !getenv("SPRK_EMB_RANK_GENERIC")) {   // one wave per query, bitonic network in registers
        const int grid = (n_queries + ERW_WAVES - 1) / ERW_WAVES;
        const int E = C <= 256 ? 4 : 16;
        const size_t lds_w = (size_t)ERW_WAVES * 64 * E * 8 + (size_t)ERW_WAVES * D * 4;
        if (E == 4)
            hipLaunchKernelGGL(k_emb_rank_wave<4>, dim3(grid), dim3(ERW_WAVES * 64), lds_w, (hipStream_t)stream, item_emb, item_has,
                               n_items, D, item_stride, query_emb, query_has, n_queries, query_stride, cand, C, scores, order);
        else
            hipLaunchKernelGGL(k_emb_rank_wave<16>, dim3(grid), dim3(ERW_WAVES * 64), lds_w, (hipStream_t)stream, item_emb, item_has,
                               n_items, D, item_stride, query_emb, query_has, n_queries, query_stride, cand, C, scores, order);
        HIP_TRY(hipGetLastError());
        return SPRK_OK;
    }
    int P = 0;
    if (order) { P = 2; while (P < C) P <<= 1; }
    const size_t lds = (size_t)P * 12 + (size_t)D * 4 + 16;
    hipLaunchKernelGGL(k_emb_rank, dim3(n_queries), dim3(ER_THREADS), lds, (hipStream_t)stream, item_emb, item_has, n_items, D,
                       item_stride, query_emb, query_has, query_stride, cand, C, P, scores, order);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

}  // extern "C" (helpers below are C++)

namespace {
struct CsvLayout {
    size_t n_cols;
    std::vector<int> id_pos, dense_pos;
};
struct CsvChunkResult {
    std::vector<int32_t> ids;
    std::vector<float> dense;
    int32_t rows = 0;
    int rc = SPRK_OK;                 // first error of the chunk, raised after `rows` good rows
    std::string msg;
};
// Rows of [p, end) appended to `out` (at most max_rows); stops at the first bad value (out.rc / out.msg).
void pack_csv_rows(const char* p, const char* end, const CsvLayout& L, const sprk_csv_col* id_cols, int n_id, const char* const* dense_names,
                   int n_dense, int32_t max_rows, CsvChunkResult& out) {
    std::vector<CsvField> fields;
    std::string scratch;
    char buf[256];
    auto line_end = [&](const char* s) { const void* q = memchr(s, '\n', (size_t)(end - s)); return q ? (const char*)q : end; };
    while (p < end && out.rows < max_rows) {
        const char* le = line_end(p);
        const char* re = (le > p && le[-1] == '\r') ? le - 1 : le;
        if (re > p) {
            split_csv_line(p, re, fields, scratch);
            if (fields.size() == L.n_cols) {                    // ignore_errors=True: other rows are dropped
                const size_t i0 = out.ids.size(), d0 = out.dense.size();
                out.ids.resize(i0 + n_id);
                out.dense.resize(d0 + n_dense);
                for (int j = 0; j < n_id; ++j) {
                    const CsvField& f = fields[L.id_pos[j]];
                    int32_t v;
                    if (id_cols[j].kind == 1) {
                        v = -1;
                        for (int g = 0; g < 19; ++g)
                            if (f.n == strlen(kGenreVocab[g]) && memcmp(f.p, kGenreVocab[g], f.n) == 0) { v = g; break; }
                        if (v >= id_cols[j].vocab) v = -1;
                    } else {
                        double d = 0.0;
                        if (f.n != 0 && !parse_number(f, &d)) {
                            snprintf(buf, sizeof(buf), "%s is not a number", id_cols[j].name);
                            out.rc = SPRK_EINVAL; out.msg = buf; out.ids.resize(i0); out.dense.resize(d0);
                            return;
                        }
                        const long long iv = (long long)d;      // int(float(v)) of the Python packer
                        if (iv < 0 || iv >= id_cols[j].vocab) {
                            snprintf(buf, sizeof(buf), "%s id %lld outside [0, %d) (reference: assert_less_than_num_buckets)", id_cols[j].name, iv,
                                     id_cols[j].vocab);
                            out.rc = SPRK_ERANGE; out.msg = buf; out.ids.resize(i0); out.dense.resize(d0);
                            return;
                        }
                        v = (int32_t)iv;
                    }
                    out.ids[i0 + j] = v;
                }
                for (int j = 0; j < n_dense; ++j) {
                    const CsvField& f = fields[L.dense_pos[j]];
                    double d = 0.0;
                    if (f.n != 0 && !parse_number(f, &d)) {
                        snprintf(buf, sizeof(buf), "%s is not a number", dense_names[j]);
                        out.rc = SPRK_EINVAL; out.msg = buf; out.ids.resize(i0); out.dense.resize(d0);
                        return;
                    }
                    out.dense[d0 + j] = (float)d;
                }
                ++out.rows;
            }
        }
        p = le < end ? le + 1 : end;
    }
}
}  // namespace

extern "C" {

int sprk_pack_csv_mt(const char* text, size_t len, const sprk_csv_col* id_cols, int32_t n_id, const char* const* dense_names,
                     int32_t n_dense, int32_t max_rows, int32_t n_threads, int32_t* ids_out, float* dense_out, int32_t* rows_out) {
    if (!text || !rows_out || n_id < 0 || n_dense < 0 || max_rows < 0) return fail(SPRK_EINVAL, "bad pack_csv arguments");
    if ((n_id > 0 && (!id_cols || !ids_out)) || (n_dense > 0 && (!dense_names || !dense_out))) return fail(SPRK_EINVAL, "NULL column list / output");
    *rows_out = 0;
    const char* p = text;
    const char* const end = text + len;
    // header
    CsvLayout L;
    {
        const void* q = memchr(p, '\n', len);
        const char* le = q ? (const char*)q : end;
        const char* he = (le > p && le[-1] == '\r') ? le - 1 : le;
        std::vector<CsvField> fields;
        std::string scratch;
        split_csv_line(p, he, fields, scratch);
        L.n_cols = fields.size();
        L.id_pos.assign(n_id, -1);
        L.dense_pos.assign(n_dense, -1);
        auto find = [&](const char* name) {
            const size_t n = strlen(name);
            for (size_t c = 0; c < L.n_cols; ++c) if (fields[c].n == n && memcmp(fields[c].p, name, n) == 0) return (int)c;
            return -1;
        };
        for (int j = 0; j < n_id; ++j) if ((L.id_pos[j] = find(id_cols[j].name)) < 0) return fail(SPRK_EINVAL, "CSV has no column %s", id_cols[j].name);
        for (int j = 0; j < n_dense; ++j) if ((L.dense_pos[j] = find(dense_names[j])) < 0) return fail(SPRK_EINVAL, "CSV has no column %s", dense_names[j]);
        p = le < end ? le + 1 : end;
    }
    // chunks of whole lines, one per thread (a text below 1 MiB is not worth a thread start)
    int T = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
    const size_t body = (size_t)(end - p);
    if (body < ((size_t)1 << 20)) T = 1;
    std::vector<const char*> cut(T + 1, end);
    cut[0] = p;
    for (int t = 1; t < T; ++t) {
        const char* c = p + body / T * t;
        if (c < cut[t - 1]) c = cut[t - 1];
        const void* q = c < end ? memchr(c, '\n', (size_t)(end - c)) : nullptr;
        cut[t] = q ? (const char*)q + 1 : end;
    }
    std::vector<CsvChunkResult> res(T);
    if (T == 1) {
        pack_csv_rows(cut[0], cut[1], L, id_cols, n_id, dense_names, n_dense, max_rows, res[0]);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t]() { pack_csv_rows(cut[t], cut[t + 1], L, id_cols, n_id, dense_names, n_dense, max_rows, res[t]); });
        for (auto& x : th) x.join();
    }
    // stitch in file order: exactly what one pass would have produced (rows beyond max_rows are never looked at)
    int32_t rows = 0;
    for (int t = 0; t < T && rows < max_rows; ++t) {
        const CsvChunkResult& r = res[t];
        const int32_t take = r.rows < max_rows - rows ? r.rows : max_rows - rows;
        if (take > 0) {
            if (n_id) memcpy(ids_out + (size_t)rows * n_id, r.ids.data(), (size_t)take * n_id * sizeof(int32_t));
            if (n_dense) memcpy(dense_out + (size_t)rows * n_dense, r.dense.data(), (size_t)take * n_dense * sizeof(float));
        }
        rows += take;
        if (r.rc != SPRK_OK && rows < max_rows) return fail(r.rc, "row %d: %s", rows, r.msg.c_str());
    }
    *rows_out = rows;
    return SPRK_OK;
}

int sprk_pack_csv(const char* text, size_t len, const sprk_csv_col* id_cols, int32_t n_id, const char* const* dense_names,
                  int32_t n_dense, int32_t max_rows, int32_t* ids_out, float* dense_out, int32_t* rows_out) {
    return sprk_pack_csv_mt(text, len, id_cols, n_id, dense_names, n_dense, max_rows, 1, ids_out, dense_out, rows_out);
}

int sprk_cross_hash(const int32_t* a, const int32_t* b, int32_t B, int64_t num_buckets, int64_t* out, void* stream) {
    if (!a || !b || !out) return fail(SPRK_EINVAL, "NULL argument");
    if (num_buckets <= 0) return fail(SPRK_EINVAL, "num_buckets must be positive");
    if (B < 0) return fail(SPRK_EINVAL, "negative batch");
    if (B == 0) return SPRK_OK;
    int blocks = (B + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_cross_hash, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, B, (unsigned long long)num_buckets, (long long*)out);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

}  // extern "C"

// ---- device ingest: the CSV text is already in HBM (k_csv_pack.h) ----
namespace {
struct DevScratch {
    void* p = nullptr;
    size_t cap = 0;
    int dev = -1;                         // the device `p` lives on: a thread that switches devices gets a new scratch there
    int ensure(size_t bytes) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return fail(SPRK_EHIP, "hipGetDevice"); }
        if (cur == dev && bytes <= cap) return SPRK_OK;
        if (p) (void)hipFree(p);          // hipFree takes a pointer of any device
        p = nullptr; cap = 0; dev = cur;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return fail(SPRK_EHIP, "device scratch of %zu bytes for the CSV tokenizer", want); }
        cap = want;
        return SPRK_OK;
    }
};
thread_local DevScratch g_csv_scratch;

// exclusive scan of n unsigned counters (in -> out, in place allowed), grand total -> *total_dev; sums = scratch of ceil(n / SCAN_TILE)
void scan_u32(const unsigned* in, unsigned* out, size_t n, unsigned* sums, unsigned* total_dev, hipStream_t st) {
    const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, sums, nb, total_dev);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, st, in, n, (const unsigned*)sums, out);
}
}  // namespace

extern "C" {

int sprk_pack_csv_device(const char* text_dev, size_t len, const sprk_csv_col* id_cols, int32_t n_id, const char* const* dense_names,
                         int32_t n_dense, int32_t max_rows, int32_t* ids_dev, float* dense_dev, int32_t* rows_out, void* stream) {
    if (!text_dev || !rows_out || n_id < 0 || n_dense < 0 || max_rows < 0) return fail(SPRK_EINVAL, "bad pack_csv_device arguments");
    if ((n_id > 0 && (!id_cols || !ids_dev)) || (n_dense > 0 && (!dense_names || !dense_dev))) return fail(SPRK_EINVAL, "NULL column list / output");
    if (n_id > CSV_MAX_OUT || n_dense > CSV_MAX_OUT) return fail(SPRK_EINVAL, "the device tokenizer packs at most %d id and %d dense columns", CSV_MAX_OUT, CSV_MAX_OUT);
    if ((uintptr_t)text_dev & 15) return fail(SPRK_EINVAL, "the CSV text must start on a 16-byte boundary in device memory");
    if (len >= ((size_t)1 << 44)) return fail(SPRK_EINVAL, "CSV text too large");
    *rows_out = 0;
    if (len == 0) return SPRK_OK;
    hipStream_t st = (hipStream_t)stream;
    // header: the first line comes back to the host and goes through the host tokenizer's own field splitter
    std::vector<char> head(len < 16384 ? len : 16384);
    HIP_TRY(hipMemcpyAsync(head.data(), text_dev, head.size(), hipMemcpyDeviceToHost, st));
    char last = 0;
    HIP_TRY(hipMemcpyAsync(&last, text_dev + len - 1, 1, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const void* q = memchr(head.data(), '\n', head.size());
    if (!q && head.size() < len) return fail(SPRK_EINVAL, "CSV header line longer than %zu bytes", head.size());
    const char* le = q ? (const char*)q : head.data() + head.size();
    const char* he = (le > head.data() && le[-1] == '\r') ? le - 1 : le;
    std::vector<CsvField> fields;
    std::string scratch;
    split_csv_line(head.data(), he, fields, scratch);
    if (fields.size() > CSV_MAX_COLS) return fail(SPRK_EINVAL, "the device tokenizer reads at most %d CSV columns (header has %zu)", CSV_MAX_COLS, fields.size());
    CsvDev L;
    memset(&L, 0, sizeof(L));
    L.n_cols = (int)fields.size(); L.n_id = n_id; L.n_dense = n_dense;
    for (int c = 0; c < CSV_MAX_COLS; ++c) { L.id_head[c] = -1; L.dense_head[c] = -1; }
    auto find = [&](const char* name) {
        const size_t n = strlen(name);
        for (size_t c = 0; c < fields.size(); ++c) if (fields[c].n == n && memcmp(fields[c].p, name, n) == 0) return (int)c;
        return -1;
    };
    for (int j = n_id - 1; j >= 0; --j) {                          // (back to front: every column's list ends up in output order)
        const int c = find(id_cols[j].name);
        if (c < 0) return fail(SPRK_EINVAL, "CSV has no column %s", id_cols[j].name);
        L.id_next[j] = L.id_head[c]; L.id_head[c] = (short)j;
        L.id_kind[j] = id_cols[j].kind; L.id_vocab[j] = id_cols[j].vocab;
    }
    for (int j = n_dense - 1; j >= 0; --j) {
        const int c = find(dense_names[j]);
        if (c < 0) return fail(SPRK_EINVAL, "CSV has no column %s", dense_names[j]);
        L.dense_next[j] = L.dense_head[c]; L.dense_head[c] = (short)j;
    }
    for (int c = 0; c < L.n_cols; ++c) {
        for (int j = L.id_head[c]; j >= 0; j = L.id_next[j]) L.role[c] |= L.id_kind[j] == 1 ? 2 : 1;
        if (L.dense_head[c] >= 0) L.role[c] |= 1;
    }
    {
        // perfect hash of the 19 genre strings into 32 slots: the first odd multiplier without a collision
        unsigned long long lo[19], hi[19];
        unsigned len[19];
        for (int g = 0; g < 19; ++g) {
            const size_t n = strlen(kGenreVocab[g]);
            lo[g] = hi[g] = 0;
            len[g] = (unsigned)n;
            for (size_t k = 0; k < n && k < 16; ++k) (k < 8 ? lo[g] : hi[g]) |= (unsigned long long)(unsigned char)kGenreVocab[g][k] << (8 * (k & 7));
        }
        unsigned long long mul = 0x9E3779B97F4A7C15ull;
        for (int tries = 0; tries < 100000; ++tries, mul += 0x632BE59BD9B4E019ull * 2) {
            unsigned used = 0;
            bool ok = true;
            for (int g = 0; g < 19 && ok; ++g) {
                const unsigned sl = csv_genre_slot(lo[g], hi[g], len[g], mul | 1);
                ok = !(used & (1u << sl));
                used |= 1u << sl;
            }
            if (ok) break;
        }
        L.g_mul = mul | 1;
        for (int sl = 0; sl < 32; ++sl) { L.gt_idx[sl] = -1; L.gt_len[sl] = -1; }
        for (int g = 0; g < 19; ++g) {
            const unsigned sl = csv_genre_slot(lo[g], hi[g], len[g], L.g_mul);
            if (L.gt_idx[sl] >= 0) return fail(SPRK_EINVAL, "no perfect hash for the genre vocabulary");
            L.gt_lo[sl] = lo[g]; L.gt_hi[sl] = hi[g]; L.gt_len[sl] = (signed char)len[g]; L.gt_idx[sl] = (signed char)g;
        }
    }
    // pass 1: newlines per chunk
    const size_t n_chunks = (len + CSV_CHUNK - 1) / CSV_CHUNK;
    const size_t sums_a = (n_chunks + SCAN_TILE - 1) / SCAN_TILE;
    const size_t fixed = 256 + sizeof(CsvErr) * 64;                 // flags | totals | first_err | n_errs, then the error records
    size_t need = fixed + (n_chunks + sums_a + 64) * sizeof(unsigned);
    if (int rc = g_csv_scratch.ensure(need)) return rc;
    auto carve = [&]() { return (char*)g_csv_scratch.p; };
    unsigned* totals = (unsigned*)(carve() + 16);                   // [0] newlines, [1] kept lines
    unsigned long long* first_err = (unsigned long long*)(carve() + 32);
    unsigned* n_errs = (unsigned*)(carve() + 48);
    CsvErr* errs = (CsvErr*)(carve() + 256);
    unsigned* counts = (unsigned*)(carve() + fixed);
    unsigned* sums = counts + n_chunks;
    HIP_TRY(hipMemsetAsync(carve(), 0, 256, st));
    HIP_TRY(hipMemsetAsync(first_err, 0xFF, sizeof(unsigned long long), st));
    const unsigned char* text = (const unsigned char*)text_dev;
    hipLaunchKernelGGL(k_csv_count, dim3((unsigned)n_chunks), dim3(256), 0, st, text, len, counts);
    scan_u32(counts, counts, n_chunks, sums, totals, st);
    unsigned h_nl = 0;
    HIP_TRY(hipMemcpyAsync(&h_nl, totals, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const size_t n_lines = (size_t)h_nl + (last != '\n' ? 1 : 0);
    if (n_lines >= ((size_t)1 << 31)) return fail(SPRK_EINVAL, "more than 2^31 lines");
    if (n_lines <= 1) return SPRK_OK;                              // header only
    // passes 2-4 need nl[] and keep[] / pos[]: grow the scratch (contents so far are carried over by redoing pass 1's scan)
    const size_t sums_b = (n_lines + SCAN_TILE - 1) / SCAN_TILE;
    const size_t off_nl = (fixed + (n_chunks + sums_a + 64) * sizeof(unsigned) + 255) & ~(size_t)255;
    const size_t off_keep = off_nl + ((size_t)h_nl + 1) * sizeof(unsigned long long);
    need = off_keep + (2 * n_lines + sums_b + 64) * sizeof(unsigned);
    if (need > g_csv_scratch.cap) {
        // (first call on a text of this size: allocate the full scratch and run pass 1 again into it)
        if (int rc = g_csv_scratch.ensure(need)) return rc;
        totals = (unsigned*)(carve() + 16); first_err = (unsigned long long*)(carve() + 32);
        n_errs = (unsigned*)(carve() + 48); errs = (CsvErr*)(carve() + 256); counts = (unsigned*)(carve() + fixed); sums = counts + n_chunks;
        HIP_TRY(hipMemsetAsync(carve(), 0, 256, st));
        HIP_TRY(hipMemsetAsync(first_err, 0xFF, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_csv_count, dim3((unsigned)n_chunks), dim3(256), 0, st, text, len, counts);
        scan_u32(counts, counts, n_chunks, sums, totals, st);
    }
    unsigned long long* nl = (unsigned long long*)(carve() + off_nl);
    unsigned* keep = (unsigned*)(carve() + off_keep);
    unsigned* pos = keep + n_lines;
    unsigned* sums2 = pos + n_lines;
    unsigned* drops = (unsigned*)(carve() + 52);
    hipLaunchKernelGGL(k_csv_mark, dim3((unsigned)n_chunks), dim3(256), 0, st, text, len, (const unsigned*)counts, nl);
    const unsigned lb = (unsigned)((n_lines + 255) / 256);
    // LDS piece per workgroup of 256 lines: twice the average, so that more workgroups share a CU when lines are short
    size_t cap = (2 * 256 * (len / n_lines + 1) + 4095) & ~(size_t)4095;
    if (cap < 8192) cap = 8192;
    if (cap > 48 * 1024) cap = 48 * 1024;
    const unsigned lds_cap = (unsigned)cap;
    unsigned h_kept = 0, h_nerr = 0, h_drops = 0;
    unsigned long long h_first = 0;
    const char* two = getenv("SPRK_CSV_TWO_PASS");              // A/B switch: "1" = always the exact keep -> scan -> parse sequence
    bool exact = two && two[0] == '1';
    if (!exact) {
        // optimistic pass over the lines that can hold the first max_rows rows if none is dropped
        const size_t lines_opt = n_lines - 1 <= (size_t)max_rows ? n_lines : (size_t)max_rows + 1;
        const unsigned lbo = (unsigned)((lines_opt + 255) / 256);
        if (lines_opt > 1)
            hipLaunchKernelGGL(k_csv_parse<true>, dim3(lbo), dim3(256), lds_cap + CSV_LDS_SLACK + CSV_LDS_GENRE, st, L, text, len, (const unsigned long long*)nl, h_nl,
                               (unsigned)lines_opt, (const unsigned*)nullptr, (const unsigned*)nullptr, (unsigned)max_rows, lds_cap, ids_dev, dense_dev,
                               first_err, errs, n_errs, drops);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_drops, drops, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_first, first_err, sizeof(h_first), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_nerr, n_errs, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        h_kept = (unsigned)(lines_opt - 1);
        if (h_drops) {                                             // some line is not a row: its successors are misplaced
            exact = true;
            HIP_TRY(hipMemsetAsync(first_err, 0xFF, sizeof(unsigned long long), st));
            HIP_TRY(hipMemsetAsync(n_errs, 0, sizeof(unsigned), st));
        }
    }
    if (exact) {
        hipLaunchKernelGGL(k_csv_keep, dim3(lb), dim3(256), lds_cap + CSV_LDS_SLACK + CSV_LDS_GENRE, st, text, len, (const unsigned long long*)nl, h_nl, (unsigned)n_lines,
                           L.n_cols, lds_cap, keep);
        scan_u32(keep, pos, n_lines, sums2, totals + 1, st);
        hipLaunchKernelGGL(k_csv_parse<false>, dim3(lb), dim3(256), lds_cap + CSV_LDS_SLACK + CSV_LDS_GENRE, st, L, text, len, (const unsigned long long*)nl, h_nl,
                           (unsigned)n_lines, (const unsigned*)keep, (const unsigned*)pos, (unsigned)max_rows, lds_cap, ids_dev, dense_dev, first_err, errs,
                           n_errs, drops);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_kept, totals + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_first, first_err, sizeof(h_first), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&h_nerr, n_errs, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (h_first != ~0ull) {
        std::vector<CsvErr> rec(h_nerr < 64 ? h_nerr : 64);
        if (!rec.empty()) HIP_TRY(hipMemcpy(rec.data(), errs, rec.size() * sizeof(CsvErr), hipMemcpyDeviceToHost));
        const unsigned row = (unsigned)(h_first >> 20);
        const int code = (int)(h_first & 15);
        const CsvErr* hit = nullptr;
        for (const CsvErr& e : rec) if (e.key == h_first) { hit = &e; break; }
        const char* name = !hit ? "a column" : (hit->is_dense ? dense_names[hit->out_col] : id_cols[hit->out_col].name);
        if (code == 1) {
            if (hit) return fail(SPRK_ERANGE, "row %u: %s id %lld outside [0, %d) (reference: assert_less_than_num_buckets)", row, name, hit->value, id_cols[hit->out_col].vocab);
            return fail(SPRK_ERANGE, "row %u: an identity id is outside its bucket range (reference: assert_less_than_num_buckets)", row);
        }
        return fail(SPRK_EKIND, "row %u: %s holds a value the device tokenizer does not convert exactly (not a plain decimal of at most 15 digits): use sprk_pack_csv", row, name);
    }
    *rows_out = (int32_t)(h_kept < (unsigned)max_rows ? h_kept : (unsigned)max_rows);
    return SPRK_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Multi-GPU: the ONE collective of the path behind the C ABI -- an all-gather of the per-rank score
// slices over RCCL (xGMI), enqueued on the caller's HIP stream.  RCCL is bound at run time (dlopen), so libsparrow_hip.so has
// no link-time dependency on it and single-GPU users never load it.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclUid { char b[SPRK_COMM_ID_BYTES]; };           // ncclUniqueId: 128 opaque bytes, passed by value
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUid*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUid, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
    if (g_rccl.lib) return SPRK_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) return fail(SPRK_EHIP, "cannot load RCCL (librccl.so.1): %s", dlerror());
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather) return fail(SPRK_EHIP, "librccl.so lacks the nccl entry points");
    g_rccl = a;
    return SPRK_OK;
}
int rccl_fail(const char* what, int rc) {
    return fail(SPRK_EHIP, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error");
}
}  // namespace

struct sprk_comm_s {
    void* comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int sprk_comm_unique_id(uint8_t id[SPRK_COMM_ID_BYTES]) {
    if (!id) return fail(SPRK_EINVAL, "id is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    RcclUid uid;
    const int nrc = g_rccl.GetUniqueId(&uid);
    if (nrc) return rccl_fail("ncclGetUniqueId", nrc);
    memcpy(id, uid.b, SPRK_COMM_ID_BYTES);
    return SPRK_OK;
}

int sprk_comm_create(const uint8_t id[SPRK_COMM_ID_BYTES], int32_t rank, int32_t world, sprk_comm* out) {
    if (!id || !out) return fail(SPRK_EINVAL, "id/out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(SPRK_EINVAL, "bad rank/world %d/%d", rank, world);
    int rc = rccl_load();
    if (rc) return rc;
    sprk_comm_s* c = new (std::nothrow) sprk_comm_s();
    if (!c) return fail(SPRK_EHIP, "out of host memory");
    c->rank = rank; c->world = world;
    RcclUid uid;
    memcpy(uid.b, id, SPRK_COMM_ID_BYTES);
    const int nrc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (nrc) { delete c; return rccl_fail("ncclCommInitRank", nrc); }
    *out = c;
    return SPRK_OK;
}

int sprk_comm_allgather_scores(sprk_comm c, const float* local, float* gathered, size_t count, void* stream) {
    if (!c || !c->comm) return fail(SPRK_EINVAL, "communicator is NULL");
    if (!local || !gathered) return fail(SPRK_EINVAL, "NULL buffer");
    if (count == 0) return SPRK_OK;
    const int nrc = g_rccl.AllGather(local, gathered, count, 7 /* ncclFloat32 */, c->comm, (hipStream_t)stream);
    return nrc ? rccl_fail("ncclAllGather", nrc) : SPRK_OK;
}

void sprk_comm_destroy(sprk_comm c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

// ---- the same exchange as direct peer writes (k_peer_gather.h) ----
}  // extern "C"

struct sprk_peer_s {
    int rank = 0, world = 1;
    size_t slot = 0;                      // floats per rank slot
    void* base = nullptr;                 // [2][world][slot] floats | [2][world] flags: one allocation, exported by IPC
    size_t flags_off = 0;
    const char* mem_kind = "";
    void* peer_base[PEER_MAX_WORLD] = {};
    unsigned* done = nullptr;             // [world] local workgroup counters
    int* err = nullptr;
    unsigned epoch = 0;
    bool connected = false;
    unsigned long long ticks = 200000000ull;   // 2 s of the 100 MHz wall clock
};

extern "C" {

int sprk_peer_create(int32_t rank, int32_t world, size_t slot_floats, uint8_t handle_out[SPRK_PEER_HANDLE_BYTES], sprk_peer* out) {
    static_assert(sizeof(hipIpcMemHandle_t) == SPRK_PEER_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    if (!out || !handle_out) return fail(SPRK_EINVAL, "NULL argument");
    *out = nullptr;
    if (world < 1 || world > PEER_MAX_WORLD || rank < 0 || rank >= world) return fail(SPRK_EINVAL, "bad rank/world %d/%d (world <= %d)", rank, world, PEER_MAX_WORLD);
    if (slot_floats == 0 || (slot_floats & 3)) return fail(SPRK_EINVAL, "slot_floats must be a positive multiple of 4 (16-byte stores)");
    sprk_peer_s* c = new sprk_peer_s;
    c->rank = rank; c->world = world; c->slot = slot_floats;
    c->flags_off = (2 * (size_t)world * slot_floats * sizeof(float) + 255) & ~(size_t)255;
    const size_t bytes = c->flags_off + 2 * (size_t)world * sizeof(unsigned);
    // peers store into this buffer over xGMI while kernels of this device poll and read it: fine-grained (uncached) memory where
    // the runtime can export it by IPC, plain device memory otherwise (same-device peers share the L2)
    const struct { unsigned flag; const char* name; } kinds[] = {
        {hipDeviceMallocUncached, "uncached"}, {hipDeviceMallocFinegrained, "fine-grained"}, {hipDeviceMallocDefault, "default"}};
    hipIpcMemHandle_t hd;
    const char* forced = getenv("SPRK_PEER_MEM");             // "default" | "fine-grained" | "uncached": pin the kind (experiments)
    for (const auto& k : kinds) {
        if (forced && strcmp(forced, k.name) != 0) continue;
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, bytes, k.flag) != hipSuccess || !p) { (void)hipGetLastError(); continue; }
        if (world == 1) memset(&hd, 0, sizeof(hd));               // a world of one exports nothing
        else if (hipIpcGetMemHandle(&hd, p) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); continue; }
        c->base = p; c->mem_kind = k.name;
        break;
    }
    if (!c->base) { delete c; return fail(SPRK_EHIP, "cannot allocate an IPC-exportable receive buffer of %zu bytes", bytes); }
    if (hipMemset(c->base, 0, bytes) != hipSuccess || hipMalloc((void**)&c->done, PEER_MAX_WORLD * sizeof(unsigned)) != hipSuccess ||
        hipMemset(c->done, 0, PEER_MAX_WORLD * sizeof(unsigned)) != hipSuccess || hipMalloc((void**)&c->err, sizeof(int)) != hipSuccess ||
        hipMemset(c->err, 0, sizeof(int)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(c->base); if (c->done) (void)hipFree(c->done); if (c->err) (void)hipFree(c->err);
        delete c;
        return fail(SPRK_EHIP, "peer buffer setup failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (const char* t = getenv("SPRK_PEER_TIMEOUT_MS")) { const long ms = atol(t); if (ms > 0) c->ticks = (unsigned long long)ms * 100000ull; }
    memcpy(handle_out, &hd, SPRK_PEER_HANDLE_BYTES);
    c->peer_base[rank] = c->base;
    *out = c;
    return SPRK_OK;
}

int sprk_peer_connect(sprk_peer c, const uint8_t* handles) {
    if (!c || !handles) return fail(SPRK_EINVAL, "NULL argument");
    if (c->connected) return fail(SPRK_ESTATE, "peer communicator is already connected");
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t hd;
        memcpy(&hd, handles + (size_t)p * SPRK_PEER_HANDLE_BYTES, sizeof(hd));
        void* q = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&q, hd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess || !q) { (void)hipGetLastError(); return fail(SPRK_EHIP, "hipIpcOpenMemHandle of rank %d's buffer: %s", p, hipGetErrorString(e)); }
        c->peer_base[p] = q;
    }
    c->connected = true;
    return SPRK_OK;
}

int sprk_peer_allgather_scores(sprk_peer c, const float* local, size_t count, const float** gathered, void* stream) {
    if (!c || !local || !gathered) return fail(SPRK_EINVAL, "NULL argument");
    if (!c->connected) return fail(SPRK_ESTATE, "peer all-gather before sprk_peer_connect");
    if (count > c->slot) return fail(SPRK_EINVAL, "count %zu exceeds the slot of %zu floats", count, c->slot);
    const unsigned e = ++c->epoch;
    const int parity = (int)(e & 1);
    *gathered = (const float*)c->base + (size_t)parity * c->world * c->slot;
    PeerPut a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < c->world; ++p) {
        a.dst[p] = (float*)c->peer_base[p] + ((size_t)parity * c->world + c->rank) * c->slot;
        a.flag[p] = (unsigned*)((char*)c->peer_base[p] + c->flags_off) + parity * c->world + c->rank;
    }
    a.src = local; a.count = count; a.epoch = e; a.world = c->world; a.done = c->done;
    long long bpp = ((long long)count / 4 + 1023) / 1024;      // ~4 sixteen-byte stores per thread
    if (bpp < 1) bpp = 1;
    if (bpp > 32) bpp = 32;
    a.blocks_per_peer = (int)bpp;
    hipLaunchKernelGGL(k_peer_put, dim3((unsigned)(c->world * bpp)), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const unsigned*)((const char*)c->base + c->flags_off) + parity * c->world, c->world, e, c->ticks, c->err);
    HIP_TRY(hipGetLastError());
    return SPRK_OK;
}

int sprk_peer_check(sprk_peer c, void* stream) {
    if (!c) return fail(SPRK_EINVAL, "communicator is NULL");
    int flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, c->err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (flag) {
        HIP_TRY(hipMemsetAsync(c->err, 0, sizeof(int), (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        return fail(SPRK_EHIP, "peer all-gather: a rank's slice did not arrive within the deadline (rank %d of %d, exchange %u)", c->rank, c->world, c->epoch);
    }
    return SPRK_OK;
}

const char* sprk_peer_memory_kind(sprk_peer c) { return c ? c->mem_kind : ""; }

void sprk_peer_destroy(sprk_peer c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank && c->peer_base[p]) (void)hipIpcCloseMemHandle(c->peer_base[p]);
    if (c->base) (void)hipFree(c->base);
    if (c->done) (void)hipFree(c->done);
    if (c->err) (void)hipFree(c->err);
    delete c;
}

}  // extern "C"

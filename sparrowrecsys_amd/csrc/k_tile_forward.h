// k_tile_forward.h -- the device-side plan, the cross hash, the plan interpreter k_tile_forward and the generic DIN stage k_din_pool.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.
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

static __global__ __launch_bounds__(256) void k_tile_forward(const DevPlan* __restrict__ P,
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
static __global__ __launch_bounds__(256) void k_din_pool(const DevPlan* __restrict__ P, const int* __restrict__ ids,
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
static __global__ __launch_bounds__(256) void k_fold_dense_rows(const float* __restrict__ table, long long vocab, int row_stride,
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
static __global__ __launch_bounds__(256) void k_zero_columns(float* __restrict__ Wt, int N, int ldw, int c0, int c1) {
    const int w = c1 - c0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < N * w; i += gridDim.x * 256) Wt[(size_t)(i / w) * ldw + c0 + i % w] = 0.f;
}


// k_operators.h -- stand-alone operator kernels (bit-exact gather, cross hash) -- closes the kernels' anonymous namespace.
// Part of sparrow_hip.hip (one translation unit); included there, not compilable on its own.

// ---------------------------------------------------------------------------------------------
// stand-alone operators
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_embedding_gather(const float* __restrict__ table, int V, int nvec,
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

static __global__ __launch_bounds__(256) void k_cross_hash(const int* __restrict__ a, const int* __restrict__ b, int B,
                                                    unsigned long long buckets, long long* __restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256)
        out[i] = (long long)cross_bucket(a[i], b[i], buckets);
}

}  // namespace sprk_dev
#pragma GCC visibility pop
using namespace sprk_dev;


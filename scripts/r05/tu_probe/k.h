#pragma once
#include <hip/hip_runtime.h>
namespace sprk_k {
template <int N>
__global__ void k_add(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += N; }
}

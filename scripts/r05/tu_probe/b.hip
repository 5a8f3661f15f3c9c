#include "k.h"
#include <cstdio>
namespace sprk_k { extern template __global__ void k_add<3>(float*, int); extern template __global__ void k_add<5>(float*, int); }
extern "C" int run(int which) {
    float* d; hipMalloc((void**)&d, 64 * 4); hipMemset(d, 0, 256);
    if (which == 3) hipLaunchKernelGGL((sprk_k::k_add<3>), dim3(1), dim3(64), 0, 0, d, 64);
    else hipLaunchKernelGGL((sprk_k::k_add<5>), dim3(1), dim3(64), 0, 0, d, 64);
    float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipFree(d);
    printf("%d -> %g %g\n", which, h[0], h[63]); return (int)h[0];
}

// util_kernels.hip -- small memory-bound helpers: stream copy (measured-peak probe),
// halo pack (gather of 16-byte elements).
#include "dev_common.h"

namespace tetsim {
namespace {

// 16 B per lane, grid-stride: one dwordx4 load + one dwordx4 store per element, 1 KiB per wave instruction.
__global__ __launch_bounds__(256) void copy16_kernel(const float4* __restrict__ src, float4* __restrict__ dst, uint64_t n) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void gather16_kernel(const float4* __restrict__ src, const int32_t* __restrict__ idx,
                                                       float4* __restrict__ dst, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

}  // namespace

void util_launch_copy(hipStream_t s, const float4* src, float4* dst, uint64_t n) {
    if (n == 0) return;
    // memory-bound: cap the grid at 256 CUs x 8 workgroups and grid-stride the rest
    const uint64_t want = (n + 255u) / 256u;
    const uint32_t grid = static_cast<uint32_t>(want < 2048u ? want : 2048u);
    hipLaunchKernelGGL(copy16_kernel, dim3(grid), dim3(256), 0, s, src, dst, n);
}
void util_launch_gather4(hipStream_t s, const float4* src, const int32_t* idx, float4* dst, uint32_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(gather16_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, src, idx, dst, n);
}

}  // namespace tetsim

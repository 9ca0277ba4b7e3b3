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

// strip float4 particles to tightly packed xyz, un-permuting the internal (Morton) numbering: out[3*a..] = src[map[a]]
__global__ __launch_bounds__(256) void pack_xyz_kernel(const float4* __restrict__ src, const uint32_t* __restrict__ map,
                                                       float* __restrict__ out, uint32_t n) {
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;
    if (a >= n) return;
    const float4 p = src[map ? map[a] : a];
    out[3 * a] = p.x; out[3 * a + 1] = p.y; out[3 * a + 2] = p.z;
}

// startGrab (Softbody.js:279-291): argmin over particles of the squared distance, evaluated in f64 exactly as the
// JS does (a0*a0 + a1*a1 + a2*a2, left to right; build with -ffp-contract=off), first minimum wins.  One candidate per
// workgroup; the host finishes over the few hundred candidates in index order.
__global__ __launch_bounds__(256) void nearest_kernel(const float4* __restrict__ pos, const uint32_t* __restrict__ map, uint32_t n,
                                                      double px, double py, double pz, double* __restrict__ best_d2,
                                                      uint32_t* __restrict__ best_id) {
    __shared__ double s_d[256];
    __shared__ uint32_t s_i[256];
    const uint32_t a = blockIdx.x * 256u + threadIdx.x;  // API particle index: ties must resolve to the smallest one
    double d2 = 1.7976931348623157e308;
    if (a < n) {
        const float4 p = pos[map ? map[a] : a];
        const double a0 = px - static_cast<double>(p.x), a1 = py - static_cast<double>(p.y), a2 = pz - static_cast<double>(p.z);
        d2 = a0 * a0 + a1 * a1 + a2 * a2;
        if (d2 != d2) d2 = 1.7976931348623157e308;  // `d2 < minD2` is false for NaN: such a particle is never picked
    }
    s_d[threadIdx.x] = d2;
    s_i[threadIdx.x] = a;
    __syncthreads();
    for (uint32_t w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            const double o = s_d[threadIdx.x + w];
            const uint32_t oi = s_i[threadIdx.x + w];
            if (o < s_d[threadIdx.x] || (o == s_d[threadIdx.x] && oi < s_i[threadIdx.x])) { s_d[threadIdx.x] = o; s_i[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { best_d2[blockIdx.x] = s_d[0]; best_id[blockIdx.x] = s_i[0]; }
}

// measurement aid of the loopback halo (TETSIM_DEBUG_LOOPBACK_DELAY_US): one wave that idles for `ticks` of the 100 MHz clock
__global__ void delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

}  // namespace

void util_launch_delay(hipStream_t s, uint32_t us) {
    if (us) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, 100ll * us);
}
void util_launch_pack_xyz(hipStream_t s, const float4* src, const uint32_t* map, float* out, uint32_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(pack_xyz_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, src, map, out, n);
}
void util_launch_nearest(hipStream_t s, const float4* pos, const uint32_t* map, uint32_t n, double px, double py, double pz,
                         double* best_d2, uint32_t* best_id) {
    if (n == 0) return;
    hipLaunchKernelGGL(nearest_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, pos, map, n, px, py, pz, best_d2, best_id);
}

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

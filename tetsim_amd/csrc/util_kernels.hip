// util_kernels.hip -- small memory-bound helpers: stream copy (measured-peak probe),
// halo pack (gather of 16-byte elements).
#include <algorithm>

#include "dev_common.h"

namespace tetsim {
namespace {

// Streaming probes: the MEASURED memory peak the roofline fractions are quoted against (MI355X_MICROARCH.md "HBM": 8 TB/s
// spec, ~6.3 TB/s achievable with a float4 copy).  A workgroup takes chunks of 256 x kUnroll float4s: every lane has kUnroll
// INDEPENDENT 16-byte accesses in flight (one load + one store per trip of a plain grid-stride loop -- the first version -- left the
// memory system a quarter idle: 4.84 TB/s at 1 GiB), optionally non-temporal (the data is touched once).  kKind: 0 copy, 1 read only
// (the sum keeps the loads alive; it is stored by no lane of a sane run), 2 write only.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int kKind, bool kNt, uint32_t kStreamUnroll>
__global__ __launch_bounds__(256) void stream_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, uint64_t n, v4f* __restrict__ sink) {
    const uint64_t chunk = 256ull * kStreamUnroll, chunks = (n + chunk - 1) / chunk;
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (uint64_t k = blockIdx.x; k < chunks; k += gridDim.x) {
        const uint64_t base = k * chunk + threadIdx.x;
        v4f a[kStreamUnroll];
        if constexpr (kKind != 2) {
#pragma unroll
            for (uint32_t u = 0; u < kStreamUnroll; u++) {
                const uint64_t i = base + 256ull * u;
                const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
                a[u] = i < n ? (kNt ? __builtin_nontemporal_load(src + i) : src[i]) : zero;
            }
        } else {
#pragma unroll
            for (uint32_t u = 0; u < kStreamUnroll; u++) { const v4f one = {1.0f, 2.0f, 3.0f, static_cast<float>(k)}; a[u] = one; }
        }
        if constexpr (kKind == 1) {
#pragma unroll
            for (uint32_t u = 0; u < kStreamUnroll; u++) acc += a[u];
        } else {
#pragma unroll
            for (uint32_t u = 0; u < kStreamUnroll; u++) {
                const uint64_t i = base + 256ull * u;
                if (i < n) { if constexpr (kNt) __builtin_nontemporal_store(a[u], dst + i); else dst[i] = a[u]; }
            }
        }
    }
    if constexpr (kKind == 1)
        if (acc.x == 123456.789f && acc.y == acc.z) sink[threadIdx.x] = acc;   // (never: the probe's buffers hold one repeated byte)
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

// tetsim_halo_p2p_probe: `reps` hand-overs with every neighbour at once -- lane k stores base + r into neighbour k's inbox word for this
// rank (peer memory, system scope) and waits for the neighbour's base + r in its own inbox; thread 0 stamps every repetition with the
// 100 MHz wall clock.  With every rank in the same loop a repetition costs ONE one-way signal latency (both directions travel at once).
__global__ void p2p_probe_kernel(P2PProbe p, uint32_t base, uint32_t reps, unsigned long long* ticks, uint32_t* error, uint32_t timeout_ms) {
    const uint32_t k = threadIdx.x;
    const long long limit = 100000ll * timeout_ms;
    for (uint32_t r = 1; r <= reps; r++) {
        const long long t0 = wall_clock64();
        if (k < p.n) {
            __hip_atomic_store(p.raise[k], base + r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            while (__hip_atomic_load(p.wait[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (base + r) > 0x7fffffffu) {   // (wrap-safe "<")
                __builtin_amdgcn_s_sleep(1);
                if (limit && wall_clock64() - t0 > limit) { __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const long long t1 = wall_clock64();
        if (k == 0) ticks[r - 1u] = static_cast<unsigned long long>(t1 - t0);
    }
}

}  // namespace

void util_launch_p2p_probe(hipStream_t s, const P2PProbe& p, uint32_t base, uint32_t reps, unsigned long long* ticks, uint32_t* error, uint32_t timeout_ms) {
    hipLaunchKernelGGL(p2p_probe_kernel, dim3(1), dim3(64), 0, s, p, base, reps, ticks, error, timeout_ms);
}
// A call's parameters into DevParams, in stream order: one lane, the 176 bytes arrive as kernel arguments.  (The copy this replaces --
// pinned slot, hipMemcpyAsync, an event per slot -- put the copy engine between two compute launches: most of the ~18 us the device idled
// in front of every call, tools/call_gap.py.)
__global__ void set_params_kernel(DevParams v, DevParams* dst) { *dst = v; }
void util_launch_set_params(hipStream_t s, DevParams* dst, const DevParams& v) { hipLaunchKernelGGL(set_params_kernel, dim3(1), dim3(1), 0, s, v, dst); }
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

// kind 0 copy / 1 read / 2 write; nt: non-temporal accesses; unroll: 4 or 8 independent 16-byte accesses per lane; grid: workgroups (0 = one per chunk)
template <int kKind, bool kNt, uint32_t kU>
static void stream_launch(hipStream_t s, uint32_t grid, const float4* src, float4* dst, uint64_t n) {
    const uint64_t chunks = (n + 256ull * kU - 1) / (256ull * kU);
    const uint32_t g = static_cast<uint32_t>(grid == 0u || chunks < grid ? std::min<uint64_t>(chunks, 0x7fffffffull) : grid);
    hipLaunchKernelGGL((stream_kernel<kKind, kNt, kU>), dim3(g), dim3(256), 0, s, reinterpret_cast<const v4f*>(src), reinterpret_cast<v4f*>(dst), n, reinterpret_cast<v4f*>(dst));
}
void util_launch_stream(hipStream_t s, int kind, bool nt, uint32_t unroll, uint32_t grid, const float4* src, float4* dst, uint64_t n) {
    if (n == 0) return;
    switch (kind * 4 + (nt ? 2 : 0) + (unroll >= 8u ? 1 : 0)) {
        case 0: stream_launch<0, false, 4>(s, grid, src, dst, n); break;
        case 1: stream_launch<0, false, 8>(s, grid, src, dst, n); break;
        case 2: stream_launch<0, true, 4>(s, grid, src, dst, n); break;
        case 3: stream_launch<0, true, 8>(s, grid, src, dst, n); break;
        case 4: stream_launch<1, false, 4>(s, grid, src, dst, n); break;
        case 5: stream_launch<1, false, 8>(s, grid, src, dst, n); break;
        case 6: stream_launch<1, true, 4>(s, grid, src, dst, n); break;
        case 7: stream_launch<1, true, 8>(s, grid, src, dst, n); break;
        case 8: stream_launch<2, false, 4>(s, grid, src, dst, n); break;
        case 9: stream_launch<2, false, 8>(s, grid, src, dst, n); break;
        case 10: stream_launch<2, true, 4>(s, grid, src, dst, n); break;
        default: stream_launch<2, true, 8>(s, grid, src, dst, n); break;
    }
}
void util_launch_copy(hipStream_t s, const float4* src, float4* dst, uint64_t n) { util_launch_stream(s, 0, false, 4u, 0u, src, dst, n); }
void util_launch_gather4(hipStream_t s, const float4* src, const int32_t* idx, float4* dst, uint32_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(gather16_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, src, idx, dst, n);
}

}  // namespace tetsim

// dev_store.h -- write-through stores the COMPILER can see (device code only).
//
// Streamed results are written write-through (sc0 sc1): a plain store leaves the line dirty in the XCD's 4 MiB L2 and the
// kernel pays the write-back at its end (pj_blocked.hip).  The first version issued them from an asm statement.  gfx9 has one
// counter (vmcnt) for loads AND stores, decremented in order, and the compiler's wait-count model cannot see a store inside
// an asm statement: wherever it later waited for an older load it emitted vmcnt(0) -- which on the hardware also waits for
// the invisible stores' full trip to HBM (>1 us per wave, e.g. between the tet kernel's result stores and its partial-sum
// reduction) -- and the asm's memory clobber made it re-read kernel parameters from memory behind every store.  A raw buffer
// store with the cache-policy bits is an ordinary store to the compiler: counted, hazard-checked, no clobber.
#pragma once
#include <hip/hip_runtime.h>

// units of 64 clocks between two looks of a wave that waits for stamped data of another workgroup (pj_blocked.hip: pjb_call_kernel,
// nh_kernels.inc: nh_sweep1_kernel / nh_call_kernel); -DTETSIM_POLL_SLEEP=n builds an A/B variant
// (profiles/r06_poll_sleep.txt: the polar call kernel does not care -- 2 / 8 / 32: level --, the Gauss-Seidel chain does: 40.9-42.3 / 41.6-42.8 /
// 44.3-45.6 us per substep)
#ifndef TETSIM_POLL_SLEEP
#define TETSIM_POLL_SLEEP 8
#endif
#ifndef TETSIM_NH_POLL_SLEEP
#define TETSIM_NH_POLL_SLEEP 2
#endif

namespace tetsim {

constexpr uint32_t kStoreWtMaxIndex = 1u << 27;  // 32-bit byte offsets: a float4 array of up to 2 GiB

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));

// base must be wave-uniform (it becomes the buffer descriptor in SGPRs); index < kStoreWtMaxIndex
__device__ __forceinline__ void store_wt(float4* base, uint32_t index, const float4& v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const v4u_t x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, static_cast<int>(index * 16u), 0, 0x11);  // aux: sc0 | sc1
}

// ... one float (the ninth of a lean-state tet record)
__device__ __forceinline__ void store_wt1(float* base, uint32_t index, float v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, static_cast<int>(index * 4u), 0, 0x11);  // aux: sc0 | sc1
}

// Exchange between workgroups that share ONE XCD (pj_blocked.hip: the frame kernel's tile partial sums when the host has placed a
// body's tiles on one XCD): the store is a plain one -- the CU's L1 writes through, the line stays in the XCD's L2 --, the load
// carries agent scope (sc1): it misses the L1 and is served by that L2.  Measured on MI355X (profiles/archive/r03_frame_kernel.txt): bit-equal
// results, 6.1 instead of 6.5 us per Dragon substep against the memory-side pair below.  NOT coherent across XCDs (a line dirty in
// another XCD's L2 is invisible), and a workgroup-scope load (sc0) is served by the stale L1 line for ever; an L1 invalidate
// (buffer_inv sc1) + plain loads works too but costs 8.1 us.
__device__ __forceinline__ void store_plain(float4* base, uint32_t index, const float4& v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const v4u_t x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, static_cast<int>(index * 16u), 0, 0);
}
__device__ __forceinline__ float4 load_l2(const float4* base, uint32_t index) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(base), 0, 0x7fffffff, 0x00020000);
    const v4u_t x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(index * 16u), 0, 0x10);   // aux: sc1
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}

// The matching load: served from the memory side (sc0 sc1), never from a line this XCD's L2 (or this CU's L1) cached before another
// workgroup's write-through store replaced it -- what a reader needs when no kernel boundary (with its cache invalidation) lies
// between the store and the load (pj_blocked.hip: the persistent frame kernel).  Same constraints as store_wt.
__device__ __forceinline__ float4 load_coherent(const float4* base, uint32_t index) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(base), 0, 0x7fffffff, 0x00020000);
    const v4u_t x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(index * 16u), 0, 0x11);   // aux: sc0 | sc1
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}

__device__ __forceinline__ float load_coherent1(const float* base, uint32_t index) {   // ... one float
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, static_cast<int>(index * 4u), 0, 0x11));   // aux: sc0 | sc1
}

}  // namespace tetsim

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

namespace tetsim {

constexpr uint32_t kStoreWtMaxIndex = 1u << 27;  // 32-bit byte offsets: a float4 array of up to 2 GiB

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));

// base must be wave-uniform (it becomes the buffer descriptor in SGPRs); index < kStoreWtMaxIndex
__device__ __forceinline__ void store_wt(float4* base, uint32_t index, const float4& v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const v4u_t x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, static_cast<int>(index * 16u), 0, 0x11);  // aux: sc0 | sc1
}

}  // namespace tetsim

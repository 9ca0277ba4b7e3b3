// handoff.hip -- what ONE hand-over of a tagged 8/16-byte value between two workgroups costs on MI355X, by how it is stored and looked at
// (development micro-benchmark behind pj_quad.hip's exchange of tile partial sums):
//   W workgroups in a ring, all on ONE XCD (block i of a grid runs on XCD i % 8: only every eighth block takes part) or spread over all;
//   round r: workgroup k stores (value, r) into its slot, then looks at slot k+1 until it carries r.  The ring runs in lock-step, so
//   cycles per round = the hand-over latency (store -> visible -> seen), the same chain a substep of the persistent frame kernel has.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/handoff.hip -o /tmp/handoff && /tmp/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000); }

enum { kPlainSc1 = 0, kWtCoherent = 1, kPlainAtomicOr = 2, kAtomicXchgAtomicOr = 3, kPlainScalarGlc = 4, kPlainAtomicAdd32x2 = 5 };

template <int kHow>
__global__ __launch_bounds__(64) void ring(unsigned long long* slots, uint32_t workgroups, uint32_t stride_blocks, uint32_t rounds, unsigned long long* out) {
    if (blockIdx.x % stride_blocks != 0) return;
    const uint32_t k = blockIdx.x / stride_blocks;
    if (k >= workgroups) return;
    unsigned long long* mine = slots + 16ull * k;                       // 128 bytes apart: one cache line per slot
    unsigned long long* next = slots + 16ull * ((k + 1u) % workgroups);
    unsigned long long polls = 0, dead = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t r = 1; r <= rounds; r++) {
        const unsigned long long word = (static_cast<unsigned long long>(r) << 32) | (k * 7u + r);   // (value, tag) in 8 bytes
        if (threadIdx.x == 0) {
            if constexpr (kHow == kPlainSc1) __builtin_amdgcn_raw_buffer_store_b64(v2u{static_cast<uint32_t>(word), static_cast<uint32_t>(word >> 32)}, rsrc(mine), 0, 0, 0);
            else if constexpr (kHow == kWtCoherent) __builtin_amdgcn_raw_buffer_store_b64(v2u{static_cast<uint32_t>(word), static_cast<uint32_t>(word >> 32)}, rsrc(mine), 0, 0, 0x11);
            else if constexpr (kHow == kAtomicXchgAtomicOr) __hip_atomic_store(mine, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *reinterpret_cast<volatile unsigned long long*>(mine) = word;
        }
        for (;;) {
            asm volatile("" ::: "memory");   // (the looks are loads of one address in a loop without stores: keep them IN the loop)
            unsigned long long seen;
            if constexpr (kHow == kPlainSc1) { const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rsrc(next), 0, 0, 0x10); seen = (static_cast<unsigned long long>(x.y) << 32) | x.x; }
            else if constexpr (kHow == kWtCoherent) { const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rsrc(next), 0, 0, 0x11); seen = (static_cast<unsigned long long>(x.y) << 32) | x.x; }
            else if constexpr (kHow == kPlainAtomicOr || kHow == kAtomicXchgAtomicOr) seen = __hip_atomic_fetch_or(next, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if constexpr (kHow == kPlainAtomicAdd32x2) {
                const uint32_t hi = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(next) + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                seen = static_cast<unsigned long long>(hi) << 32;
            } else {
                unsigned long long v;
                asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(next) : "memory");
                seen = v;
            }
            polls++;
            if (static_cast<uint32_t>(seen >> 32) >= r) break;
            if (polls > 40ull * rounds + 100000ull) { r = rounds; dead = 1; break; }   // (a look that is served stale for ever must not wedge the GPU)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2ull * k] = dead ? 0ull : t1 - t0; out[2ull * k + 1] = polls; }
}

template <int kHow>
void run(const char* name, unsigned long long* slots, unsigned long long* out, uint32_t workgroups, uint32_t stride, uint32_t rounds) {
    hipMemset(slots, 0, 128ull * workgroups);
    hipLaunchKernelGGL(ring<kHow>, dim3(workgroups * stride), dim3(64), 0, 0, slots, workgroups, stride, rounds, out);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%-58s FAILED\n", name); return; }
    std::vector<unsigned long long> h(2ull * workgroups);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, polls = 0;
    bool dead = false;
    for (uint32_t k = 0; k < workgroups; k++) { cyc += static_cast<double>(h[2 * k]); polls += static_cast<double>(h[2 * k + 1]); dead = dead || h[2 * k] == 0; }
    if (dead) { printf("%-58s %2u workgroups %s: NEVER SEEN (the look is served stale)\n", name, workgroups, stride == 8 ? "on one XCD  " : "over all XCDs"); fflush(stdout); return; }
    // (s_memtime counts shader-clock cycles, ~1.9-2.1 GHz while this runs)
    printf("%-58s %2u workgroups %s: %7.0f cycles per hand-over, %5.2f looks\n", name, workgroups, stride == 8 ? "on one XCD  " : "over all XCDs", cyc / workgroups / rounds,
           polls / workgroups / rounds);
    fflush(stdout);
}

int main() {
    unsigned long long *slots, *out;
    hipMalloc(&slots, 128 * 1024);
    hipMalloc(&out, 16 * 1024);
    const uint32_t rounds = 2000;
    for (uint32_t stride : {8u, 1u})
        for (uint32_t w : {2u, 16u, 62u}) {
            run<kPlainSc1>("plain store, sc1 load (pj_quad local)", slots, out, w, stride, rounds);
            run<kWtCoherent>("sc0 sc1 store, sc0 sc1 load (memory side)", slots, out, w, stride, rounds);
            run<kPlainAtomicOr>("plain store, returning 64-bit atomic or", slots, out, w, stride, rounds);
            run<kAtomicXchgAtomicOr>("agent atomic store, returning 64-bit atomic or", slots, out, w, stride, rounds);
            run<kPlainAtomicAdd32x2>("plain store, returning 32-bit atomic add (tag only)", slots, out, w, stride, rounds);
            if (stride == 8u) run<kPlainScalarGlc>("plain store, scalar load glc", slots, out, w, stride, rounds);
        }
    return 0;
}

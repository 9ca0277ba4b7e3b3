// handover_latency.hip -- how long does ONE stamped hand-over between two workgroups take on this chip?  (measurement tool, not product)
// Two waves play ping-pong through 16-byte records {x, y, z, stamp}: A stores stamp 2i+1 and waits for 2i+2, B the other way round.
// Half the round trip = store -> visible -> seen by a polling load: the floor under every link of the Gauss-Seidel chain
// (nh_kernels.inc: nh_call_kernel) and of the polar call / frame kernels' exchanges.
//   variants: cache policy of the store / of the load, the partner on the same XCD or another, 1 or 64 lanes (64 records in 64 lines), s_sleep between looks
// build: hipcc -O3 --offload-arch=gfx950 -o tools/micro/bin/handover_latency tools/micro/handover_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kAux> __device__ __forceinline__ void store16(float4* base, uint32_t index, v4u_t x) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(x, rsrc, static_cast<int>(index * 16u), 0, kAux);
}
template <int kAux> __device__ __forceinline__ v4u_t load16(float4* base, uint32_t index) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(index * 16u), 0, kAux);
}

// blocks a_block and b_block play; every other block leaves at once.  lanes: how many lanes of the wave take part (each its own record, `stride` records apart)
template <int kStoreAux, int kLoadAux, int kSleep>
__global__ __launch_bounds__(64) void pingpong(float4* ab, float4* ba, uint32_t a_block, uint32_t b_block, uint32_t rounds, uint32_t lanes, uint32_t stride, long long* ticks) {
    const bool is_a = blockIdx.x == a_block, is_b = blockIdx.x == b_block;
    if (!is_a && !is_b) return;
    const uint32_t lane = threadIdx.x;
    const bool on = lane < lanes;
    float4* const mine = is_a ? ab : ba;
    float4* const theirs = is_a ? ba : ab;
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    for (uint32_t i = 0; i < rounds; i++) {
        const uint32_t send = 2u * i + (is_a ? 1u : 2u), want = 2u * i + (is_a ? 2u : 1u);
        if (is_a && on) store16<kStoreAux>(mine, lane * stride, v4u_t{lane, i, 7u, send});
        bool pend = on;
        do {
            if (kSleep) __builtin_amdgcn_s_sleep(kSleep);
            asm volatile("" ::: "memory");
            if (pend) { const v4u_t x = load16<kLoadAux>(theirs, lane * stride); pend = x.w != want; }
        } while (__builtin_amdgcn_ballot_w64(pend) != 0ull);
        if (is_b && on) store16<kStoreAux>(mine, lane * stride, v4u_t{lane, i, 7u, send});
    }
    if (lane == 0 && is_a) { ticks[0] = wall_clock64() - w0; ticks[1] = __builtin_readcyclecounter() - t0; }
}

template <int kStoreAux, int kLoadAux, int kSleep>
double run(float4* ab, float4* ba, long long* ticks, uint32_t a_block, uint32_t b_block, uint32_t lanes, uint32_t stride) {
    const uint32_t rounds = 2000;
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipMemset(ab, 0, 64 * 64 * 16));
        CHECK(hipMemset(ba, 0, 64 * 64 * 16));
        hipLaunchKernelGGL((pingpong<kStoreAux, kLoadAux, kSleep>), dim3(16), dim3(64), 0, 0, ab, ba, a_block, b_block, rounds, lanes, stride, ticks);
        CHECK(hipDeviceSynchronize());
        long long h[2];
        CHECK(hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost));
        const double us = h[0] / 100.0 / (2.0 * rounds);   // 100 MHz clock; one way
        if (us < best) best = us;
    }
    return best;
}

int main() {
    float4 *ab, *ba;
    long long* ticks;
    CHECK(hipMalloc(&ab, 64 * 64 * 16));
    CHECK(hipMalloc(&ba, 64 * 64 * 16));
    CHECK(hipMalloc(&ticks, 16));
    printf("one-way hand-over latency (store -> seen by the partner's polling load), best of 5 x 2000 round trips, microseconds\n");
    printf("%-44s %10s %10s %10s %10s\n", "store / load policy, sleep", "xcd 1 lane", "xcd 64 ln", "same 1 ln", "same 64 ln");
    // blocks 0 and 1: neighbouring XCDs; blocks 0 and 8: the same XCD (round-robin dispatch over 8 XCDs)
#define ROW(name, S, L, SL) printf("%-44s %10.3f %10.3f %10.3f %10.3f\n", name, run<S, L, SL>(ab, ba, ticks, 0, 1, 1, 1), run<S, L, SL>(ab, ba, ticks, 0, 1, 64, 37), \
                                   run<S, L, SL>(ab, ba, ticks, 0, 8, 1, 1), run<S, L, SL>(ab, ba, ticks, 0, 8, 64, 37)); fflush(stdout)
    ROW("store sc0 sc1 / load sc0 sc1, no sleep", 0x11, 0x11, 0);
    ROW("store sc0 sc1 / load sc0 sc1, s_sleep 2", 0x11, 0x11, 2);
    ROW("store sc0 sc1 / load sc0 sc1, s_sleep 8", 0x11, 0x11, 8);
    ROW("store sc1 / load sc0 sc1, no sleep", 0x10, 0x11, 0);
    ROW("store sc0 sc1 nt / load sc0 sc1 nt, no sleep", 0x13, 0x13, 0);
    return 0;
}

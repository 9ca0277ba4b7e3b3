// xq_value_ops.hip -- hand-over between two streams through a device word (development micro-benchmark):
//   (a) one-wave signal / wait KERNELS (what tetsim_halo.hip uses)   (b) hipStreamWriteValue32 / hipStreamWaitValue32
// Ping-pong: stream A: work, signal(i) ... stream B: wait(i), work, signal'(i) ... A: wait'(i) -- per hop = 2 hand-overs + 2 kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xq_value_ops.hip -o /tmp/xqv && /tmp/xqv
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(int* p, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (p && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1);
}
__global__ void sig(uint32_t* w, uint32_t v) { if (threadIdx.x == 0) __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void wt(const uint32_t* w, uint32_t v) {
    if (threadIdx.x == 0) while (static_cast<int32_t>(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) __builtin_amdgcn_s_sleep(4);
}
int main() {
    int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
    uint32_t *w = nullptr, *w2 = nullptr;   // (signal memory comes in 8-byte allocations)
    if (hipExtMallocWithFlags(reinterpret_cast<void**>(&w), 8, hipMallocSignalMemory) != hipSuccess ||
        hipExtMallocWithFlags(reinterpret_cast<void**>(&w2), 8, hipMallocSignalMemory) != hipSuccess) { printf("signal memory alloc failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    hipMemset(w, 0, 8); hipMemset(w2, 0, 8);
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    const int hops = 200; const long long cyc = 2000;  // 20 us of "work" per kernel (100 MHz wall clock): the host runs ahead, the hand-overs' DEVICE cost shows
    uint32_t seq = 0;
    auto run = [&](int mode) {   // 0: kernels, 1: value ops, 2: no hand-over at all (two independent streams, lower bound)
        for (int i = 0; i < hops; i++) {
            ++seq;
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, a, d, cyc);
            if (mode == 0) { hipLaunchKernelGGL(sig, dim3(1), dim3(64), 0, a, w, seq); hipLaunchKernelGGL(wt, dim3(1), dim3(64), 0, b, w, seq); }
            if (mode == 1) { hipStreamWriteValue32(a, w, seq, 0); hipStreamWaitValue32(b, w, seq, hipStreamWaitValueGte, 0xffffffffu); }
            hipLaunchKernelGGL(work, dim3(256), dim3(256), 0, b, d, cyc);
            if (mode == 0) { hipLaunchKernelGGL(sig, dim3(1), dim3(64), 0, b, w2, seq); hipLaunchKernelGGL(wt, dim3(1), dim3(64), 0, a, w2, seq); }
            if (mode == 1) { hipStreamWriteValue32(b, w2, seq, 0); hipStreamWaitValue32(a, w2, seq, hipStreamWaitValueGte, 0xffffffffu); }
        }
    };
    const char* names[3] = {"one-wave signal/wait kernels", "hipStreamWrite/WaitValue32", "no hand-over (independent streams)"};
    for (int mode : {2, 0, 1, 0, 1}) {
        run(mode); hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; r++) run(mode);
        const auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        const auto t2 = std::chrono::steady_clock::now();
        printf("%-38s %6.2f us per hop (2 work kernels + 2 hand-overs), host enqueue %6.2f us per hop\n", names[mode],
               std::chrono::duration<double, std::micro>(t2 - t0).count() / (5.0 * hops), std::chrono::duration<double, std::micro>(t1 - t0).count() / (5.0 * hops));
    }
    return 0;
}

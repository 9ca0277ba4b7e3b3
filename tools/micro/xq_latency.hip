// xq_latency.hip -- what a dependency between two kernels costs on this stack (development micro-benchmark):
//   same stream | alternating between two streams (event record + wait per hop), each eagerly and from a captured graph.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/xq_latency.hip -o /tmp/xq && /tmp/xq
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(int* p, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (p && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1);
}
int main() {
    const int hops = 200;
    hipStream_t s[3];
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(hops);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    const long long cyc = 200;  // ~2 us at 100 MHz clock64 ... whatever it is, the same in every variant
    auto chain = [&](int nstreams) {
        for (int i = 0; i < hops; i++) {
            hipStream_t st = s[i % nstreams];
            if (nstreams > 1 && i > 0) hipStreamWaitEvent(st, ev[i - 1], 0);
            hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, cyc);
            if (nstreams > 1) hipEventRecord(ev[i], st);
        }
    };
    auto timed = [&](const char* name, int nstreams, bool graph) {
        hipGraphExec_t exec = nullptr;
        if (graph) {
            hipGraph_t g;
            hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal);
            chain(nstreams);
            if (nstreams > 1) for (int k = 1; k < nstreams; k++) hipStreamWaitEvent(s[0], ev[hops - 1 - ((hops - 1) % nstreams == k ? 0 : 0)], 0);
            if (nstreams > 1) hipStreamWaitEvent(s[0], ev[hops - 1], 0), hipStreamWaitEvent(s[0], ev[hops - 2], 0), hipStreamWaitEvent(s[0], ev[hops - 3], 0);
            if (hipStreamEndCapture(s[0], &g) != hipSuccess) { printf("%s: capture failed\n", name); return; }
            hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
            hipGraphLaunch(exec, s[0]);
            hipDeviceSynchronize();
        } else { chain(nstreams); hipDeviceSynchronize(); }
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < 5; r++) { if (graph) hipGraphLaunch(exec, s[0]); else chain(nstreams); }
        hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * hops);
        printf("%-34s %6.2f us per kernel (kernel itself + dependency)\n", name, us);
    };
    timed("1 stream, eager", 1, false);
    timed("1 stream, graph", 1, true);
    timed("2 streams alternating, eager", 2, false);
    timed("2 streams alternating, graph", 2, true);
    timed("3 streams rotating, eager", 3, false);
    timed("3 streams rotating, graph", 3, true);
    return 0;
}

// Do event-record nodes inside a captured HIP graph give usable timestamps on this stack?  (bench: in-graph duration of a kernel)
//   hipcc --offload-arch=gfx950 -O2 -o bin/graph_events graph_events.hip && bin/graph_events
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(float* p, int n) { float a = p[threadIdx.x]; for (int i = 0; i < n; i++) a = a * 1.0001f + 0.5f; p[threadIdx.x + blockIdx.x * blockDim.x] = a; }
int main() {
    float* d; CK(hipMalloc(&d, 4096 * 256 * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int K = 20;
    std::vector<hipEvent_t> ev(2 * K);
    for (auto& e : ev) CK(hipEventCreate(&e));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipGraph_t g; hipGraphExec_t ge;
    // plain graph: K kernels back to back
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < K; i++) spin<<<4096, 256, 0, s>>>(d, 2000);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; r++) { CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("plain graph: %.2f us per kernel slot\n", ms * 1e3 / K); }
    // graph with event-record nodes around every kernel
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < K; i++) { CK(hipEventRecord(ev[2 * i], s)); spin<<<4096, 256, 0, s>>>(d, 2000); CK(hipEventRecord(ev[2 * i + 1], s)); }
    CK(hipStreamEndCapture(s, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge2, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("graph with event nodes: %.2f us per kernel slot;", ms * 1e3 / K);
        float sum = 0; for (int i = 0; i < K; i++) { float k; CK(hipEventElapsedTime(&k, ev[2 * i], ev[2 * i + 1])); sum += k; }
        printf(" kernels by their event nodes: %.2f us each\n", sum * 1e3 / K);
    }
    // eager with hipExtLaunchKernelGGL events
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < K; i++) hipExtLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, d, 2000);
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); printf("eager, per-launch events: %.2f us per kernel slot;", ms * 1e3 / K);
        float sum = 0; for (int i = 0; i < K; i++) { float k; CK(hipEventElapsedTime(&k, ev[2 * i], ev[2 * i + 1])); sum += k; }
        printf(" kernels by their events: %.2f us each\n", sum * 1e3 / K);
    }
    return 0;
}

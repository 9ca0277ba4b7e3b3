// valu_rate.hip -- issue rate of plain vs packed f32 FMA and of the transcendentals on gfx950 (development micro-benchmark).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
// Each wave runs N instructions on 8 independent accumulator chains; waves per SIMD are swept (1, 2, 4, 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;

__global__ void k_fma(float* out, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pk(float* out, float a, float b) {
    v2f x[8]; v2f aa = {a, a}, bb = {b, b};
    for (int i = 0; i < 8; i++) { x[i].x = threadIdx.x * 1e-3f + i; x[i].y = x[i].x + 0.5f; }
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(aa), "v"(bb));
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define TRANS_KERNEL(name, insn)                                                              \
    __global__ void name(float* out, float a, float b) {                                      \
        float x[8];                                                                           \
        for (int i = 0; i < 8; i++) x[i] = 1.0f + threadIdx.x * 1e-3f + i;                    \
        for (int it = 0; it < kIters; it++) {                                                 \
            _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(insn " %0, %0" : "+v"(x[i])); \
        }                                                                                     \
        float s = 0; for (int i = 0; i < 8; i++) s += x[i];                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                       \
    }
TRANS_KERNEL(k_rcp, "v_rcp_f32")
TRANS_KERNEL(k_rsq, "v_rsq_f32")
TRANS_KERNEL(k_sin, "v_sin_f32")
__global__ void k_mix(float* out, float a, float b) {  // 3 fma : 1 rcp, independent chains
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = 1.0f + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x[6]));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x[7]));
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int waves_per_simd, double flops_per_insn) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int threads = 64 * waves_per_simd * 4;  // one workgroup per CU filling 4 SIMDs
    float* out; hipMalloc(&out, sizeof(float) * cus * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, out, 0.999f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, out, 0.999f, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insn_per_simd = double(kIters) * 8 * waves_per_simd;
    const double clk = p.clockRate * 1e3;  // Hz
    printf("%-8s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction at %.0f MHz (%.1f TFLOP/s)\n", name, waves_per_simd, ms,
           ms * 1e-3 * clk / insn_per_simd, clk / 1e6, flops_per_insn * 64 * insn_per_simd * cus * 4 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run("fma", k_fma, w, 2); run("pk_fma", k_pk, w, 4); run("rcp", k_rcp, w, 0); run("rsq", k_rsq, w, 0); run("sin", k_sin, w, 0); run("6fma+2rcp", k_mix, w, 0);
    }
    return 0;
}

// skin_kernels.hip -- embedded visual-mesh skinning on the device.
// Reference: updateVisMesh (Softbody.js:259-277, CPU, f64 arithmetic with f32 stores) and the vertex-shader
// injection of SoftbodyGPU.js:424-448 (f32; normals rotated by the tet quaternion, :440).
// Build with -ffp-contract=off: the JS-order variant must not fuse multiply-add.
#include "dev_common.h"

namespace tetsim {
namespace {

template <bool JS>
__global__ __launch_bounds__(256) void skin_kernel(SkinDev d, const float4* __restrict__ pos, const float4* __restrict__ quat) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= d.nvis) return;
    const int4 c = d.corner[i];
    const float4 w = d.weight[i];
    const float4 p0 = pos[c.x], p1 = pos[c.y], p2 = pos[c.z], p3 = pos[c.w];
    float4 o;
    if constexpr (JS) {
        // b3 = 1.0 - b0 - b1 - b2 in f64; positions = 0; then four `a += p*b` steps, each f64 multiply + add, f32 store
        const double b0 = w.x, b1 = w.y, b2 = w.z, b3 = 1.0 - b0 - b1 - b2;
        auto acc = [&](float a0, float a1, float a2, float a3) {
            float a = 0.0f;
            a = static_cast<float>(static_cast<double>(a) + static_cast<double>(a0) * b0);
            a = static_cast<float>(static_cast<double>(a) + static_cast<double>(a1) * b1);
            a = static_cast<float>(static_cast<double>(a) + static_cast<double>(a2) * b2);
            a = static_cast<float>(static_cast<double>(a) + static_cast<double>(a3) * b3);
            return a;
        };
        o = make_float4(acc(p0.x, p1.x, p2.x, p3.x), acc(p0.y, p1.y, p2.y, p3.y), acc(p0.z, p1.z, p2.z, p3.z), 0.0f);
    } else {
        // lastTetWeight = 1.0 - (y + z + w); ((p0*b0 + p1*b1) + p2*b2) + p3*b3, f32 (SoftbodyGPU.js:431-435)
        const float b3 = 1.0f - ((w.x + w.y) + w.z);
        o = make_float4(((p0.x * w.x + p1.x * w.y) + p2.x * w.z) + p3.x * b3,
                        ((p0.y * w.x + p1.y * w.y) + p2.y * w.z) + p3.y * b3,
                        ((p0.z * w.x + p1.z * w.y) + p2.z * w.z) + p3.z * b3, 0.0f);
    }
    d.out_pos[i] = o;
    if (d.out_nrm && d.normal0 && quat) {  // transformedNormal = Rotate(objectNormal, tetQuaternion), :428,440
        const float4 q = quat[d.qidx[i]];
        const float4 n = d.normal0[i];
        // v + 2 * cross(q.xyz, cross(q.xyz, v) + q.w * v)
        const float ix = (q.y * n.z - n.y * q.z) + q.w * n.x;
        const float iy = (q.z * n.x - n.z * q.x) + q.w * n.y;
        const float iz = (q.x * n.y - n.x * q.y) + q.w * n.z;
        d.out_nrm[i] = make_float4(n.x + 2.0f * (q.y * iz - iy * q.z), n.y + 2.0f * (q.z * ix - iz * q.x),
                                   n.z + 2.0f * (q.x * iy - ix * q.y), 0.0f);
    }
}

}  // namespace

void skin_launch(hipStream_t s, const SkinDev& d, const float4* pos, const float4* quat, bool js_order) {
    if (d.nvis == 0) return;
    const dim3 grid((d.nvis + 255u) / 256u), block(256);
    if (js_order) hipLaunchKernelGGL(skin_kernel<true>, grid, block, 0, s, d, pos, quat);
    else hipLaunchKernelGGL(skin_kernel<false>, grid, block, 0, s, d, pos, quat);
}

}  // namespace tetsim

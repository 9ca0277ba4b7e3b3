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

// three.js r160 BufferGeometry.computeVertexNormals (indexed branch) + normalizeNormals, as a gather: the reference walks the
// triangles in order and adds each face normal to its three vertices through Float32Array stores, so vertex v's normal is the
// f32-rounded running sum over ITS triangles in triangle order -- which one lane per vertex reproduces exactly.
//   cb = pC - pB; ab = pA - pB; cb.cross(ab)  (Vector3: f64);   n[v] = f32(f64(n[v]) + cb);   n = f32(n * (1 / (|n| || 1)))
__global__ __launch_bounds__(256) void vertex_normals_kernel(SkinDev d) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= d.nvis) return;
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;
    const float4* const P = d.tri_pos ? d.tri_pos : d.out_pos;
    for (uint32_t q = d.vt_off[v]; q < d.vt_off[v + 1]; q++) {
        const int4 t = d.tri[d.vt_tri[q]];
        const float4 a = P[t.x], b = P[t.y], c = P[t.z];
        const double cbx = static_cast<double>(c.x) - static_cast<double>(b.x), cby = static_cast<double>(c.y) - static_cast<double>(b.y),
                     cbz = static_cast<double>(c.z) - static_cast<double>(b.z);
        const double abx = static_cast<double>(a.x) - static_cast<double>(b.x), aby = static_cast<double>(a.y) - static_cast<double>(b.y),
                     abz = static_cast<double>(a.z) - static_cast<double>(b.z);
        const double x = cby * abz - cbz * aby, y = cbz * abx - cbx * abz, z = cbx * aby - cby * abx;   // Vector3.cross
        nx = static_cast<float>(static_cast<double>(nx) + x);
        ny = static_cast<float>(static_cast<double>(ny) + y);
        nz = static_cast<float>(static_cast<double>(nz) + z);
    }
    const double X = nx, Y = ny, Z = nz;
    double len = sqrt(X * X + Y * Y + Z * Z);          // Vector3.length
    if (len == 0.0 || len != len) len = 1.0;            // `length() || 1` (0 and NaN are falsy)
    const double s = 1.0 / len;                         // divideScalar(s) = multiplyScalar(1 / s)
    d.out_vnrm[v] = make_float4(static_cast<float>(X * s), static_cast<float>(Y * s), static_cast<float>(Z * s), 0.0f);
}

}  // namespace

void skin_launch_vertex_normals(hipStream_t s, const SkinDev& d) {
    if (d.nvis == 0) return;
    hipLaunchKernelGGL(vertex_normals_kernel, dim3((d.nvis + 255u) / 256u), dim3(256), 0, s, d);
}

void skin_launch(hipStream_t s, const SkinDev& d, const float4* pos, const float4* quat, bool js_order) {
    if (d.nvis == 0) return;
    const dim3 grid((d.nvis + 255u) / 256u), block(256);
    if (js_order) hipLaunchKernelGGL(skin_kernel<true>, grid, block, 0, s, d, pos, quat);
    else hipLaunchKernelGGL(skin_kernel<false>, grid, block, 0, s, d, pos, quat);
}

}  // namespace tetsim

// pj_blocked.hip -- POLAR_JACOBI, blocked formulation, FAST arithmetic (gfx950, wave64).
//
// Same substep as pj_kernels.inc (P3+P4 per tet, P5+P6+P7 (+P1,P2 of the next substep) per particle), but the
// scatter between the two kernels is restructured around workgroup tiles so that HBM traffic drops from
// ~270 B to ~180 B per tet-solve and almost all random 16-byte gathers become LDS reads:
//
//   pjb_tet_kernel     one workgroup per tile (<= 256 tets, <= 256 distinct particles):
//     1. the tile's particle positions are loaded once into LDS (one gather per distinct particle instead of
//        one per corner: 131 instead of 1024 on the lattice);
//     2. each lane solves one tet from LDS-resident corners (rotation extraction, goals), streams its carried
//        rest shape (48 B, packed) and quaternion in and out with 16-byte coalesced accesses, and leaves
//        (V*goal, V) for its 4 corners in LDS;
//     3. lanes switch roles -- one lane per tile PARTICLE -- and add up the corner goals of that particle in
//        LDS, in a fixed host-built order (deterministic, no atomics), storing ONE partial sum per
//        (tile, particle), coalesced.
//   pjb_vertex_kernel  one lane per particle: adds the 1..9 (2.9 on average) partial sums of the tiles that touch
//        it instead of gathering ~23 goals, then collides / integrates exactly like the gather formulation.
//
// Result differs from the gather formulation only by summation order (tile partials) -- tolerance-level, FAST
// mode only; PRECISE keeps the reference's slot order.
#define TETSIM_FAST 1
#include "dev_common.h"

namespace tetsim {
namespace {

#include "pj_math.inc"

__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t tiles_per_xcd) { return (b & 7u) * tiles_per_xcd + (b >> 3); }

constexpr uint32_t kTile = 256;

__global__ __launch_bounds__(256) void pjb_tet_kernel(PJBlk d, uint32_t tiles_per_xcd) {
    __shared__ float4 s_pos[kTile];        // staged particle positions, later reused for nothing else
    __shared__ float4 s_goal[4 * kTile];   // (V*goal, V) per corner, plane-major: [corner][tet]
    __shared__ uint2 s_ent[kTile];         // the tile's reduction order, 4 x u16 per tet position

    const uint32_t b = xcd_tile(blockIdx.x, tiles_per_xcd);
    if (b >= d.nb) return;  // whole workgroup leaves together
    const uint32_t tid = threadIdx.x;
    const uint32_t t0 = d.blk_tet_off[b], ntb = d.blk_tet_off[b + 1] - t0;
    const uint32_t v0 = d.blk_vert_off[b], nu = d.blk_vert_off[b + 1] - v0;

    // 1. stage the tile's particles; issue this lane's streaming loads before the barrier so they overlap it
    if (tid < nu) s_pos[tid] = d.pos_pred[d.blk_verts[v0 + tid]];
    const bool has_tet = tid < ntb;
    const uint32_t e = t0 + (has_tet ? tid : 0u);
    uchar4 li = make_uchar4(0, 0, 0, 0);
    float4 ra, rb, rc, q_old;
    float V = 0.0f;
    if (has_tet) {
        li = d.tet_lidx[e];
        ra = d.rest_a[e]; rb = d.rest_b[e]; rc = d.rest_c[e];
        q_old = d.quat[e];
        V = d.vol[e];
        s_ent[tid] = d.lc_ent[e];
    }
    __syncthreads();

    // 2. solve
    if (has_tet) {
        f3 cur[4], rest[4], goal[4];
        cur[0] = xyz(s_pos[li.x]); cur[1] = xyz(s_pos[li.y]); cur[2] = xyz(s_pos[li.z]); cur[3] = xyz(s_pos[li.w]);
        rest[0] = F3(ra.x, ra.y, ra.z); rest[1] = F3(ra.w, rb.x, rb.y);
        rest[2] = F3(rb.z, rb.w, rc.x); rest[3] = F3(rc.y, rc.z, rc.w);
        float4 q_new;
        pj_solve_tet(cur, rest, q_old, q_new, goal);
        d.quat[e] = q_new;
        d.rest_a[e] = make_float4(goal[0].x, goal[0].y, goal[0].z, goal[1].x);
        d.rest_b[e] = make_float4(goal[1].y, goal[1].z, goal[2].x, goal[2].y);
        d.rest_c[e] = make_float4(goal[2].z, goal[3].x, goal[3].y, goal[3].z);
#pragma unroll
        for (int k = 0; k < 4; k++) s_goal[k * kTile + tid] = make_float4(goal[k].x * V, goal[k].y * V, goal[k].z * V, V);
    }
    __syncthreads();

    // 3. one lane per tile particle: fixed-order sum of its corner goals
    if (tid < nu) {
        const uint32_t range = d.lc_range[v0 + tid];
        const uint32_t first = range & 0xffffu, last = range >> 16;
        const uint16_t* ent = reinterpret_cast<const uint16_t*>(s_ent);
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (uint32_t i = first; i < last; i++) {
            const uint32_t en = ent[i];
            const float4 g = s_goal[(en & 3u) * kTile + (en >> 2)];
            acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
        }
        d.partial[v0 + tid] = acc;
    }
}

__global__ __launch_bounds__(256) void pjb_vertex_kernel(PJBlk d, uint32_t first, uint32_t count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const uint32_t v = first + i;
    const DevParams& P = *d.params;

    // P5 (SoftbodyGPU.js:302-320) from tile partial sums, ascending tile order.  The index lists are ELL
    // (column-major, coalesced, no offset lookup first), fetched 4 columns at a time so 4 gathers are in flight.
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t* col = d.vp_ell + v;
    for (uint32_t j0 = 0; j0 < d.vp_cols; j0 += 4u) {
        uint32_t idx[4];
        float4 g[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) idx[j] = (j0 + j < d.vp_cols) ? col[static_cast<size_t>(j0 + j) * d.nv_pad] : 0xffffffffu;
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) g[j] = idx[j] != 0xffffffffu ? d.partial[idx[j]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) { acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; acc.w += g[j].w; }
    }
    const float rw = __builtin_amdgcn_rcpf(acc.w);
    f3 p = F3(acc.x * rw, acc.y * rw, acc.z * rw);  // 0 * inf = NaN for a particle without tets, as in the reference

    // P6, :340-355
    const f3 prev = xyz(d.pos_final[v]);
    if (static_cast<int32_t>(v) == P.grab_local) p = F3(P.grab[0], P.grab[1], P.grab[2]);
    p.x = fminf(fmaxf(p.x, P.lo[0]), P.hi[0]);
    p.y = fminf(fmaxf(p.y, P.lo[1]), P.hi[1]);
    p.z = fminf(fmaxf(p.z, P.lo[2]), P.hi[2]);
    if (p.y < 0.0f) {
        p.y = 0.0f;
        const f3 F = prev - p;
        const float fr = fminf(1.0f, P.dt * P.friction);
        p.x += F.x * fr;
        p.z += F.z * fr;
    }
    // P7, :364-372, then P1 + P2 of the next substep
    const float rdt = __builtin_amdgcn_rcpf(P.dt);
    const f3 vel = (p - prev) * rdt + F3(0.0f, P.gravity, 0.0f) * P.dt;
    d.pos_final[v] = make_float4(p.x, p.y, p.z, 0.0f);
    d.vel[v] = make_float4(vel.x, vel.y, vel.z, 0.0f);
    const f3 pred = p + vel * P.dt;
    d.pos_pred[v] = make_float4(pred.x, pred.y, pred.z, 0.0f);
}

__global__ __launch_bounds__(256) void pjb_repredict_kernel(PJBlk d) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= d.nv_owned) return;
    const f3 pred = xyz(d.pos_final[v]) + xyz(d.vel[v]) * d.params->dt;
    d.pos_pred[v] = make_float4(pred.x, pred.y, pred.z, 0.0f);
}

}  // namespace

void pjb_launch_tet(hipStream_t s, const PJBlk& d) {
    if (d.nb == 0) return;
    const uint32_t per_xcd = (d.nb + 7u) / 8u;
    hipLaunchKernelGGL(pjb_tet_kernel, dim3(per_xcd * 8u), dim3(256), 0, s, d, per_xcd);
}
void pjb_launch_vertex(hipStream_t s, const PJBlk& d, uint32_t first, uint32_t count) {
    if (count == 0) return;
    hipLaunchKernelGGL(pjb_vertex_kernel, dim3((count + 255u) / 256u), dim3(256), 0, s, d, first, count);
}
void pjb_launch_repredict(hipStream_t s, const PJBlk& d) {
    if (d.nv_owned == 0) return;
    hipLaunchKernelGGL(pjb_repredict_kernel, dim3((d.nv_owned + 255u) / 256u), dim3(256), 0, s, d);
}

}  // namespace tetsim

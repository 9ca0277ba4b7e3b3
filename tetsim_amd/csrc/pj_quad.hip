// pj_quad.hip -- POLAR_JACOBI, FAST, SMALL bodies: one tet (and one particle) on FOUR lanes (gfx950, wave64).
//
// The reference's own workload is the Dragon (3,840 tets; /root/reference/src/main.js:26-27,79-84).  On this chip such a body is
// not bandwidth but a chain: with one tet per lane and 256-tet tiles (pj_blocked.hip) it is 60 waves, each alone on its SIMD, and a
// lone wave issues one DEPENDENT vector instruction every ~5.5-7 cycles -- the ~970 instructions of a tet solve were 48% of a
// substep (profiles/archive/r03_frame_kernel.txt).  The only lever is fewer instructions per lane.  Here:
//   * tiles hold <= 64 tets touching <= 64 particles (host_prep.h kQuadTile): a workgroup is 64 QUADS = 4 waves = one wave per SIMD
//     of a CU, and the Dragon's 62 tiles keep 248 SIMDs busy instead of 60;
//   * lanes 4i..4i+2 of quad i hold the x / y / z COMPONENT of everything vector-valued of tet i (P3 + P4, SoftbodyGPU.js:80-262)
//     and, in the particle phases, of tile particle i (P5 + P6 + P7 + P1 + P2, :272-376).  Lane 4i+3 carries zeros (it reads the
//     unused fourth float of every record) and follows along.  What couples the components goes through quad_perm DPP operands:
//       - lane c keeps ROW c of the covariance A and of the rotation R, the columns in rotated order (c, c+1, c+2): with
//         (a, b, d, w) = (q[c], q[c+1], q[c+2], q.w) the row is R[c][c] = 1/2 - (b^2 + d^2), R[c][c+1] = ab - wd,
//         R[c][c+2] = ad + wb for every lane alike (R/2, as pj_math.inc);
//       - the cross products sum_i R_i x A_i need rows c+1 and c+2: R's come as DPP operands of the FMAs that consume them, A's
//         rotated copies are made once per substep;
//       - the three dot-product sums (denominator, |omega|^2, the quaternion products' scalar part) are added up in FIXED order
//         (x + y) + z from broadcasts, so every lane of a quad holds the same bits and takes the same decisions.
//     ~490 instead of ~970 instructions per lane and tet.
// Three kernels share every arithmetic function below (this unit is built with -ffp-contract=off and every fused multiply-add is
// spelled out, so they agree BIT FOR BIT by construction):
//   pjq_frame_kernel    one launch per tetsim_step_n call: every tile's workgroup stays resident for the n substeps, tet and
//                       particle state in registers, the tile partial sums carry the substep's sequence number in their fourth
//                       float and are exchanged without a barrier (the choreography of pjb_frame_kernel, pj_blocked.hip);
//   pjq_tet_kernel, pjq_vertex_kernel    the same substep as two launches through memory: tetsim_profile, and what a body falls back
//                       to if a frame kernel's bounded wait ever gives up (tetsim_step is the frame kernel for n = 1).
// Differences from the other FAST formulations are summation order only (64-tet tiles, component-wise sums): tolerance level.
#include <cstdint>

#include "dev_common.h"
#include "dev_store.h"
#include "host_prep.h"

namespace tetsim {
namespace {

constexpr uint32_t kQT = kQuadTile;          // tets / particle slots per tile = quads per workgroup
constexpr uint32_t kQThreads = 4u * kQT;
constexpr float kRefRotExitW2ForQuad = 1.0e-18f;   // (1e-9)^2, SoftbodyGPU.js:131: iteration 1 always ends a tet there (pj_math.inc)

// ---- quad-lane plumbing -------------------------------------------------------------------------------------------------------
template <int kCtrl>
__device__ __forceinline__ float dpp(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), kCtrl, 0xf, 0xf, true)); }
constexpr int kNext = 0xC9, kNext2 = 0xD2;                        // quad_perm [1,2,0,3] / [2,0,1,3]: component c+1 / c+2 (lane 3: itself)
constexpr int kLane0 = 0x00, kLane1 = 0x55, kLane2 = 0xAA;        // broadcasts of lane 0 / 1 / 2
__device__ __forceinline__ float nxt(float v) { return dpp<kNext>(v); }
__device__ __forceinline__ float nx2(float v) { return dpp<kNext2>(v); }
// x + y + z of a per-component value, the same bits in all four lanes
__device__ __forceinline__ float sum_xyz(float v) { return (dpp<kLane0>(v) + dpp<kLane1>(v)) + dpp<kLane2>(v); }
// component c of a float4 record (lane 3: 0)
__device__ __forceinline__ float comp(const float4& v, uint32_t c) { return c == 0u ? v.x : c == 1u ? v.y : c == 2u ? v.z : 0.0f; }

// A quaternion as the lanes hold it: a = q[c], b = q[c+1], d = q[c+2], w = q.w (lane 3: a = b = d = 0)
struct QRot { float a, b, d, w; };
__device__ __forceinline__ QRot qrot(const float4& q, uint32_t c) {
    QRot r;
    r.a = comp(q, c); r.b = c == 0u ? q.y : c == 1u ? q.z : c == 2u ? q.x : 0.0f; r.d = c == 0u ? q.z : c == 1u ? q.x : c == 2u ? q.y : 0.0f; r.w = q.w;
    return r;
}
// dq (x) q, quat_mult of SoftbodyGPU.js:114-121: component c = dw q[c] + dq[c] qw + dq[c+1] q[c+2] - dq[c+2] q[c+1]
__device__ __forceinline__ QRot qmul(float da, float dw, const QRot& q) {
    QRot r;
    r.a = fmaf(dw, q.a, fmaf(da, q.w, fmaf(nxt(da), q.d, -(nx2(da) * q.b))));
    r.w = fmaf(dw, q.w, -sum_xyz(da * q.a));
    r.b = nxt(r.a); r.d = nx2(r.a);
    return r;
}
__device__ __forceinline__ QRot qnormalize(const QRot& q) {   // v_rsq_f32 as it is (pj_math.inc: normalize4)
    const float r = __builtin_amdgcn_rsqf(fmaf(q.w, q.w, sum_xyz(q.a * q.a)));
    QRot o;
    o.a = q.a * r; o.b = q.b * r; o.d = q.d * r; o.w = q.w * r;
    return o;
}

// ---- P3 + P4 for one component of one tet --------------------------------------------------------------------------------------
// cur[k]: component c of the 4 predicted corners; rest[k]: component c of the carried shape (centred), replaced by the goal shape;
// q: the tet's quaternion, replaced; vg[k]: component c of V * (goal_k + centroid) for the tile's reduction.
__device__ __forceinline__ void pjq_solve(float cur[4], float rest[4], QRot& q, const float V, const float exit_w2, float vg[4]) {
    // centroid of the current corners, SoftbodyGPU.js:162-175
    const float cc = (((cur[0] + cur[1]) + cur[2]) + cur[3]) * 0.25f;
    float rn[4], rm[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { cur[k] -= cc; rn[k] = nxt(rest[k]); rm[k] = nx2(rest[k]); }
    // TransposeMult (:90-105), row c of A in rotated column order: Ar[j] = A[c][c+j] = sum_k rest_k[c+j] cur_k[c]
    const float Ar0 = fmaf(rest[3], cur[3], fmaf(rest[2], cur[2], fmaf(rest[1], cur[1], rest[0] * cur[0])));
    const float Ar1 = fmaf(rn[3], cur[3], fmaf(rn[2], cur[2], fmaf(rn[1], cur[1], rn[0] * cur[0])));
    const float Ar2 = fmaf(rm[3], cur[3], fmaf(rm[2], cur[2], fmaf(rm[1], cur[1], rm[0] * cur[0])));
    // sum_i R_i x A_i, component c = sum_a R[c+1][a] A[c+2][a] - sum_a R[c+2][a] A[c+1][a].  Every lane L forms the two dot products of ITS
    // row of R with the rows L+1 and L+2 of A -- whose copies, in the column order of its own R row, it makes once per substep -- and
    // component c picks nxt(S) - nx2(T): 6 local FMAs in two independent chains and 2 DPP operands per iteration (with R's rows
    // fetched from the neighbours instead it was 6 DPP moves and a 6-deep chain).
    const float As0 = nxt(Ar2), As1 = nxt(Ar0), As2 = nxt(Ar1);    // A[L+1][L+j]
    const float At0 = nx2(Ar1), At1 = nx2(Ar2), At2 = nx2(Ar0);    // A[L+2][L+j]
    const float trA = sum_xyz(Ar0);

    // extractRotation from identity, <= 9 iterations (:122-139); wave-uniform loop, a done tet stops changing (pj_math.inc)
    QRot r;
    r.a = r.b = r.d = 0.0f; r.w = 1.0f;
    float w2;
    {   // iteration 1, R = I: omega = (A[c+2][c+1] - A[c+1][c+2]) / |tr A + 1e-9|
        const float omega = (nxt(As0) - nx2(At0)) * __builtin_amdgcn_rcpf(fabsf(trA + 0.000000001f));
        w2 = sum_xyz(omega * omega);
        if (w2 >= kRefRotExitW2ForQuad) {
            const float winv = __builtin_amdgcn_rsqf(w2), rev = (w2 * winv) * 0.07957747f;   // (|omega| / 2) in revolutions
            r.a = omega * (__builtin_amdgcn_sinf(rev) * winv);
            r.w = __builtin_amdgcn_sinf(rev + 0.24987326f);                                    // cos(h) written sin(h + 1.57), :106-110
            r.b = nxt(r.a); r.d = nx2(r.a);
        }
    }
    if (__builtin_amdgcn_ballot_w64(w2 >= kRefRotExitW2ForQuad) != 0ull) {
        for (int iter = 1; iter < 9; iter++) {
            // row L of R/2 (rotated columns), then omega = sum_i R_i x A_i / |sum_i R_i . A_i + 1e-9| with both halved
            const float R0 = 0.5f - fmaf(r.d, r.d, r.b * r.b), R1 = fmaf(-r.w, r.d, r.a * r.b), R2 = fmaf(r.w, r.b, r.a * r.d);
            const float S = fmaf(R2, As2, fmaf(R1, As1, R0 * As0));
            const float T = fmaf(R2, At2, fmaf(R1, At1, R0 * At0));
            const float D = fmaf(R2, Ar2, fmaf(R1, Ar1, R0 * Ar0));
            // (the denominator's three addends meet in ROTATED order, two operations instead of three: its last bit may differ from lane
            // to lane, which only scales that lane's component of omega; what must be the same bits in every lane -- |omega|^2, which
            // decides, and the scalar part of the quaternion, which is state -- is summed in fixed order)
            const float den = fabsf(((D + nxt(D)) + nx2(D)) + 0.0000000005f);
            const float omega = (nxt(S) - nx2(T)) * __builtin_amdgcn_rcpf(den);
            w2 = sum_xyz(omega * omega);
            if (w2 >= exit_w2) {   // (the same bits in the four lanes of a quad: one decision per tet)
                const float winv = __builtin_amdgcn_rsqf(w2), rev = (w2 * winv) * 0.07957747f;
                r = qmul(omega * (__builtin_amdgcn_sinf(rev) * winv), __builtin_amdgcn_sinf(rev + 0.24987326f), r);
            }
            if (__builtin_amdgcn_fcmpf(w2, exit_w2, 3) == 0ull) break;   // no tet of this wave moved
        }
    }
    // quat' = normalize(rot (x) quat), :181;  rel = normalize(rot) == normalize(quat' (x) conj(quat)) in exact arithmetic, :237-239
    {
        QRot t;
        t.a = fmaf(r.w, q.a, fmaf(r.a, q.w, fmaf(r.b, q.d, -(r.d * q.b))));
        t.w = fmaf(r.w, q.w, -sum_xyz(r.a * q.a));
        t.b = nxt(t.a); t.d = nx2(t.a);
        q = qnormalize(t);
    }
    const QRot l = qnormalize(r);
    // goal_k = Rotate(rest_k, rel) + centroid (:253-256) through row c of R(rel); the goal shape is the next carried shape
    const float G0 = 1.0f - 2.0f * fmaf(l.d, l.d, l.b * l.b), G1 = 2.0f * fmaf(-l.w, l.d, l.a * l.b), G2 = 2.0f * fmaf(l.w, l.b, l.a * l.d);
    const float vcc = cc * V;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        rest[k] = fmaf(G2, rm[k], fmaf(G1, rn[k], G0 * rest[k]));
        vg[k] = fmaf(rest[k], V, vcc);
    }
}

// ---- P5's division + P6 + P7 (+ P1, P2 of the next substep) for one component of one particle ---------------------------------
struct QParams { float dt, rdt, fr, g_dt, lo, hi, grab; int32_t grab0, grab1; };
__device__ __forceinline__ QParams load_qparams(const DevParams& P, uint32_t c) {
    QParams o;
    o.dt = P.dt;
    o.rdt = __builtin_amdgcn_rcpf(P.dt);
    o.fr = fminf(1.0f, P.dt * P.friction);
    o.g_dt = P.dt * (c == 1u ? P.gravity : 0.0f);            // F3(0, gravity, 0) * dt, :371
    o.lo = c == 0u ? P.lo[0] : c == 1u ? P.lo[1] : P.lo[2];
    o.hi = c == 0u ? P.hi[0] : c == 1u ? P.hi[1] : P.hi[2];
    o.grab = c == 0u ? P.grab[0] : c == 1u ? P.grab[1] : P.grab[2];
    o.grab0 = P.grab_local; o.grab1 = P.grab_local2;
    return o;
}
struct QVertex { float p, vel, pred; };
__device__ __forceinline__ QVertex pjq_vertex_update(float acc, float wsum, float prev, const QParams& P, uint32_t vid, uint32_t c) {
    float p = acc * __builtin_amdgcn_rcpf(wsum);   // 0 * inf = NaN for a particle without tets, as in the reference
    // P6, SoftbodyGPU.js:340-355
    if (static_cast<int32_t>(vid) == P.grab0 || static_cast<int32_t>(vid) == P.grab1) p = P.grab;
    p = fminf(fmaxf(p, P.lo), P.hi);
    if (dpp<kLane1>(p) < 0.0f) p = c == 1u ? 0.0f : fmaf(prev - p, P.fr, p);   // below the floor: y = 0, x and z rubbed back
    // P7, :364-372, then P1 + P2 of the next substep
    QVertex o;
    o.p = p;
    o.vel = fmaf(P.rdt, p - prev, P.g_dt);
    o.pred = fmaf(P.dt, o.vel, p);
    return o;
}

enum { kModeFrame = 0, kModeTet = 1, kModeVertex = 2 };
constexpr uint32_t kQMaxSrc = kQuadMaxPartials;   // partial sums per particle at most (host_prep.h); longer lists keep the 256-tet tiles

// kLocal (frame kernel): every tile of a body sits on ONE XCD: the exchange is coherent in that XCD's L2 (dev_store.h)
template <int kMode, bool kLocal>
__device__ __forceinline__ void pjq_body(const PJBlk& d, const DevParams& P, const uint32_t n, const uint32_t b, float4* const pbuf0, float4* const pbuf1, uint32_t* const err,
                                         const uint32_t timeout_ms) {
    __shared__ float s_pos[4 * kQT];          // staged particle positions, xyzw per slot
    __shared__ float s_g[3 * 4 * kQT];        // V * goal, [component][corner][tet]
    __shared__ uint16_t s_ent[4 * kQT];       // the tile's reduction order (word offsets corner * kQT + tet)

    const uint32_t tid = threadIdx.x, c = tid & 3u, qd = tid >> 2;
#ifdef TETSIM_ABLATION   // development build: thread 0 adds up the cycles of each phase over the call (TETSIM_DEBUG_TRACE, tools/attic/frame_trace.py)
    uint32_t fr_acc[5] = {0, 0, 0, 0, 0}, fr_last = 0, fr_polls = 0;   // (32-bit: a call is a few hundred thousand cycles, and the kernel has 128 registers)
#define QSTAMP(i) do { if (kMode == kModeFrame && d.trace && tid == 0) { const uint32_t now_ = static_cast<uint32_t>(__builtin_amdgcn_s_memtime()); if ((i) > 0) fr_acc[(i) > 0 ? (i) - 1 : 0] += now_ - fr_last; fr_last = now_; } } while (0)
#define QPOLL() do { if (tid == 0) fr_polls++; } while (0)
    // TETSIM_QUAD_POLL_DELAY=-1 (development): ABSOLUTE s_memtime stamps of substeps n-3 and n-2 instead of the phase sums -- gather
    // done, solve done, sum stored -- to read the tiles' timelines against each other (one XCD: one clock)
    unsigned long long fr_abs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define QABS(slot) do { if (kMode == kModeFrame && d.trace && tid == 0 && P.poll_delay < 0 && n >= 4u && (s == n - 3u || s == n - 2u)) fr_abs[(s == n - 3u ? 0 : 4) + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QABS(slot) do { } while (0)
#define QSTAMP(i) do { } while (0)
#define QPOLL() do { } while (0)
#endif
    const uint32_t t0 = d.blk_tet_off[b], ntb = d.blk_tet_off[b + 1] - t0;
    const uint32_t v0 = d.blk_vert_off[b], nu = d.blk_vert_off[b + 1] - v0;
    const bool has_slot = qd < nu, has_tet = qd < ntb;
    const uint32_t slot = v0 + (has_slot ? qd : 0u), e = t0 + (has_tet ? qd : 0u);

    // ---- what stays for the whole call (ids first: the particle loads depend on them) ------------------------------------------
    const uint32_t vid = static_cast<uint32_t>(d.blk_verts[slot]);
    const uint32_t range = has_slot ? d.lc_range[slot] : 0u;
    uint32_t src[kQMaxSrc];
    if constexpr (kMode != kModeTet) {
        const uint32_t maxsrc = d.blk_maxsrc[b];
        const uint32_t* col = d.slot_src + slot;
#pragma unroll
        for (uint32_t j = 0; j < kQMaxSrc; j++) src[j] = (has_slot && j < maxsrc) ? col[static_cast<size_t>(j) * d.ns_pad] : 0xffffffffu;
    }
    // (the longest list in this WAVE bounds the gather's trips over list positions with scalar branches: most waves hold lists of 3-6)
    uint32_t wmax = 0;
    if constexpr (kMode != kModeTet) {
#pragma unroll
        for (uint32_t j = 0; j < kQMaxSrc; j++) wmax = __builtin_amdgcn_ballot_w64(src[j] != 0xffffffffu) != 0ull ? j + 1u : wmax;
    }
    uchar4 li = make_uchar4(0, 0, 0, 0);
    float rest[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    QRot q;
    q.a = q.b = q.d = 0.0f; q.w = 1.0f;
    float V = 0.0f;
    if constexpr (kMode != kModeVertex) {
        li = d.tet_lidx[e];
        const float4 ra = d.rest_a[e], rb = d.rest_b[e], rc = d.rest_c[e];   // 12 floats: corner k's component c is float 3k + c
        q = qrot(d.quat[e], c);
        V = d.vol[e];
        rest[0] = comp(ra, c);
        rest[1] = c == 0u ? ra.w : c == 1u ? rb.x : c == 2u ? rb.y : 0.0f;
        rest[2] = c == 0u ? rb.z : c == 1u ? rb.w : c == 2u ? rc.x : 0.0f;
        rest[3] = c == 0u ? rc.y : c == 1u ? rc.z : c == 2u ? rc.w : 0.0f;
        const uint16_t* ent = reinterpret_cast<const uint16_t*>(d.lc_ent);
        s_ent[tid] = tid < 4u * ntb ? ent[4ull * t0 + tid] : static_cast<uint16_t>(0);
    }
    float prev = 0.0f, wsum = 1.0f, stage = 0.0f;
    if constexpr (kMode != kModeTet) { prev = comp(d.pos_final[vid], c); wsum = d.wsum[vid]; }
    if constexpr (kMode != kModeVertex) stage = comp(d.pos_pred[vid], c);   // substep 0 starts from the prediction the previous call left
    const QParams qp = load_qparams(P, c);
    const uint32_t epoch = d.epoch ? d.epoch : P.epoch;   // (a direct launch -- tetsim_step -- brings its own block of sequence numbers)
    const int32_t poll_delay = P.poll_delay;
    const uint32_t first = range & 0x7ffu, last = range >> 16;
    const bool owner = has_slot && ((range >> 15) & 1u);
    float4* const pbuf[2] = {pbuf0, pbuf1};
    const long long limit = 100000ll * timeout_ms;   // 100 MHz ticks; 0 = unbounded

    // the particle update of substep s from the partial sums of every tile that touches the particle (this one included), ascending
    // tile order.  Frame kernel: a sum is there when its fourth float carries the substep's sequence number; only the lanes (and list
    // positions) that still miss one ask again.
    auto gather_update = [&](const uint32_t s) -> QVertex {
        const float4* buf = kMode == kModeFrame ? pbuf[s & 1u] : d.partial;
        float g[kQMaxSrc];
        uint32_t pend = 0;
#pragma unroll
        for (uint32_t j = 0; j < kQMaxSrc; j++) { g[j] = 0.0f; pend |= (src[j] != 0xffffffffu ? 1u : 0u) << j; }
        if constexpr (kMode == kModeFrame) {
            const uint32_t expect = epoch + s;
            long long w0 = 0ll;
            uint32_t trips = 0;
            for (int32_t i = 0; i < poll_delay; i++) __builtin_amdgcn_s_sleep(1);   // (a look that comes too early costs a whole trip)
            while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
                QPOLL();
                // (8 list positions per trip, the rare 9th..12th behind them: 12 sums in flight would be 48 registers, and the kernel's
                // budget is 128 -- four workgroups per CU, the Dragon's 62 tiles on ONE XCD)
#pragma unroll
                for (uint32_t j0 = 0; j0 < kQMaxSrc; j0 += 8u) {
                    if (j0 != 0u && __builtin_amdgcn_ballot_w64((pend >> j0) != 0u) == 0ull) break;
                    float4 t[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8u; j++)
                        if (j0 + j < kQMaxSrc && j0 + j < wmax && ((pend >> (j0 + j)) & 1u)) t[j] = kLocal ? load_l2(buf, src[j0 + j]) : load_coherent(buf, src[j0 + j]);
#pragma unroll
                    for (uint32_t j = 0; j < 8u; j++)
                        if (j0 + j < kQMaxSrc && j0 + j < wmax && ((pend >> (j0 + j)) & 1u) && __float_as_uint(t[j].w) == expect) { g[j0 + j] = comp(t[j], c); pend &= ~(1u << (j0 + j)); }
                }
                // (the clock is a scalar memory read of its own: first looked at on the 16th trip -- a wait that long is not a normal one --
                // and on every 16th from there)
                if (limit && (++trips & 15u) == 0u) {
                    const long long now = wall_clock64();
                    if (trips == 16u) w0 = now;
                    else if (pend != 0u && now - w0 > limit) {   // never in a correct run; a wedged GPU helps nobody
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        pend = 0u;
                    }
                }
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < kQMaxSrc; j++)
                if ((pend >> j) & 1u) g[j] = comp(buf[src[j]], c);
        }
        float acc = 0.0f;
#pragma unroll
        for (uint32_t j = 0; j < kQMaxSrc; j++) acc += g[j];   // absent = +0
        return pjq_vertex_update(acc, wsum, prev, qp, vid, c);
    };
    auto store_particle = [&](const QVertex& o) {   // one writer per particle: lane 0 of the owning slot's quad
        const float4 p4 = make_float4(o.p, nxt(o.p), nx2(o.p), 0.0f), v4 = make_float4(o.vel, nxt(o.vel), nx2(o.vel), 0.0f),
                     x4 = make_float4(o.pred, nxt(o.pred), nx2(o.pred), 0.0f);
        if (owner && c == 0u) { store_wt(d.pos_final, vid, p4); store_wt(d.vel, vid, v4); store_wt(d.pos_pred, vid, x4); }
    };
    if constexpr (kMode == kModeVertex) {
        store_particle(gather_update(0u));
        return;
    }

    for (uint32_t s = 0; s < n; s++) {
        QSTAMP(0);
        QABS(0);
        if constexpr (kMode == kModeFrame) {
            if (s > 0u) {
                const QVertex o = gather_update(s - 1u);
                prev = c < 3u ? o.p : 0.0f;      // (lane 3 keeps carrying zeros)
                stage = c < 3u ? o.pred : 0.0f;
            }
        }
        QSTAMP(1);
        QABS(1);
        if (has_slot) s_pos[tid] = stage;      // (lane 3: the fourth float, 0)
        __syncthreads();
        QSTAMP(2);
        if (has_tet) {
            float cur[4] = {s_pos[4u * li.x + c], s_pos[4u * li.y + c], s_pos[4u * li.z + c], s_pos[4u * li.w + c]};
            float vg[4];
            pjq_solve(cur, rest, q, V, d.rot_exit_w2, vg);
            if (c < 3u) {
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) s_g[(c * 4u + k) * kQT + qd] = vg[k];
            }
        }
        QSTAMP(3);
        QABS(2);
        __syncthreads();
        QSTAMP(4);
        {   // The tile's reduction, one slot per quad: lane l adds up entries first + l, first + l + 4, ... of ALL three planes (a slot of
            // the Dragon has up to 26 entries, and an entry is two dependent LDS trips: one lane per component was a chain of 26 of
            // them -- 3.7 k cycles for the slowest tile, which every neighbour then waits for), the four partial sums meet in fixed
            // order (l0 + l1) + (l2 + l3).  Every lane of the quad ends up with the same three totals.
            float ax = 0.0f, ay = 0.0f, az = 0.0f;
            if (has_slot)
                for (uint32_t i = first + c; i < last; i += 4u) {
                    const uint32_t o = s_ent[i];
                    ax += s_g[o]; ay += s_g[4u * kQT + o]; az += s_g[8u * kQT + o];
                }
            auto quad_sum = [](float v) { return (dpp<kLane0>(v) + dpp<kLane1>(v)) + (dpp<kLane2>(v) + dpp<0xFF>(v)); };
            const float4 part = make_float4(quad_sum(ax), quad_sum(ay), quad_sum(az), kMode == kModeFrame ? __uint_as_float(epoch + s) : 0.0f);
            if (has_slot && c == 0u) {
                float4* const out = kMode == kModeFrame ? pbuf[s & 1u] : d.partial;
                if constexpr (kMode == kModeFrame && kLocal) store_plain(out, v0 + qd, part);
                else store_wt(out, v0 + qd, part);
            }
        }
        QSTAMP(5);
        QABS(3);
        // (no barrier here: the next trip writes s_pos, last read before the second barrier above, and the planes are rewritten only
        // behind the next trip's first barrier, which every reducing lane reaches after its reads)
    }
    if constexpr (kMode == kModeFrame) store_particle(gather_update(n - 1u));
#ifdef TETSIM_ABLATION
    if (kMode == kModeFrame && d.trace && tid == 0 && P.poll_delay < 0) { for (int i = 0; i < 8; i++) d.trace[8ull * b + i] = fr_abs[i]; }
    else if (kMode == kModeFrame && d.trace && tid == 0) { for (int i = 0; i < 5; i++) d.trace[8ull * b + i] = fr_acc[i]; d.trace[8ull * b + 5] = fr_polls; d.trace[8ull * b + 6] = n; d.trace[8ull * b + 7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }   // XCC_ID
#endif
#undef QSTAMP
#undef QABS
#undef QPOLL
    // every tet back to memory: lane 0 of its quad collects the components
    {
        const float4 q4 = make_float4(q.a, q.b, q.d, q.w);
        float ry[4], rz[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { ry[k] = nxt(rest[k]); rz[k] = nx2(rest[k]); }
        if (has_tet && c == 0u) {
            store_wt(d.quat, e, q4);
            store_wt(d.rest_a, e, make_float4(rest[0], ry[0], rz[0], rest[1]));
            store_wt(d.rest_b, e, make_float4(ry[1], rz[1], rest[2], ry[2]));
            store_wt(d.rest_c, e, make_float4(rz[2], rest[3], ry[3], rz[3]));
        }
    }
}

// (four waves per SIMD, i.e. at most 128 registers: four workgroups per CU put the Dragon's 62 tiles on ONE XCD -- 64 is half of an XCD's 128
// slots -- where the exchange is an L2 trip; at 130 registers the tiles spread over all XCDs and the gather phase took 6.1 k instead of
// 4.5 k cycles per substep, profiles/r04_quad_lanes.txt)
// (The call's parameters arrive by value, with the launch -- no upload in front of a call, nothing between two calls: pjb_call_kernel,
// pj_blocked.hip; the first workgroup leaves them in DevParams for the kernels behind this one.)
template <bool kLocal>
__global__ __launch_bounds__(kQThreads, 4) void pjq_frame_kernel(PJBlk d, uint32_t n, const int32_t* block_tile, float4* pbuf0, float4* pbuf1, uint32_t* err,
                                                              uint32_t timeout_ms, DevParams pv, DevParams* pdev) {
    if (blockIdx.x == 0u && threadIdx.x == 0u) *pdev = pv;
    const int32_t bt = block_tile[blockIdx.x];
    if (bt < 0) return;   // (a block that only pads the grid so that the others land on the intended XCDs)
    pjq_body<kModeFrame, kLocal>(d, pv, n, static_cast<uint32_t>(bt), pbuf0, pbuf1, err, timeout_ms);
}
__global__ __launch_bounds__(kQThreads) void pjq_tet_kernel(PJBlk d) { pjq_body<kModeTet, false>(d, *d.params, 1u, blockIdx.x, nullptr, nullptr, nullptr, 0u); }
__global__ __launch_bounds__(kQThreads) void pjq_vertex_kernel(PJBlk d) { pjq_body<kModeVertex, false>(d, *d.params, 1u, blockIdx.x, nullptr, nullptr, nullptr, 0u); }

}  // namespace

void pjq_launch_frame(hipStream_t s, const PJBlk& d, uint32_t n, const int32_t* block_tile, uint32_t blocks, bool local, float4* pbuf0, float4* pbuf1,
                      uint32_t* err, uint32_t timeout_ms, const DevParams& params, DevParams* params_dev, hipEvent_t e0, hipEvent_t e1) {
    if (d.nb == 0 || n == 0 || blocks == 0) return;
    auto* kernel = local ? pjq_frame_kernel<true> : pjq_frame_kernel<false>;
    if (e0) hipExtLaunchKernelGGL(kernel, dim3(blocks), dim3(kQThreads), 0, s, e0, e1, 0, d, n, block_tile, pbuf0, pbuf1, err, timeout_ms, params, params_dev);
    else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kQThreads), 0, s, d, n, block_tile, pbuf0, pbuf1, err, timeout_ms, params, params_dev);
}
void pjq_launch_tet(hipStream_t s, const PJBlk& d, hipEvent_t e0, hipEvent_t e1) {
    if (d.nb == 0) return;
    if (e0) hipExtLaunchKernelGGL(pjq_tet_kernel, dim3(d.nb), dim3(kQThreads), 0, s, e0, e1, 0, d);
    else hipLaunchKernelGGL(pjq_tet_kernel, dim3(d.nb), dim3(kQThreads), 0, s, d);
}
void pjq_launch_vertex(hipStream_t s, const PJBlk& d, hipEvent_t e0, hipEvent_t e1) {
    if (d.nb == 0) return;
    if (e0) hipExtLaunchKernelGGL(pjq_vertex_kernel, dim3(d.nb), dim3(kQThreads), 0, s, e0, e1, 0, d);
    else hipLaunchKernelGGL(pjq_vertex_kernel, dim3(d.nb), dim3(kQThreads), 0, s, d);
}
uint32_t pjq_frame_capacity(uint32_t* compute_units) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    int per_cu_any = 0;   // (the smaller answer of the two placements' kernels: which one a body launches is decided after this query)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pjq_frame_kernel<true>, static_cast<int>(kQThreads), 0) != hipSuccess || per_cu <= 0) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_any, pjq_frame_kernel<false>, static_cast<int>(kQThreads), 0) != hipSuccess || per_cu_any <= 0) return 0;
    per_cu = per_cu < per_cu_any ? per_cu : per_cu_any;
    if (compute_units) *compute_units = static_cast<uint32_t>(prop.multiProcessorCount);
    return static_cast<uint32_t>(per_cu);
}

}  // namespace tetsim

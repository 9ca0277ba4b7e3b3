// pj_blocked.hip -- POLAR_JACOBI, blocked formulation, FAST arithmetic (gfx950, wave64).
//
// Same substep as pj_kernels.inc (P3+P4 per tet, P5+P6+P7 (+P1,P2 of the next substep) per particle), but the
// scatter between the two kernels is restructured around workgroup tiles so that HBM traffic drops from
// ~270 B to ~180 B per tet-solve and almost all random 16-byte gathers become LDS reads:
//
//   pjb_tet_kernel     one workgroup per tile (<= 256 tets, <= 256 distinct particles), 8 workgroups per CU:
//     1. the tile's particle positions are loaded once into LDS (one gather per distinct particle instead of
//        one per corner: 131 instead of 1024 on the lattice);
//     2. each lane solves one tet from LDS-resident corners (rotation extraction, goals), streams its carried
//        rest shape (48 B, packed) and quaternion in and out with 16-byte coalesced accesses, and leaves
//        (V*goal, V) for its 4 corners in LDS;
//     3. lanes switch roles -- one lane per tile PARTICLE -- and add up the corner goals of that particle in
//        LDS, in a fixed host-built order (deterministic, no atomics), storing ONE partial sum per
//        (tile, particle), coalesced.
//   pjb_vertex_kernel  one lane per particle: adds the 1..9 (2.9 on average; irregular meshes: more) partial sums of the tiles that touch
//        it instead of gathering ~23 goals, then collides / integrates exactly like the gather formulation.
//
// Tried and dropped in round 2 (profiles/archive/r02b_kernel_variants.txt): the staged positions as three f32 planes instead of float4
// (12 ds_read_b32 instead of 4 ds_read_b128 per tet: 29.5 vs 29.0 us), and 128-tet tiles (-DTETSIM_TILE=128: 16 workgroups per
// CU, tet kernel 28.3-30.5 us against 29.0, but 15% more partial sums: particle pass 6.5 vs 5.8 us, no gain overall).
// ... and a second attempt at a persistent, software-pipelined kernel (commit 645542b: workgroups resident over several tiles, the next
// tile's record AND position gather in flight during the solve -- ids two tiles ahead, headers through scalar loads, partial sums
// leaving one tile late so that the one drain point per tile finds everything landed; correct, 86 registers, 5 waves per SIMD):
// 33.4-35.0 us against 28.3-29.2 (profiles/archive/r02j_pipelined_kernel_ab.txt).  What eight independent workgroups per CU overlap for
// free, one pipelined workgroup pays for in registers, occupancy and rotation moves.
// Tried and dropped (measured on the 1 M-tet lattice): (a) a persistent variant prefetching tile i+1
// during tile i's solve, with scalar tile headers, a 3-deep index pipeline, a peeled first trip and counted
// vmcnt waits: 49 us vs 42-44 us, its 0-iteration base is already slower (32.5 vs 28.5 us); (b) having the particle
// pass scatter positions into the tile cells (no staging gather here): tet kernel 40.9 us but the particle pass
// 9.8 -> 18.7 us; (c) staggering the first round of workgroups with s_sleep: strictly slower; (d) two tets per lane on
// packed f32 (every add/mul/fma a v_pk_*_f32, 128-thread workgroups, 114 VGPRs, no spills): the rotation loop got 14 %
// cheaper (1.37 vs 1.60 us per iteration) but the 0-iteration base 1.2 us dearer, 37.9 vs 35.0 us overall.  On gfx950 a
// wave64 v_pk_fma_f32 issues in 4.8 cycles against 2.8 for v_fma_f32 (tools/micro/valu_rate.hip,
// profiles/archive/r01e_valu_issue_rates.txt): packing buys at most 16 % of f32 throughput, not 2x.
//
// Result differs from the gather formulation only by summation order (tile partials) -- tolerance-level, FAST
// mode only; PRECISE keeps the reference's slot order.
#define TETSIM_FAST 1
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dev_common.h"
#include "dev_store.h"
#include "host_prep.h"
#include "pj_lab.h"

namespace tetsim {
namespace {

#include "pj_math.inc"
#include "pj_blocked_lab.inc"   // empty unless -DTETSIM_ABLATION: the iteration histogram, the run-time mode word

// Streamed outputs (carried rest shape, quaternion, partial sums, particle state) are written WRITE-THROUGH
// (sc0 sc1, dev_store.h): a plain store leaves the line dirty in the XCD's 4 MiB L2, and a kernel that streams ~70 MB of
// results then pays the write-back of whatever is still dirty at its end, outside the CUs' busy time (per-CU
// s_memtime timelines: ~65k busy cycles inside a ~96k-cycle kernel).  With write-through the bytes leave while the
// kernel is still computing.  The data is next read by OTHER CUs in the next kernel, so L2 residency is not needed.

__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t tiles_per_xcd) { return (b & 7u) * tiles_per_xcd + (b >> 3); }

// P5's division + P6 + P7 (+ P1, P2 of the next substep) for one particle, shared by the particle kernel and by the fused
// staging of the tet kernel (same function, same contraction: the two must agree bit for bit -- tetsim_step runs the former,
// tetsim_step_n the latter).  acc = sum of V*goal, wsum = sum of V (a constant), prev = end of the previous substep.
struct VertexOut { f3 p, vel, pred; };
// (Every multiply-add below is spelled out: with -ffp-contract=fast the compiler decides per call site which products to fuse, and this
// function is inlined into several kernels whose results must agree bit for bit -- a fourth call site (profiles/archive/r03_tile_finish.txt)
// came out an ulp off.  The spelling is the one the particle kernel has always compiled to.)
__device__ __forceinline__ VertexOut pjb_vertex_update(f3 acc, float wsum, f3 prev, const DevParams& P, uint32_t v) {
    const float rw = __builtin_amdgcn_rcpf(wsum);
    f3 p = F3(acc.x * rw, acc.y * rw, acc.z * rw);  // 0 * inf = NaN for a particle without tets, as in the reference
    // P6, SoftbodyGPU.js:340-355
    if (static_cast<int32_t>(v) == P.grab_local || static_cast<int32_t>(v) == P.grab_local2) p = F3(P.grab[0], P.grab[1], P.grab[2]);
    p.x = fminf(fmaxf(p.x, P.lo[0]), P.hi[0]);
    p.y = fminf(fmaxf(p.y, P.lo[1]), P.hi[1]);
    p.z = fminf(fmaxf(p.z, P.lo[2]), P.hi[2]);
    if (p.y < 0.0f) {
        p.y = 0.0f;
        const float Fx = prev.x - p.x, Fz = prev.z - p.z;
        const float fr = fminf(1.0f, P.dt * P.friction);
        p.x = fmaf(Fx, fr, p.x);
        p.z = fmaf(Fz, fr, p.z);
    }
    // P7, :364-372, then P1 + P2 of the next substep
    const float dt = P.dt;
    const float rdt = __builtin_amdgcn_rcpf(dt);
    VertexOut o;
    o.p = p;
    const float gx = dt * 0.0f, gy = dt * P.gravity, gz = dt * 0.0f;   // F3(0, gravity, 0) * dt
    o.vel = F3(fmaf(rdt, p.x - prev.x, gx), fmaf(rdt, p.y - prev.y, gy), fmaf(rdt, p.z - prev.z, gz));
    o.pred = F3(fmaf(dt, o.vel.x, p.x), fmaf(dt, o.vel.y, p.y), fmaf(dt, o.vel.z, p.z));
    return o;
}

constexpr uint32_t kTile = kBlockTile;   // tets (= threads) per workgroup tile; host_prep.cpp cuts the tiles with the same constant

// LDS per workgroup is 18 KB (4 + 12 + 2) so that 8 workgroups fit a CU's 160 KB: with 22.5 KB only 7
// fit and the 3900 tiles of the 1 M-tet lattice need 2.18 "rounds" of the chip instead of 1.9.
// Timing ablations (fewer rotation iterations, no rest-shape write-back, unpeeled first iteration), per-tile phase stamps and the
// rotation-iteration histogram change the physics or the timing and exist only in the separate development build
// (-DTETSIM_ABLATION -> libtetsim_hip_ablation.so): pj_lab.h defines every TETSIM_LAB_* / TETSIM_DBG_* name used below as nothing
// (or as the product's compile-time constant: 9 iterations, peeled, every store) unless that build is being made.
// kLean: the constant-rest-shape formulation (TETSIM_FLAG_CONSTANT_REST_SHAPE), a compile-time choice: as a run-time flag it
// cost the default path 12 register moves per tet at the join of the two variants.
// kAlt: ghost particles (id >= nv_owned) are staged from d.ghost_alt instead of pos_pred's tail (peer-to-peer halo, odd substeps)
// kHaloWait: the halo-side tiles of a peer-to-peer body do their queue's hand-overs themselves (pjb_tet_kernel_x<.., TetHwait>): the first wave of
// every tile looks at V and at the neighbours' "arrived" words -- after the record loads are out, before the positions are asked for --
// and every position comes from the memory side: the kernel may have started before the data it waits for was written.
// TETSIM_FLAG_LEAN_STATE: the carried shape is relative to its own centroid, its fourth corner is minus the sum of the other three
// (one association for every kernel of this unit: the per-substep, fused and persistent kernels agree bit for bit)
__device__ __forceinline__ f3 lean_fourth_corner(const f3 r[4]) {
    return F3(-((r[0].x + r[1].x) + r[2].x), -((r[0].y + r[1].y) + r[2].y), -((r[0].z + r[1].z) + r[2].z));
}
struct PJHaloWait { const PJPeerSync* w; const uint32_t* vflag; uint32_t* error; uint32_t timeout_ms; const float4* ghosts; };
// kMode: what a tet's record in HBM is (blk_mode(): a property of the body, a compile-time choice of the kernel)
//   0  the reference's formulation: the carried shape (48 B in, 48 B out) and the quaternion (16 B in, 16 B out) -- 148 B per tet
//   1  TETSIM_FLAG_CONSTANT_REST_SHAPE (kLean): the constant rest shape in, the quaternion in and out -- 100 B per tet
//   2  TETSIM_FLAG_LEAN_STATE: THREE corners of the carried shape in and out (36 B each way: the shape is relative to its own
//      centroid, so the fourth corner is minus the sum of the other three) and NO quaternion -- in FAST arithmetic the substep's
//      rotation `rel` is normalize(q) of this substep alone (pj_math.inc), the accumulated quaternion is pure output, and the carried
//      shape IS the accumulated rotation applied to the rest shape: tetsim_read_quats / the visual mesh / checkpoints recover it
//      from there when they are asked (skin_kernels.hip: pjb_recover_quat_kernel) -- 92 B per tet
constexpr int kModeCarried = 0, kModeConstantRest = 1, kModeLeanState = 2;
inline int blk_mode(const PJBlk& d) { return pjb_mode(d); }
// stamp: what goes into the (otherwise unused) fourth float of every partial sum -- 0, or the sequence number of the substep when the
// particle pass runs in the SAME launch and looks for it (pjb_substep_kernel below)
// kPoll (pjb_call_kernel below): the tile runs in a launch that spans SEVERAL substeps -- its block index inside its substep is poll->block,
// and the predictions it stages were written by particle waves of the same launch: memory-side loads, looked at until they carry the
// previous substep's sequence number poll->want (0: the launch's first substep, whose predictions a completed launch left)
struct PJPoll { uint32_t block, want; uint32_t* error; uint32_t timeout_ms; };
template <int kMode, bool kFused, bool kAlt = false, bool kHaloWait = false, bool kPoll = false>
__device__ __forceinline__ void pjb_tet_body(const PJBlk& d, uint32_t tile_first, uint32_t tile_count, uint32_t tiles_per_xcd TETSIM_DBG_PARAM,
                                             [[maybe_unused]] const PJHaloWait* hw = nullptr, const uint32_t stamp = 0u, [[maybe_unused]] const PJPoll* poll = nullptr) {
    __shared__ float4 s_pos[kTile];        // staged particle positions
    __shared__ float s_gx[4 * kTile];      // V*goal per corner, plane-major [corner][tet], one plane per component
    __shared__ float s_gy[4 * kTile];
    __shared__ float s_gz[4 * kTile];
    __shared__ uint2 s_ent[kTile];         // the tile's reduction order, 4 x u16 per tet position

    constexpr bool kLean = kMode == kModeConstantRest;
    uint32_t block_index = blockIdx.x;
    if constexpr (kPoll) block_index = poll->block;
    const uint32_t rel = xcd_tile(block_index, tiles_per_xcd);
    if (rel >= tile_count) return;  // whole workgroup leaves together
    const uint32_t b = tile_first + rel;
    const uint32_t tid = threadIdx.x;
    TETSIM_LAB_TILE_BEGIN();
    const uint32_t t0 = d.blk_tet_off[b], ntb = d.blk_tet_off[b + 1] - t0;
    const uint32_t v0 = d.blk_vert_off[b], nu = d.blk_vert_off[b + 1] - v0;

    // 1. stage the tile's particles.  Order of ISSUE matters: the particle gather is a dependent pair (slot -> particle id ->
    // position) and the tet record is independent of it, so the ids are requested first, then the whole tet record, and only
    // then the positions (which wait for the ids alone: vmcnt is in order, the tet loads stay in flight) -- two memory round
    // trips per tile.  Written in source order "gather, then tet record" the compiler has to finish the gather (its LDS store
    // needs the data) before it may issue the tet loads: three round trips, the load phase being a third of a tile's life.
    // All of it is issued UNCONDITIONALLY, lanes without a slot / without a tet re-reading element 0 of the tile (one
    // request per wave, results unused): with the loads inside `if`s the wait-count pass sees paths with different numbers of
    // loads in flight and, at the join, waits for the position gather's operand with the most conservative count -- which
    // lets (almost) the whole tet record land first, i.e. serialises the round trips again.
    const bool has_slot = tid < nu, has_tet = tid < ntb;
    const uint32_t slot = v0 + (has_slot ? tid : 0u), e = t0 + (has_tet ? tid : 0u);
    const uint32_t vid = static_cast<uint32_t>(d.blk_verts[slot]);   // (unsigned: no sign extension right behind the load)
    const uint32_t range = has_slot ? d.lc_range[slot] : 0u;
    // kFused: the particle update of the PREVIOUS substep happens here, for this tile's own particles (a particle in k tiles is
    // updated k times, identically; the tile holding the first of its partial sums -- `owner` -- writes the result back): the
    // slot's list of partial sums is requested together with its particle id (round trip 1), then the sums themselves, the
    // particle's previous position and weight together with the tet record (round trip 2).  One kernel per substep instead of
    // two: the particle kernel's launch, its 30 MB and a second dependency bubble per substep are gone, for one more gather
    // level here.  Partial sums and positions are double buffered (other tiles still read the old ones).
    uint32_t src[8];
    uint32_t src8 = 0xffffffffu;
    [[maybe_unused]] uint32_t maxsrc = 0;
    if constexpr (kFused) {
        maxsrc = d.blk_maxsrc[b];
        const uint32_t* col = d.slot_src + slot;
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) src[j] = (j < maxsrc) ? col[static_cast<size_t>(j) * d.ns_pad] : 0xffffffffu;   // (uniform)
        if (maxsrc > 8u) src8 = col[8ull * d.ns_pad];
    }
    const uchar4 li = d.tet_lidx[e];
    // (kPoll: the record was written by this tile's workgroup of the PREVIOUS substep of the same launch -- no kernel boundary has emptied
    // this CU's first-level cache of the line it may have read two substeps ago: memory-side loads, like the looks at the predictions)
    // ... and only AFTER the looks at the predictions have succeeded (sub > 0): a tile stores its partial sums behind the completion of its
    // record stores (below), the particle lanes store their predictions behind their looks at those sums, so a stamped prediction says the
    // record is there -- a record requested together with the predictions could be the substep before's.  One more dependent trip per
    // tile, hidden by the seven other workgroups of the CU; the launch's first substep keeps the two trips.
    auto rec4 = [](const float4* a, uint32_t i) { if constexpr (kPoll) return load_coherent(a, i); else return a[i]; };
    float4 ra = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rb = ra, rc = ra, q_old = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    auto load_record = [&]() {
        ra = rec4(d.rest_a, e); rb = rec4(d.rest_b, e);
        if constexpr (kMode == kModeLeanState) { if constexpr (kPoll) rc.x = load_coherent1(d.rest_c1, e); else rc.x = d.rest_c1[e]; }   // (r2.z: the ninth float)
        else { rc = rec4(d.rest_c, e); q_old = rec4(d.quat, e); }
    };
    bool record_late = false;
    if constexpr (kPoll) record_late = poll->want != 0u;   // (uniform)
    if (!record_late) load_record();
    const float V = d.vol[e];
    const uint2 ent_row = d.lc_ent[e];
    float4 pos_stage;
    if constexpr (kFused) {
        // (a ghost would keep its received prediction; fused bodies have none)
        f3 g[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {   // ascending tile order, absent = +0: the particle kernel's order of additions
            const float4 t = d.partial_prev[src[j] == 0xffffffffu ? 0u : src[j]];
            g[j] = src[j] == 0xffffffffu ? F3(0.0f, 0.0f, 0.0f) : xyz(t);
        }
        const float4 t8 = d.partial_prev[src8 == 0xffffffffu ? 0u : src8];
        const float4 prev4 = d.fin_in[vid];
        const float wsum = d.wsum[vid];
        f3 acc = F3(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) { acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; }
        if (src8 != 0xffffffffu) { acc.x += t8.x; acc.y += t8.y; acc.z += t8.z; }
        if (maxsrc > 9u) {   // (uniform, rare: a tile with a particle at the corner of more than nine tiles -- irregular meshes; one trip per entry)
            const uint32_t* col = d.slot_src + slot;
            for (uint32_t j = 9u; j < maxsrc; j++) {
                const uint32_t sj = col[static_cast<size_t>(j) * d.ns_pad];
                const float4 tj = d.partial_prev[sj == 0xffffffffu ? 0u : sj];
                if (sj != 0xffffffffu) { acc.x += tj.x; acc.y += tj.y; acc.z += tj.z; }
            }
        }
        const VertexOut o = pjb_vertex_update(acc, wsum, xyz(prev4), *d.params, vid);
        pos_stage = make_float4(o.pred.x, o.pred.y, o.pred.z, 0.0f);
        if (has_slot && ((range >> 15) & 1u)) {   // one writer per particle
            store_wt(d.fin_out, vid, make_float4(o.p.x, o.p.y, o.p.z, 0.0f));
            if (d.vel) store_wt(d.vel, vid, make_float4(o.vel.x, o.vel.y, o.vel.z, 0.0f));
        }
    } else if constexpr (kHaloWait) {
        if (tid < 64u) {
            const PJPeerSync& w = *hw->w;
            const uint32_t* word = tid == 0u ? hw->vflag : (tid <= w.n_wait ? w.wait[tid - 1u] : nullptr);
            bool pending = word != nullptr;
            const long long t0 = wall_clock64(), limit = 100000ll * hw->timeout_ms;   // 100 MHz ticks
            while (true) {   // (relaxed looks: an acquire load is a load plus a cache invalidation)
                if (pending) pending = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u;
                if (__builtin_amdgcn_ballot_w64(pending) == 0ull) break;
                __builtin_amdgcn_s_sleep(16);
                if (limit && wall_clock64() - t0 > limit) { if (tid == 0) __hip_atomic_store(hw->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
            if (w.delay_us && w.n_wait) { const long long d0 = wall_clock64(); while (wall_clock64() - d0 < 100ll * w.delay_us) __builtin_amdgcn_s_sleep(8); }
        }
        __syncthreads();
        const bool ghost = vid >= d.nv_owned;
        const float4 own = load_coherent(d.pos_pred, ghost ? 0u : vid), gh = load_coherent(hw->ghosts, ghost ? vid - d.nv_owned : 0u);
        pos_stage = ghost ? gh : own;
    } else if constexpr (kAlt) {
        const uint32_t g = vid - d.nv_owned;   // (ghost index; two-layer regions keep the layers in separate buffers)
        const float4* src = vid >= d.nv_owned ? (g < d.n_ghost1 ? d.ghost_alt + g : d.ghost2 + (g - d.n_ghost1)) : d.pos_pred + vid;
        pos_stage = *src;
    } else if constexpr (kPoll) {
        pos_stage = load_coherent(d.pos_pred, vid);
        if (poll->want != 0u) {
            bool pend = has_slot && __float_as_uint(pos_stage.w) != poll->want;
            if (__builtin_amdgcn_ballot_w64(pend) != 0ull) {
                const long long t0 = wall_clock64(), limit = 100000ll * poll->timeout_ms;   // 100 MHz ticks; 0 = unbounded
                do {
                    __builtin_amdgcn_s_sleep(TETSIM_POLL_SLEEP);
                    asm volatile("" ::: "memory");   // (every look is a fresh load)
                    if (pend) { pos_stage = load_coherent(d.pos_pred, vid); pend = __float_as_uint(pos_stage.w) != poll->want; }
                    if (limit && pend && wall_clock64() - t0 > limit) { __hip_atomic_store(poll->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); pend = false; }
                } while (__builtin_amdgcn_ballot_w64(pend) != 0ull);
            }
            __syncthreads();   // (every wave's looks have succeeded: every particle of the tile is stamped, i.e. every neighbouring tile's record too)
            load_record();
        }
    } else {
        pos_stage = d.pos_pred[vid];
    }
    if (has_slot) s_pos[tid] = pos_stage;
    if (has_tet) s_ent[tid] = ent_row;
    TETSIM_STAMP(1);  // loads issued (and landed, for this wave)
    // Every load above has been consumed by now on the path that has tets; say so for ALL paths.  Without this, the
    // wait-count model keeps "load into v[..] pending" alive through the path that skips the solve, and protects the reuse
    // of those registers after the join with vmcnt(0) -- which on gfx9 (one in-order counter for loads and stores) makes
    // every wave sit out its own write-through result stores' trip to HBM before the partial-sum reduction.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    TETSIM_STAMP(2);  // tile staged

    // 2. solve
    if (has_tet) {
        f3 cur[4], rest[4], goal[4];
        cur[0] = xyz(s_pos[li.x]); cur[1] = xyz(s_pos[li.y]); cur[2] = xyz(s_pos[li.z]); cur[3] = xyz(s_pos[li.w]);
        rest[0] = F3(ra.x, ra.y, ra.z); rest[1] = F3(ra.w, rb.x, rb.y);
        rest[2] = F3(rb.z, rb.w, rc.x); rest[3] = F3(rc.y, rc.z, rc.w);
        if constexpr (kMode == kModeLeanState) rest[3] = lean_fourth_corner(rest);
        float4 q_new;
        // The carried shape lives in HBM relative to its own centroid (see pj_solve_tet): `goal` comes back centred, the
        // world-space goal is goal + cc.  Same bytes as the reference's world-space shape, 21 instructions fewer per tet
        // (no rest centroid, no subtraction), and without the add-then-subtract of a position-sized number every substep.
        f3 cc;
        TETSIM_LAB_SOLVE_TET(cur, rest, q_old, q_new, goal, cc);   // pj_solve_tet(..., 9 iterations, peeled, kLean, centred, &cc, d.rot_exit_w2)
        TETSIM_STAMP(3);  // solved
        // LDS staging first, global results after it: nothing below may have to wait for the write-through stores
        const f3 vcc = cc * V;
#pragma unroll
        for (int k = 0; k < 4; k++) {  // V * (goal + cc)
            s_gx[k * kTile + tid] = fmaf(goal[k].x, V, vcc.x);
            s_gy[k * kTile + tid] = fmaf(goal[k].y, V, vcc.y);
            s_gz[k * kTile + tid] = fmaf(goal[k].z, V, vcc.z);
        }
        if constexpr (kMode != kModeLeanState) store_wt(d.quat, e, q_new);
        if (!kLean && TETSIM_DBG_STORE_REST) {  // constant-rest-shape bodies never write the shape back
            store_wt(d.rest_a, e, make_float4(goal[0].x, goal[0].y, goal[0].z, goal[1].x));
            store_wt(d.rest_b, e, make_float4(goal[1].y, goal[1].z, goal[2].x, goal[2].y));
            if constexpr (kMode == kModeLeanState) store_wt1(d.rest_c1, e, goal[2].z);
            else store_wt(d.rest_c, e, make_float4(goal[2].z, goal[3].x, goal[3].y, goal[3].z));
        }
    }
    TETSIM_STAMP(4);  // stores issued
    if constexpr (kPoll) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's record stores have LANDED (write-through: at the memory side) before any partial sum of the tile is stored
    __syncthreads();
    TETSIM_STAMP(5);

    // 3. one lane per tile particle: fixed-order sum of its corner goals -> one partial per (tile, particle)
    if (tid < nu) {
        const uint32_t first = range & 0x7ffu, last = range >> 16;   // (bit 15: owner flag)
        const uint16_t* ent = reinterpret_cast<const uint16_t*>(s_ent);
        // An entry is the word offset corner * kTile + tet of a corner goal in the planes (host_prep.cpp), so its byte offset
        // and the byte offset of its tet's weight are one shift and one mask.  4 entries per trip: the 4 index reads, then
        // the 16 value reads, are independent LDS ops in flight together (one entry at a time is a ~9-deep chain of
        // dependent LDS round trips: 4.9k cycles per tile in the s_memtime trace); whole groups of 4 run unmasked, the
        // 0-3 left over one by one.  Accumulation order stays entry order, i.e. deterministic.
        auto plane = [](const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
        // The weight sum of a particle (sum of V over its entries, then over its tiles) never changes -- rest volumes are
        // constants -- so it is not reduced here at all: the host adds it up once, in this very order (tetsim_create.hip ->
        // PJBlk::wsum), and the particle pass reads it.  Three planes per entry instead of four.
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t i = first;
        for (; i + 4u <= last; i += 4u) {
            uint32_t o[4];
            float gx[4], gy[4], gz[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) o[j] = static_cast<uint32_t>(ent[i + j]) << 2;
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) { gx[j] = plane(s_gx, o[j]); gy[j] = plane(s_gy, o[j]); gz[j] = plane(s_gz, o[j]); }
#pragma unroll
            for (uint32_t j = 0; j < 4u; j++) { acc.x += gx[j]; acc.y += gy[j]; acc.z += gz[j]; }
        }
        for (; i < last; i++) {
            const uint32_t o = static_cast<uint32_t>(ent[i]) << 2;
            acc.x += plane(s_gx, o); acc.y += plane(s_gy, o); acc.z += plane(s_gz, o);
        }
        acc.w = __uint_as_float(stamp);
        store_wt(d.partial, v0 + tid, acc);
    }
    TETSIM_STAMP(6);
}

__global__ __launch_bounds__(kTile, 2) void pjb_tet_kernel(PJBlk d, uint32_t tile_first, uint32_t tile_count,
                                                        uint32_t tiles_per_xcd TETSIM_DBG_PARAM) {
    pjb_tet_body<kModeCarried, false>(d, tile_first, tile_count, tiles_per_xcd TETSIM_DBG_ARG);
}
__global__ __launch_bounds__(kTile, 2) void pjb_tet_kernel_constant_rest(PJBlk d, uint32_t tile_first, uint32_t tile_count,
                                                                      uint32_t tiles_per_xcd TETSIM_DBG_PARAM) {
    pjb_tet_body<kModeConstantRest, false>(d, tile_first, tile_count, tiles_per_xcd TETSIM_DBG_ARG);
}
__global__ __launch_bounds__(kTile, 2) void pjb_tet_kernel_lean(PJBlk d, uint32_t tile_first, uint32_t tile_count,
                                                             uint32_t tiles_per_xcd TETSIM_DBG_PARAM) {
    pjb_tet_body<kModeLeanState, false>(d, tile_first, tile_count, tiles_per_xcd TETSIM_DBG_ARG);
}
// The variants partitioned and mid-sized bodies need are ONE kernel template over what happens around the tile's solve:
//   TetRaise  raises a hand-over word as it STARTS (partitioned bodies): a kernel starts only when everything in front
//             of it in its in-order queue is complete, so "the previous kernel of this queue is done" costs one store of one thread here
//             instead of a signal kernel of its own (~2.7 us of queue time each, and the two-queue substep had two of them) -- and, in
//             front of the raise, puts back a word whose waiters were the waves of the kernel in front of this one in its queue (the
//             interior particle kernel that waits for G itself, pjb_vertex_kernel_await: all of its waves are through when this starts);
//   TetAlt    reads the ghosts from the second buffer (peer-to-peer halo; halo-side tiles of odd substeps);
//   TetHwait  does the halo queue's hand-overs itself (peer-to-peer halo, one rank per process): as the kernel STARTS its first thread
//             tells the neighbours "my boundary predictions of the previous substep are in your ghost range" (the kernel in front of
//             it in this queue is this rank's boundary-particle kernel), every tile waits for V and the neighbours' words (kHaloWait),
//             and nobody consumes them: the boundary-particle kernel behind this one puts them back as it starts
//             (pjb_vertex_kernel_peer).  The one-wave wait kernel this replaces was a launch boundary on the chain that decides how
//             much wire latency a rank can hide;
//   TetFused  the previous substep's particle update fused into the staging (unpartitioned bodies of < 2,048 tiles, tetsim_step_n).
// The plain kernel of the headline keeps its own two names above (profiles and counters are keyed by them).
__device__ __forceinline__ void clear_then_raise(uint32_t* clear, uint32_t* sig) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (clear) __hip_atomic_store(clear, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sig) __hip_atomic_store(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}
struct TetRaise { uint32_t* sig; uint32_t* clear; };
struct TetAlt { uint32_t unused; };
struct TetHwait { PJPeerSync w; const uint32_t* vflag; uint32_t* error; uint32_t timeout_ms; const float4* ghosts; };
struct TetFused { uint32_t unused; };
template <int kMode, class Extra>
__global__ __launch_bounds__(kTile, 2) void pjb_tet_kernel_x(PJBlk d, uint32_t tile_first, uint32_t tile_count, uint32_t tiles_per_xcd, Extra x TETSIM_DBG_PARAM) {
    if constexpr (std::is_same_v<Extra, TetRaise>) {
        clear_then_raise(x.clear, x.sig);
        pjb_tet_body<kMode, false>(d, tile_first, tile_count, tiles_per_xcd TETSIM_DBG_ARG);
    } else if constexpr (std::is_same_v<Extra, TetAlt>) {
        pjb_tet_body<kMode, false, true>(d, tile_first, tile_count, tiles_per_xcd TETSIM_DBG_ARG);
    } else if constexpr (std::is_same_v<Extra, TetHwait>) {
        if (blockIdx.x == 0 && threadIdx.x < x.w.n_raise) __hip_atomic_store(x.w.raise[threadIdx.x], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const PJHaloWait hw = {&x.w, x.vflag, x.error, x.timeout_ms, x.ghosts};
        pjb_tet_body<kMode, false, false, true>(d, tile_first, tile_count, tiles_per_xcd TETSIM_DBG_ARG, &hw);
    } else {
        pjb_tet_body<kMode, true>(d, tile_first, tile_count, tiles_per_xcd TETSIM_DBG_ARG);
    }
}
// one launcher for all of them: the kernel by (constant rest shape?, variant), its own begin / end events on request
template <class Extra>
void launch_tet_x(hipStream_t s, const PJBlk& d, uint32_t tile_first, uint32_t tile_count, const Extra& x, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    if (tile_count == 0) return;
    const uint32_t per_xcd = (tile_count + 7u) / 8u;
    const int mode = blk_mode(d);
    auto* kernel = mode == kModeConstantRest ? pjb_tet_kernel_x<kModeConstantRest, Extra> : mode == kModeLeanState ? pjb_tet_kernel_x<kModeLeanState, Extra> : pjb_tet_kernel_x<kModeCarried, Extra>;
    if (e0) hipExtLaunchKernelGGL(kernel, dim3(per_xcd * 8u), dim3(kTile), 0, s, e0, e1, 0, d, tile_first, tile_count, per_xcd, x TETSIM_DBG_LAUNCH);
    else hipLaunchKernelGGL(kernel, dim3(per_xcd * 8u), dim3(kTile), 0, s, d, tile_first, tile_count, per_xcd, x TETSIM_DBG_LAUNCH);
}

// ---- one persistent launch per tetsim_step_n call (small bodies) -------------------------------------------
// A Dragon-sized body (15 tiles) occupies 15 of the chip's 2,048 workgroup slots: its substep is launch boundaries and dependent
// memory trips and nothing else.  Here every tile's workgroup stays resident for all n substeps of a call:
//   * a lane keeps ITS tet's record -- carried rest shape, quaternion, volume, corner slots -- in registers from substep to
//     substep (read once, written back once), the tile's reduction order stays in LDS;
//   * a lane keeps ITS tile slot's particle -- previous position, weight, list of partial sums -- in registers too; positions and
//     velocities go to memory once, at the end of the call (by the slot that owns the particle);
//   * the only per-substep traffic between workgroups is the tile partial sums: written write-through with the substep's SEQUENCE
//     NUMBER in the unused fourth float, and read by the tiles sharing the particle with loads that bypass the caches
//     (dev_store.h: load_coherent) in a loop that ends when every sum carries the expected number.  Data and flag are one 16-byte
//     store / one 16-byte load: a substep's exchange is ONE memory trip, there is no barrier among workgroups and no flag array
//     (the persistent Gauss-Seidel sweep of round 2 paid three trips per colour for flag-then-data).  Buffers alternate by
//     substep parity: a tile overwrites its sums of substep s-2 only after it has consumed its neighbours' sums of s-1, which they
//     wrote after consuming this tile's sums of s-2.
// The arithmetic is the fused kernel's, operation for operation (same pjb_vertex_update, same order of additions), so a call
// equals the same number of tetsim_step calls bit for bit.  Sequence numbers start at DevParams::epoch, which the host advances
// by n per call (stale sums of an earlier call never match).  Every workgroup must be resident at once (a waiting tile holds its
// slot): the host only uses this kernel for bodies of at most pjb_frame_capacity() tiles; a wait is bounded all the same.
// kLocal: every group of tiles that exchange sums (a body) sits on ONE XCD (the host maps blocks to tiles accordingly and has
// verified the dispatcher's block -> XCD rule with pjb_probe_xcd), so the exchange only has to be coherent in that XCD's L2: plain
// stores (the L1 writes through) and agent-scope loads (miss the L1, served by the L2) -- dev_store.h.  !kLocal: write-through
// stores and cache-bypassing loads, coherent at the memory side (any placement).
constexpr int kFrameIters = TETSIM_LAB_FRAME_ITERS;   // the product's compile-time constant, 9 (tools/mutation_check.sh mutates it for both kernels)
template <int kMode, bool kLocal>
__device__ __forceinline__ void pjb_frame_body(const PJBlk& d, const DevParams& P, const uint32_t n, const int32_t* const block_tile, float4* const pbuf0, float4* const pbuf1,
                                               uint32_t* const err, const uint32_t timeout_ms) {
    __shared__ float4 s_pos[kTile];
    __shared__ float s_gx[4 * kTile];
    __shared__ float s_gy[4 * kTile];
    __shared__ float s_gz[4 * kTile];
    __shared__ uint2 s_ent[kTile];

    constexpr bool kLean = kMode == kModeConstantRest;
    const int32_t bt = block_tile[blockIdx.x];
    if (bt < 0) return;   // (a block that only pads the grid so that the others land on the intended XCDs)
    const uint32_t b = static_cast<uint32_t>(bt);
    const uint32_t tid = threadIdx.x;
    TETSIM_LAB_FRAME_BEGIN();
    const uint32_t t0 = d.blk_tet_off[b], ntb = d.blk_tet_off[b + 1] - t0;
    const uint32_t v0 = d.blk_vert_off[b], nu = d.blk_vert_off[b + 1] - v0;
    const bool has_slot = tid < nu, has_tet = tid < ntb;
    const uint32_t slot = v0 + (has_slot ? tid : 0u), e = t0 + (has_tet ? tid : 0u);
    // everything that stays for the whole call, requested together (ids first: the particle loads depend on them)
    const uint32_t vid = static_cast<uint32_t>(d.blk_verts[slot]);
    const uint32_t range = has_slot ? d.lc_range[slot] : 0u;
    const uint32_t maxsrc = d.blk_maxsrc[b];
    uint32_t src[9];
    {
        const uint32_t* col = d.slot_src + slot;
#pragma unroll
        for (uint32_t j = 0; j < 9u; j++) src[j] = (has_slot && j < maxsrc) ? col[static_cast<size_t>(j) * d.ns_pad] : 0xffffffffu;
    }
    const uchar4 li = d.tet_lidx[e];
    const float4 ra = d.rest_a[e], rb = d.rest_b[e];
    float4 rc = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if constexpr (kMode == kModeLeanState) rc.x = d.rest_c1[e];   // (lean state: three corners, no quaternion -- pjb_tet_body)
    else { rc = d.rest_c[e]; q = d.quat[e]; }
    const float V = d.vol[e];
    s_ent[tid] = has_tet ? d.lc_ent[e] : make_uint2(0u, 0u);
    f3 rest[4];
    rest[0] = F3(ra.x, ra.y, ra.z); rest[1] = F3(ra.w, rb.x, rb.y);
    rest[2] = F3(rb.z, rb.w, rc.x); rest[3] = F3(rc.y, rc.z, rc.w);
    f3 prev = xyz(d.pos_final[vid]);
    const float wsum = d.wsum[vid];
    f3 stage = xyz(d.pos_pred[vid]);     // substep 0 starts from the prediction the previous call left
    const uint32_t epoch = d.epoch ? d.epoch : P.epoch;   // (a direct launch -- tetsim_step -- brings its own block of sequence numbers)
    const uint32_t first = range & 0x7ffu, last = range >> 16;
    const bool owner = has_slot && ((range >> 15) & 1u);
    float4* const pbuf[2] = {pbuf0, pbuf1};
    const long long limit = 100000ll * timeout_ms;   // 100 MHz ticks; 0 = unbounded

    // the particle update of substep s from the partial sums of every tile that touches the particle (this one included):
    // ascending tile order, absent = +0 -- the particle kernel's order of additions
    auto gather_update = [&](const uint32_t s) -> VertexOut {
        const float4* buf = pbuf[s & 1u];
        const uint32_t expect = epoch + s;
        f3 g[9];
        uint32_t pend = 0;
#pragma unroll
        for (uint32_t j = 0; j < 9u; j++) { g[j] = F3(0.0f, 0.0f, 0.0f); pend |= (src[j] != 0xffffffffu ? 1u : 0u) << j; }
        const long long w0 = limit ? wall_clock64() : 0ll;
        while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
            // only the lanes (and list positions) that still miss a sum issue a request: a trip of the loop is as long as its
            // slowest request, and a wave that asked for 9 x 64 sums where ~100 are missing waited for 576 round trips' tail
            float4 t[9];
#pragma unroll
            for (uint32_t j = 0; j < 9u; j++)
                if ((pend >> j) & 1u) t[j] = kLocal ? load_l2(buf, src[j]) : load_coherent(buf, src[j]);
            FRAME_POLL();
#pragma unroll
            for (uint32_t j = 0; j < 9u; j++)
                if (((pend >> j) & 1u) && __float_as_uint(t[j].w) == expect) { g[j] = xyz(t[j]); pend &= ~(1u << j); }
            if (limit && pend != 0u && wall_clock64() - w0 > limit) {   // never in a correct run; a wedged GPU helps nobody
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                pend = 0u;
            }
        }
        f3 acc = F3(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) { acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; }
        if (src[8] != 0xffffffffu) { acc.x += g[8].x; acc.y += g[8].y; acc.z += g[8].z; }
        if (maxsrc > 9u) {   // (uniform, rare: lists of more than nine partial sums -- irregular meshes; entry by entry, the ids re-read from the table)
            const uint32_t* col = d.slot_src + slot;
            for (uint32_t j = 9u; j < maxsrc; j++) {
                const uint32_t sj = has_slot ? col[static_cast<size_t>(j) * d.ns_pad] : 0xffffffffu;
                bool miss = sj != 0xffffffffu;
                f3 gj = F3(0.0f, 0.0f, 0.0f);
                while (__builtin_amdgcn_ballot_w64(miss) != 0ull) {
                    if (miss) {
                        const float4 t = kLocal ? load_l2(buf, sj) : load_coherent(buf, sj);
                        if (__float_as_uint(t.w) == expect) { gj = xyz(t); miss = false; }
                    }
                    if (limit && miss && wall_clock64() - w0 > limit) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        miss = false;
                    }
                }
                acc.x += gj.x; acc.y += gj.y; acc.z += gj.z;
            }
        }
        return pjb_vertex_update(acc, wsum, prev, P, vid);
    };

    for (uint32_t s = 0; s < n; s++) {
        FRAME_STAMP(0);
        if (s > 0u) {
            const VertexOut o = gather_update(s - 1u);
            prev = o.p;
            stage = o.pred;
        }
        FRAME_STAMP(1);
        if (has_slot) s_pos[tid] = make_float4(stage.x, stage.y, stage.z, 0.0f);
        __syncthreads();
        FRAME_STAMP(2);
        if (has_tet) {
            f3 cur[4], r[4], goal[4];
            cur[0] = xyz(s_pos[li.x]); cur[1] = xyz(s_pos[li.y]); cur[2] = xyz(s_pos[li.z]); cur[3] = xyz(s_pos[li.w]);
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = rest[k];
            if constexpr (kMode == kModeLeanState) r[3] = lean_fourth_corner(rest);
            float4 q_new;
            f3 cc;
            pj_solve_tet(cur, r, q, q_new, goal, kFrameIters, true, kLean, !kLean, &cc, d.rot_exit_w2);
            if constexpr (kMode != kModeLeanState) q = q_new;
            if (!kLean) {
#pragma unroll
                for (int k = 0; k < 4; k++) rest[k] = goal[k];   // the carried shape stays in registers
            }
            const f3 vcc = cc * V;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                s_gx[k * kTile + tid] = fmaf(goal[k].x, V, vcc.x);
                s_gy[k * kTile + tid] = fmaf(goal[k].y, V, vcc.y);
                s_gz[k * kTile + tid] = fmaf(goal[k].z, V, vcc.z);
            }
        }
        FRAME_STAMP(3);
        __syncthreads();
        FRAME_STAMP(4);
        if (has_slot) {   // the tet kernel's reduction, entry for entry
            const uint16_t* ent = reinterpret_cast<const uint16_t*>(s_ent);
            auto plane = [](const float* base, uint32_t byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            uint32_t i = first;
            for (; i + 4u <= last; i += 4u) {
                uint32_t o[4];
                float gx[4], gy[4], gz[4];
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) o[j] = static_cast<uint32_t>(ent[i + j]) << 2;
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) { gx[j] = plane(s_gx, o[j]); gy[j] = plane(s_gy, o[j]); gz[j] = plane(s_gz, o[j]); }
#pragma unroll
                for (uint32_t j = 0; j < 4u; j++) { acc.x += gx[j]; acc.y += gy[j]; acc.z += gz[j]; }
            }
            for (; i < last; i++) {
                const uint32_t o = static_cast<uint32_t>(ent[i]) << 2;
                acc.x += plane(s_gx, o); acc.y += plane(s_gy, o); acc.z += plane(s_gz, o);
            }
            acc.w = __uint_as_float(epoch + s);   // data and "substep s is here" in one 16-byte store
            if constexpr (kLocal) store_plain(pbuf[s & 1u], v0 + tid, acc);
            else store_wt(pbuf[s & 1u], v0 + tid, acc);
        }
        FRAME_STAMP(5);
        // (no barrier here: the next trip writes s_pos, last read before the second barrier above, and the planes are rewritten
        // only behind the next trip's first barrier, which every reducing lane reaches after its reads)
    }
    // the particle update that ends the call, and the state back to memory: one writer per particle, every lane its own tet
    const VertexOut o = gather_update(n - 1u);
    TETSIM_LAB_FRAME_END();
    if (owner) {
        store_wt(d.pos_final, vid, make_float4(o.p.x, o.p.y, o.p.z, 0.0f));
        store_wt(d.vel, vid, make_float4(o.vel.x, o.vel.y, o.vel.z, 0.0f));
        store_wt(d.pos_pred, vid, make_float4(o.pred.x, o.pred.y, o.pred.z, 0.0f));
    }
    if (has_tet) {
        if constexpr (kMode != kModeLeanState) store_wt(d.quat, e, q);
        if (!kLean) {
            store_wt(d.rest_a, e, make_float4(rest[0].x, rest[0].y, rest[0].z, rest[1].x));
            store_wt(d.rest_b, e, make_float4(rest[1].y, rest[1].z, rest[2].x, rest[2].y));
            if constexpr (kMode == kModeLeanState) store_wt1(d.rest_c1, e, rest[2].z);
            else store_wt(d.rest_c, e, make_float4(rest[2].z, rest[3].x, rest[3].y, rest[3].z));
        }
    }
}
#define TETSIM_FRAME_KERNEL(name, mode, local)                                                                                         \
    __global__ __launch_bounds__(kTile, 2) void name(PJBlk d, uint32_t n, const int32_t* block_tile, float4* pbuf0, float4* pbuf1, uint32_t* err, \
                                                     uint32_t timeout_ms, DevParams pv, DevParams* pdev) {                             \
        if (blockIdx.x == 0u && threadIdx.x == 0u) *pdev = pv; /* (parameters by value: see pjb_call_kernel) */                      \
        pjb_frame_body<mode, local>(d, pv, n, block_tile, pbuf0, pbuf1, err, timeout_ms);                                              \
    }
TETSIM_FRAME_KERNEL(pjb_frame_kernel, kModeCarried, false)
TETSIM_FRAME_KERNEL(pjb_frame_kernel_constant_rest, kModeConstantRest, false)
TETSIM_FRAME_KERNEL(pjb_frame_kernel_lean, kModeLeanState, false)
TETSIM_FRAME_KERNEL(pjb_frame_kernel_local, kModeCarried, true)
TETSIM_FRAME_KERNEL(pjb_frame_kernel_constant_rest_local, kModeConstantRest, true)
TETSIM_FRAME_KERNEL(pjb_frame_kernel_lean_local, kModeLeanState, true)
#undef TETSIM_FRAME_KERNEL
// which XCD runs block i of a grid: the dispatcher hands consecutive workgroups to consecutive XCDs (round-robin), which the
// host verifies with this kernel before it relies on it (hardware register XCC_ID)
__global__ void pjb_probe_xcd_kernel(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
}

// ---- cross-queue hand-over of partitioned bodies ----------------------------------------------------------
// A partitioned substep runs on two queues (halo-side tiles, boundary particles and the transfer on the halo stream, everything
// else on the main stream; tetsim_halo.hip: enqueue_phase_a).  A dependency between the queues costs ~15 us as an event (eager) and ~6 us as a fork/join edge of a captured graph
// on this stack (tools/micro/xq_latency.hip), and the cycle  particles(s) -> transfer(s) -> H tiles(s+1) -> particles(s+1)
// has two of them.  Here the producer queue sets a word behind its kernel (in-order queue: the kernel and its agent-scope
// release are complete) -- as the first store of the NEXT kernel of that queue (the *_raise kernels below), or from a one-wave
// signal kernel where no kernel follows -- and the consumer queue runs a one-wave wait kernel in front of the dependent kernel
// (whose own start then performs the agent-scope acquire): ~2.7 us of queue time per one-wave kernel, nothing else is delayed.
// The word is a BINARY SEMAPHORE: signal stores 1, wait spins until it reads non-zero and stores 0 again.  That needs no
// sequence number -- the dependency cycle itself makes producer and consumer alternate strictly (the next signal of a word is
// behind the completion of the wait that consumed the previous one) -- so both kernels take CONSTANT arguments and the two
// queues' chains can be replayed from captured graphs (tetsim_step_n; a per-launch sequence number was what kept this path
// eager, and a counter kept in device memory cost two more memory round trips per hand-over).
// The wait is bounded (timeout_ms, 30 s by default; then *error is raised and it carries on): a wedged peer must not wedge
// this GPU.  The host submits every signal before the matching wait, so even a single shared hardware queue stays live in
// eager mode (graph replay is only used when a probe at set-up found the two streams on independent hardware queues).
// Tried and dropped: the hand-over inside the compute kernels (last-workgroup detection / every workgroup polling on entry).
// Any per-workgroup device-scope atomic -- read-modify-write or load, one word or 64 words a cache line apart -- costs
// ~20 ns and serialises: the 2,744-workgroup particle pass went from 7 us to 60-400 us.
__device__ __forceinline__ void await_done(const PJSync& y) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64(), limit = 100000ll * y.timeout_ms;   // 100 MHz ticks
        while (__hip_atomic_load(y.flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(4);
            if (limit && wall_clock64() - t0 > limit) {
                __hip_atomic_store(y.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
        __hip_atomic_store(y.flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed
    }
}

// 64-thread workgroups: 175,616 particles are only 2,744 waves (2.7 per SIMD); one-wave workgroups spread over the
// 256 CUs evenly (10.7 per CU) where 256-thread ones leave some CUs with 3 and others with 2.
// kCoherent: the partial sums are read from the memory side (dev_store.h: load_coherent) -- for the kernel that starts BEFORE the
// halo-side tiles of its substep are known to be done (pjb_vertex_kernel_await): a line this XCD's L2 cached earlier must not be served
template <bool kPeer = false, bool kCoherent = false>
__device__ __forceinline__ void pjb_vertex_body(const PJBlk& d, uint32_t first, uint32_t count, const PJPeer* peer = nullptr) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= count) return;
    const uint32_t v = first + i;

    // P5 (SoftbodyGPU.js:302-320) from tile partial sums, ascending tile order.  The index lists are ELL
    // (column-major, coalesced, no offset lookup first), fetched 8 columns at a time: the kernel is two dependent memory
    // round trips (indices, then partial sums) and little else, so a trip must cover almost every particle -- 8 columns
    // do on the lattice (up to 9 partials; 18 particles have 9), where 4 columns made every wave pay three trips.
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t* col = d.vp_ell + v;
    for (uint32_t j0 = 0; j0 < d.vp_cols; j0 += 8u) {
        uint32_t idx[8];
        float4 g[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) idx[j] = (j0 + j < d.vp_cols) ? col[static_cast<size_t>(j0 + j) * d.nv_pad] : 0xffffffffu;
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {
            if constexpr (kCoherent) { const float4 t = load_coherent(d.partial, idx[j] != 0xffffffffu ? idx[j] : 0u); g[j] = idx[j] != 0xffffffffu ? t : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
            else g[j] = idx[j] != 0xffffffffu ? d.partial[idx[j]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) { acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; }
        if (__all(idx[7] == 0xffffffffu)) break;  // lists are front-packed: nobody in this wave has a ninth partial
    }
    // (the weight: sum of the rest volumes of the particle's live corners -- a constant, added up on the host in the tiles' entry order)
    const VertexOut o = pjb_vertex_update(xyz(acc), d.wsum[v], xyz(d.fin_in[v]), *d.params, v);
    store_wt(d.fin_out, v, make_float4(o.p.x, o.p.y, o.p.z, 0.0f));
    if (d.vel) store_wt(d.vel, v, make_float4(o.vel.x, o.vel.y, o.vel.z, 0.0f));   // (null between two substeps of one call: nothing reads it there)
    store_wt(d.pos_pred, v, make_float4(o.pred.x, o.pred.y, o.pred.z, 0.0f));
    if constexpr (kPeer) {
        // peer-to-peer halo: the prediction also goes straight into the ghost range of every neighbour that reads this particle
        // (write-through, system scope: peer memory over xGMI, or this device's own memory for partitions sharing a GPU).  The
        // "it is there" word follows when the next kernel of this queue starts (pjb_wait_peers_kernel).
        const float4 pred4 = make_float4(o.pred.x, o.pred.y, o.pred.z, 0.0f), fin4 = make_float4(o.p.x, o.p.y, o.p.z, 0.0f);
        const uint32_t* col = peer->slots + v;
        for (uint32_t c = 0; c < peer->cols; c++) {
            const uint32_t e = col[static_cast<size_t>(c) * peer->stride];
            for (uint32_t k = 0; k < peer->n; k++) {   // (the store wants a wave-uniform base: one neighbour at a time, the others' lanes masked)
                if (peer->ghost[k] && e != 0xffffffffu && (e >> 24) == k) store_wt(peer->ghost[k], e & 0xffffffu, pred4);
                if (peer->fin[k] && e != 0xffffffffu && (e >> 24) == k) store_wt(peer->fin[k], e & 0xffffffu, fin4);
            }
        }
        if (peer->slots2) {
            const uint32_t* col2 = peer->slots2 + v;
            for (uint32_t c = 0; c < peer->cols2; c++) {
                const uint32_t e = col2[static_cast<size_t>(c) * peer->stride];
                for (uint32_t k = 0; k < peer->n; k++)
                    if (peer->ghost2[k] && e != 0xffffffffu && (e >> 24) == k) store_wt(peer->ghost2[k], e & 0xffffffu, pred4);
            }
        }
    }
}

__global__ __launch_bounds__(64) void pjb_vertex_kernel(PJBlk d, uint32_t first, uint32_t count) { pjb_vertex_body(d, first, count); }

// ---- a CALL as one launch (large unpartitioned bodies: tetsim_step_n) -------------------------------------------------------------------
// All n substeps of a call in one grid: per substep the tiles' workgroups, then the particles' (four waves of 64 per workgroup), substep
// after substep.  The dispatcher hands out workgroups in grid order, so whatever a wave waits for has been dispatched before it -- what
// it cannot know is whether it is DONE.  So data carries its stamp, as the persistent frame kernel's partial sums do (pjb_frame_body):
//   * every partial sum has the substep's sequence number in its fourth float (one 16-byte store = data + "it is there"); a particle wave
//     looks at its particle's (up to nine) sums until they carry it;
//   * every prediction has it too; a tile of the NEXT substep looks at the predictions it stages until they carry it.
// All looks are memory-side loads (the writer may sit on another XCD, no kernel boundary in between) and bounded.  Nothing is double
// buffered: a tile's partial sums are read by the particle lanes of its own particles only, and the tile overwrites them after it has
// staged those particles' next predictions, which their lanes wrote after reading the sums; a prediction is overwritten by its lane after
// all tiles around the particle delivered their sums, i.e. after they staged it.  No launch boundary inside a call, and the particle
// pass -- two dependent memory trips and little else -- and the tiles' staging run under the neighbouring passes' compute.  The
// arithmetic is pjb_tet_body's and pjb_vertex_update's, operation for operation: a call equals the same substeps through the two
// kernels (tetsim_step, tetsim_profile) bit for bit.
//   The call's parameters arrive BY VALUE, with the launch: the 176-byte upload in front of every call (a copy, its event, the switch between
// the copy engine and the compute queue) was most of the ~18 us the device idled between two calls (tools/call_gap.py).  The first workgroup
// leaves them in DevParams for the kernels behind this one (tetsim_step's pair, the read-outs).
template <int kMode>
__global__ __launch_bounds__(kTile, 2) void pjb_call_kernel(PJBlk d, uint32_t n_sub, uint32_t tile_count, uint32_t tiles_per_xcd, uint32_t tet_blocks, uint32_t blocks_per_sub,
                                                           uint32_t* err, uint32_t timeout_ms, DevParams pv, DevParams* pdev TETSIM_DBG_PARAM) {
    const uint32_t sub = blockIdx.x / blocks_per_sub, r = blockIdx.x - sub * blocks_per_sub;   // (blocks_per_sub is a multiple of 8: r and blockIdx.x land on the same XCD)
    const uint32_t stamp = (d.epoch ? d.epoch : pv.epoch) + sub + 1u;
    if (blockIdx.x == 0u && threadIdx.x == 0u) *pdev = pv;
    if (r < tet_blocks) {
        const PJPoll poll = {r, sub ? stamp - 1u : 0u, err, timeout_ms};
        pjb_tet_body<kMode, false, false, false, true>(d, 0u, tile_count, tiles_per_xcd TETSIM_DBG_ARG, nullptr, stamp, &poll);
        return;
    }
    const uint32_t v = (r - tet_blocks) * kTile + threadIdx.x;
    if (v >= d.nv_owned) return;
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t* col = d.vp_ell + v;
    const long long limit = 100000ll * timeout_ms;   // 100 MHz ticks; 0 = unbounded
    const float wsum = d.wsum[v];
    // (the store of this particle's lane one substep before -- another wave of the same launch: past the caches, and looked at again below
    // until it carries that substep's number: it was stored ~a tile's lifetime ago, the first look finds it)
    float4 prev = load_coherent(d.fin_in, v);
    for (uint32_t j0 = 0; j0 < d.vp_cols; j0 += 8u) {   // (pjb_vertex_body's gather, every sum looked at until it is this substep's)
        uint32_t idx[8];
        float4 g[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) idx[j] = (j0 + j < d.vp_cols) ? col[static_cast<size_t>(j0 + j) * d.nv_pad] : 0xffffffffu;
        uint32_t pend = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {
            const float4 t = load_coherent(d.partial, idx[j] != 0xffffffffu ? idx[j] : 0u);
            g[j] = idx[j] != 0xffffffffu ? t : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            pend |= (idx[j] != 0xffffffffu && __float_as_uint(t.w) != stamp ? 1u : 0u) << j;
        }
        if (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) {
            const long long t0 = wall_clock64();
            do {
                __builtin_amdgcn_s_sleep(TETSIM_POLL_SLEEP);
                asm volatile("" ::: "memory");   // (every look is a fresh load)
#pragma unroll
                for (uint32_t j = 0; j < 8u; j++)
                    if ((pend >> j) & 1u) {
                        const float4 t = load_coherent(d.partial, idx[j]);
                        if (__float_as_uint(t.w) == stamp) { g[j] = t; pend &= ~(1u << j); }
                    }
                if (limit && pend != 0u && wall_clock64() - t0 > limit) {   // never in a correct run; a wedged GPU helps nobody
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    pend = 0u;
                }
            } while (__builtin_amdgcn_ballot_w64(pend != 0u) != 0ull);
        }
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) { acc.x += g[j].x; acc.y += g[j].y; acc.z += g[j].z; }
        if (__all(idx[7] == 0xffffffffu)) break;
    }
    if (sub) {
        const long long t0 = wall_clock64();
        while (__float_as_uint(prev.w) != stamp - 1u) {
            __builtin_amdgcn_s_sleep(4);
            asm volatile("" ::: "memory");
            prev = load_coherent(d.fin_in, v);
            if (limit && wall_clock64() - t0 > limit) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
    }
    const VertexOut o = pjb_vertex_update(xyz(acc), wsum, xyz(prev), pv, v);
    store_wt(d.fin_out, v, make_float4(o.p.x, o.p.y, o.p.z, __uint_as_float(stamp)));
    if (d.vel && sub + 1u == n_sub) store_wt(d.vel, v, make_float4(o.vel.x, o.vel.y, o.vel.z, 0.0f));   // (the velocity array: behind a call's last substep only)
    store_wt(d.pos_pred, v, make_float4(o.pred.x, o.pred.y, o.pred.z, __uint_as_float(stamp)));
}
__global__ __launch_bounds__(64) void pjb_vertex_kernel_raise(PJBlk d, uint32_t first, uint32_t count, uint32_t* sig, uint32_t* clear) {
    clear_then_raise(clear, sig);
    pjb_vertex_body(d, first, count);
}
// ... waiting for a hand-over word ITSELF: every wave looks at the word before it touches a partial sum (one lane, agent-scope loads,
// bounded like the wait kernels) and nobody puts it back -- the next kernel of this queue does, as it starts (clear_then_raise).  The
// one-wave wait kernel this replaces cost the main queue a third launch boundary per substep.
__global__ __launch_bounds__(64) void pjb_vertex_kernel_await(PJBlk d, uint32_t first, uint32_t count, uint32_t* flag, uint32_t* error, uint32_t timeout_ms) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64(), limit = 100000ll * timeout_ms;   // 100 MHz ticks
        // (RELAXED: an acquire load is a load plus a cache invalidation, and thousands of waves invalidating their XCD's L2 on every
        // look took the substep from 37 to 53-150 us)
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(16);
            if (limit && wall_clock64() - t0 > limit) { __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // (no cache invalidation here either -- 2,744 waves invalidating their XCD's L2 took the substep from 37 to 71 us: the partial sums, the
    // only input another queue's kernel may still have been writing when this kernel started, are read past the caches instead)
    pjb_vertex_body<false, true>(d, first, count);
}
__global__ __launch_bounds__(64) void pjb_vertex_kernel_peer(PJBlk d, uint32_t first, uint32_t count, PJPeer peer, uint32_t* sig, PJClear clr) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // (the words the halo-side tiles in front of this kernel looked at: all of them are through)
        for (uint32_t i = 0; i < clr.n; i++) __hip_atomic_store(clr.word[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (sig) __hip_atomic_store(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    pjb_vertex_body<true>(d, first, count, &peer);
}
// One wave in front of the halo-side tiles of a peer-to-peer body: (1) as it STARTS, the boundary-particle kernel in front of it
// in this queue is complete, i.e. this rank's predictions are in the neighbours' ghost ranges -- tell them (one system-scope store
// per neighbour, into THEIR memory); (2) wait for the local word (V: this rank's interior particles) if there is one; (3) wait for
// the words the neighbours raise here, and clear them.  Words alternate by substep parity (the host passes the right pair): the
// raise of substep s+2 lands on the word of substep s only after its consumer has cleared it -- the dependency cycle orders
// them -- whereas ONE word per neighbour could see two raises before a wait (s+1's while this wave still waits for V).
__global__ void pjb_wait_peers_kernel(uint32_t* flag, uint32_t* error, uint32_t timeout_ms, PJPeerSync w) {
    PJSync y;
    y.flag = flag; y.error = error; y.timeout_ms = timeout_ms;
    const uint32_t t = threadIdx.x;
    if (t < w.n_raise) __hip_atomic_store(w.raise[t], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (y.flag) await_done(y);
    if (t < w.n_wait) {
        const long long t0 = wall_clock64(), limit = 100000ll * y.timeout_ms;
        while (__hip_atomic_load(w.wait[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
            __builtin_amdgcn_s_sleep(4);
            if (limit && wall_clock64() - t0 > limit) { __hip_atomic_store(y.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
        __hip_atomic_store(w.wait[t], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (w.delay_us) { const long long d0 = wall_clock64(); while (wall_clock64() - d0 < 100ll * w.delay_us) __builtin_amdgcn_s_sleep(8); }
    }
}
// (scalar arguments, not the PJSync struct: the unit is built with kernel-argument preload, which hands leading SCALARS to the wave in
// SGPRs -- these one-wave kernels sit on the substep's critical chains and have nothing to hide a scalar load behind)
__global__ void pjb_wait_kernel(uint32_t* flag, uint32_t* error, uint32_t timeout_ms) {
    PJSync y;
    y.flag = flag; y.error = error; y.timeout_ms = timeout_ms;
    await_done(y);
}
__global__ void pjb_signal_kernel(uint32_t* flag, uint32_t* clear) {
    if (threadIdx.x == 0) {
        if (clear) __hip_atomic_store(clear, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// TETSIM_FLAG_LEAN_STATE: the accumulated quaternion of every tet, recovered from its carried shape when somebody asks for it
// (tetsim_read_quats, the visual mesh, a checkpoint) instead of being read, multiplied up and written back by every substep.
// The reference accumulates q <- normalize(rot (x) q) (SoftbodyGPU.js:181) and rotates the carried shape by the same `rot`
// (:253-262), so in exact arithmetic carried_k = R(q) rest0_k for the centred corners: with S0 = [s0 s1 s2] and C = [c0 c1 c2]
// (the fourth corner is minus their sum on both sides) R = C S0^-1 -- formed in f64, turned into a quaternion by the
// largest-diagonal rule, normalised, and given the sign that keeps it next to the quaternion this array held before (q and -q are
// the same rotation; the reference's product never jumps).  Differs from the multiplied-up quaternion by the rounding the two
// have gathered on their separate ways (1e-6 after hundreds of substeps, tests/test_gpu_lean_state.py).  A degenerate tet
// (det S0 = 6V/4 = 0: the reference's zero-volume case) keeps what the array held.
__global__ __launch_bounds__(256) void pjb_recover_quat_kernel(const float4* __restrict__ r0a, const float4* __restrict__ r0b, const float4* __restrict__ r0c,
                                                              const float4* __restrict__ ca, const float4* __restrict__ cb, const float* __restrict__ cc1,
                                                              float4* __restrict__ quat, uint32_t nt) {
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= nt) return;
    const float4 a0 = r0a[e], b0 = r0b[e], c0 = r0c[e], a = ca[e], b = cb[e];
    const float c1 = cc1[e];
    const float4 qp = quat[e];
    // columns of S0 and C
    const double s[3][3] = {{a0.x, a0.y, a0.z}, {a0.w, b0.x, b0.y}, {b0.z, b0.w, c0.x}};
    const double c[3][3] = {{a.x, a.y, a.z}, {a.w, b.x, b.y}, {b.z, b.w, c1}};
    // inverse of S0 (columns s[0], s[1], s[2]) through the cross products of its columns: rows of the inverse are (s1 x s2, s2 x s0, s0 x s1) / det
    auto cr = [](const double* u, const double* v, double* o) { o[0] = u[1] * v[2] - u[2] * v[1]; o[1] = u[2] * v[0] - u[0] * v[2]; o[2] = u[0] * v[1] - u[1] * v[0]; };
    double inv[3][3];
    cr(s[1], s[2], inv[0]); cr(s[2], s[0], inv[1]); cr(s[0], s[1], inv[2]);
    const double det = s[0][0] * inv[0][0] + s[0][1] * inv[0][1] + s[0][2] * inv[0][2];
    if (!(fabs(det) > 1.0e-300)) return;
    const double rd = 1.0 / det;
    double R[3][3];   // R[i][j] = sum_k c[k][i] * inv[k][j] / det
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i][j] = (c[0][i] * inv[0][j] + c[1][i] * inv[1][j] + c[2][i] * inv[2][j]) * rd;
    const double tr = R[0][0] + R[1][1] + R[2][2];
    double x, y, z, w;
    if (tr > 0.0) { w = tr + 1.0; x = R[2][1] - R[1][2]; y = R[0][2] - R[2][0]; z = R[1][0] - R[0][1]; }
    else if (R[0][0] >= R[1][1] && R[0][0] >= R[2][2]) { x = 1.0 + R[0][0] - R[1][1] - R[2][2]; y = R[0][1] + R[1][0]; z = R[0][2] + R[2][0]; w = R[2][1] - R[1][2]; }
    else if (R[1][1] >= R[2][2]) { y = 1.0 + R[1][1] - R[0][0] - R[2][2]; x = R[0][1] + R[1][0]; z = R[1][2] + R[2][1]; w = R[0][2] - R[2][0]; }
    else { z = 1.0 + R[2][2] - R[0][0] - R[1][1]; x = R[0][2] + R[2][0]; y = R[1][2] + R[2][1]; w = R[1][0] - R[0][1]; }
    const double n2 = x * x + y * y + z * z + w * w;
    if (!(n2 > 0.0) || !(n2 < 1.0e300)) return;   // (NaN / inf shape: a body that has blown up keeps its last quaternion)
    double k = 1.0 / sqrt(n2);
    if (x * qp.x + y * qp.y + z * qp.z + w * qp.w < 0.0) k = -k;
    quat[e] = make_float4(static_cast<float>(x * k), static_cast<float>(y * k), static_cast<float>(z * k), static_cast<float>(w * k));
}

__global__ __launch_bounds__(256) void pjb_repredict_kernel(PJBlk d) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= d.nv_owned) return;
    const f3 pred = xyz(d.pos_final[v]) + xyz(d.vel[v]) * d.params->dt;
    d.pos_pred[v] = make_float4(pred.x, pred.y, pred.z, 0.0f);
}

}  // namespace

void pjb_launch_tet(hipStream_t s, const PJBlk& d, uint32_t tile_first, uint32_t tile_count, hipEvent_t e0, hipEvent_t e1, uint32_t* raise_word,
                    uint32_t* clear_word) {
    if (tile_count == 0) return;   // (callers with a word to raise or clear check this themselves)
    if (raise_word || clear_word) return launch_tet_x(s, d, tile_first, tile_count, TetRaise{raise_word, clear_word}, e0, e1);
    const uint32_t per_xcd = (tile_count + 7u) / 8u;
    const int mode = blk_mode(d);
    auto* kernel = mode == kModeConstantRest ? pjb_tet_kernel_constant_rest : mode == kModeLeanState ? pjb_tet_kernel_lean : pjb_tet_kernel;
    if (e0) hipExtLaunchKernelGGL(kernel, dim3(per_xcd * 8u), dim3(kTile), 0, s, e0, e1, 0, d, tile_first, tile_count, per_xcd TETSIM_DBG_LAUNCH);
    else hipLaunchKernelGGL(kernel, dim3(per_xcd * 8u), dim3(kTile), 0, s, d, tile_first, tile_count, per_xcd TETSIM_DBG_LAUNCH);
}
void pjb_launch_call(hipStream_t s, const PJBlk& d, uint32_t n, uint32_t* err, uint32_t timeout_ms, const DevParams& params, DevParams* params_dev) {
    if (d.nb == 0 || n == 0) return;
    const uint32_t per_xcd = (d.nb + 7u) / 8u, tet_blocks = per_xcd * 8u, per_sub = (tet_blocks + (d.nv_owned + kTile - 1u) / kTile + 7u) & ~7u;
    const int mode = blk_mode(d);
    auto* kernel = mode == kModeConstantRest ? pjb_call_kernel<kModeConstantRest> : mode == kModeLeanState ? pjb_call_kernel<kModeLeanState> : pjb_call_kernel<kModeCarried>;
    hipLaunchKernelGGL(kernel, dim3(per_sub * n), dim3(kTile), 0, s, d, n, d.nb, per_xcd, tet_blocks, per_sub, err, timeout_ms, params, params_dev TETSIM_DBG_LAUNCH);
}
void pjb_launch_tet_fused(hipStream_t s, const PJBlk& d, hipEvent_t e0, hipEvent_t e1) { launch_tet_x(s, d, 0u, d.nb, TetFused{0u}, e0, e1); }
void pjb_launch_frame(hipStream_t s, const PJBlk& d, uint32_t n, const int32_t* block_tile, uint32_t blocks, bool local, float4* pbuf0, float4* pbuf1,
                      uint32_t* err, uint32_t timeout_ms, const DevParams& params, DevParams* params_dev, hipEvent_t e0, hipEvent_t e1) {
    if (d.nb == 0 || n == 0 || blocks == 0) return;
    const int mode = blk_mode(d);
    auto* kernel = local ? (mode == kModeConstantRest ? pjb_frame_kernel_constant_rest_local : mode == kModeLeanState ? pjb_frame_kernel_lean_local : pjb_frame_kernel_local)
                         : (mode == kModeConstantRest ? pjb_frame_kernel_constant_rest : mode == kModeLeanState ? pjb_frame_kernel_lean : pjb_frame_kernel);
    if (e0) hipExtLaunchKernelGGL(kernel, dim3(blocks), dim3(kTile), 0, s, e0, e1, 0, d, n, block_tile, pbuf0, pbuf1, err, timeout_ms, params, params_dev);
    else hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kTile), 0, s, d, n, block_tile, pbuf0, pbuf1, err, timeout_ms, params, params_dev);
}
uint32_t pjb_frame_capacity(int mode, uint32_t* compute_units) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    // the smaller answer of the two placements' kernels: which one a body launches is decided after this query (a body too large for the
    // one-XCD placement takes the other), and their register counts need not stay equal
    int per_cu_any = 0;
    auto* k_local = mode == kModeConstantRest ? pjb_frame_kernel_constant_rest_local : mode == kModeLeanState ? pjb_frame_kernel_lean_local : pjb_frame_kernel_local;
    auto* k_any = mode == kModeConstantRest ? pjb_frame_kernel_constant_rest : mode == kModeLeanState ? pjb_frame_kernel_lean : pjb_frame_kernel;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_local, static_cast<int>(kTile), 0);
    const hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_any, k_any, static_cast<int>(kTile), 0);
    if (e != hipSuccess || e2 != hipSuccess || per_cu <= 0 || per_cu_any <= 0) return 0;
    per_cu = std::min(per_cu, per_cu_any);
    if (compute_units) *compute_units = static_cast<uint32_t>(prop.multiProcessorCount);
    return static_cast<uint32_t>(per_cu);
}
// How many waiting workgroups the in-kernel hand-overs may put on the current device (tetsim_halo.hip: folded waits).  A wave that looks
// at a word keeps its slot while it waits, and the kernel that raises the word needs slots too: the waiting kernels may hold HALF of
// what the device keeps resident of pjb_vertex_kernel_await (one-wave workgroups) and a QUARTER of what it keeps of
// the TetHwait tet kernel -- measured limits, not constants: a compute partition with fewer CUs (or a kernel that grew) shrinks them.
void pjb_wait_capacity(int mode, uint32_t* vertex_waves, uint32_t* hwait_blocks) {
    *vertex_waves = 0; *hwait_blocks = 0;
    int per_cu_v = 0, per_cu_t = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_v, pjb_vertex_kernel_await, 64, 0) != hipSuccess) per_cu_v = 0;
    auto* k_hwait = mode == kModeConstantRest ? pjb_tet_kernel_x<kModeConstantRest, TetHwait> : mode == kModeLeanState ? pjb_tet_kernel_x<kModeLeanState, TetHwait> : pjb_tet_kernel_x<kModeCarried, TetHwait>;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_t, k_hwait, static_cast<int>(kTile), 0);
    if (e != hipSuccess) per_cu_t = 0;
    const uint32_t cus = static_cast<uint32_t>(std::max(prop.multiProcessorCount, 0));
    *vertex_waves = static_cast<uint32_t>(std::max(per_cu_v, 0)) * cus / 2u;
    *hwait_blocks = static_cast<uint32_t>(std::max(per_cu_t, 0)) * cus / 4u;
}
// 8 if block i of a grid runs on XCD i % 8 (whatever the XCDs' numbering) on this device, else 0: a grid of `blocks` one-wave
// workgroups reports its XCC_ID register
uint32_t pjb_probe_xcd(hipStream_t s, uint32_t blocks) {
    uint32_t* d_out = nullptr;
    if (blocks < 16u || hipMalloc(reinterpret_cast<void**>(&d_out), blocks * sizeof(uint32_t)) != hipSuccess) return 0;
    std::vector<uint32_t> out(blocks, 0xffffffffu);
    hipLaunchKernelGGL(pjb_probe_xcd_kernel, dim3(blocks), dim3(64), 0, s, d_out);
    const bool ok = hipStreamSynchronize(s) == hipSuccess && hipMemcpy(out.data(), d_out, blocks * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d_out);
    if (!ok) return 0;
    for (uint32_t i = 0; i < 8u; i++)
        for (uint32_t j = 0; j < i; j++) if (out[i] == out[j]) return 0;          // the first eight blocks: eight different XCDs
    for (uint32_t i = 8u; i < blocks; i++) if (out[i] != out[i & 7u]) return 0;    // ... and the pattern repeats
    return 8;
}
void pjb_launch_wait_peers(hipStream_t s, const PJSync& y, const PJPeerSync& w) { hipLaunchKernelGGL(pjb_wait_peers_kernel, dim3(1), dim3(64), 0, s, y.flag, y.error, y.timeout_ms, w); }
void pjb_launch_vertex_peer(hipStream_t s, const PJBlk& d, uint32_t first, uint32_t count, const PJPeer& peer, uint32_t* raise_word, const PJClear& clr) {
    if (count == 0) return;
    hipLaunchKernelGGL(pjb_vertex_kernel_peer, dim3((count + 63u) / 64u), dim3(64), 0, s, d, first, count, peer, raise_word, clr);
}
void pjb_launch_tet_hwait(hipStream_t s, const PJBlk& d, uint32_t tile_first, uint32_t tile_count, const PJSync& yv, const PJPeerSync& w, const float4* ghosts) {
    launch_tet_x(s, d, tile_first, tile_count, TetHwait{w, yv.flag, yv.error, yv.timeout_ms, ghosts});
}
void pjb_launch_tet_alt(hipStream_t s, const PJBlk& d, uint32_t tile_first, uint32_t tile_count) { launch_tet_x(s, d, tile_first, tile_count, TetAlt{0u}); }
void pjb_launch_wait(hipStream_t s, const PJSync& y) { hipLaunchKernelGGL(pjb_wait_kernel, dim3(1), dim3(64), 0, s, y.flag, y.error, y.timeout_ms); }
void pjb_launch_signal(hipStream_t s, const PJSync& y, uint32_t* clear_word) { hipLaunchKernelGGL(pjb_signal_kernel, dim3(1), dim3(64), 0, s, y.flag, clear_word); }
void pjb_launch_vertex_await(hipStream_t s, const PJBlk& d, uint32_t first, uint32_t count, const PJSync& y, hipEvent_t e0, hipEvent_t e1) {
    if (count == 0) return;
    if (e0) hipExtLaunchKernelGGL(pjb_vertex_kernel_await, dim3((count + 63u) / 64u), dim3(64), 0, s, e0, e1, 0, d, first, count, y.flag, y.error, y.timeout_ms);
    else hipLaunchKernelGGL(pjb_vertex_kernel_await, dim3((count + 63u) / 64u), dim3(64), 0, s, d, first, count, y.flag, y.error, y.timeout_ms);
}
void pjb_launch_vertex(hipStream_t s, const PJBlk& d, uint32_t first, uint32_t count, hipEvent_t e0, hipEvent_t e1, uint32_t* raise_word, uint32_t* clear_word) {
    if (count == 0) return;
    if (raise_word || clear_word) {
        if (e0) hipExtLaunchKernelGGL(pjb_vertex_kernel_raise, dim3((count + 63u) / 64u), dim3(64), 0, s, e0, e1, 0, d, first, count, raise_word, clear_word);
        else hipLaunchKernelGGL(pjb_vertex_kernel_raise, dim3((count + 63u) / 64u), dim3(64), 0, s, d, first, count, raise_word, clear_word);
        return;
    }
    if (e0) hipExtLaunchKernelGGL(pjb_vertex_kernel, dim3((count + 63u) / 64u), dim3(64), 0, s, e0, e1, 0, d, first, count);
    else hipLaunchKernelGGL(pjb_vertex_kernel, dim3((count + 63u) / 64u), dim3(64), 0, s, d, first, count);
}
void pjb_launch_recover_quats(hipStream_t s, const PJBlk& d, const float4* rest0_a, const float4* rest0_b, const float4* rest0_c) {
    if (d.nt == 0 || !d.lean_state) return;
    hipLaunchKernelGGL(pjb_recover_quat_kernel, dim3((d.nt + 255u) / 256u), dim3(256), 0, s, rest0_a, rest0_b, rest0_c, d.rest_a, d.rest_b, d.rest_c1, d.quat, d.nt);
}
void pjb_launch_repredict(hipStream_t s, const PJBlk& d) {
    if (d.nv_owned == 0) return;
    hipLaunchKernelGGL(pjb_repredict_kernel, dim3((d.nv_owned + 255u) / 256u), dim3(256), 0, s, d);
}

}  // namespace tetsim

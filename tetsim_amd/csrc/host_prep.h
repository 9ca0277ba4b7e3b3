// host_prep.h -- host-side preprocessing shared by the C ABI (no HIP dependency).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace tetsim {

constexpr int kRefSlots = 36;  // 9 RGBA tables, SoftbodyGPU.js:29-37

// Softbody.js:60-87 in JS number semantics (f64 arithmetic, f32 stores).
void prep_rest(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, double density,
               float* inv_mass, float* inv_rest_pose, float* inv_rest_volume);

// Rest volume as SoftbodyGPU.js:579-589 computes it: fround(1 / (det(f32 edge matrix) / 6)).
float pj_inv_rest_volume(const float* verts, const int32_t* tet);

uint32_t prep_levels(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* level);
uint32_t prep_colours(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* colour);
// NEOHOOKEAN_GS, TETSIM_ORDER_CLUSTERED.  Tets are grouped into clusters of <= kClusterTets tets over <= kClusterVerts
// distinct vertices (on a cell-major lattice: the 6 tets of one cell); clusters are coloured so that clusters of one colour
// share no vertex.  One kernel launch per cluster colour: a lane sweeps ITS cluster tet by tet with the cluster's vertices
// held in LDS, so the memory round trips and the launch are paid once per cluster colour (8 on the lattice) instead of once
// per tet colour (31), for 6 x 8 = 48 sequential tet solves instead of 31.
// `pre` is the sequential order whose result the schedule reproduces (colour, cluster, position in cluster): any two tets
// sharing a vertex are either in one cluster (kept in order by the lane) or in clusters of different colours (kept in order
// by the launches).
constexpr uint32_t kClusterTets = 8, kClusterVerts = 8;
struct ClusterPlan {
    std::vector<int32_t> pre;            // [nt] sequential position -> caller's tet id
    std::vector<uint32_t> exec_pos;      // [nt] storage slot -> sequential position
    std::vector<uint32_t> corner_slots;  // [nt] per storage slot: the 4 corners' cluster-local vertex slots, one byte each
    std::vector<uint32_t> launch_off;    // [launches+1] storage range of each launch (= cluster colour)
    std::vector<uint32_t> step_off;      // [launches+1] range in step_first/step_count
    std::vector<uint32_t> step_first, step_count;  // step j of a launch: lanes [0, count) solve storage slots first + lane
    std::vector<uint32_t> vid_off;       // [launches+1] range in slot_vid; a launch's block is [kClusterVerts][clusters] column-major
    std::vector<int32_t> slot_vid;       // vertex id of (slot, cluster), -1 = unused
    uint32_t num_clusters = 0;
};
ClusterPlan prep_clusters(const int32_t* tets, uint32_t nt, uint32_t nv);
// returns dropped contributions
uint32_t prep_slot_table(const int32_t* tets, uint32_t nt, uint32_t nv, bool ref_quirk, int32_t* slots);

// Per-vertex incident (tet,corner) lists in tet order, CSR.  ref_quirk: drop (tet 0, corner 0) as the
// reference's `<= 0.0` slot test does (only meaningful when local tet 0 IS global tet 0); ref_cap: keep at
// most 36 incidences per vertex.  With both, the lists equal the reference's 36-slot rows.
struct Incidence {
    std::vector<uint32_t> offset;  // [nv+1]
    std::vector<int32_t> slot;     // 4*tet + corner
    uint32_t max_valence = 0;
    uint32_t dropped = 0;
};
Incidence build_incidence(const int32_t* tets, uint32_t nt, uint32_t nv, bool ref_quirk, bool ref_cap);

// Domain decomposition plan for one partition.
struct Partition {
    uint32_t nv_global = 0, nt_global = 0;
    int part_count = 1, part_index = 0;
    std::vector<int32_t> local_to_global_vert;  // [owned boundary | owned interior | ghosts by (owner,id)]
    uint32_t n_owned = 0, n_boundary = 0;
    std::vector<int32_t> local_to_global_tet;   // ascending global id
    std::vector<int32_t> local_tets;            // [4*nt_local] local vertex ids
    uint32_t owned_tets = 0;                    // tets whose lowest-owner rule assigns them here
    struct Neighbour {
        int rank;
        std::vector<int32_t> send_local;  // owned local ids, ascending global id
        std::vector<int32_t> send_global;
        uint32_t recv_start = 0, recv_count = 0;  // contiguous ghost range in local numbering
        std::vector<int32_t> recv_global;
        bool send_contiguous = false;
        // depth 2: the SECOND ghost layer of this neighbour's particles (received), and the owned particles that are the
        // neighbour's second layer (sent) -- disjoint from the first-layer lists above
        std::vector<int32_t> send2_local, send2_global;
        uint32_t recv2_start = 0, recv2_count = 0;
        std::vector<int32_t> recv2_global;
    };
    std::vector<Neighbour> neigh;
    // depth 2 (a two-layer ghost region: the partition can advance its first ghost layer itself, so that ghosts need to cross
    // only every other substep -- DESIGN.md 7): ghosts [n_owned, n_owned + n_ghost1) are the first layer (share a tet with an
    // owned particle), the rest the second (share a tet with a first-layer ghost); local tets with tet_layer 1 touch no owned
    // particle (first-layer ghost tets' outer neighbours).  depth 1: n_ghost1 = all ghosts, tet_layer all 0.
    int depth = 1;
    uint32_t n_ghost1 = 0;
    std::vector<uint8_t> tet_layer;
};
// The built-in vertex partitioner (partitioner.cpp): recursive bisection by breadth-first pseudo-diameter keys (and by x / y / z when
// `verts` is given), then k-way boundary refinement; weights 1 + valence, parts within +-3% of the mean.  out_owner [nv].
std::string prep_partition(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, int parts, int32_t* out_owner);
// What a vertex -> part map costs, per part (the counts build_partition would produce for it, without building the plans).
struct PartQuality {
    uint32_t owned_particles = 0, ghost_particles = 0, boundary_particles = 0;   // ghosts: read, not owned; boundary: owned, read by others
    uint32_t local_elems = 0, owned_elems = 0;                                    // tets solved here; tets counted once (lowest-owner rule)
    uint32_t num_neighbours = 0;
};
std::string partition_quality(const int32_t* tets, uint32_t nt, uint32_t nv, int parts, const int32_t* owner, PartQuality* out);
// vert_owner may be null: prep_partition without coordinates (the plan entry points have none). Returns "" or an error message.
std::string build_partition(const int32_t* tets, uint32_t nt, uint32_t nv, int part_count, int part_index,
                            const int32_t* vert_owner, Partition* out, int depth = 1);

// Permutation that sorts points [first, first+count) of `xyz` along a Morton (Z-order) curve; entries outside that
// range map to themselves.  order[new] = old.  Used to renumber particles internally so that particles that are
// close in space are close in memory (tile staging and partial-sum gathers then touch near-contiguous runs).
std::vector<uint32_t> morton_vertex_order(const float* xyz, uint32_t n, uint32_t first, uint32_t count);

// Workgroup tiling for the blocked POLAR_JACOBI formulation.
// Tets are sorted along a Morton curve of their rest centroids and cut into tiles of <= 256 tets that touch
// <= 256 distinct vertices, so a tile's vertex set fits an LDS tile addressed by 8-bit local indices.
#ifndef TETSIM_TILE
#define TETSIM_TILE 256
#endif
constexpr uint32_t kBlockTile = TETSIM_TILE;   // tets per workgroup tile of the blocked polar kernel (pj_blocked.hip kTile)
constexpr uint32_t kQuadTile = 64;             // ... of the four-lanes-per-tet kernels of small bodies (pj_quad.hip): 64 quads = 256 threads
constexpr int32_t kQuadPollDelay = 0;           // s_sleep units between a tile's store and its first poll of the neighbours' sums (TETSIM_QUAD_POLL_DELAY overrides: A/B)
constexpr uint32_t kQuadMaxPartials = 12;      // ... which take lists of at most this many partial sums per particle (the Dragon: 10)
struct BlockPlan {
    uint32_t num_blocks = 0;
    uint32_t tile = kBlockTile;          // tets (and particle slots) per tile at most: 256 (pj_blocked.hip) or kQuadTile (pj_quad.hip, small bodies)
    std::vector<int32_t> tet_perm;       // new tet position -> input tet index
    std::vector<uint32_t> blk_tet_off;   // [num_blocks+1]
    std::vector<uint32_t> blk_vert_off;  // [num_blocks+1] into blk_verts / partial sums
    std::vector<int32_t> blk_verts;      // vertex ids of each tile's LDS slots
    std::vector<uint8_t> tet_lidx;       // [4*nt] LDS slot of every corner (new tet order)
    std::vector<uint32_t> lc_range;      // per tile vertex: first | owner << 15 | (last+1) << 16 into the tile's lc_ent (first, last+1 <= 1024);
                                         //   owner = this slot is the FIRST of its particle's partial sums (the fused kernel's one writer)
    std::vector<uint16_t> lc_ent;        // [4*nt] per tile: corner*256 + tetLocal (word offset into the goal planes) grouped by LDS slot, tet order inside
    std::vector<uint32_t> vp_off;        // [nv_sum+1] per summed vertex: range into vp_idx
    std::vector<uint32_t> vp_idx;        // indices into the partial-sum array, ascending tile
    std::vector<uint32_t> vp_ell;        // the same lists as ELL [max_partials][nv_pad], 0xffffffff = none
    // fused particle pass (pj_blocked.hip): per tile slot, the partial sums of ITS particle (= the particle's vp list) as ELL
    // [max_partials][ns_pad], 0xffffffff = none; per tile, the longest such list among its slots
    std::vector<uint32_t> slot_src;
    std::vector<uint32_t> blk_maxsrc;    // [num_blocks]
    uint32_t ns_pad = 0;
    bool every_owned_particle_has_a_partial = true;   // else some particle is summed by no tile: the fused pass cannot serve it
    uint32_t nv_pad = 0;
    uint32_t max_tile_verts = 0, max_partials = 0;
    uint32_t num_interior_blocks = 0;    // tiles [0, num_interior_blocks) touch no ghost (id >= nv_sum) and no boundary particle
    uint32_t num_first_blocks = 0;       // tiles [0, num_first_blocks) hold tets of classes 0 and 1; the rest (two-layer ghost regions) the second-layer ghost tets
};
// `inc` (build_incidence) decides WHICH (tet,corner) contributions count (reference quirk / cap); contributions
// it drops are left out of lc_ent.  Only vertices < nv_sum get vp lists (the owned ones).
// Batches (tetsim_create_batch: several independent bodies behind one handle): body_first_tet / body_first_vert [bodies + 1]
// give each body's tet and vertex range; tiles never span two bodies and every body is tiled exactly as it would be alone
// (its own bounding box for the Morton codes), so that each body's results equal its solo run bit for bit.  NULL = one body.
// nv_boundary: particles [0, nv_boundary) are the partition's BOUNDARY particles (the ones it sends to neighbours).  A tile is
// "halo-side" -- ordered last, counted out of num_interior_blocks -- if it touches a ghost (id >= nv_sum) OR a boundary particle:
// then every tile that contributes to a boundary particle is halo-side, and the halo queue can finish the boundary particles
// and start the transfer without waiting for the interior tiles.
// Partitions: tet_class[e] (0..2) keeps tets of different classes in different tiles, class after class.  1 = the tets that touch a
// boundary or a ghost particle: with tiles of their own the halo-side tiles are exactly those tets -- two or three cell layers
// along an interface instead of every cube-shaped tile that happens to reach it (14.5% -> 7% of a 1 M-tet slab's tets on the halo
// queue's critical chain).  2 = the second-layer ghost tets of a two-layer ghost region, ordered last (solved on every other substep
// only); nv_owned (default nv_sum) is then smaller than nv_sum -- the first ghost layer is summed here too -- and "halo-side" means
// touching a particle >= nv_owned or < nv_boundary.
void build_blocks(const float* verts, const int32_t* tets, uint32_t nt, uint32_t nv, uint32_t nv_sum,
                  const Incidence& inc, BlockPlan* out, const uint32_t* body_first_tet = nullptr,
                  const uint32_t* body_first_vert = nullptr, uint32_t bodies = 1, uint32_t nv_boundary = 0,
                  const uint8_t* tet_class = nullptr, uint32_t nv_owned = 0xffffffffu, uint32_t tile = kBlockTile);

std::string validate_mesh(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, bool forbid_repeats);

}  // namespace tetsim

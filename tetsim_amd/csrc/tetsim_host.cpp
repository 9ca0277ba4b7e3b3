// tetsim_host.cpp -- entry points that never touch a device: preprocessing, partition plans, the .tetsim container.
#include "body.h"

using namespace tetsim;

extern "C" {

// ---- .tetsim mesh container ---------------------------------------------------------------------------------------
struct tetsim_mesh_file { tetsim::MeshFile* m; };

int tetsim_mesh_write(const char* path, const TetSimMeshArrays* a) {
    if (!a) return fail(nullptr, TETSIM_EINVAL, "arrays is null");
    const std::string e = mesh_write(path, *a);
    return e.empty() ? TETSIM_OK : fail(nullptr, TETSIM_EINVAL, e);
}
int tetsim_mesh_open(const char* path, tetsim_mesh* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "out is null");
    *out = nullptr;
    tetsim::MeshFile* m = nullptr;
    const std::string e = mesh_open(path, &m);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    *out = new tetsim_mesh_file{m};
    return TETSIM_OK;
}
int tetsim_mesh_arrays(tetsim_mesh m, TetSimMeshArrays* out) {
    if (!m || !out) return fail(nullptr, TETSIM_EINVAL, "null argument");
    *out = mesh_arrays(m->m);
    return TETSIM_OK;
}
int tetsim_mesh_close(tetsim_mesh m) {
    if (!m) return TETSIM_OK;
    mesh_close(m->m);
    delete m;
    return TETSIM_OK;
}

// ---- host-only preprocessing ---------------------------------------------------------------------------------
int tetsim_prep_levels(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* level, uint32_t* num_levels) {
    if ((nt && (!tets || !level)) || !num_levels) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    *num_levels = prep_levels(tets, nt, nv, level);
    return 0;
}
int tetsim_prep_colours(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* colour, uint32_t* num_colours) {
    if ((nt && (!tets || !colour)) || !num_colours) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    *num_colours = prep_colours(tets, nt, nv, colour);
    return 0;
}
int tetsim_prep_clusters(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* order, int32_t* launch, int32_t* lane, int32_t* step,
                         uint32_t* num_launches, uint32_t* num_clusters) {
    if ((nt && (!tets || !order || !launch || !lane || !step)) || !num_launches) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    const ClusterPlan P = prep_clusters(tets, nt, nv);
    for (uint32_t i = 0; i < nt; i++) order[i] = P.pre[i];
    const uint32_t nl = static_cast<uint32_t>(P.launch_off.size() - 1);
    for (uint32_t l = 0; l < nl; l++)
        for (uint32_t j = P.step_off[l]; j < P.step_off[l + 1]; j++)
            for (uint32_t i = 0; i < P.step_count[j]; i++) {
                const uint32_t pos = P.exec_pos[P.step_first[j] + i];
                launch[pos] = static_cast<int32_t>(l);
                lane[pos] = static_cast<int32_t>(i);
                step[pos] = static_cast<int32_t>(j - P.step_off[l]);
            }
    *num_launches = nl;
    if (num_clusters) *num_clusters = P.num_clusters;
    return 0;
}
int tetsim_prep_tiles(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, const uint32_t* body_first_tet,
                      const uint32_t* body_first_vert, uint32_t bodies, int32_t* tile_tets, uint32_t* tile_off, uint8_t* corner_slot,
                      uint32_t* num_tiles) {
    if (!num_tiles) return TETSIM_EINVAL;
    std::string e = validate_mesh(verts, nv, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    if (body_first_tet && body_first_vert && bodies) {
        if (body_first_tet[0] != 0 || body_first_vert[0] != 0 || body_first_tet[bodies] != nt || body_first_vert[bodies] != nv)
            return fail(nullptr, TETSIM_EINVAL, "body ranges must cover the whole mesh");
        for (uint32_t b = 0; b < bodies; b++) {
            if (body_first_tet[b] > body_first_tet[b + 1] || body_first_vert[b] >= body_first_vert[b + 1]) return fail(nullptr, TETSIM_EINVAL, "body ranges must ascend");
            for (uint64_t i = 4ull * body_first_tet[b]; i < 4ull * body_first_tet[b + 1]; i++)
                if (static_cast<uint32_t>(tets[i]) < body_first_vert[b] || static_cast<uint32_t>(tets[i]) >= body_first_vert[b + 1])
                    return fail(nullptr, TETSIM_EINVAL, "a tet references a particle of another body");
        }
    }
    const Incidence inc = build_incidence(tets, nt, nv, false, false);
    BlockPlan B;
    build_blocks(verts, tets, nt, nv, nv, inc, &B, body_first_tet, body_first_vert, bodies);
    *num_tiles = B.num_blocks;
    if (tile_tets) std::copy(B.tet_perm.begin(), B.tet_perm.end(), tile_tets);
    if (tile_off) std::copy(B.blk_tet_off.begin(), B.blk_tet_off.end(), tile_off);
    if (corner_slot) std::copy(B.tet_lidx.begin(), B.tet_lidx.end(), corner_slot);
    return 0;
}
int tetsim_prep_slot_table(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t ref_quirk, int32_t* slots, uint32_t* dropped) {
    if ((nt && !tets) || !slots) return TETSIM_EINVAL;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv ? nv : 1, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    const uint32_t d = prep_slot_table(tets, nt, nv, ref_quirk != 0, slots);
    if (dropped) *dropped = d;
    return 0;
}
int tetsim_prep_ref_grab_texels(int32_t grab_id, uint32_t num_elems, uint32_t num_particles, int32_t out[2]) {
    if (!out) return TETSIM_EINVAL;
    ref_grab_texels(grab_id, num_elems, num_particles, out);
    return 0;
}
int tetsim_prep_rest(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, double density, float* inv_mass, float* inv_rest_pose, float* inv_rest_volume) {
    if (!inv_mass || (nt && (!inv_rest_pose || !inv_rest_volume))) return TETSIM_EINVAL;
    std::string e = validate_mesh(verts, nv, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    prep_rest(verts, nv, tets, nt, density, inv_mass, inv_rest_pose, inv_rest_volume);
    return 0;
}

// ---- the built-in partitioner (partitioner.cpp) -------------------------------------------------------------------
int tetsim_prep_partition(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, int32_t part_count, int32_t* vert_owner_out) {
    if ((nt && !tets) || (nv && !vert_owner_out) || part_count < 1) return fail(nullptr, TETSIM_EINVAL, "tetsim_prep_partition: null argument or part_count < 1");
    if (verts)
        for (uint64_t i = 0; i < 3ull * nv; i++)
            if (!std::isfinite(verts[i])) return fail(nullptr, TETSIM_EINVAL, "non-finite vertex coordinate");
    const std::string e = prep_partition(verts, nv, tets, nt, part_count, vert_owner_out);
    return e.empty() ? TETSIM_OK : fail(nullptr, TETSIM_EINVAL, e);
}
int tetsim_prep_partition_quality(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t part_count, const int32_t* vert_owner, TetSimPartQuality* out) {
    if ((nt && !tets) || !out || part_count < 1) return fail(nullptr, TETSIM_EINVAL, "tetsim_prep_partition_quality: null argument or part_count < 1");
    std::vector<int32_t> own;
    if (!vert_owner) {   // what tetsim_create / tetsim_plan_create pick by themselves
        own.resize(nv);
        const std::string e = prep_partition(nullptr, nv, tets, nt, part_count, own.data());
        if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
        vert_owner = own.data();
    }
    std::vector<PartQuality> q(part_count);
    const std::string e = partition_quality(tets, nt, nv, part_count, vert_owner, q.data());
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    for (int32_t r = 0; r < part_count; r++) {
        out[r].owned_particles = q[r].owned_particles; out[r].ghost_particles = q[r].ghost_particles; out[r].boundary_particles = q[r].boundary_particles;
        out[r].local_elems = q[r].local_elems; out[r].owned_elems = q[r].owned_elems; out[r].num_neighbours = q[r].num_neighbours;
    }
    return TETSIM_OK;
}

// ---- partition plan (host only) ---------------------------------------------------------------------------------
struct tetsim_plan_s { Partition P; };

int tetsim_plan_create(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t part_count, int32_t part_index,
                       const int32_t* vert_owner, tetsim_plan* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "null plan pointer");
    *out = nullptr;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    tetsim_plan_s* p = new tetsim_plan_s();
    e = build_partition(tets, nt, nv, part_count, part_index, vert_owner, &p->P);
    if (!e.empty()) { delete p; return fail(nullptr, TETSIM_EINVAL, e); }
    *out = p;
    return 0;
}
int tetsim_plan_create_deep(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t part_count, int32_t part_index,
                            const int32_t* vert_owner, int32_t depth, tetsim_plan* out) {
    if (!out) return fail(nullptr, TETSIM_EINVAL, "null plan pointer");
    *out = nullptr;
    std::string e = validate_mesh(reinterpret_cast<const float*>(tets), nv, tets, nt, false);
    if (!e.empty()) return fail(nullptr, TETSIM_EINVAL, e);
    tetsim_plan_s* p = new tetsim_plan_s();
    e = build_partition(tets, nt, nv, part_count, part_index, vert_owner, &p->P, depth);
    if (!e.empty()) { delete p; return fail(nullptr, TETSIM_EINVAL, e); }
    *out = p;
    return 0;
}
int tetsim_plan_layers(tetsim_plan p, uint32_t* first_layer_ghosts, uint8_t* tet_layer) {
    if (!p) return TETSIM_EINVAL;
    if (first_layer_ghosts) *first_layer_ghosts = p->P.n_ghost1;
    if (tet_layer) std::copy(p->P.tet_layer.begin(), p->P.tet_layer.end(), tet_layer);
    return 0;
}
int tetsim_plan_neighbour_layer2(tetsim_plan p, uint32_t i, uint32_t* send_count, uint32_t* recv_start, uint32_t* recv_count) {
    if (!p || i >= p->P.neigh.size()) return TETSIM_EINVAL;
    const auto& nb = p->P.neigh[i];
    if (send_count) *send_count = static_cast<uint32_t>(nb.send2_local.size());
    if (recv_start) *recv_start = nb.recv2_start;
    if (recv_count) *recv_count = nb.recv2_count;
    return 0;
}
int tetsim_plan_neighbour_layer2_ids(tetsim_plan p, uint32_t i, int32_t* send_local, int32_t* send_global, int32_t* recv_global) {
    if (!p || i >= p->P.neigh.size()) return TETSIM_EINVAL;
    const auto& nb = p->P.neigh[i];
    if (send_local) std::copy(nb.send2_local.begin(), nb.send2_local.end(), send_local);
    if (send_global) std::copy(nb.send2_global.begin(), nb.send2_global.end(), send_global);
    if (recv_global) std::copy(nb.recv2_global.begin(), nb.recv2_global.end(), recv_global);
    return 0;
}
void tetsim_plan_destroy(tetsim_plan p) { delete p; }
int tetsim_plan_sizes(tetsim_plan p, TetSimPlanSizes* out) {
    if (!p || !out) return TETSIM_EINVAL;
    out->owned_particles = p->P.n_owned;
    out->boundary_particles = p->P.n_boundary;
    out->local_particles = static_cast<uint32_t>(p->P.local_to_global_vert.size());
    out->local_elems = static_cast<uint32_t>(p->P.local_to_global_tet.size());
    out->owned_elems = p->P.owned_tets;
    out->num_neighbours = static_cast<uint32_t>(p->P.neigh.size());
    return 0;
}
int tetsim_plan_arrays(tetsim_plan p, int32_t* l2gv, int32_t* l2gt, int32_t* ltets) {
    if (!p) return TETSIM_EINVAL;
    if (l2gv) std::copy(p->P.local_to_global_vert.begin(), p->P.local_to_global_vert.end(), l2gv);
    if (l2gt) std::copy(p->P.local_to_global_tet.begin(), p->P.local_to_global_tet.end(), l2gt);
    if (ltets) std::copy(p->P.local_tets.begin(), p->P.local_tets.end(), ltets);
    return 0;
}
int tetsim_plan_neighbour(tetsim_plan p, uint32_t i, int32_t* rank, uint32_t* send_count, uint32_t* recv_start,
                          uint32_t* recv_count, int32_t* contiguous) {
    if (!p || i >= p->P.neigh.size()) return TETSIM_EINVAL;
    const auto& nb = p->P.neigh[i];
    if (rank) *rank = nb.rank;
    if (send_count) *send_count = static_cast<uint32_t>(nb.send_local.size());
    if (recv_start) *recv_start = nb.recv_start;
    if (recv_count) *recv_count = nb.recv_count;
    if (contiguous) *contiguous = nb.send_contiguous ? 1 : 0;
    return 0;
}
int tetsim_plan_neighbour_ids(tetsim_plan p, uint32_t i, int32_t* send_local, int32_t* send_global, int32_t* recv_global) {
    if (!p || i >= p->P.neigh.size()) return TETSIM_EINVAL;
    const auto& nb = p->P.neigh[i];
    if (send_local) std::copy(nb.send_local.begin(), nb.send_local.end(), send_local);
    if (send_global) std::copy(nb.send_global.begin(), nb.send_global.end(), send_global);
    if (recv_global) std::copy(nb.recv_global.begin(), nb.recv_global.end(), recv_global);
    return 0;
}

}  // extern "C"

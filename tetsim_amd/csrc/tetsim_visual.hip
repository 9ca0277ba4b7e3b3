// tetsim_visual.hip -- C ABI, part 3 (include/tetsim.h): the embedded visual mesh (skinning, three.js computeVertexNormals) and
// the grab interface (pin a particle; nearest-particle query on the device).  See body.h.
#include "body.h"

using namespace tetsim;

extern "C" {

int tetsim_set_visual_mesh(tetsim_handle h, const float* vis_verts, uint32_t nvis, const float* rest_normals) {
    if (!h || (nvis && !vis_verts)) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->vis_attached) return fail(h, TETSIM_ESTATE, "a visual mesh is already attached");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    const uint32_t nt = h->info.num_elems;
    // A PARTITION takes the same (global) list on every rank and keeps the visual vertices whose tet it OWNS by the lowest-owner rule
    // (TetSimInfo.owned_elems: every tet has exactly one such partition, and that partition solves it, so the tet's quaternion is
    // local); the union over the partitions is the whole visual mesh, tetsim_get_visual_ids says which rows a partition kept.
    // Corners of a kept tet are owned or ghost particles: the ghosts' end-of-substep positions are fetched by
    // tetsim_read_visual_mesh (tetsim_halo_refresh_final).
    std::vector<int32_t> g2l_tet, g2l_vert;
    if (h->partitioned) {
        g2l_tet.assign(nt, -1);
        for (size_t i = 0; i < h->part.local_to_global_tet.size(); i++) g2l_tet[h->part.local_to_global_tet[i]] = static_cast<int32_t>(i);
        g2l_vert.assign(h->info.num_particles, -1);
        for (size_t i = 0; i < h->part.local_to_global_vert.size(); i++) g2l_vert[h->part.local_to_global_vert[i]] = static_cast<int32_t>(i);
    }
    std::vector<int32_t> tet_pos;  // (local) tet id -> device tet position (quaternion index)
    if (pjs) {
        const uint32_t ntl = h->partitioned ? h->info.local_elems : nt;
        tet_pos.resize(ntl);
        for (uint32_t i = 0; i < ntl; i++) tet_pos[h->blocked ? h->tet_perm[i] : i] = static_cast<int32_t>(i);
    }
    std::vector<int4> corner;
    std::vector<float4> weight, n0;
    std::vector<int32_t> qidx, kept;
    for (uint32_t i = 0; i < nvis; i++) {
        const float tn = vis_verts[4 * i];
        if (!(tn >= 0.0f) || tn >= static_cast<float>(nt) || tn != std::floor(tn)) return fail(h, TETSIM_EINVAL, "visual vertex " + std::to_string(i) + " references a tet outside the mesh");
        const uint32_t e = static_cast<uint32_t>(tn);
        uint32_t el = e;   // the tet in this handle's numbering
        if (h->partitioned) {
            int lowest = h->opt.part_count;
            bool all_local = g2l_tet[e] >= 0;
            for (int k = 0; k < 4 && all_local; k++) {
                const int32_t lv = g2l_vert[h->h_tets[4 * e + k]];
                if (lv < 0) { all_local = false; break; }
                // owner of a local particle: this partition for the owned range, the neighbour whose receive range holds it otherwise
                int owner = h->opt.part_index;
                // (two-layer ghost regions keep the second layer in receive ranges of its own: a tet with such a corner belongs to the
                // neighbour too -- found in neither range it would pass for this partition's and be kept twice; advisor, round 5)
                if (static_cast<uint32_t>(lv) >= h->part.n_owned)
                    for (const auto& nb : h->part.neigh)
                        if ((static_cast<uint32_t>(lv) >= nb.recv_start && static_cast<uint32_t>(lv) < nb.recv_start + nb.recv_count) ||
                            (static_cast<uint32_t>(lv) >= nb.recv2_start && static_cast<uint32_t>(lv) < nb.recv2_start + nb.recv2_count)) owner = nb.rank;
                lowest = std::min(lowest, owner);
            }
            if (!all_local || lowest != h->opt.part_index) continue;   // another partition's row
            el = static_cast<uint32_t>(g2l_tet[e]);
        }
        int32_t c[4];
        for (int k = 0; k < 4; k++) {
            int32_t v = h->h_tets[4 * e + k];
            if (h->partitioned) v = g2l_vert[v];
            c[k] = (pjs && !h->api2dev.empty()) ? static_cast<int32_t>(h->api2dev[v]) : v;
        }
        corner.push_back(make_int4(c[0], c[1], c[2], c[3]));
        weight.push_back(make_float4(vis_verts[4 * i + 1], vis_verts[4 * i + 2], vis_verts[4 * i + 3], 0.0f));
        qidx.push_back(pjs ? tet_pos[el] : 0);
        if (rest_normals) n0.push_back(make_float4(rest_normals[3 * i], rest_normals[3 * i + 1], rest_normals[3 * i + 2], 0.0f));
        kept.push_back(static_cast<int32_t>(i));
    }
    const uint32_t nk = static_cast<uint32_t>(kept.size());
    SkinDev& k = h->skin;
    int4* dc; float4 *dw, *dn = nullptr; int32_t* dq;
    int rc;
    if ((rc = dev_alloc(h, &dc, nk))) return rc;
    if ((rc = dev_alloc(h, &dw, nk))) return rc;
    if ((rc = dev_alloc(h, &dq, nk))) return rc;
    if ((rc = dev_alloc(h, &k.out_pos, nk))) return rc;
    if ((rc = upload(h, dc, corner))) return rc;
    if ((rc = upload(h, dw, weight))) return rc;
    if ((rc = upload(h, dq, qidx))) return rc;
    if (rest_normals && pjs) {
        if ((rc = dev_alloc(h, &dn, nk))) return rc;
        if ((rc = dev_alloc(h, &k.out_nrm, nk))) return rc;
        if ((rc = upload(h, dn, n0))) return rc;
    }
    k.corner = dc; k.weight = dw; k.qidx = dq; k.normal0 = dn;
    k.nvis = nk;
    h->vis_global = kept;
    h->vis_attached = true;
    h->info.num_vis_verts = nk;
    return 0;
}

int tetsim_get_visual_ids(tetsim_handle h, int32_t* out) {
    if (!h || !out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->vis_attached) return fail(h, TETSIM_ESTATE, "no visual mesh attached (tetsim_set_visual_mesh)");
    std::copy(h->vis_global.begin(), h->vis_global.end(), out);
    return 0;
}

int tetsim_read_visual_mesh(tetsim_handle h, float* positions_out, float* normals_out) {
    if (!h || !positions_out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->vis_attached) return fail(h, TETSIM_ESTATE, "no visual mesh attached (tetsim_set_visual_mesh)");
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    if (normals_out && h->skin.nvis && !h->skin.out_nrm) return fail(h, TETSIM_ESTATE, "normals need POLAR_JACOBI and rest normals at tetsim_set_visual_mesh");
    HIPCHK(h, hipSetDevice(h->opt.device));
    if (h->partitioned && !h->neigh.empty() && !h->final_ghosts_fresh)
        // The corners this partition does not own: their end-of-substep positions come from the neighbours, by an EXPLICIT call every rank
        // makes.  (Round 5 ran the RCCL exchange from inside this read, gated by a per-rank flag: one rank reading twice per frame, or only
        // some ranks having refreshed, left the others alone inside a collective -- a hang.  A read never communicates.)
        return fail(h, TETSIM_ESTATE, "the ghost particles' end-of-substep positions are stale: after the frame's last substep every rank calls tetsim_halo_refresh_final "
                                      "(RCCL; in-process groups: tetsim_group_refresh_final) before it reads the visual mesh of a partition");
    if (!h->skin.nvis) return 0;   // (a partition that owns no tet with a visual vertex)
    // Softbody.js arithmetic for the solver that mirrors Softbody.js, the vertex-shader arithmetic for the other
    if (pjs) { if (int rc = ensure_quats(h)) return rc; }   // (lean-state bodies: the quaternions the skinning shader reads, SoftbodyGPU.js:440)
    skin_launch(h->stream, h->skin, pjs ? h->pj.pos_final : h->nh.pos, pjs ? h->pj.quat : nullptr, !pjs);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t n = h->skin.nvis;
    std::vector<float4> tmp(n);
    HIPCHK(h, hipMemcpy(tmp.data(), h->skin.out_pos, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) { positions_out[3 * i] = tmp[i].x; positions_out[3 * i + 1] = tmp[i].y; positions_out[3 * i + 2] = tmp[i].z; }
    if (normals_out) {
        HIPCHK(h, hipMemcpy(tmp.data(), h->skin.out_nrm, n * sizeof(float4), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n; i++) { normals_out[3 * i] = tmp[i].x; normals_out[3 * i + 1] = tmp[i].y; normals_out[3 * i + 2] = tmp[i].z; }
    }
    return 0;
}

int tetsim_set_visual_triangles(tetsim_handle h, const int32_t* tri_ids, uint32_t ntri) {
    if (!h || (ntri && !tri_ids)) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->partitioned) return fail(h, TETSIM_ESTATE, "visual triangles (computeVertexNormals) are supported on unpartitioned bodies only: a triangle's corners may be "
                                                      "skinned by different partitions; partitions deliver the quaternion-rotated normals of tetsim_read_visual_mesh");
    if (!h->skin.nvis) return fail(h, TETSIM_ESTATE, "no visual mesh attached (tetsim_set_visual_mesh)");
    if (h->skin.vt_off) return fail(h, TETSIM_ESTATE, "visual triangles are already attached");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const uint32_t nvis = h->skin.nvis;
    std::vector<int4> tri(ntri);
    std::vector<uint32_t> off(nvis + 1, 0);
    for (uint32_t t = 0; t < ntri; t++) {
        for (int k = 0; k < 3; k++) {
            const int32_t v = tri_ids[3 * t + k];
            if (v < 0 || static_cast<uint32_t>(v) >= nvis) return fail(h, TETSIM_EINVAL, "triangle " + std::to_string(t) + " references a visual vertex outside the mesh");
            off[v + 1]++;
        }
        tri[t] = make_int4(tri_ids[3 * t], tri_ids[3 * t + 1], tri_ids[3 * t + 2], 0);
    }
    for (uint32_t v = 0; v < nvis; v++) off[v + 1] += off[v];
    std::vector<uint32_t> ent(3ull * ntri), fill(off.begin(), off.end() - 1);
    for (uint32_t t = 0; t < ntri; t++)   // triangle order, corner order: the order of the reference's accumulation
        for (int k = 0; k < 3; k++) ent[fill[tri_ids[3 * t + k]]++] = t;
    SkinDev& k = h->skin;
    int4* dt; uint32_t *doff, *dent;
    int rc;
    if ((rc = dev_alloc(h, &dt, ntri))) return rc;
    if ((rc = dev_alloc(h, &doff, off.size()))) return rc;
    if ((rc = dev_alloc(h, &dent, ent.size()))) return rc;
    if ((rc = dev_alloc(h, &k.out_vnrm, nvis))) return rc;
    if ((rc = upload(h, dt, tri))) return rc;
    if ((rc = upload(h, doff, off))) return rc;
    if ((rc = upload(h, dent, ent))) return rc;
    k.ntri = ntri; k.tri = dt; k.vt_tri = dent;
    k.vt_off = doff;
    return 0;
}

int tetsim_read_visual_vertex_normals(tetsim_handle h, float* normals_out) {
    if (!h || !normals_out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->skin.vt_off) return fail(h, TETSIM_ESTATE, "no visual triangles attached (tetsim_set_visual_triangles)");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    if (pjs) { if (int rc = ensure_quats(h)) return rc; }
    skin_launch(h->stream, h->skin, pjs ? h->pj.pos_final : h->nh.pos, pjs ? h->pj.quat : nullptr, !pjs);
    skin_launch_vertex_normals(h->stream, h->skin);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t n = h->skin.nvis;
    std::vector<float4> tmp(n);
    HIPCHK(h, hipMemcpy(tmp.data(), h->skin.out_vnrm, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) { normals_out[3 * i] = tmp[i].x; normals_out[3 * i + 1] = tmp[i].y; normals_out[3 * i + 2] = tmp[i].z; }
    return 0;
}

int tetsim_set_grab(tetsim_handle h, int32_t id, const float xyz[3]) {
    if (!h) return TETSIM_EINVAL;
    if (id >= static_cast<int32_t>(h->info.num_particles)) return fail(h, TETSIM_EINVAL, "grab id out of range");
    h->grab_global = id < 0 ? -1 : id;
    ref_grab_texels(h->grab_global, h->info.num_elems, h->info.num_particles, h->grab_ref);
    if (xyz) std::memcpy(h->grab_pos, xyz, 3 * sizeof(float));
    return 0;
}
namespace {
// argmin of Softbody.js:279-291 over this handle's OWNED particles, on the device: one (d2, index) candidate per 256
// particles comes back.  *local receives the API-local index (first minimum), *best its squared distance (f64).
int nearest_owned(tetsim_body* h, const float xyz[3], int32_t* local, double* best_out) {
    HIPCHK(h, hipSetDevice(h->opt.device));
    const uint32_t n = h->info.owned_particles, nblk = (n + 255u) / 256u;
    int rc;
    if (!h->d_best) {
        if ((rc = dev_alloc(h, &h->d_best, nblk))) return rc;
        if ((rc = dev_alloc(h, &h->d_best_id, nblk))) return rc;
        if ((rc = ensure_index_map(h))) return rc;
    }
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    util_launch_nearest(h->stream, current_positions(h), h->d_api2dev, n, static_cast<double>(xyz[0]), static_cast<double>(xyz[1]),
                        static_cast<double>(xyz[2]), h->d_best, h->d_best_id);
    std::vector<double> bd(nblk);
    std::vector<uint32_t> bi(nblk);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (nblk) {
        HIPCHK(h, hipMemcpy(bd.data(), h->d_best, nblk * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(bi.data(), h->d_best_id, nblk * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    double best = 1.7976931348623157e308;
    int32_t id = -1;
    for (uint32_t b = 0; b < nblk; b++)  // blocks are in ascending particle order: `<` keeps the first minimum
        if (bd[b] < best) { best = bd[b]; id = static_cast<int32_t>(bi[b]); }
    *local = id;
    *best_out = best;
    return 0;
}
}  // namespace

int tetsim_start_grab(tetsim_handle h, const float xyz[3], int32_t* id_out) {
    if (!h || !xyz) return fail(h, TETSIM_EINVAL, "null argument");
    if (h->partitioned) return fail(h, TETSIM_ESTATE, "start_grab on a partitioned body: combine tetsim_nearest_particle over the partitions on the host, then tetsim_set_grab on each");
    int32_t id = -1;
    double best = 0.0;
    int rc = nearest_owned(h, xyz, &id, &best);
    if (rc) return rc;
    h->grab_global = id;
    ref_grab_texels(h->grab_global, h->info.num_elems, h->info.num_particles, h->grab_ref);
    std::memcpy(h->grab_pos, xyz, 3 * sizeof(float));
    if (id_out) *id_out = id;
    return 0;
}

int tetsim_nearest_particle(tetsim_handle h, const float xyz[3], int32_t* global_id, double* dist2) {
    if (!h || !xyz || !global_id || !dist2) return fail(h, TETSIM_EINVAL, "null argument");
    int32_t local = -1;
    int rc = nearest_owned(h, xyz, &local, dist2);
    if (rc) return rc;
    *global_id = local < 0 ? -1 : (h->partitioned ? h->part.local_to_global_vert[local] : local);
    return 0;
}

}  // extern "C"

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
    h->vis_total = nvis;
    h->vis_attached = true;
    h->info.num_vis_verts = nk;
    h->info.total_vis_verts = nvis;
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
    if (!h->vis_attached) return fail(h, TETSIM_ESTATE, "no visual mesh attached (tetsim_set_visual_mesh)");
    if (h->skin.vt_off) return fail(h, TETSIM_ESTATE, "visual triangles are already attached");
    HIPCHK(h, hipSetDevice(h->opt.device));
    // A PARTITION takes the same, global triangle list on every rank (ids = rows of the caller's visVerts) and keeps, for each row it
    // skins, that row's triangles in triangle order: a triangle's other corners may be skinned by other ranks, so the normals are computed
    // from the ranks' skins put together (tetsim_visual_vertex_normals_from) -- per vertex the same sums in the same order as unpartitioned.
    const bool part = h->partitioned;
    const uint32_t nrows = part ? h->vis_total : h->skin.nvis;      // what a triangle id may address
    const uint32_t nvis = h->skin.nvis;                              // the rows this handle computes normals for
    std::vector<int32_t> row_of;                                     // partitions: global row -> own row, or -1
    if (part) {
        row_of.assign(nrows, -1);
        for (uint32_t j = 0; j < nvis; j++) row_of[h->vis_global[j]] = static_cast<int32_t>(j);
    }
    auto own = [&](int32_t g) { return part ? row_of[g] : g; };
    std::vector<int4> tri(ntri);
    std::vector<uint32_t> off(nvis + 1, 0);
    for (uint32_t t = 0; t < ntri; t++) {
        for (int k = 0; k < 3; k++) {
            const int32_t v = tri_ids[3 * t + k];
            if (v < 0 || static_cast<uint32_t>(v) >= nrows) return fail(h, TETSIM_EINVAL, "triangle " + std::to_string(t) + " references a visual vertex outside the mesh");
            if (own(v) >= 0) off[own(v) + 1]++;
        }
        tri[t] = make_int4(tri_ids[3 * t], tri_ids[3 * t + 1], tri_ids[3 * t + 2], 0);
    }
    for (uint32_t v = 0; v < nvis; v++) off[v + 1] += off[v];
    std::vector<uint32_t> ent(off[nvis]), fill(off.begin(), off.end() - 1);
    for (uint32_t t = 0; t < ntri; t++)   // triangle order, corner order: the order of the reference's accumulation
        for (int k = 0; k < 3; k++) { const int32_t o = own(tri_ids[3 * t + k]); if (o >= 0) ent[fill[o]++] = t; }
    SkinDev& k = h->skin;
    int4* dt; uint32_t *doff, *dent;
    int rc;
    if ((rc = dev_alloc(h, &dt, ntri))) return rc;
    if ((rc = dev_alloc(h, &doff, off.size()))) return rc;
    if ((rc = dev_alloc(h, &dent, ent.size()))) return rc;
    if ((rc = dev_alloc(h, &k.out_vnrm, nvis))) return rc;
    if ((rc = dev_alloc(h, &h->d_vis_full, nrows))) return rc;   // (tetsim_visual_vertex_normals_from; unpartitioned bodies may use it too)
    if ((rc = upload(h, dt, tri))) return rc;
    if ((rc = upload(h, doff, off))) return rc;
    if ((rc = upload(h, dent, ent))) return rc;
    k.ntri = ntri; k.tri = dt; k.vt_tri = dent;
    k.vt_off = doff;
    return 0;
}

namespace {
int read_vnrm(tetsim_body* h, float* normals_out) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t n = h->skin.nvis;
    std::vector<float4> tmp(n);
    if (n) HIPCHK(h, hipMemcpy(tmp.data(), h->skin.out_vnrm, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) { normals_out[3 * i] = tmp[i].x; normals_out[3 * i + 1] = tmp[i].y; normals_out[3 * i + 2] = tmp[i].z; }
    return 0;
}
}  // namespace

int tetsim_read_visual_vertex_normals(tetsim_handle h, float* normals_out) {
    if (!h || !normals_out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->skin.vt_off) return fail(h, TETSIM_ESTATE, "no visual triangles attached (tetsim_set_visual_triangles)");
    if (h->partitioned) return fail(h, TETSIM_ESTATE, "a partition skins only its own rows, and a triangle's corners may belong to other ranks: put the ranks' skins together "
                                                      "(tetsim_read_visual_mesh + tetsim_get_visual_ids) and call tetsim_visual_vertex_normals_from, or tetsim_group_read_visual_vertex_normals");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const bool pjs = h->opt.solver == TETSIM_SOLVER_POLAR_JACOBI;
    if (pjs) { if (int rc = ensure_quats(h)) return rc; }
    skin_launch(h->stream, h->skin, pjs ? h->pj.pos_final : h->nh.pos, pjs ? h->pj.quat : nullptr, !pjs);
    skin_launch_vertex_normals(h->stream, h->skin);
    return read_vnrm(h, normals_out);
}

// computeVertexNormals (Softbody.js:273) of THIS handle's rows from a complete set of visual positions -- for a partition the ranks'
// skins put together by the host (every rank's tetsim_read_visual_mesh scattered by tetsim_get_visual_ids), for an unpartitioned body
// any positions of its visual mesh.  Per vertex the face normals of its triangles are added in triangle order exactly as three.js
// does, so the partitions' rows equal the unpartitioned body's normals bit for bit.
int tetsim_visual_vertex_normals_from(tetsim_handle h, const float* all_positions, float* normals_out) {
    if (!h || !all_positions || !normals_out) return fail(h, TETSIM_EINVAL, "null argument");
    if (!h->skin.vt_off) return fail(h, TETSIM_ESTATE, "no visual triangles attached (tetsim_set_visual_triangles)");
    HIPCHK(h, hipSetDevice(h->opt.device));
    const uint32_t nrows = h->partitioned ? h->vis_total : h->skin.nvis;
    std::vector<float4> full(nrows);
    for (uint32_t i = 0; i < nrows; i++) full[i] = make_float4(all_positions[3 * i], all_positions[3 * i + 1], all_positions[3 * i + 2], 0.0f);
    HIPCHK(h, hipStreamSynchronize(h->stream));   // (a previous call's kernel may still read the buffer)
    if (nrows) HIPCHK(h, hipMemcpy(h->d_vis_full, full.data(), nrows * sizeof(float4), hipMemcpyHostToDevice));
    SkinDev k = h->skin;
    k.tri_pos = h->d_vis_full;
    skin_launch_vertex_normals(h->stream, k);
    return read_vnrm(h, normals_out);
}

// The partitions of ONE process: every member skins its rows (their ghosts' end-of-substep positions must be fresh:
// tetsim_group_refresh_final), the rows are put together, every member computes the normals of its rows from the whole.  positions_out /
// normals_out [3 * rows of visVerts], either may be NULL.  Equal to the unpartitioned body's tetsim_read_visual_mesh /
// tetsim_read_visual_vertex_normals bit for bit (PRECISE).
int tetsim_group_read_visual_vertex_normals(tetsim_handle* hs, uint32_t count, float* positions_out, float* normals_out) {
    group_begin(hs, count);
    auto impl = [&]() -> int {
        if (!hs || count == 0) return TETSIM_EINVAL;
        for (uint32_t i = 0; i < count; i++)
            if (!hs[i] || hs[i]->opt.part_count != static_cast<int32_t>(count) || hs[i]->opt.part_index != static_cast<int32_t>(i))
                return fail(hs[i], TETSIM_EINVAL, "handles[i] must be partition i of a count-way decomposition");
        const uint32_t total = hs[0]->vis_total;
        std::vector<float> full(3ull * total, 0.0f), rows, nrm;
        for (uint32_t i = 0; i < count; i++) {
            tetsim_body* h = hs[i];
            if (!h->vis_attached || h->vis_total != total) return fail(h, TETSIM_ESTATE, "every member needs the same visual mesh (tetsim_set_visual_mesh)");
            rows.resize(3ull * h->skin.nvis);
            if (int rc = tetsim_read_visual_mesh(h, rows.data(), nullptr)) return rc;
            for (uint32_t j = 0; j < h->skin.nvis; j++) std::memcpy(&full[3ull * h->vis_global[j]], &rows[3ull * j], 3 * sizeof(float));
        }
        if (positions_out) std::memcpy(positions_out, full.data(), full.size() * sizeof(float));
        if (!normals_out) return 0;
        for (uint32_t i = 0; i < count; i++) {
            tetsim_body* h = hs[i];
            nrm.resize(3ull * h->skin.nvis);
            if (int rc = tetsim_visual_vertex_normals_from(h, full.data(), nrm.data())) return rc;
            for (uint32_t j = 0; j < h->skin.nvis; j++) std::memcpy(&normals_out[3ull * h->vis_global[j]], &nrm[3ull * j], 3 * sizeof(float));
        }
        return 0;
    };
    return group_result(hs, count, impl());
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

// tetsim_create.hip -- construction: host preprocessing (host_prep.cpp) -> device state of the two solvers.
#include "body.h"

namespace tetsim {

// ---- construction ----------------------------------------------------------------------------------------
int create_polar(tetsim_body* h, const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt) {
    const TetSimOptions& o = h->opt;
    const bool ref_table = (o.flags & TETSIM_FLAG_REF_SLOT_TABLE) != 0;
    std::vector<int32_t> ltets;   // local connectivity
    std::vector<int32_t> l2g_v, l2g_t;
    uint32_t nvl = nv, nvo = nv, nvb = 0, ntl = nt;
    h->partitioned = o.part_count > 1;
    h->deep = h->partitioned && (o.flags & TETSIM_FLAG_DEEP_GHOSTS);
    {   // how this body's halo is synchronised and replayed: read per body, so that a caller can rebuild a body more conservatively
        // after a transport failure (bench.py's N > 1 retry ladder) without restarting the process
        const char* sy = getenv("TETSIM_HALO_SYNC");
        const char* gr = getenv("TETSIM_HALO_GRAPH");
        h->halo_use_flags = !(sy && sy[0] == 'e');
        h->halo_use_graph = !(gr && gr[0] == '0');
    }
    if ((o.flags & TETSIM_FLAG_DEEP_GHOSTS) && !(h->partitioned && h->fast && !(o.flags & TETSIM_FLAG_GATHER_FORMULATION)))
        return fail(h, TETSIM_EINVAL, "TETSIM_FLAG_DEEP_GHOSTS needs a partitioned POLAR_JACOBI body in the blocked FAST formulation");
    uint32_t nvg1 = 0;   // first-layer ghosts that this partition advances itself (two-layer ghost regions)
    if (h->partitioned) {
        std::string e = build_partition(tets, nt, nv, o.part_count, o.part_index, o.vert_owner, &h->part, h->deep ? 2 : 1);
        if (!e.empty()) return fail(h, TETSIM_EINVAL, e);
        if (h->deep) {
            nvg1 = h->part.n_ghost1;
            h->n_ghost1 = nvg1;
            h->g2l_ghost1.assign(nv, -1);
            for (uint32_t i = 0; i < nvg1; i++) h->g2l_ghost1[h->part.local_to_global_vert[h->part.n_owned + i]] = static_cast<int32_t>(h->part.n_owned + i);
        }
        const Partition& P = h->part;
        ltets = P.local_tets;
        l2g_v = P.local_to_global_vert;
        l2g_t = P.local_to_global_tet;
        nvl = static_cast<uint32_t>(l2g_v.size());
        nvo = P.n_owned;
        nvb = P.n_boundary;
        ntl = static_cast<uint32_t>(l2g_t.size());
        h->g2l_owned.assign(nv, -1);
        for (uint32_t i = 0; i < nvo; i++) h->g2l_owned[l2g_v[i]] = static_cast<int32_t>(i);
        h->info.owned_elems = P.owned_tets;
    } else {
        ltets.assign(tets, tets + 4ull * nt);
        h->info.owned_elems = nt;
    }
    // device numbering: Morton order inside the interior segment [nvb, nvo); boundary (halo sends stay contiguous
    // runs) and ghosts (receive ranges) keep the plan's order.  Batches: every body's segment on its own (as it would be alone).
    const bool batch = !h->batch_first_vert.empty();
    const uint32_t bodies = batch ? static_cast<uint32_t>(h->batch_first_vert.size() - 1) : 1u;
    {
        std::vector<float> lv(3ull * nvl);
        for (uint32_t i = 0; i < nvl; i++) {
            const uint32_t g = h->partitioned ? static_cast<uint32_t>(l2g_v[i]) : i;
            lv[3 * i] = verts[3 * g]; lv[3 * i + 1] = verts[3 * g + 1]; lv[3 * i + 2] = verts[3 * g + 2];
        }
        if (!batch) h->dev2api = morton_vertex_order(lv.data(), nvl, nvb, nvo - nvb);
        else {
            h->dev2api.resize(nvl);
            for (uint32_t b = 0; b < bodies; b++) {
                const uint32_t vb = h->batch_first_vert[b], n = h->batch_first_vert[b + 1] - vb;
                const std::vector<uint32_t> ord = morton_vertex_order(lv.data() + 3ull * vb, n, 0, n);
                for (uint32_t i = 0; i < n; i++) h->dev2api[vb + i] = vb + ord[i];
            }
        }
        h->api2dev.resize(nvl);
        for (uint32_t dv = 0; dv < nvl; dv++) h->api2dev[h->dev2api[dv]] = dv;
        for (auto& id : ltets) id = static_cast<int32_t>(h->api2dev[id]);
    }
    // incidence (which (tet, corner) contributions each particle sums).  The reference's `<= 0.0` quirk drops the contribution of
    // ITS tet 0 / corner 0: in a batch that is every body's own first tet, so the table is built body by body.
    Incidence inc;
    if (!batch) {
        const bool quirk_here = ref_table && ntl > 0 && (!h->partitioned || l2g_t[0] == 0);
        inc = build_incidence(ltets.data(), ntl, nvl, quirk_here, ref_table);
    } else {
        inc.offset.assign(nvl + 1, 0);
        for (uint32_t b = 0; b < bodies; b++) {
            const uint32_t vb = h->batch_first_vert[b], nvb_ = h->batch_first_vert[b + 1] - vb;
            const uint32_t tb = h->batch_first_tet[b], ntb_ = h->batch_first_tet[b + 1] - tb;
            std::vector<int32_t> bt(ltets.begin() + 4ull * tb, ltets.begin() + 4ull * (tb + ntb_));
            for (auto& id : bt) id -= static_cast<int32_t>(vb);   // a body's particles occupy [vb, vb + nvb_) in device numbering too
            const Incidence bi = build_incidence(bt.data(), ntb_, nvb_, ref_table && ntb_ > 0, ref_table);
            const uint32_t base = static_cast<uint32_t>(inc.slot.size());
            for (uint32_t v = 0; v < nvb_; v++) inc.offset[vb + v + 1] = base + bi.offset[v + 1];
            for (int32_t enc : bi.slot) inc.slot.push_back(enc + static_cast<int32_t>(4u * tb));
            inc.max_valence = std::max(inc.max_valence, bi.max_valence);
            inc.dropped += bi.dropped;
        }
    }
    // Only owned vertices are averaged here; the table rows of ghosts are never read.
    uint32_t maxv = 0;
    for (uint32_t v = 0; v < nvo + nvg1; v++) maxv = std::max(maxv, inc.offset[v + 1] - inc.offset[v]);

    PJDev& d = h->pj;
    d.nv_local = nvl; d.nv_owned = nvo; d.nv_boundary = nvb; d.nt = ntl;
    d.nt_pad = (ntl + 63u) & ~63u;
    d.nv_pad = (nvo + 63u) & ~63u;
    d.max_valence = maxv;
    h->info.owned_particles = nvo;
    h->info.local_particles = nvl;
    h->info.local_elems = ntl;
    h->info.max_valence = maxv;
    h->info.dropped_slots = inc.dropped;

    int rc;
    if ((rc = dev_alloc(h, &d.pos_pred, nvl))) return rc;
    if ((rc = dev_alloc(h, &d.pos_final, nvl))) return rc;
    if ((rc = dev_alloc(h, &d.vel, nvl))) return rc;
    d.params = h->d_params;
    // FAST: the correction iterations 2..9 of a tet end below 1e-6 rad unless the caller wants the reference's 1e-9 (pj_math.inc)
    d.rot_exit_w2 = (h->fast && !(o.flags & TETSIM_FLAG_REF_ROTATION_EXIT)) ? 1.0e-12f : 1.0e-18f;

    std::vector<float4> pos(nvl);
    std::vector<float> lverts(3ull * nvl);
    for (uint32_t i = 0; i < nvl; i++) {  // i = device index
        const uint32_t a = h->dev2api[i];
        const uint32_t g = h->partitioned ? static_cast<uint32_t>(l2g_v[a]) : a;
        pos[i] = make_float4(verts[3 * g], verts[3 * g + 1], verts[3 * g + 2], 0.0f);
        lverts[3 * i] = verts[3 * g]; lverts[3 * i + 1] = verts[3 * g + 1]; lverts[3 * i + 2] = verts[3 * g + 2];
    }
    if ((rc = upload(h, d.pos_pred, pos))) return rc;
    if ((rc = upload(h, d.pos_final, pos))) return rc;
    h->final_ghosts_fresh = true;   // (pos_final's ghost range holds the rest positions: a fresh partition can skin its visual mesh -- SoftBodyHIP.js does, in its constructor)
    HIPCHK(h, hipMemset(d.vel, 0, std::max<size_t>(nvl, 1) * sizeof(float4)));

    // the weight the reference's P4 writes into elems.w: 1.0 / texture(invRestVolume).x, in f32
    // (SoftbodyGPU.js:220,259-262 with invRestVolume = fround(1/V), :582-589)
    auto rest_weight = [&](uint32_t local_tet) {
        const uint32_t ge = h->partitioned ? static_cast<uint32_t>(l2g_t[local_tet]) : local_tet;
        return 1.0f / pj_inv_rest_volume(verts, &tets[4 * ge]);
    };

    h->blocked = h->fast && !(o.flags & TETSIM_FLAG_GATHER_FORMULATION);
    if ((o.flags & TETSIM_FLAG_CONSTANT_REST_SHAPE) && !h->blocked)
        return fail(h, TETSIM_EINVAL, "TETSIM_FLAG_CONSTANT_REST_SHAPE needs POLAR_JACOBI + TETSIM_FAST without TETSIM_FLAG_GATHER_FORMULATION");
    if ((o.flags & TETSIM_FLAG_LEAN_STATE) && !h->blocked)
        return fail(h, TETSIM_EINVAL, "TETSIM_FLAG_LEAN_STATE needs POLAR_JACOBI + TETSIM_FAST without TETSIM_FLAG_GATHER_FORMULATION");
    const bool lean_state = (o.flags & TETSIM_FLAG_LEAN_STATE) != 0;
    if (h->blocked) {
        BlockPlan B;
        // partitions: the tets that touch a boundary or a ghost particle get tiles of their own (class 1), and so do the second-layer
        // ghost tets of a two-layer ghost region (class 2) -- host_prep.h.  TETSIM_HALO_ALIGNED_TILES=0 (read here): the round-2 tiling,
        // where every tile that reaches the interface is halo-side (A/B)
        std::vector<uint8_t> tet_class;
        if (h->partitioned) {
            const char* e = lab_env("TETSIM_HALO_ALIGNED_TILES");
            const bool aligned = !(e && e[0] == '0');
            tet_class.assign(ntl, 0);
            for (uint32_t i = 0; i < ntl; i++) {
                if (h->deep && h->part.tet_layer[i]) { tet_class[i] = 2; continue; }
                if (!aligned) continue;
                for (int c = 0; c < 4; c++) {
                    const uint32_t v = static_cast<uint32_t>(ltets[4ull * i + c]);
                    if (v < nvb || v >= nvo) tet_class[i] = 1;
                }
            }
        }
        // SMALL bodies (the reference's own workload, main.js:79-84) are tiled into 64-tet tiles and solved with one tet / one particle
        // on FOUR lanes (pj_quad.hip): every tile's workgroup must be resident at once for the frame kernel, half the device's capacity
        // at most (another body's kernels may hold slots too).  TETSIM_QUAD=0 keeps the 256-tet tiles (development A/B).
        static const bool allow_quad = [] { const char* e = getenv("TETSIM_QUAD"); return !(e && e[0] == '0'); }();
        uint32_t quad_cus = 0, quad_per_cu = 0;
        bool quad = allow_quad && !h->partitioned && !(o.flags & (TETSIM_FLAG_CONSTANT_REST_SHAPE | TETSIM_FLAG_LEAN_STATE)) && ntl > 0 && nvo == nvl;
        if (quad) {
            quad_per_cu = pjq_frame_capacity(&quad_cus);
            quad = quad_per_cu != 0u && (static_cast<uint64_t>(ntl) + kQuadTile - 1u) / kQuadTile <= static_cast<uint64_t>(quad_per_cu) * quad_cus / 2u;
        }
        for (;;) {
            build_blocks(lverts.data(), ltets.data(), ntl, nvl, nvo + nvg1, inc, &B, batch ? h->batch_first_tet.data() : nullptr,
                         batch ? h->batch_first_vert.data() : nullptr, bodies, nvb, h->partitioned ? tet_class.data() : nullptr, nvo, quad ? kQuadTile : kBlockTile);
            // (what the quad kernels take: every particle summed by some tile, lists of at most kQuadMaxPartials partial sums, all tiles resident)
            if (!quad || (B.every_owned_particle_has_a_partial && B.max_partials <= kQuadMaxPartials && B.num_blocks <= quad_per_cu * quad_cus / 2u)) break;
            quad = false;
        }
        h->quad = quad;
        h->tet_perm = B.tet_perm;
        h->nb_first = B.num_first_blocks;
        // the two-queue halo path (tetsim_halo.hip) rests on this: an interior tile touches neither a ghost nor a boundary particle
        for (uint32_t b = 0; b < B.num_interior_blocks; b++)
            for (uint32_t e = B.blk_tet_off[b]; e < B.blk_tet_off[b + 1]; e++)
                for (int c = 0; c < 4; c++) {
                    const uint32_t v = static_cast<uint32_t>(ltets[4ull * static_cast<uint32_t>(B.tet_perm[e]) + c]);
                    if (v < nvb || v >= nvo) return fail(h, TETSIM_ESTATE, "internal error in the tile plan: an interior tile touches a boundary or a ghost particle");
                }
        PJBlk& k = h->blk;
        h->interior_tets = B.blk_tet_off[B.num_interior_blocks];
        k.nb = B.num_blocks; k.nb_interior = B.num_interior_blocks; k.nt = ntl; k.nv_local = nvl; k.nv_owned = nvo; k.nv_boundary = nvb;
        {   // (tetsim_halo.hip: interior_particles) needs interior tiles to put G back and interior particles to look at it
            // Used with the peer-to-peer halo (switched on when it is connected: 38.2 against 39.5 us per substep in loopback, 42.1
            // against 42.9 with 20 us of injected latency); with RCCL's transfer kernel on the halo queue it gains 0.7 us at +0 and
            // loses 1.7 us at +20 us, so there it stays off.  TETSIM_HALO_FOLD_WAIT=0 / 1: never / also with RCCL (development A/B).
            const char* fw = getenv("TETSIM_HALO_FOLD_WAIT");
            h->fold_possible = !(fw && fw[0] == '0') && h->partitioned && B.num_interior_blocks > 0 && B.num_interior_blocks < B.num_blocks && nvo > nvb;
            if (h->fold_possible) pjb_wait_capacity((o.flags & TETSIM_FLAG_CONSTANT_REST_SHAPE) ? 1 : lean_state ? 2 : 0, &h->fold_wave_limit, &h->fold_tile_limit);
            h->fold_wait = h->fold_possible && fw && fw[0] == '1';
            h->fold_halo = h->fold_wait;   // (RCCL bodies: only when forced; the peer-to-peer connection switches both on)
        }
        k.pos_pred = d.pos_pred; k.pos_final = d.pos_final; k.vel = d.vel; k.params = h->d_params;
        k.lean = (o.flags & TETSIM_FLAG_CONSTANT_REST_SHAPE) != 0;
        k.lean_state = lean_state;
        k.rot_exit_w2 = d.rot_exit_w2;
        uint32_t *bto, *bvo, *lcr, *vpe;
        int32_t* bv;
        uchar4* lidx;
        float* vol;
        uint2* lce;
        const size_t nslots = B.blk_verts.size();
        if ((rc = dev_alloc(h, &bto, B.blk_tet_off.size()))) return rc;
        if ((rc = dev_alloc(h, &bvo, B.blk_vert_off.size()))) return rc;
        if ((rc = dev_alloc(h, &bv, nslots))) return rc;
        if (ntl >= kStoreWtMaxIndex || nslots >= kStoreWtMaxIndex || nvl >= kStoreWtMaxIndex)
            return fail(h, TETSIM_EINVAL, "body too large for one handle (2^27 tets / particles / partial sums: 32-bit store offsets, dev_store.h); partition it");
        if ((rc = dev_alloc(h, &lidx, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.rest_a, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.rest_b, ntl))) return rc;
        if (lean_state) {   // three carried corners (a, b, c1); the constant rest shape stays beside them for the quaternion's recovery
            if ((rc = dev_alloc(h, &k.rest_c1, ntl))) return rc;
            if ((rc = dev_alloc(h, &h->rest0_a, ntl))) return rc;
            if ((rc = dev_alloc(h, &h->rest0_b, ntl))) return rc;
            if ((rc = dev_alloc(h, &h->rest0_c, ntl))) return rc;
        } else if ((rc = dev_alloc(h, &k.rest_c, ntl))) return rc;
        if ((rc = dev_alloc(h, &vol, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.quat, ntl))) return rc;
        if ((rc = dev_alloc(h, &lcr, nslots))) return rc;
        if ((rc = dev_alloc(h, &lce, ntl))) return rc;
        if ((rc = dev_alloc(h, &k.partial, nslots))) return rc;
        if ((rc = dev_alloc(h, &vpe, B.vp_ell.size()))) return rc;
        std::vector<float4> ra(ntl), rb(ntl), rcv(ntl), quat(ntl, make_float4(0, 0, 0, 1));
        std::vector<float> volh(ntl);
        std::vector<uchar4> lidxh(ntl);
        std::vector<uint2> lceh(ntl);
        for (uint32_t i = 0; i < ntl; i++) {
            const uint32_t lt = static_cast<uint32_t>(B.tet_perm[i]);
            const int32_t* c = &ltets[4 * lt];
            const float4 p0 = pos[c[0]], p1 = pos[c[1]], p2 = pos[c[2]], p3 = pos[c[3]];
            float4 r0 = p0, r1 = p1, r2 = p2, r3 = p3;
            {  // the blocked kernels keep the (carried or constant) shape relative to its centroid; f32, the kernel's association
                const float cx = (((p0.x + p1.x) + p2.x) + p3.x) * 0.25f, cy = (((p0.y + p1.y) + p2.y) + p3.y) * 0.25f,
                            cz = (((p0.z + p1.z) + p2.z) + p3.z) * 0.25f;
                r0 = make_float4(p0.x - cx, p0.y - cy, p0.z - cz, 0.0f); r1 = make_float4(p1.x - cx, p1.y - cy, p1.z - cz, 0.0f);
                r2 = make_float4(p2.x - cx, p2.y - cy, p2.z - cz, 0.0f); r3 = make_float4(p3.x - cx, p3.y - cy, p3.z - cz, 0.0f);
            }
            ra[i] = make_float4(r0.x, r0.y, r0.z, r1.x);
            rb[i] = make_float4(r1.y, r1.z, r2.x, r2.y);
            rcv[i] = make_float4(r2.z, r3.x, r3.y, r3.z);
            volh[i] = rest_weight(lt);
            lidxh[i] = make_uchar4(B.tet_lidx[4ull * i], B.tet_lidx[4ull * i + 1], B.tet_lidx[4ull * i + 2], B.tet_lidx[4ull * i + 3]);
            const uint16_t* en = &B.lc_ent[4ull * i];
            lceh[i] = make_uint2(en[0] | (static_cast<uint32_t>(en[1]) << 16), en[2] | (static_cast<uint32_t>(en[3]) << 16));
        }
        // Weight sums (PJBlk::wsum): rest volumes are constants, so the denominator of the volume-weighted average
        // (SoftbodyGPU.js:306-319) is added up HERE, once, in exactly the order the kernels add the numerator: a tile slot's
        // entries in entry order, then the particle's tile partial sums in ascending tile order, f32 throughout.
        std::vector<float> wsum(std::max<size_t>(B.nv_pad, 1), 0.0f);
        {
            std::vector<float> slot_w(nslots, 0.0f);
            for (uint32_t b = 0; b < B.num_blocks; b++) {
                const uint32_t t0 = B.blk_tet_off[b], v0 = B.blk_vert_off[b], nu = B.blk_vert_off[b + 1] - v0;
                for (uint32_t u = 0; u < nu; u++) {
                    const uint32_t first = B.lc_range[v0 + u] & 0x7ffu, last = B.lc_range[v0 + u] >> 16;   // (bit 15: owner flag)
                    float w = 0.0f;
                    for (uint32_t i = first; i < last; i++) w += volh[t0 + (B.lc_ent[4ull * t0 + i] % B.tile)];
                    slot_w[v0 + u] = w;
                }
            }
            for (uint32_t v = 0; v < nvo + nvg1; v++) {
                float w = 0.0f;
                for (uint32_t j = 0; j < B.max_partials; j++) {
                    const uint32_t idx = B.vp_ell[static_cast<size_t>(j) * B.nv_pad + v];
                    if (idx != 0xffffffffu) w += slot_w[idx];
                }
                wsum[v] = w;
            }
        }
        float* dws;
        if ((rc = dev_alloc(h, &dws, wsum.size()))) return rc;
        if ((rc = upload(h, dws, wsum))) return rc;
        k.wsum = dws;
        // fused particle pass: unpartitioned bodies whose every particle is summed by some tile (lists of up to 9 partial sums are
        // gathered in one trip, 8 + 1; longer ones -- irregular meshes -- entry by entry behind them), TETSIM_FUSED_PARTICLE_PASS=0 keeps
        // the two-kernel substep (development A/B)
        static const bool allow_fused = [] { const char* e = getenv("TETSIM_FUSED_PARTICLE_PASS"); return !(e && e[0] == '0'); }();
        // ... and only where it pays: a body of fewer tiles than the chip has workgroup slots (2,048) is bound by launches and
        // dependency bubbles, and one kernel per substep instead of two is worth +21% on the Dragon (15 tiles; profiles/archive/r02h_*); the
        // 1 M-tet lattice (3,900 tiles) gains nothing -- the fused kernel costs what the particle kernel and its bubble cost
        // (32.0 us against 25.6 + 5.8), needs 74 registers instead of 49 (6 waves per SIMD instead of 8) and reads lower on the
        // roofline -- and keeps the two-kernel substep.  TETSIM_FUSED_PARTICLE_PASS=1 forces it on (A/B).
        static const bool force_fused = [] { const char* e = getenv("TETSIM_FUSED_PARTICLE_PASS"); return e && e[0] == '1'; }();
        h->fused = allow_fused && !h->quad && !h->partitioned && nvo == nvl && B.every_owned_particle_has_a_partial && ntl > 0 &&
                   (B.num_blocks < 2048u || force_fused);
        k.fin_in = d.pos_final; k.fin_out = d.pos_final;
        h->info.fused_particle_pass = h->fused ? 1u : 0u;
        {   // bodies that keep two kernels per substep (>= 2,048 tiles): both in ONE launch inside tetsim_step_n (pjb_substep_kernel)
            const char* e = getenv("TETSIM_PJ_ONE_LAUNCH");   // (read at every creation: tests build both in one process)
            h->pj_one_launch = !(e && e[0] == '0') && !h->fused && !h->quad && !h->partitioned && nvo == nvl && ntl > 0 && B.num_blocks > 0;
            if (h->pj_one_launch) {
                h->info.fused_particle_pass = 5u;
                if ((rc = dev_alloc(h, &h->d_substep_err, 1))) return rc;
                HIPCHK(h, hipMemset(h->d_substep_err, 0, sizeof(uint32_t)));
            }
        }
        if (h->fused || h->quad) {
            uint32_t *dsrc, *dmax;
            if ((rc = dev_alloc(h, &dsrc, B.slot_src.size()))) return rc;
            if ((rc = dev_alloc(h, &dmax, B.blk_maxsrc.size()))) return rc;
            if ((rc = dev_alloc(h, &h->partial_b, nslots))) return rc;
            if ((rc = upload(h, dsrc, B.slot_src))) return rc;
            if ((rc = upload(h, dmax, B.blk_maxsrc))) return rc;
            HIPCHK(h, hipMemset(h->partial_b, 0, std::max<size_t>(nslots, 1) * sizeof(float4)));
            h->partial_slots = nslots;
            if (h->fused) {   // (the fused per-substep kernel double-buffers the end-of-substep positions; the quad kernels do not)
                if ((rc = dev_alloc(h, &h->pos_final_b, nvl))) return rc;
                if ((rc = upload(h, h->pos_final_b, pos))) return rc;
            }
            k.slot_src = dsrc; k.blk_maxsrc = dmax; k.ns_pad = B.ns_pad;
            // ... and bodies small enough for every tile's workgroup to be resident at once run a whole tetsim_step_n call as ONE
            // persistent launch (pj_blocked.hip: pjb_frame_kernel; TETSIM_FRAME_KERNEL=0 keeps one kernel per substep: A/B).  Half
            // the device's capacity at most: another body's kernels may hold slots too.
            static const bool allow_frame = [] { const char* e = getenv("TETSIM_FRAME_KERNEL"); return !(e && e[0] == '0'); }();
            // Placement.  The tiles of one body exchange partial sums every substep; tiles of different bodies (a batch) never do.
            // If the dispatcher hands block i of a grid to XCD i % 8 (verified once per device with a probe kernel: the XCC_ID
            // register of every block of a test grid), each body's tiles are put on ONE XCD -- bodies spread over the XCDs,
            // largest first -- and the exchange only has to be coherent in that XCD's L2 (TETSIM_FRAME_LOCAL=0: never).  Else,
            // or if a body is too large for half an XCD's resident workgroups, tiles spread over all XCDs and the exchange goes
            // through the memory side.  Either way at most half the resident workgroups the device offers are used: another body's
            // kernels may hold slots too, and a waiting tile keeps its slot.
            uint32_t cus = 0;
            const uint32_t per_cu = !allow_frame ? 0u : h->quad ? pjq_frame_capacity(&cus) : pjb_frame_capacity(pjb_mode(k), &cus);
            const uint32_t nbk = B.num_blocks;
            std::vector<int32_t> block_tile;
            // Up to HALF the device's resident workgroups two such bodies fit side by side whatever the dispatcher does.  A body that
            // needs more (up to all of them: 100 k-200 k tets) still takes the persistent launch -- 6.9 against 9.0 us per substep at
            // 131 k tets, 7.5 against 9.6 at 197 k -- but is `exclusive`: while one lives on a device, the persistent launches of ALL
            // bodies of that device take turns (tetsim_step_n), so that two of them are never half resident next to each other.
            // An exclusive body may take up to 90% of the slots (not all: a compute-unit mask, or a kernel of another queue holding a
            // few, must not leave a tile without a slot), and none at all on a device where a partitioned body's folded halo waits
            // may be holding slots of their own (tetsim_halo.hip: pjb_wait_capacity).
            const uint32_t slots = per_cu * cus, exclusive_max = slots - slots / 10u;
            if (per_cu != 0u && nbk != 0u && nbk <= (h->partitioned ? slots / 2u : exclusive_max)) {
                h->frame_exclusive = nbk > slots / 2u;
                static const bool allow_local = [] { const char* e = lab_env("TETSIM_FRAME_LOCAL"); return !(e && e[0] == '0'); }();
                static int xcd_rule[64] = {};   // per device: 0 = not probed, 1 = round-robin over 8 XCDs verified, 2 = no
                int& rule = xcd_rule[o.device & 63];
                if (allow_local && rule == 0) rule = pjb_probe_xcd(h->stream, 256) == 8u ? 1 : 2;
                // group the tiles by body
                std::vector<std::vector<uint32_t>> groups(bodies);
                for (uint32_t b = 0; b < nbk; b++) {
                    uint32_t body = 0;
                    if (batch) {
                        const uint32_t lt = static_cast<uint32_t>(B.tet_perm[B.blk_tet_off[b]]);
                        body = static_cast<uint32_t>(std::upper_bound(h->batch_first_tet.begin(), h->batch_first_tet.end(), lt) - h->batch_first_tet.begin()) - 1u;
                    }
                    groups[std::min(body, bodies - 1u)].push_back(b);
                }
                const uint32_t budget = per_cu * (cus / 8u) / 2u;   // workgroups of this body per XCD
                std::vector<uint32_t> by_size(bodies);
                for (uint32_t g = 0; g < bodies; g++) by_size[g] = g;
                std::stable_sort(by_size.begin(), by_size.end(), [&](uint32_t a, uint32_t c) { return groups[a].size() > groups[c].size(); });
                std::vector<std::vector<uint32_t>> on_xcd(8);
                bool local = allow_local && rule == 1 && cus % 8u == 0u;
                for (uint32_t g : by_size) {
                    if (!local) break;
                    uint32_t best = 0;
                    for (uint32_t x = 1; x < 8u; x++) if (on_xcd[x].size() < on_xcd[best].size()) best = x;
                    if (on_xcd[best].size() + groups[g].size() > budget) { local = false; break; }
                    on_xcd[best].insert(on_xcd[best].end(), groups[g].begin(), groups[g].end());
                }
                if (!local) {   // any placement: consecutive tiles on consecutive blocks
                    for (auto& v : on_xcd) v.clear();
                    for (uint32_t b = 0; b < nbk; b++) on_xcd[b & 7u].push_back(b);
                }
                uint32_t most = 0;
                for (auto& v : on_xcd) most = std::max<uint32_t>(most, static_cast<uint32_t>(v.size()));
                block_tile.assign(8ull * most, -1);
                for (uint32_t x = 0; x < 8u; x++)
                    for (uint32_t j = 0; j < on_xcd[x].size(); j++) block_tile[8ull * j + x] = static_cast<int32_t>(on_xcd[x][j]);
                h->frame = true;
                frame_turn_enter(h);   // (an exclusive body is counted from its creation on: tetsim_api.hip, FrameTurn)
                h->frame_local = local;
                h->frame_blocks = static_cast<uint32_t>(block_tile.size());
            }
            if (h->frame) {
                if ((rc = dev_alloc(h, &h->d_block_tile, block_tile.size()))) return rc;
                if ((rc = upload(h, h->d_block_tile, block_tile))) return rc;
                if ((rc = dev_alloc(h, &h->d_frame_err, 1))) return rc;
                HIPCHK(h, hipMemset(h->d_frame_err, 0, sizeof(uint32_t)));
                h->info.fused_particle_pass = h->quad ? 3u : 2u;
            }
        }
        if ((rc = upload(h, bto, B.blk_tet_off))) return rc;
        if ((rc = upload(h, bvo, B.blk_vert_off))) return rc;
        if ((rc = upload(h, bv, B.blk_verts))) return rc;
        if ((rc = upload(h, lidx, lidxh))) return rc;
        if ((rc = upload(h, k.rest_a, ra))) return rc;
        if ((rc = upload(h, k.rest_b, rb))) return rc;
        if (lean_state) {
            std::vector<float> rc1(ntl);
            for (uint32_t i = 0; i < ntl; i++) rc1[i] = rcv[i].x;
            if ((rc = upload(h, k.rest_c1, rc1))) return rc;
            if ((rc = upload(h, h->rest0_a, ra))) return rc;
            if ((rc = upload(h, h->rest0_b, rb))) return rc;
            if ((rc = upload(h, h->rest0_c, rcv))) return rc;
        } else if ((rc = upload(h, k.rest_c, rcv))) return rc;
        if ((rc = upload(h, vol, volh))) return rc;
        if ((rc = upload(h, k.quat, quat))) return rc;
        if ((rc = upload(h, lcr, B.lc_range))) return rc;
        if ((rc = upload(h, lce, lceh))) return rc;
        if ((rc = upload(h, vpe, B.vp_ell))) return rc;
        HIPCHK(h, hipMemset(k.partial, 0, std::max<size_t>(nslots, 1) * sizeof(float4)));
        k.blk_tet_off = bto; k.blk_vert_off = bvo; k.blk_verts = bv; k.tet_lidx = lidx; k.vol = vol;
        k.lc_range = lcr; k.lc_ent = lce; k.vp_ell = vpe; k.vp_cols = B.max_partials; k.nv_pad = B.nv_pad;
        d.quat = k.quat;  // tetsim_read_quats
#ifdef TETSIM_ABLATION
        if (lab_env("TETSIM_DEBUG_ITER_HIST")) {  // development: rotation-iteration statistics of every tet-kernel launch (pjb_log_iterations)
            if ((rc = dev_alloc(h, &k.iter_hist, 278))) return rc;
            HIPCHK(h, hipMemset(k.iter_hist, 0, 278 * sizeof(unsigned long long)));
        }
#endif
        if (lab_env("TETSIM_DEBUG_TRACE")) {  // development: per-tile phase timestamps of the LAST tet-kernel launch
            if ((rc = dev_alloc(h, &k.trace, 8ull * B.num_blocks))) return rc;
            HIPCHK(h, hipMemset(k.trace, 0, 8ull * B.num_blocks * sizeof(unsigned long long)));
        }
    } else {
        if ((rc = dev_alloc(h, &d.tet_idx, ntl))) return rc;
        if ((rc = dev_alloc(h, &d.elem, 4ull * d.nt_pad))) return rc;
        if ((rc = dev_alloc(h, &d.quat, ntl))) return rc;
        if ((rc = dev_alloc(h, &d.slot_tab, static_cast<size_t>(std::max(maxv, 1u)) * d.nv_pad))) return rc;
        if ((rc = dev_alloc(h, &d.slot_cnt, d.nv_pad))) return rc;
        std::vector<int4> idx(ntl);
        std::vector<float4> elem(4ull * d.nt_pad, make_float4(0, 0, 0, 0)), quat(ntl, make_float4(0, 0, 0, 1));
        for (uint32_t e = 0; e < ntl; e++) {
            const int32_t* lt = &ltets[4 * e];
            idx[e] = make_int4(lt[0], lt[1], lt[2], lt[3]);
            const float w = rest_weight(e);
            for (int k = 0; k < 4; k++) {
                const float4 p = pos[lt[k]];
                elem[static_cast<size_t>(k) * d.nt_pad + e] = make_float4(p.x, p.y, p.z, w);
            }
        }
        if ((rc = upload(h, d.tet_idx, idx))) return rc;
        if ((rc = upload(h, d.elem, elem))) return rc;
        if ((rc = upload(h, d.quat, quat))) return rc;

        std::vector<int32_t> tab(static_cast<size_t>(std::max(maxv, 1u)) * d.nv_pad, 0);
        std::vector<uint32_t> cnt(d.nv_pad, 0);
        for (uint32_t v = 0; v < nvo; v++) {
            const uint32_t c = inc.offset[v + 1] - inc.offset[v];
            cnt[v] = c;
            for (uint32_t sl = 0; sl < c; sl++) {
                const int32_t enc = inc.slot[inc.offset[v] + sl];
                tab[static_cast<size_t>(sl) * d.nv_pad + v] = static_cast<int32_t>((enc & 3) * d.nt_pad + (enc >> 2));
            }
        }
        if ((rc = upload(h, d.slot_tab, tab))) return rc;
        if ((rc = upload(h, d.slot_cnt, cnt))) return rc;
    }

    if (h->partitioned) {
        for (const auto& nb : h->part.neigh) {
            NeighDev nd;
            nd.rank = nb.rank;
            nd.send_count = static_cast<uint32_t>(nb.send_local.size());
            nd.recv_start = nb.recv_start;
            nd.recv_count = nb.recv_count;
            nd.contiguous = nb.send_contiguous;
            nd.send_first = nd.send_count ? static_cast<uint32_t>(nb.send_local[0]) : 0;
            nd.send_global = nb.send_global;
            nd.recv_global = nb.recv_global;
            nd.send_local = nb.send_local;
            if (nd.send_count) {  // staging is always available (tetsim_halo_export, non-contiguous sends)
                if ((rc = dev_alloc(h, &nd.send_idx, nd.send_count))) return rc;
                if ((rc = dev_alloc(h, &nd.send_buf, nd.send_count))) return rc;
                if ((rc = upload(h, nd.send_idx, nb.send_local))) return rc;
            }
            h->neigh.push_back(std::move(nd));
        }
        h->info.num_neighbours = static_cast<uint32_t>(h->neigh.size());
    }
    return 0;
}

int create_neohookean(tetsim_body* h, const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt) {
    const TetSimOptions& o = h->opt;
    if (o.part_count > 1) return fail(h, TETSIM_EINVAL, "NEOHOOKEAN_GS does not partition (one halo per colour would be needed); use POLAR_JACOBI");
    // 1. element order
    std::vector<int32_t> pre(nt);
    for (uint32_t e = 0; e < nt; e++) pre[e] = static_cast<int32_t>(e);
    if (o.order == TETSIM_ORDER_COLOURED) {
        std::vector<int32_t> colour(nt);
        if (h->tet_colour.size() == nt) colour = h->tet_colour;  // caller-supplied (TetSimOptions.tet_colour)
        else prep_colours(tets, nt, nv, colour.data());
        std::stable_sort(pre.begin(), pre.end(), [&](int32_t a, int32_t b) { return colour[a] < colour[b]; });
    }
    ClusterPlan plan;
    const bool clustered = o.order == TETSIM_ORDER_CLUSTERED;
    if (clustered) {
        plan = prep_clusters(tets, nt, nv);
        pre = plan.pre;
    }
    std::vector<int32_t> ptets(4ull * nt);
    for (uint32_t i = 0; i < nt; i++) std::memcpy(&ptets[4 * i], &tets[4 * pre[i]], 4 * sizeof(int32_t));
    // 2. rest data in the order the reference would see (mass accumulation is order dependent, Softbody.js:74-78)
    std::vector<float> irp(9ull * nt), irv(nt);
    h->h_inv_mass.assign(nv, 0.0f);
    prep_rest(verts, nv, ptets.data(), nt, o.density, h->h_inv_mass.data(), irp.data(), irv.data());
    // 3. dependency levels of that order; solve order = stable sort by level
    std::vector<int32_t> pos_in(nt);
    std::vector<uint32_t> body_of;   // level schedules: the body of the tet at sequential position i (batches)
    uint32_t nl = 0;
    if (clustered) {  // the plan IS the schedule: one launch per cluster colour, storage order = step after step
        nl = static_cast<uint32_t>(plan.launch_off.size() - 1);
        for (uint32_t i = 0; i < nt; i++) pos_in[i] = static_cast<int32_t>(plan.exec_pos[i]);
        h->level_off = plan.launch_off;
    } else {
        std::vector<int32_t> level(nt);
        nl = prep_levels(ptets.data(), nt, nv, level.data());
        for (uint32_t i = 0; i < nt; i++) pos_in[i] = static_cast<int32_t>(i);
        // (a batch: inside a level the tets lie body by body -- any order inside a level gives the sequential result, and the
        // single-workgroup launch below walks ITS body's piece of every level)
        body_of.assign(nt, 0);
        if (!h->batch_first_tet.empty())
            for (uint32_t i = 0; i < nt; i++)
                body_of[i] = static_cast<uint32_t>(std::upper_bound(h->batch_first_tet.begin(), h->batch_first_tet.end(), static_cast<uint32_t>(pre[i])) - h->batch_first_tet.begin()) - 1u;
        std::stable_sort(pos_in.begin(), pos_in.end(), [&](int32_t a, int32_t b) { return level[a] != level[b] ? level[a] < level[b] : body_of[a] < body_of[b]; });
        h->level_off.assign(nl + 1, 0);
        for (uint32_t i = 0; i < nt; i++) h->level_off[level[i] + 1]++;
        for (uint32_t l = 0; l < nl; l++) h->level_off[l + 1] += h->level_off[l];
    }
    h->order.resize(nt);  // solve position -> caller's tet id, for the PERMUTED sequence the reference must be fed
    std::vector<int32_t> seq(nt);
    for (uint32_t i = 0; i < nt; i++) seq[i] = pre[i];
    h->order = seq;  // tetsim_get_tet_order: the sequential order whose result we reproduce
    h->info.num_levels = nl;
    h->info.owned_particles = h->info.local_particles = nv;
    h->info.local_elems = h->info.owned_elems = nt;

    if (nv >= kStoreWtMaxIndex) return fail(h, TETSIM_EINVAL, "more than 2^27 particles (32-bit store offsets, dev_store.h)");
    NHDev& d = h->nh;
    d.nv = nv; d.nt = nt;
    int rc;
    if ((rc = dev_alloc(h, &d.pos, nv))) return rc;
    if ((rc = dev_alloc(h, &d.prev, nv))) return rc;
    if ((rc = dev_alloc(h, &d.vel, nv))) return rc;
    if ((rc = dev_alloc(h, &d.tet_idx, nt))) return rc;
    if ((rc = dev_alloc(h, &d.irp_a, nt))) return rc;
    if ((rc = dev_alloc(h, &d.irp_b, nt))) return rc;
    if ((rc = dev_alloc(h, &d.irp_c, nt))) return rc;
    if ((rc = dev_alloc(h, &d.vol_err, nt))) return rc;
    if ((rc = dev_alloc(h, &d.order, nt))) return rc;
    d.params = h->d_params;

    std::vector<float4> pos(nv);
    for (uint32_t i = 0; i < nv; i++) pos[i] = make_float4(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2], h->h_inv_mass[i]);
    if ((rc = upload(h, d.pos, pos))) return rc;
    if ((rc = upload(h, d.prev, pos))) return rc;
    HIPCHK(h, hipMemset(d.vel, 0, std::max<size_t>(nv, 1) * sizeof(float4)));
    HIPCHK(h, hipMemset(d.vol_err, 0, std::max<size_t>(nt, 1) * sizeof(double)));
    std::vector<int4> idx(nt);
    std::vector<float4> a(nt), b(nt), c(nt);
    std::vector<int32_t> ord(nt);
    for (uint32_t s = 0; s < nt; s++) {
        const uint32_t i = static_cast<uint32_t>(pos_in[s]);  // position in the permuted sequential order
        const int32_t* t = &ptets[4 * i];
        idx[s] = make_int4(t[0], t[1], t[2], t[3]);
        const float* m = &irp[9 * i];
        a[s] = make_float4(m[0], m[1], m[2], m[3]);
        b[s] = make_float4(m[4], m[5], m[6], m[7]);
        c[s] = make_float4(m[8], irv[i], 0.0f, 0.0f);
        ord[s] = static_cast<int32_t>(i);  // vol_err is indexed by sequential position
    }
    if ((rc = upload(h, d.tet_idx, idx))) return rc;
    if ((rc = upload(h, d.irp_a, a))) return rc;
    if ((rc = upload(h, d.irp_b, b))) return rc;
    if ((rc = upload(h, d.irp_c, c))) return rc;
    if ((rc = upload(h, d.order, ord))) return rc;
    if (clustered) {
        if ((rc = dev_alloc(h, &d.corner_slots, nt))) return rc;
        if ((rc = upload(h, d.corner_slots, plan.corner_slots))) return rc;
        if ((rc = dev_alloc(h, &h->d_slot_vid, plan.slot_vid.size()))) return rc;
        if ((rc = upload(h, h->d_slot_vid, plan.slot_vid))) return rc;
        // who touches a particle FIRST in a sweep (launch order; the clusters of one launch share no particle): that lane also does
        // the particle pass between two substeps of a run (TETSIM_NH_FOLD=0, read here: a pass of its own as in round 2 -- A/B)
        std::vector<uint8_t> first_mask;
        std::vector<uint32_t> mask_off(nl + 1, 0);
        {
            const char* e = lab_env("TETSIM_NH_FOLD");
            const char* q = lab_env("TETSIM_NH_QUADS");
            // FAST on four lanes per cluster only: a lane folds two particles there; with one lane per cluster (PRECISE) it would fold
            // eight one after the other, in f64 -- as long as the pass it replaces (measured: 132 against 120 us per substep)
            h->nh_fold = h->fast && !(q && q[0] == '0') && !(e && e[0] == '0');
        }
        for (uint32_t l = 0; l < nl; l++) {   // (colour l's clusters sit at [mask_off[l], mask_off[l + 1]) of the per-cluster tables)
            const uint32_t nsteps = plan.step_off[l + 1] - plan.step_off[l];
            mask_off[l + 1] = mask_off[l] + (nsteps ? plan.step_count[plan.step_off[l]] : 0);
        }
        if (h->nh_fold) {
            std::vector<uint8_t> seen(nv, 0);
            for (uint32_t l = 0; l < nl; l++) {
                const uint32_t clusters = mask_off[l + 1] - mask_off[l];
                first_mask.resize(mask_off[l + 1], 0);
                for (uint32_t k = 0; k < kClusterVerts; k++)
                    for (uint32_t i = 0; i < clusters; i++) {
                        const int32_t v = plan.slot_vid[plan.vid_off[l] + static_cast<size_t>(k) * clusters + i];
                        if (v >= 0 && !seen[v]) { seen[v] = 1; first_mask[mask_off[l] + i] |= static_cast<uint8_t>(1u << k); }
                    }
            }
            std::vector<uint32_t> untouched;
            for (uint32_t v = 0; v < nv; v++) if (!seen[v]) untouched.push_back(v);
            h->nh_untouched = static_cast<uint32_t>(untouched.size());
            if ((rc = dev_alloc(h, &h->d_first_mask, first_mask.size()))) return rc;
            if ((rc = upload(h, h->d_first_mask, first_mask))) return rc;
            if ((rc = dev_alloc(h, &h->d_nh_untouched, untouched.size()))) return rc;
            if ((rc = upload(h, h->d_nh_untouched, untouched))) return rc;
        }
        for (uint32_t l = 0; l < nl; l++) {
            NHClusterLaunch L;
            if (h->nh_fold) L.first_mask = h->d_first_mask + mask_off[l];
            L.nsteps = plan.step_off[l + 1] - plan.step_off[l];
            for (uint32_t j = 0; j < L.nsteps; j++) {
                L.first[j] = plan.step_first[plan.step_off[l] + j];
                L.count[j] = plan.step_count[plan.step_off[l] + j];
            }
            L.clusters = L.nsteps ? L.count[0] : 0;
            L.slot_vid = h->d_slot_vid + plan.vid_off[l];
            h->cluster_launch.push_back(L);
        }
        // The sweep as ONE launch (nh_kernels.inc: nh_sweep1_kernel): per cluster and slot, how many colours back the previous toucher of
        // the slot's particle sits (0 = first touch of the sweep: first_mask's bit) and whether this cluster is the last to touch it;
        // the inverse masses as part of the cluster record; one exchange cell per particle.  The folded particle pass is part of it.
        const bool allow_one_launch = [] { const char* e = getenv("TETSIM_NH_ONE_LAUNCH"); return !(e && e[0] == '0'); }();   // (read at every creation: tests build both in one process)
        // (FAST, four lanes per cluster with the folded particle pass.  PRECISE -- one lane per cluster in f64, bit-exact -- was built the same
        // way for the 1 M-tet lattice and measured SLOWER: 128.6 against 116.7 us per substep with one launch per colour; every look at a
        // handed-on particle is a memory-side load where the per-colour launches hit the L2, 8 per lane: removed, HISTORY.md round 6)
        if (allow_one_launch && h->nh_fold && nl >= 2 && nl <= 255u && nv > 0) {
            const size_t ncl = mask_off[nl];
            std::vector<float> slot_im(plan.slot_vid.size(), 0.0f);
            std::vector<uint2> delta(ncl, make_uint2(0u, 0u));
            std::vector<uint8_t> last_mask(ncl, 0);
            std::vector<int32_t> last_colour(nv, -1);
            std::vector<uint32_t> last_cell(nv, 0);   // (cluster index in mask order, slot) of the latest toucher: index * 8 + slot
            for (uint32_t l = 0; l < nl; l++) {
                const uint32_t clusters = mask_off[l + 1] - mask_off[l];
                for (uint32_t k = 0; k < kClusterVerts; k++)
                    for (uint32_t i = 0; i < clusters; i++) {
                        const size_t at = plan.vid_off[l] + static_cast<size_t>(k) * clusters + i;
                        const int32_t v = plan.slot_vid[at];
                        if (v < 0) continue;
                        slot_im[at] = h->h_inv_mass[v];
                        if (last_colour[v] >= 0) {
                            const uint32_t back = l - static_cast<uint32_t>(last_colour[v]);
                            uint2& dl = delta[mask_off[l] + i];
                            if (k < 4u) dl.x |= back << (8u * k); else dl.y |= back << (8u * (k - 4u));
                        }
                        last_colour[v] = static_cast<int32_t>(l);
                        last_cell[v] = static_cast<uint32_t>((mask_off[l] + i) * 8u + k);
                    }
            }
            for (uint32_t v = 0; v < nv; v++) if (last_colour[v] >= 0) last_mask[last_cell[v] >> 3] |= static_cast<uint8_t>(1u << (last_cell[v] & 7u));
            // ... and, for a call as ONE launch (nh_call_kernel): how far back a first toucher finds its particle's last toucher of the sweep before
            const bool call = h->nh_untouched == 0 && nl <= 127u;
            std::vector<uint2> delta_first(call ? ncl : 0, make_uint2(0u, 0u));
            if (call)
                for (uint32_t l = 0; l < nl; l++) {
                    const uint32_t clusters = mask_off[l + 1] - mask_off[l];
                    for (uint32_t k = 0; k < kClusterVerts; k++)
                        for (uint32_t i = 0; i < clusters; i++) {
                            const int32_t v = plan.slot_vid[plan.vid_off[l] + static_cast<size_t>(k) * clusters + i];
                            if (v < 0 || !((first_mask[mask_off[l] + i] >> k) & 1u)) continue;
                            const uint32_t back = l + nl - static_cast<uint32_t>(last_colour[v]);
                            uint2& dfv = delta_first[mask_off[l] + i];
                            if (k < 4u) dfv.x |= back << (8u * k); else dfv.y |= back << (8u * (k - 4u));
                        }
                }
            float* d_im; uint2 *d_delta, *d_dfirst = nullptr; uint8_t* d_last; NHSweepColour* d_cols;
            if (call) {
                if ((rc = dev_alloc(h, &d_dfirst, delta_first.size()))) return rc;
                if ((rc = upload(h, d_dfirst, delta_first))) return rc;
            }
            if ((rc = dev_alloc(h, &d_im, slot_im.size()))) return rc;
            if ((rc = upload(h, d_im, slot_im))) return rc;
            if ((rc = dev_alloc(h, &d_delta, delta.size()))) return rc;
            if ((rc = upload(h, d_delta, delta))) return rc;
            if ((rc = dev_alloc(h, &d_last, last_mask.size()))) return rc;
            if ((rc = upload(h, d_last, last_mask))) return rc;
            std::vector<NHSweepColour> cols(nl);
            uint32_t blocks = 0;
            for (uint32_t l = 0; l < nl; l++) {
                cols[l].L = h->cluster_launch[l];
                cols[l].slot_im = d_im + plan.vid_off[l];
                cols[l].delta = d_delta + mask_off[l];
                cols[l].last_mask = d_last + mask_off[l];
                cols[l].delta_first = call ? d_dfirst + mask_off[l] : nullptr;
                cols[l].first_block = blocks;
                blocks += (h->cluster_launch[l].clusters + 63u) / 64u;
            }
            if ((rc = dev_alloc(h, &d_cols, cols.size()))) return rc;
            if ((rc = upload(h, d_cols, cols))) return rc;
            NHSweep& w = h->nh_sweep1;
            w.colours = d_cols; w.ncolours = nl; w.blocks = blocks;
            if ((rc = dev_alloc(h, &w.exchange, nv))) return rc;
            HIPCHK(h, hipMemset(w.exchange, 0, static_cast<size_t>(nv) * sizeof(float4)));
            if ((rc = dev_alloc(h, &w.error, 1))) return rc;
            HIPCHK(h, hipMemset(w.error, 0, sizeof(uint32_t)));
            w.timeout_ms = halo_timeout_ms(h);
            h->nh_one_launch = true;
            h->nh_call = call;
        }
    }
    // Small bodies with a level schedule (the reference's own workload, main.js:26-27): every particle of a body fits one CU's LDS (40 B
    // each) and tetsim_step_n / tetsim_step run a whole call as ONE launch, one workgroup per body (nh_kernels.inc: nh_frame_kernel;
    // tetsim_create_batch: every body of the batch must fit); tetsim_profile keeps the level kernels, whose arithmetic it shares.
    // TETSIM_NH_FRAME=0: never (development A/B).
    // (PRECISE: bodies of up to 12 k tets -- f64 at half rate on ONE CU is throughput-bound beyond that: 157 us per substep at 20 k tets
    // against 117 with one launch per level, 92 against ~115 at 10 k; FAST stays ahead up to the LDS limit: 60 against 87 us at 20 k
    // tets -- tools/nh_size_sweep.py)
    static const bool allow_nh_frame = [] { const char* e = lab_env("TETSIM_NH_FRAME"); return !(e && e[0] == '0'); }();
    if (allow_nh_frame && !clustered && nv > 0 && nt > 0 && nl > 0) {
        std::vector<uint32_t> first_vert = h->batch_first_vert, first_tet = h->batch_first_tet;
        if (first_vert.empty()) { first_vert = {0u, nv}; first_tet = {0u, nt}; }
        const uint32_t bodies = static_cast<uint32_t>(first_vert.size() - 1);
        uint32_t most_v = 0, most_t = 0;
        for (uint32_t b = 0; b < bodies; b++) { most_v = std::max(most_v, first_vert[b + 1] - first_vert[b]); most_t = std::max(most_t, first_tet[b + 1] - first_tet[b]); }
        const uint32_t lds_limit = h->fast ? nh_frame_lds_limit_fast() : nh_frame_lds_limit_precise();
        if (most_v > 0 && static_cast<uint64_t>(most_v) * 40u <= lds_limit && (h->fast || most_t <= 12288u) && bodies <= 65535u) {
            // per level, where every body's tets begin (solve positions; the tets of a level lie body by body)
            std::vector<uint32_t> seg(2ull * nl * bodies);   // (first, end) pairs
            for (uint32_t l = 0; l < nl; l++) {
                uint32_t sp = h->level_off[l];
                for (uint32_t b = 0; b < bodies; b++) {
                    while (sp < h->level_off[l + 1] && body_of[static_cast<uint32_t>(pos_in[sp])] < b) sp++;
                    uint32_t ep = sp;
                    while (ep < h->level_off[l + 1] && body_of[static_cast<uint32_t>(pos_in[ep])] == b) ep++;
                    seg[2ull * (static_cast<size_t>(l) * bodies + b)] = sp;
                    seg[2ull * (static_cast<size_t>(l) * bodies + b) + 1u] = ep;
                    sp = ep;
                }
            }
            uint32_t *dseg = nullptr, *dfv = nullptr;
            if ((rc = dev_alloc(h, &dseg, seg.size()))) return rc;
            if ((rc = upload(h, dseg, seg))) return rc;
            if ((rc = dev_alloc(h, &dfv, first_vert.size()))) return rc;
            if ((rc = upload(h, dfv, first_vert))) return rc;
            h->nh_seg = seg;
            NHFrameLaunch& f = h->nh_frame_launch;
            f.seg = dseg; f.first_vert = dfv; f.levels = nl; f.bodies = bodies; f.max_body_particles = most_v;
            f.block = 512u;   // 128 quads for the narrow levels (kNHQuadLevel), one lane per tet for the wide ones; wider than 512: several trips
            h->nh_frame = true;
            h->info.fused_particle_pass = 4u;
        }
    }
    return 0;
}

}  // namespace tetsim

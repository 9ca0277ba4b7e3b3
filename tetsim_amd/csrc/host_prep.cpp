// host_prep.cpp -- host-side preprocessing for libtetsim_hip.so (no HIP dependency; unit-tested on CPU).
//
// Everything that the reference does once in its constructors (Softbody.js:60-87 initPhysics,
// SoftbodyGPU.js:487-608 initPhysics) plus what a parallel schedule needs on top: Gauss-Seidel
// dependency levels, greedy colouring, the particle->(tet,corner) incidence table and the
// domain-decomposition plan.
//
// Must be compiled with -ffp-contract=off: prep_rest reproduces JS number semantics (f64 arithmetic,
// an f32 rounding at every Float32Array store) and JS never fuses multiply-add.
#include "host_prep.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace tetsim {

namespace {
inline float fround(double x) { return static_cast<float>(x); }

// f32-stored edge matrix (column-major: column k = p_{k+1} - p_0), Softbody.js:69-71
inline void edge_matrix(const float* verts, const int32_t* t, float m[9]) {
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++)
            m[3 * k + c] = fround(static_cast<double>(verts[3 * t[k + 1] + c]) - static_cast<double>(verts[3 * t[0] + c]));
}
// Softbody.js:381-387, term order as written
inline double det3(const float* A) {
    double a11 = A[0], a12 = A[3], a13 = A[6];
    double a21 = A[1], a22 = A[4], a23 = A[7];
    double a31 = A[2], a32 = A[5], a33 = A[8];
    return a11 * a22 * a33 + a12 * a23 * a31 + a13 * a21 * a32 - a13 * a22 * a31 - a12 * a21 * a33 - a11 * a23 * a32;
}
}  // namespace

void prep_rest(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, double density,
               float* inv_mass, float* irp, float* irv) {
    std::fill(inv_mass, inv_mass + nv, 0.0f);
    const uint64_t total = 9ull * nt;
    for (uint32_t e = 0; e < nt; e++) {
        const int32_t* t = &tets[4 * e];
        float* A = &irp[9 * e];
        edge_matrix(verts, t, A);
        const double det = det3(A);
        const double V = det / 6.0;
        if (det == 0.0) {
            // Softbody.js:391-394 clears A[e .. e+8] of the WHOLE array (the index lacks the *9);
            // out-of-range typed-array writes are dropped.
            for (uint64_t i = 0; i < 9; i++)
                if (e + i < total) irp[e + i] = 0.0f;
        } else {
            const double invDet = 1.0 / det;
            const double a11 = A[0], a12 = A[3], a13 = A[6];
            const double a21 = A[1], a22 = A[4], a23 = A[7];
            const double a31 = A[2], a32 = A[5], a33 = A[8];
            A[0] = fround((a22 * a33 - a23 * a32) * invDet);
            A[3] = fround(-(a12 * a33 - a13 * a32) * invDet);
            A[6] = fround((a12 * a23 - a13 * a22) * invDet);
            A[1] = fround(-(a21 * a33 - a23 * a31) * invDet);
            A[4] = fround((a11 * a33 - a13 * a31) * invDet);
            A[7] = fround(-(a11 * a23 - a13 * a21) * invDet);
            A[2] = fround((a21 * a32 - a22 * a31) * invDet);
            A[5] = fround(-(a11 * a32 - a12 * a31) * invDet);
            A[8] = fround((a11 * a22 - a12 * a21) * invDet);
        }
        const double pm = V / 4.0 * density;
        for (int k = 0; k < 4; k++) inv_mass[t[k]] = fround(static_cast<double>(inv_mass[t[k]]) + pm);
        irv[e] = fround(1.0 / V);
    }
    for (uint32_t i = 0; i < nv; i++)
        if (inv_mass[i] != 0.0f) inv_mass[i] = fround(1.0 / static_cast<double>(inv_mass[i]));
}

float pj_inv_rest_volume(const float* verts, const int32_t* tet) {
    float m[9];
    edge_matrix(verts, tet, m);
    return fround(1.0 / (det3(m) / 6.0));
}

uint32_t prep_levels(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* level) {
    std::vector<int32_t> last(nv, -1);  // level of the latest tet touching each vertex
    int32_t top = -1;
    for (uint32_t e = 0; e < nt; e++) {
        const int32_t* t = &tets[4 * e];
        int32_t l = std::max(std::max(last[t[0]], last[t[1]]), std::max(last[t[2]], last[t[3]])) + 1;
        level[e] = l;
        last[t[0]] = last[t[1]] = last[t[2]] = last[t[3]] = l;
        top = std::max(top, l);
    }
    return static_cast<uint32_t>(top + 1);
}

uint32_t prep_colours(const int32_t* tets, uint32_t nt, uint32_t nv, int32_t* colour) {
    // per-vertex bitset of used colours, grown in 64-colour words
    std::vector<std::vector<uint64_t>> used(nv);
    uint32_t ncol = 0;
    for (uint32_t e = 0; e < nt; e++) {
        const int32_t* t = &tets[4 * e];
        size_t words = 0;
        for (int k = 0; k < 4; k++) words = std::max(words, used[t[k]].size());
        int32_t c = -1;
        for (size_t w = 0; w <= words && c < 0; w++) {
            uint64_t m = 0;
            for (int k = 0; k < 4; k++)
                if (w < used[t[k]].size()) m |= used[t[k]][w];
            if (~m) c = static_cast<int32_t>(64 * w + __builtin_ctzll(~m));
        }
        colour[e] = c;
        for (int k = 0; k < 4; k++) {
            auto& u = used[t[k]];
            if (u.size() <= static_cast<size_t>(c / 64)) u.resize(c / 64 + 1, 0);
            u[c / 64] |= 1ull << (c % 64);
        }
        ncol = std::max<uint32_t>(ncol, c + 1);
    }
    return ncol;
}

ClusterPlan prep_clusters(const int32_t* tets, uint32_t nt, uint32_t nv) {
    ClusterPlan P;
    // vertex -> incident tets (CSR, ascending tet id)
    std::vector<uint32_t> voff(nv + 1, 0);
    for (uint64_t i = 0; i < 4ull * nt; i++) voff[tets[i] + 1]++;
    for (uint32_t v = 0; v < nv; v++) voff[v + 1] += voff[v];
    std::vector<uint32_t> vtet(4ull * nt), fill(voff.begin(), voff.end() - 1);
    for (uint32_t e = 0; e < nt; e++)
        for (int k = 0; k < 4; k++) vtet[fill[tets[4 * e + k]]++] = e;

    // 1. greedy clusters: seed = lowest unassigned tet; grow by the unassigned neighbour that adds the fewest new vertices
    //    (ties: closest tet id to the seed, which keeps the cells of a cell-major lattice together)
    std::vector<int32_t> cluster_of(nt, -1);
    struct Cluster { uint32_t n = 0, nvert = 0; uint32_t tet[kClusterTets]; int32_t vert[kClusterVerts]; };
    std::vector<Cluster> clusters;
    std::vector<uint32_t> cand;
    std::vector<int32_t> stamp(nt, -1);    // cluster id for which `covered` is valid
    std::vector<uint8_t> covered(nt, 0);   // corners of a candidate tet that are cluster vertices already
    for (uint32_t seed = 0; seed < nt; seed++) {
        if (cluster_of[seed] >= 0) continue;
        Cluster c;
        const int32_t id = static_cast<int32_t>(clusters.size());
        cand.clear();
        auto add = [&](uint32_t e) {
            c.tet[c.n++] = e;
            cluster_of[e] = id;
            for (int k = 0; k < 4; k++) {
                const int32_t v = tets[4 * e + k];
                bool have = false;
                for (uint32_t j = 0; j < c.nvert; j++) have |= c.vert[j] == v;
                if (have) continue;
                c.vert[c.nvert++] = v;
                for (uint32_t q = voff[v]; q < voff[v + 1]; q++) {  // (a tet listing v twice is visited twice: two corners covered)
                    const uint32_t t = vtet[q];
                    if (cluster_of[t] >= 0) continue;
                    if (stamp[t] != id) { stamp[t] = id; covered[t] = 0; cand.push_back(t); }
                    covered[t]++;
                }
            }
        };
        add(seed);  // (at most 4 distinct vertices: always fits)
        while (c.n < kClusterTets) {
            uint32_t best = nt, best_new = 5;
            uint64_t best_dist = ~0ull;
            for (uint32_t e : cand) {
                if (cluster_of[e] >= 0) continue;
                const uint32_t fresh = 4u - covered[e];
                if (c.nvert + fresh > kClusterVerts) continue;
                const uint64_t dist = e > seed ? e - seed : seed - e;
                if (fresh < best_new || (fresh == best_new && dist < best_dist)) { best = e; best_new = fresh; best_dist = dist; }
            }
            if (best == nt) break;
            add(best);
        }
        std::sort(c.tet, c.tet + c.n);  // inside a cluster: the caller's order
        clusters.push_back(c);
    }
    const uint32_t nc = static_cast<uint32_t>(clusters.size());
    P.num_clusters = nc;

    // 2. greedy colouring of the clusters (conflict = a shared vertex), per-vertex bitsets of used colours
    std::vector<std::vector<uint64_t>> used(nv);
    std::vector<uint32_t> colour(nc);
    uint32_t ncol = 0;
    for (uint32_t ci = 0; ci < nc; ci++) {
        const Cluster& c = clusters[ci];
        size_t words = 0;
        for (uint32_t j = 0; j < c.nvert; j++) words = std::max(words, used[c.vert[j]].size());
        int32_t col = -1;
        for (size_t w = 0; w <= words && col < 0; w++) {
            uint64_t m = 0;
            for (uint32_t j = 0; j < c.nvert; j++)
                if (w < used[c.vert[j]].size()) m |= used[c.vert[j]][w];
            if (~m) col = static_cast<int32_t>(64 * w + __builtin_ctzll(~m));
        }
        colour[ci] = static_cast<uint32_t>(col);
        for (uint32_t j = 0; j < c.nvert; j++) {
            auto& u = used[c.vert[j]];
            if (u.size() <= static_cast<size_t>(col / 64)) u.resize(col / 64 + 1, 0);
            u[col / 64] |= 1ull << (col % 64);
        }
        ncol = std::max<uint32_t>(ncol, col + 1);
    }

    // 3. per colour: clusters by size, largest first (so step j is the lane range [0, count_j)); layout
    std::vector<uint32_t> by(nc);
    for (uint32_t i = 0; i < nc; i++) by[i] = i;
    std::stable_sort(by.begin(), by.end(), [&](uint32_t a, uint32_t b) {
        if (colour[a] != colour[b]) return colour[a] < colour[b];
        return clusters[a].n > clusters[b].n;
    });
    P.pre.reserve(nt);
    P.exec_pos.assign(nt, 0);
    P.corner_slots.assign(nt, 0);
    P.launch_off.assign(1, 0);
    P.step_off.assign(1, 0);
    P.vid_off.assign(1, 0);
    uint32_t storage = 0;
    for (uint32_t b0 = 0; b0 < nc;) {
        uint32_t b1 = b0;
        while (b1 < nc && colour[by[b1]] == colour[by[b0]]) b1++;
        const uint32_t n0 = b1 - b0;
        // sequential order: cluster after cluster
        std::vector<uint32_t> seq_first(n0);
        for (uint32_t i = 0; i < n0; i++) {
            const Cluster& c = clusters[by[b0 + i]];
            seq_first[i] = static_cast<uint32_t>(P.pre.size());
            for (uint32_t j = 0; j < c.n; j++) P.pre.push_back(static_cast<int32_t>(c.tet[j]));
        }
        // storage order: step after step
        for (uint32_t j = 0; j < clusters[by[b0]].n; j++) {
            uint32_t count = 0;
            while (count < n0 && clusters[by[b0 + count]].n > j) count++;
            P.step_first.push_back(storage);
            P.step_count.push_back(count);
            for (uint32_t i = 0; i < count; i++) {
                const Cluster& c = clusters[by[b0 + i]];
                uint32_t packed = 0;
                for (int k = 0; k < 4; k++) {
                    uint32_t slot = 0;
                    while (c.vert[slot] != tets[4 * c.tet[j] + k]) slot++;
                    packed |= slot << (8 * k);
                }
                P.exec_pos[storage + i] = seq_first[i] + j;
                P.corner_slots[storage + i] = packed;
            }
            storage += count;
        }
        const size_t base = P.slot_vid.size();
        P.slot_vid.resize(base + static_cast<size_t>(kClusterVerts) * n0, -1);
        for (uint32_t i = 0; i < n0; i++) {
            const Cluster& c = clusters[by[b0 + i]];
            for (uint32_t k = 0; k < c.nvert; k++) P.slot_vid[base + static_cast<size_t>(k) * n0 + i] = c.vert[k];
        }
        P.launch_off.push_back(storage);
        P.step_off.push_back(static_cast<uint32_t>(P.step_first.size()));
        P.vid_off.push_back(static_cast<uint32_t>(P.slot_vid.size()));
        b0 = b1;
    }
    return P;
}

Incidence build_incidence(const int32_t* tets, uint32_t nt, uint32_t nv, bool ref_quirk, bool ref_cap) {
    Incidence inc;
    std::vector<uint32_t> count(nv, 0);
    for (uint64_t i = 0; i < 4ull * nt; i++) count[tets[i]]++;
    // SoftbodyGPU.js:568: a slot holding the encoded value 0 (tet 0, corner 0) passes the `<= 0.0`
    // "empty" test, so the particle's NEXT incidence overwrites it: that contribution is lost iff the
    // particle has another incident tet.
    const int32_t quirk_vertex = (ref_quirk && nt > 0 && count[tets[0]] >= 2) ? tets[0] : -1;
    inc.offset.assign(nv + 1, 0);
    for (uint32_t v = 0; v < nv; v++) {
        uint32_t c = count[v];
        if (static_cast<int32_t>(v) == quirk_vertex) c--;
        if (ref_cap && c > kRefSlots) c = kRefSlots;  // valence beyond 36 is silently dropped
        inc.offset[v + 1] = inc.offset[v] + c;
        inc.max_valence = std::max(inc.max_valence, c);
    }
    inc.slot.assign(inc.offset[nv], -1);
    std::vector<uint32_t> fill(nv, 0);
    uint64_t kept = 0;
    for (uint32_t e = 0; e < nt; e++)
        for (int k = 0; k < 4; k++) {
            const int32_t v = tets[4 * e + k];
            if (v == quirk_vertex && e == 0 && k == 0) continue;
            const uint32_t cap = inc.offset[v + 1] - inc.offset[v];
            if (fill[v] >= cap) continue;
            inc.slot[inc.offset[v] + fill[v]++] = static_cast<int32_t>(4 * e + k);
            kept++;
        }
    inc.dropped = static_cast<uint32_t>(4ull * nt - kept);
    return inc;
}

uint32_t prep_slot_table(const int32_t* tets, uint32_t nt, uint32_t nv, bool ref_quirk, int32_t* slots) {
    std::fill(slots, slots + static_cast<size_t>(nv) * kRefSlots, -1);
    // direct emulation of the reference's fill loop (kept independent of build_incidence on purpose:
    // the unit tests cross-check the two).
    uint32_t dropped = 0;
    for (uint32_t e = 0; e < nt; e++)
        for (int k = 0; k < 4; k++) {
            int32_t* row = &slots[static_cast<size_t>(tets[4 * e + k]) * kRefSlots];
            bool placed = false;
            for (int s = 0; s < kRefSlots; s++) {
                const bool empty = ref_quirk ? (row[s] <= 0) : (row[s] < 0);
                if (empty) {
                    if (row[s] == 0) dropped++;  // overwriting a live 0 entry
                    row[s] = static_cast<int32_t>(4 * e + k);
                    placed = true;
                    break;
                }
            }
            if (!placed) dropped++;
        }
    return dropped;
}

namespace {
inline uint32_t spread3(uint32_t x) {  // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
}  // namespace

namespace {
struct Quantiser {
    float lo[3], inv;
    Quantiser(const float* xyz, uint32_t n) {
        float hi[3] = {-1e30f, -1e30f, -1e30f};
        lo[0] = lo[1] = lo[2] = 1e30f;
        for (uint32_t v = 0; v < n; v++)
            for (int c = 0; c < 3; c++) { lo[c] = std::min(lo[c], xyz[3 * v + c]); hi[c] = std::max(hi[c], xyz[3 * v + c]); }
        float ext = 1e-30f;
        for (int c = 0; c < 3; c++) ext = std::max(ext, hi[c] - lo[c]);
        inv = 1024.0f / ext;
    }
    uint32_t code(float x, float y, float z) const {
        const float p[3] = {x, y, z};
        uint32_t c = 0;
        for (int a = 0; a < 3; a++) {
            const uint32_t qv = static_cast<uint32_t>(std::min(1023.0f, std::max(0.0f, (p[a] - lo[a]) * inv)));
            c |= spread3(qv) << a;
        }
        return c;
    }
};
}  // namespace

std::vector<uint32_t> morton_vertex_order(const float* xyz, uint32_t n, uint32_t first, uint32_t count) {
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    if (count < 2) return order;
    const Quantiser Q(xyz, n);
    std::vector<uint64_t> key(count);
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t v = first + i;
        key[i] = (static_cast<uint64_t>(Q.code(xyz[3 * v], xyz[3 * v + 1], xyz[3 * v + 2])) << 32) | v;
    }
    std::sort(key.begin(), key.end());
    for (uint32_t i = 0; i < count; i++) order[first + i] = static_cast<uint32_t>(key[i] & 0xffffffffu);
    return order;
}

void build_blocks(const float* verts, const int32_t* tets, uint32_t nt, uint32_t nv, uint32_t nv_sum,
                  const Incidence& inc, BlockPlan* out, const uint32_t* body_first_tet, const uint32_t* body_first_vert, uint32_t bodies,
                  uint32_t nv_boundary, const uint8_t* tet_class, uint32_t nv_owned, uint32_t tile) {
    BlockPlan& B = *out;
    B = BlockPlan();
    B.tile = tile;
    if (nv_owned > nv_sum) nv_owned = nv_sum;
    const uint32_t kMaxTets = tile, kMaxVerts = tile;   // = tets (and LDS particle slots) of one workgroup: pj_blocked.hip kTile, or pj_quad.hip kQuadTile
    const uint32_t one_t[2] = {0u, nt}, one_v[2] = {0u, nv};
    if (!body_first_tet || !body_first_vert || bodies == 0) { body_first_tet = one_t; body_first_vert = one_v; bodies = 1; }
    // 1. Morton order of rest centroids (quantised to 10 bits per axis over the body's bounding box), body after body
    std::vector<uint64_t> key(nt);
    std::vector<uint32_t> body_of_pos(nt);   // body of the tet at each sorted position
    B.tet_perm.resize(nt);
    for (uint32_t b = 0; b < bodies; b++) {
        const uint32_t tb = body_first_tet[b], te = body_first_tet[b + 1], vb = body_first_vert[b], ve = body_first_vert[b + 1];
        const Quantiser Q(verts + 3ull * vb, ve - vb);
        for (uint32_t e = tb; e < te; e++) {
            float m[3] = {0.0f, 0.0f, 0.0f};
            for (int k = 0; k < 4; k++)
                for (int c = 0; c < 3; c++) m[c] += verts[3 * tets[4 * e + k] + c];
            key[e] = (static_cast<uint64_t>(Q.code(0.25f * m[0], 0.25f * m[1], 0.25f * m[2])) << 32) | (e - tb);  // ties keep the caller's order
            if (tet_class) key[e] |= static_cast<uint64_t>(tet_class[e] & 3u) << 62;   // classes one after the other (the code uses bits 32..61)
        }
        std::sort(key.begin() + tb, key.begin() + te);
        for (uint32_t i = tb; i < te; i++) { B.tet_perm[i] = static_cast<int32_t>(tb + (key[i] & 0xffffffffu)); body_of_pos[i] = b; }
    }

    // which (tet,corner) contributions are live (the incidence table may drop some: reference quirk / cap)
    std::vector<uint8_t> live(4ull * nt, 0);
    for (int32_t enc : inc.slot) live[enc] = 1;

    // 2. greedy tiling along the curve: tile = a run [begin, end) of the sorted tets
    std::vector<int32_t> slot_of(nv, -1);
    std::vector<int32_t> touched;
    struct Run { uint32_t begin, end; bool ghost; uint8_t cls; };
    auto cls_of = [&](uint32_t sorted_pos) -> uint8_t { return tet_class ? tet_class[B.tet_perm[sorted_pos]] : 0; };
    std::vector<Run> runs;
    {
        uint32_t i = 0;
        while (i < nt) {
            const uint32_t t0 = i;
            touched.clear();
            bool ghost = false;
            while (i < nt && i - t0 < kMaxTets && body_of_pos[i] == body_of_pos[t0] && cls_of(i) == cls_of(t0)) {   // a tile never spans two bodies, nor two tet classes
                const int32_t* t = &tets[4 * B.tet_perm[i]];
                uint32_t fresh = 0;
                for (int k = 0; k < 4; k++) {
                    bool seen = slot_of[t[k]] >= 0;
                    for (int j = 0; j < k && !seen; j++) seen = t[j] == t[k];
                    if (!seen) fresh++;
                }
                if (touched.size() + fresh > kMaxVerts) break;
                for (int k = 0; k < 4; k++)
                    if (slot_of[t[k]] < 0) {
                        slot_of[t[k]] = 0; touched.push_back(t[k]);
                        ghost |= static_cast<uint32_t>(t[k]) >= nv_owned || static_cast<uint32_t>(t[k]) < nv_boundary;   // halo-side
                    }
                i++;
            }
            for (int32_t v : touched) slot_of[v] = -1;
            runs.push_back({t0, i, ghost, cls_of(t0)});
        }
    }
    // tiles that touch a ghost or a boundary particle go last: a partitioned body solves the others while the halo is in flight
    std::stable_sort(runs.begin(), runs.end(), [](const Run& a, const Run& b) { return a.cls != b.cls ? a.cls < b.cls : a.ghost < b.ghost; });
    {
        std::vector<int32_t> perm2(nt);
        uint32_t o = 0;
        for (Run& r : runs) {
            std::copy(B.tet_perm.begin() + r.begin, B.tet_perm.begin() + r.end, perm2.begin() + o);
            const uint32_t len = r.end - r.begin;
            r.begin = o; r.end = o + len;
            o += len;
        }
        B.tet_perm.swap(perm2);
    }

    // 3. emit per-tile tables
    B.tet_lidx.resize(4ull * nt);
    B.lc_ent.resize(4ull * nt);
    B.blk_tet_off.push_back(0);
    B.blk_vert_off.push_back(0);
    std::vector<std::vector<uint32_t>> vert_partials(nv_sum);
    for (const Run& r : runs) {
        const uint32_t t0 = r.begin, i = r.end;
        if (!r.ghost && r.cls == 0) B.num_interior_blocks++;
        if (r.cls <= 1) B.num_first_blocks++;
        touched.clear();
        for (uint32_t j = t0; j < i; j++)
            for (int k = 0; k < 4; k++) {
                const int32_t v = tets[4 * B.tet_perm[j] + k];
                if (slot_of[v] < 0) { slot_of[v] = 0; touched.push_back(v); }
            }
        // LDS slots in ascending particle id: with Morton-numbered particles a tile's slots are a few contiguous
        // runs of the position array, and a particle's partial sums sit next to its neighbours'
        std::sort(touched.begin(), touched.end());
        for (size_t u = 0; u < touched.size(); u++) slot_of[touched[u]] = static_cast<int32_t>(u);
        for (uint32_t j = t0; j < i; j++)
            for (int k = 0; k < 4; k++) B.tet_lidx[4ull * j + k] = static_cast<uint8_t>(slot_of[tets[4 * B.tet_perm[j] + k]]);
        const uint32_t ntb = i - t0, nu = static_cast<uint32_t>(touched.size());
        const uint32_t v0 = B.blk_vert_off.back();
        // per-slot entry lists: counting sort of the live (tetLocal, corner) pairs, tet order within a slot
        std::vector<uint32_t> cnt(nu + 1, 0);
        for (uint32_t j = 0; j < ntb; j++)
            for (int k = 0; k < 4; k++)
                if (live[4ull * B.tet_perm[t0 + j] + k]) cnt[B.tet_lidx[4ull * (t0 + j) + k] + 1]++;
        for (uint32_t u = 0; u < nu; u++) cnt[u + 1] += cnt[u];
        for (uint32_t u = 0; u < nu; u++) B.lc_range.push_back(cnt[u] | (cnt[u + 1] << 16));
        std::vector<uint32_t> fillp(cnt.begin(), cnt.end() - 1);
        for (uint32_t j = 0; j < ntb; j++)
            for (int k = 0; k < 4; k++)
                if (live[4ull * B.tet_perm[t0 + j] + k]) {
                    const uint32_t u = B.tet_lidx[4ull * (t0 + j) + k];
                    B.lc_ent[4ull * t0 + fillp[u]++] = static_cast<uint16_t>(kMaxTets * k + j);  // word offset into the [corner][tet] goal planes
                }
        for (uint32_t u = 0; u < nu; u++) {
            const int32_t v = touched[u];
            B.blk_verts.push_back(v);
            if (static_cast<uint32_t>(v) < nv_sum && cnt[u + 1] > cnt[u]) vert_partials[v].push_back(v0 + u);
            slot_of[v] = -1;
        }
        B.blk_tet_off.push_back(i);
        B.blk_vert_off.push_back(v0 + nu);
        B.max_tile_verts = std::max(B.max_tile_verts, nu);
    }
    B.num_blocks = static_cast<uint32_t>(B.blk_tet_off.size() - 1);
    B.vp_off.assign(nv_sum + 1, 0);
    for (uint32_t v = 0; v < nv_sum; v++) {
        B.vp_off[v + 1] = B.vp_off[v] + static_cast<uint32_t>(vert_partials[v].size());
        B.max_partials = std::max<uint32_t>(B.max_partials, static_cast<uint32_t>(vert_partials[v].size()));
    }
    B.vp_idx.reserve(B.vp_off[nv_sum]);
    for (uint32_t v = 0; v < nv_sum; v++) B.vp_idx.insert(B.vp_idx.end(), vert_partials[v].begin(), vert_partials[v].end());
    B.nv_pad = (nv_sum + 63u) & ~63u;
    B.vp_ell.assign(static_cast<size_t>(std::max(B.max_partials, 1u)) * B.nv_pad, 0xffffffffu);
    for (uint32_t v = 0; v < nv_sum; v++)
        for (size_t j = 0; j < vert_partials[v].size(); j++) B.vp_ell[j * B.nv_pad + v] = vert_partials[v][j];
    // fused particle pass: every tile slot carries its particle's list of partial sums; the list's first slot is the owner
    const uint32_t ns = static_cast<uint32_t>(B.blk_verts.size());
    B.ns_pad = (ns + 63u) & ~63u;
    B.slot_src.assign(static_cast<size_t>(std::max(B.max_partials, 1u)) * B.ns_pad, 0xffffffffu);
    B.blk_maxsrc.assign(B.num_blocks, 0);
    for (uint32_t v = 0; v < nv_sum; v++)
        if (vert_partials[v].empty()) B.every_owned_particle_has_a_partial = false;
    for (uint32_t b = 0; b < B.num_blocks; b++)
        for (uint32_t g = B.blk_vert_off[b]; g < B.blk_vert_off[b + 1]; g++) {
            const uint32_t v = static_cast<uint32_t>(B.blk_verts[g]);
            if (v >= nv_sum) continue;   // ghosts are not integrated here
            const std::vector<uint32_t>& list = vert_partials[v];
            for (size_t j = 0; j < list.size(); j++) B.slot_src[j * B.ns_pad + g] = list[j];
            B.blk_maxsrc[b] = std::max<uint32_t>(B.blk_maxsrc[b], static_cast<uint32_t>(list.size()));
            if (!list.empty() && list[0] == g) B.lc_range[g] |= 1u << 15;
        }
}

std::string validate_mesh(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, bool forbid_repeats) {
    if ((nv && !verts) || (nt && !tets)) return "null mesh pointer";
    if (nv == 0) return "mesh has no vertices";
    if (nv > 0x3fffffffu || nt > 0x1fffffffu) return "mesh too large for 32-bit slot encoding";
    for (uint32_t e = 0; e < nt; e++) {
        const int32_t* t = &tets[4 * e];
        for (int k = 0; k < 4; k++)
            if (t[k] < 0 || static_cast<uint32_t>(t[k]) >= nv)
                return "tet " + std::to_string(e) + " references vertex " + std::to_string(t[k]) + " outside [0," + std::to_string(nv) + ")";
        if (forbid_repeats && (t[0] == t[1] || t[0] == t[2] || t[0] == t[3] || t[1] == t[2] || t[1] == t[3] || t[2] == t[3]))
            return "tet " + std::to_string(e) + " repeats a vertex (unsupported by the parallel Gauss-Seidel schedule)";
    }
    return "";
}

std::string build_partition(const int32_t* tets, uint32_t nt, uint32_t nv, int part_count, int part_index,
                            const int32_t* vert_owner, Partition* out, int depth) {
    Partition& P = *out;
    P = Partition();
    P.nv_global = nv;
    P.nt_global = nt;
    P.part_count = part_count;
    P.part_index = part_index;
    P.depth = depth;
    if (part_count < 1 || part_index < 0 || part_index >= part_count) return "bad part_index/part_count";
    if (depth != 1 && depth != 2) return "ghost depth must be 1 or 2";

    std::vector<int32_t> owner(nv);
    if (vert_owner) {
        for (uint32_t v = 0; v < nv; v++) {
            if (vert_owner[v] < 0 || vert_owner[v] >= part_count) return "vert_owner out of range";
            owner[v] = vert_owner[v];
        }
    } else {
        const std::string perr = prep_partition(nullptr, nv, tets, nt, part_count, owner.data());
        if (!perr.empty()) return perr;
    }
    const int me = part_index;

    // Which ranks read a particle as a ghost.  near1[v]: ranks r != owner(v) that own a particle sharing a tet with v -- v is in
    // r's FIRST ghost layer.  near2[v] (depth 2): ranks r that neither own v nor have it in their first layer, but have a
    // first-layer ghost sharing a tet with v -- v is in r's SECOND layer.  (Small sorted vectors; empty for interior particles.)
    std::vector<std::vector<int32_t>> near1(nv), near2(depth == 2 ? nv : 0);
    auto add = [](std::vector<int32_t>& s, int32_t r) { if (std::find(s.begin(), s.end(), r) == s.end()) s.push_back(r); };
    for (uint32_t e = 0; e < nt; e++) {
        const int32_t* t = &tets[4 * e];
        const int32_t o[4] = {owner[t[0]], owner[t[1]], owner[t[2]], owner[t[3]]};
        if (o[0] == o[1] && o[1] == o[2] && o[2] == o[3]) continue;
        for (int a = 0; a < 4; a++)
            for (int b = 0; b < 4; b++)
                if (o[b] != o[a]) add(near1[t[a]], o[b]);
    }
    if (depth == 2)
        for (uint32_t e = 0; e < nt; e++) {
            const int32_t* t = &tets[4 * e];
            for (int a = 0; a < 4; a++)
                for (int32_t r : near1[t[a]])            // t[a] is a first-layer ghost of r ...
                    for (int b = 0; b < 4; b++) {        // ... so every other corner r does not already hold is in its second layer
                        const int32_t v = t[b];
                        if (owner[v] != r && std::find(near1[v].begin(), near1[v].end(), r) == near1[v].end()) add(near2[v], r);
                    }
        }
    auto reads = [&](const std::vector<std::vector<int32_t>>& near, uint32_t v, int32_t r) {
        return !near.empty() && std::find(near[v].begin(), near[v].end(), r) != near[v].end();
    };

    // local tets (ascending global id): every tet with an owned corner, and -- depth 2 -- every tet with a first-layer ghost corner
    for (uint32_t e = 0; e < nt; e++) {
        const int32_t* t = &tets[4 * e];
        bool mine = false, first_layer = false;
        int lowest = part_count;
        for (int k = 0; k < 4; k++) {
            if (owner[t[k]] == me) mine = true;
            else if (reads(near1, static_cast<uint32_t>(t[k]), me)) first_layer = true;
            lowest = std::min(lowest, owner[t[k]]);
        }
        if (!mine && !(depth == 2 && first_layer)) continue;
        P.local_to_global_tet.push_back(static_cast<int32_t>(e));
        P.tet_layer.push_back(mine ? 0 : 1);
        if (mine && lowest == me) P.owned_tets++;
    }

    // local vertex numbering: [owned boundary | owned interior | first-layer ghosts by (owner, id) | second-layer ghosts by (owner, id)];
    // boundary = an owned particle some other rank reads (in either layer)
    std::vector<int32_t> g2l(nv, -1);
    auto is_boundary = [&](uint32_t v) { return !near1[v].empty() || (depth == 2 && !near2[v].empty()); };
    for (uint32_t v = 0; v < nv; v++)
        if (owner[v] == me && is_boundary(v)) { g2l[v] = static_cast<int32_t>(P.local_to_global_vert.size()); P.local_to_global_vert.push_back(v); }
    P.n_boundary = static_cast<uint32_t>(P.local_to_global_vert.size());
    for (uint32_t v = 0; v < nv; v++)
        if (owner[v] == me && !is_boundary(v)) { g2l[v] = static_cast<int32_t>(P.local_to_global_vert.size()); P.local_to_global_vert.push_back(v); }
    P.n_owned = static_cast<uint32_t>(P.local_to_global_vert.size());
    std::vector<int32_t> ghosts1, ghosts2;
    for (uint32_t v = 0; v < nv; v++) {
        if (owner[v] == me) continue;
        if (reads(near1, v, me)) ghosts1.push_back(static_cast<int32_t>(v));
        else if (reads(near2, v, me)) ghosts2.push_back(static_cast<int32_t>(v));
    }
    auto by_owner = [&](int32_t a, int32_t b) { return owner[a] < owner[b]; };
    std::stable_sort(ghosts1.begin(), ghosts1.end(), by_owner);
    std::stable_sort(ghosts2.begin(), ghosts2.end(), by_owner);
    for (int32_t v : ghosts1) { g2l[v] = static_cast<int32_t>(P.local_to_global_vert.size()); P.local_to_global_vert.push_back(v); }
    P.n_ghost1 = static_cast<uint32_t>(ghosts1.size());
    for (int32_t v : ghosts2) { g2l[v] = static_cast<int32_t>(P.local_to_global_vert.size()); P.local_to_global_vert.push_back(v); }

    P.local_tets.resize(4 * P.local_to_global_tet.size());
    for (size_t i = 0; i < P.local_to_global_tet.size(); i++)
        for (int k = 0; k < 4; k++) {
            const int32_t l = g2l[tets[4 * P.local_to_global_tet[i] + k]];
            if (l < 0) return "internal error in the partition plan: a local tet has a corner outside the ghost layers";
            P.local_tets[4 * i + k] = l;
        }

    // halo lists, per neighbour: received = my ghosts it owns (one contiguous range per layer); sent = my owned particles it reads
    // (per layer, ascending global id) -- exactly its ghosts owned by me, in its order
    size_t g1 = 0, g2 = 0;
    for (int r = 0; r < part_count; r++) {
        if (r == me) continue;
        Partition::Neighbour nb;
        nb.rank = r;
        nb.recv_start = P.n_owned + static_cast<uint32_t>(g1);
        while (g1 < ghosts1.size() && owner[ghosts1[g1]] == r) { nb.recv_global.push_back(ghosts1[g1]); g1++; }
        nb.recv_count = static_cast<uint32_t>(nb.recv_global.size());
        nb.recv2_start = P.n_owned + P.n_ghost1 + static_cast<uint32_t>(g2);
        while (g2 < ghosts2.size() && owner[ghosts2[g2]] == r) { nb.recv2_global.push_back(ghosts2[g2]); g2++; }
        nb.recv2_count = static_cast<uint32_t>(nb.recv2_global.size());
        for (uint32_t v = 0; v < nv; v++) {
            if (owner[v] != me) continue;
            if (reads(near1, v, r)) { nb.send_global.push_back(static_cast<int32_t>(v)); nb.send_local.push_back(g2l[v]); }
            else if (reads(near2, v, r)) { nb.send2_global.push_back(static_cast<int32_t>(v)); nb.send2_local.push_back(g2l[v]); }
        }
        nb.send_contiguous = !nb.send_local.empty();
        for (size_t i = 1; i < nb.send_local.size(); i++)
            if (nb.send_local[i] != nb.send_local[i - 1] + 1) { nb.send_contiguous = false; break; }
        if (nb.recv_count || nb.recv2_count || !nb.send_local.empty() || !nb.send2_local.empty()) P.neigh.push_back(std::move(nb));
    }
    return "";
}

}  // namespace tetsim

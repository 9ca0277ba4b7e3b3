// partitioner.cpp -- the built-in vertex partitioner for general meshes (host only, no HIP dependency).
//
// SURVEY.md 8(e) "General meshes (Dragon-class): host-side graph partition (greedy BFS/METIS-like, build's own) with the same
// ghost-tet rule".  What a cut must respect is the coupling of the Jacobi average: a particle reads the goals of every tet
// incident to it (/root/reference/src/SoftbodyGPU.js:563-577 builds those lists, :306-319 consumes them), so a partition solves
// every tet that touches a particle it owns and every particle of such a tet it does not own is a ghost that crosses the wire
// each substep (host_prep.cpp: build_partition).  The objective is therefore: equal work per part, few ghosts.
//
// Algorithm (deterministic, O(log(parts) * tets)):
//   1. recursive bisection of the vertex set, weights w(v) = 1 + valence(v) (a part's tets ~ the corners it owns / 4).  A subset
//      is split at the weighted median of a scalar key; several keys are tried and the one that cuts the fewest tets wins:
//        * d_A - d_B, d_A, d_B: breadth-first distances through the subset's own tets from the two ends A, B of a pseudo-diameter
//          (A = farthest from the subset's first vertex, B = farthest from A) -- topology only, always available: level sets of
//          d_A - d_B are the "planes" half-way between the two ends of an elongated body;
//        * x, y, z: when the caller hands coordinates (tetsim_prep_partition) -- recursive coordinate bisection's candidates; on
//          a lattice they give the planar cuts.
//   2. k-way boundary refinement, up to kRefinePasses sweeps in vertex order: a vertex moves to the neighbouring part that holds
//      more of its tet-mates than its own part does (strictly: every move lowers the number of (corner, corner) pairs of a tet
//      with different owners, so the sweeps terminate), as long as both parts stay within +-3% of the mean weight.
// tetsim_create / tetsim_plan_create with part_count > 1 and vert_owner == NULL use prep_partition WITHOUT coordinates (the plan
// entry points have none, and a plan must equal what tetsim_create builds); a host that wants the geometric candidates calls
// tetsim_prep_partition itself and passes the result to both.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "host_prep.h"

namespace tetsim {
namespace {

constexpr int kRefinePasses = 4;

struct Graph {
    const int32_t* tets;
    uint32_t nt, nv;
    std::vector<uint32_t> off;   // [nv + 1] incident tets of every vertex, CSR, ascending tet id
    std::vector<uint32_t> inc;
    std::vector<uint32_t> weight;
};

Graph make_graph(const int32_t* tets, uint32_t nt, uint32_t nv) {
    Graph g{tets, nt, nv, std::vector<uint32_t>(nv + 1, 0), std::vector<uint32_t>(4ull * nt), std::vector<uint32_t>(nv, 1)};
    for (uint64_t i = 0; i < 4ull * nt; i++) g.off[tets[i] + 1]++;
    for (uint32_t v = 0; v < nv; v++) { g.weight[v] += g.off[v + 1]; g.off[v + 1] += g.off[v]; }
    std::vector<uint32_t> fill(g.off.begin(), g.off.end() - 1);
    for (uint32_t e = 0; e < nt; e++)
        for (int k = 0; k < 4; k++) g.inc[fill[tets[4 * e + k]]++] = e;
    return g;
}

// Breadth-first distances inside the subset `set_id` (member[v] == set_id), starting at `start`; components the front never
// reaches are appended one after the other (restart at the lowest unvisited vertex of `verts`, distances continuing), so that
// every vertex of the subset gets a finite key.  Returns the last vertex reached in the FIRST component.
uint32_t bfs(const Graph& g, const std::vector<uint32_t>& verts, const std::vector<int32_t>& member, int32_t set_id, uint32_t start,
             std::vector<uint32_t>& dist, std::vector<uint32_t>& tet_stamp, uint32_t& stamp, std::vector<uint32_t>& queue) {
    constexpr uint32_t kUnseen = 0xffffffffu;
    for (uint32_t v : verts) dist[v] = kUnseen;
    queue.clear();
    uint32_t last_first = start, next_seed = 0, base = 0;
    bool first = true;
    for (;;) {
        stamp++;
        size_t head = queue.size();
        dist[start] = base;
        queue.push_back(start);
        while (head < queue.size()) {
            const uint32_t v = queue[head++];
            for (uint32_t i = g.off[v]; i < g.off[v + 1]; i++) {
                const uint32_t e = g.inc[i];
                if (tet_stamp[e] == stamp) continue;   // all four corners were pushed when the tet was first met
                tet_stamp[e] = stamp;
                for (int k = 0; k < 4; k++) {
                    const uint32_t u = static_cast<uint32_t>(g.tets[4 * e + k]);
                    if (member[u] == set_id && dist[u] == kUnseen) { dist[u] = dist[v] + 1; queue.push_back(u); }
                }
            }
        }
        if (first) { last_first = queue.back(); first = false; }
        base = dist[queue.back()] + 1;
        while (next_seed < verts.size() && dist[verts[next_seed]] != kUnseen) next_seed++;
        if (next_seed == verts.size()) break;
        start = verts[next_seed];
    }
    return last_first;
}

struct Bisector {
    const Graph& g;
    const float* xyz;                 // may be null
    std::vector<int32_t>& owner;      // result
    std::vector<int32_t> member;      // subset id of every vertex during the recursion
    std::vector<uint32_t> dA, dB, tet_stamp, queue;
    std::vector<uint8_t> side;
    uint32_t stamp = 0;
    int32_t next_set = 1;

    Bisector(const Graph& gr, const float* coords, std::vector<int32_t>& out)
        : g(gr), xyz(coords), owner(out), member(gr.nv, 0), dA(gr.nv), dB(gr.nv), tet_stamp(gr.nt, 0), side(gr.nv, 0) {}

    // tets of the subset with corners on both sides (corners outside the subset do not count)
    uint64_t cut_tets(const std::vector<uint32_t>& verts, int32_t set_id) {
        stamp++;
        uint64_t cut = 0;
        for (uint32_t v : verts)
            for (uint32_t i = g.off[v]; i < g.off[v + 1]; i++) {
                const uint32_t e = g.inc[i];
                if (tet_stamp[e] == stamp) continue;
                tet_stamp[e] = stamp;
                bool s0 = false, s1 = false;
                for (int k = 0; k < 4; k++) {
                    const uint32_t u = static_cast<uint32_t>(g.tets[4 * e + k]);
                    if (member[u] == set_id) (side[u] ? s1 : s0) = true;
                }
                cut += s0 && s1;
            }
        return cut;
    }

    void split(std::vector<uint32_t> verts, int parts, int first_part) {   // verts ascending
        if (parts == 1 || verts.empty()) {
            for (uint32_t v : verts) owner[v] = first_part;
            return;
        }
        const int32_t set_id = next_set++;
        uint64_t total = 0;
        for (uint32_t v : verts) { member[v] = set_id; total += g.weight[v]; }
        const int left_parts = parts / 2;
        const uint64_t target = total * static_cast<uint64_t>(left_parts) / static_cast<uint64_t>(parts);

        const uint32_t a = bfs(g, verts, member, set_id, verts[0], dA, tet_stamp, stamp, queue);
        const uint32_t b = bfs(g, verts, member, set_id, a, dA, tet_stamp, stamp, queue);
        bfs(g, verts, member, set_id, b, dB, tet_stamp, stamp, queue);

        // candidate keys; a split takes the vertices in ascending (key, id) order until the target weight is reached
        std::vector<uint32_t> order(verts), best_left;
        uint64_t best_cut = ~0ull;
        const int candidates = xyz ? 6 : 3;
        for (int c = 0; c < candidates; c++) {
            auto key = [&](uint32_t v) -> double {
                switch (c) {
                    case 0: return static_cast<double>(dA[v]) - static_cast<double>(dB[v]);
                    case 1: return static_cast<double>(dA[v]);
                    case 2: return static_cast<double>(dB[v]);
                    default: return static_cast<double>(xyz[3ull * v + (c - 3)]);
                }
            };
            std::sort(order.begin(), order.end(), [&](uint32_t p, uint32_t q) {
                const double kp = key(p), kq = key(q);
                if (kp != kq) return kp < kq;
                if (c == 0 && dA[p] != dA[q]) return dA[p] < dA[q];
                return p < q;
            });
            uint64_t acc = 0;
            size_t n_left = 0;
            while (n_left < order.size() && acc + g.weight[order[n_left]] / 2 < target) acc += g.weight[order[n_left++]];
            n_left = std::min(std::max<size_t>(n_left, 1), order.size() - 1);   // (both sides non-empty whenever there are two vertices)
            for (size_t i = 0; i < order.size(); i++) side[order[i]] = i >= n_left;
            const uint64_t cut = cut_tets(verts, set_id);
            if (cut < best_cut) { best_cut = cut; best_left.assign(order.begin(), order.begin() + static_cast<std::ptrdiff_t>(n_left)); }
        }
        for (uint32_t v : verts) side[v] = 1;
        for (uint32_t v : best_left) side[v] = 0;
        std::vector<uint32_t> left, right;
        for (uint32_t v : verts) (side[v] ? right : left).push_back(v);
        if (verts.size() == 1) { left = verts; right.clear(); }
        std::vector<uint32_t>().swap(verts);
        std::vector<uint32_t>().swap(order);
        split(std::move(left), left_parts, first_part);
        split(std::move(right), parts - left_parts, first_part + left_parts);
    }
};

void refine(const Graph& g, int parts, std::vector<int32_t>& owner) {
    std::vector<uint64_t> load(parts, 0);
    uint64_t total = 0;
    for (uint32_t v = 0; v < g.nv; v++) { load[owner[v]] += g.weight[v]; total += g.weight[v]; }
    const double mean = static_cast<double>(total) / parts;
    const double hi = mean * 1.03, lo = mean * 0.97;
    std::vector<uint32_t> cnt(parts, 0);
    std::vector<int32_t> touched;
    for (int pass = 0; pass < kRefinePasses; pass++) {
        uint64_t moves = 0;
        for (uint32_t v = 0; v < g.nv; v++) {
            const int32_t a = owner[v];
            touched.clear();
            for (uint32_t i = g.off[v]; i < g.off[v + 1]; i++) {
                const int32_t* t = &g.tets[4ull * g.inc[i]];
                for (int k = 0; k < 4; k++) {
                    if (static_cast<uint32_t>(t[k]) == v) continue;
                    const int32_t r = owner[t[k]];
                    if (cnt[r]++ == 0) touched.push_back(r);
                }
            }
            int32_t best = a;
            for (int32_t r : touched)
                if (r != a && (cnt[r] > cnt[best] || (cnt[r] == cnt[best] && best != a && r < best))) best = r;
            const bool gain = best != a && cnt[best] > cnt[a];
            for (int32_t r : touched) cnt[r] = 0;
            if (!gain) continue;
            const uint64_t w = g.weight[v];
            if (static_cast<double>(load[best] + w) > hi || static_cast<double>(load[a] - w) < lo) continue;
            load[a] -= w; load[best] += w;
            owner[v] = best;
            moves++;
        }
        if (moves == 0) break;
    }
}

}  // namespace

std::string prep_partition(const float* verts, uint32_t nv, const int32_t* tets, uint32_t nt, int parts, int32_t* out_owner) {
    if (parts < 1) return "part_count must be at least 1";
    if (!out_owner && nv) return "null output";
    for (uint64_t i = 0; i < 4ull * nt; i++)
        if (tets[i] < 0 || static_cast<uint32_t>(tets[i]) >= nv) return "tet vertex id out of range";
    std::vector<int32_t> owner(nv, 0);
    if (parts > 1 && nv > 0) {
        const Graph g = make_graph(tets, nt, nv);
        {
            Bisector bis(g, verts, owner);
            std::vector<uint32_t> all(nv);
            std::iota(all.begin(), all.end(), 0u);
            bis.split(std::move(all), parts, 0);
        }
        refine(g, parts, owner);
    }
    std::copy(owner.begin(), owner.end(), out_owner);
    return "";
}

std::string partition_quality(const int32_t* tets, uint32_t nt, uint32_t nv, int parts, const int32_t* owner, PartQuality* out) {
    if (parts < 1 || !owner || !out) return "bad argument";
    for (uint32_t v = 0; v < nv; v++)
        if (owner[v] < 0 || owner[v] >= parts) return "vert_owner out of range";
    for (int r = 0; r < parts; r++) out[r] = PartQuality();
    for (uint32_t v = 0; v < nv; v++) out[owner[v]].owned_particles++;
    // readers[v]: parts other than owner(v) that solve a tet of v -- v is their ghost (build_partition's near1)
    std::vector<std::vector<int32_t>> readers(nv);
    for (uint32_t e = 0; e < nt; e++) {
        const int32_t* t = &tets[4ull * e];
        int32_t o[4], distinct[4];
        int nd = 0, lowest = parts;
        for (int k = 0; k < 4; k++) {
            if (t[k] < 0 || static_cast<uint32_t>(t[k]) >= nv) return "tet vertex id out of range";
            o[k] = owner[t[k]];
            lowest = std::min(lowest, o[k]);
            if (std::find(distinct, distinct + nd, o[k]) == distinct + nd) distinct[nd++] = o[k];
        }
        out[lowest].owned_elems++;
        for (int i = 0; i < nd; i++) out[distinct[i]].local_elems++;
        if (nd == 1) continue;
        for (int k = 0; k < 4; k++)
            for (int i = 0; i < nd; i++)
                if (distinct[i] != o[k] && std::find(readers[t[k]].begin(), readers[t[k]].end(), distinct[i]) == readers[t[k]].end())
                    readers[t[k]].push_back(distinct[i]);
    }
    std::vector<std::vector<int32_t>> neigh(parts);
    for (uint32_t v = 0; v < nv; v++) {
        if (readers[v].empty()) continue;
        out[owner[v]].boundary_particles++;
        for (int32_t r : readers[v]) {
            out[r].ghost_particles++;
            if (std::find(neigh[r].begin(), neigh[r].end(), owner[v]) == neigh[r].end()) neigh[r].push_back(owner[v]);
        }
    }
    for (int r = 0; r < parts; r++) out[r].num_neighbours = static_cast<uint32_t>(neigh[r].size());
    return "";
}

}  // namespace tetsim

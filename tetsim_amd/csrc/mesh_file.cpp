// mesh_file.cpp -- the .tetsim binary mesh container, GPU-free.
//
// The reference ships its demo mesh as five JavaScript array literals (Dragon.js:1,311,1080,1705,11640: tet vertices,
// tet ids, tet edge ids, embedded visual vertices [tetNr,b0,b1,b2], visual triangle ids) that the browser parses on every
// load, and rebuilds every derived table in the constructor.  This container carries the same five arrays as raw
// little-endian sections plus the optional preprocessing a multi-GPU / coloured run wants to pin down: a tet colouring
// (NEOHOOKEAN_GS order) and a vertex->partition map (POLAR_JACOBI domain decomposition).  Readers mmap the file: the
// arrays are used in place, nothing is parsed.
//
// layout:  Header (64 B) | Section[nsec] (32 B each) | data, every section 64-byte aligned
#include "mesh_file.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace tetsim {
namespace {

constexpr char kMagic[8] = {'T', 'E', 'T', 'S', 'I', 'M', 1, '\n'};
constexpr uint32_t kVersion = 1;
enum : uint32_t { DT_F32 = 0, DT_I32 = 1 };

struct Header {
    char magic[8];
    uint32_t version, nsec;
    uint64_t file_bytes;
    uint32_t part_count;
    uint32_t reserved[9];
};
static_assert(sizeof(Header) == 64, "header layout");
struct Section {
    char tag[8];
    uint32_t dtype, ncols;
    uint64_t nrows, offset;
};
static_assert(sizeof(Section) == 32, "section layout");

struct Spec { const char* tag; uint32_t dtype, ncols; };
constexpr Spec kVerts{"verts", DT_F32, 3}, kTets{"tets", DT_I32, 4}, kEdges{"edges", DT_I32, 2}, kVisVerts{"visverts", DT_F32, 4},
    kVisTris{"vistris", DT_I32, 3}, kColour{"colour", DT_I32, 1}, kOwner{"owner", DT_I32, 1};

uint64_t align64(uint64_t x) { return (x + 63u) & ~uint64_t(63); }

}  // namespace

struct MeshFile {
    void* map = nullptr;
    size_t bytes = 0;
    TetSimMeshArrays arrays{};
};

std::string mesh_write(const char* path, const TetSimMeshArrays& a) {
    if (!path) return "path is null";
    if (!a.verts || a.num_particles == 0) return "mesh file needs vertices";
    if (a.num_elems && !a.tets) return "tets is null";
    struct Item { Spec spec; const void* data; uint64_t nrows; };
    std::vector<Item> items;
    items.push_back({kVerts, a.verts, a.num_particles});
    items.push_back({kTets, a.tets, a.num_elems});
    if (a.edge_ids) items.push_back({kEdges, a.edge_ids, a.num_edges});
    if (a.vis_verts) items.push_back({kVisVerts, a.vis_verts, a.num_vis_verts});
    if (a.vis_tri_ids) items.push_back({kVisTris, a.vis_tri_ids, a.num_vis_tris});
    if (a.tet_colour) items.push_back({kColour, a.tet_colour, a.num_elems});
    if (a.vert_owner) {
        if (a.part_count < 1) return "vert_owner needs part_count >= 1";
        for (uint32_t v = 0; v < a.num_particles; v++)
            if (a.vert_owner[v] < 0 || static_cast<uint32_t>(a.vert_owner[v]) >= a.part_count) return "vert_owner entry out of [0, part_count)";
        items.push_back({kOwner, a.vert_owner, a.num_particles});
    }
    Header h{};
    std::memcpy(h.magic, kMagic, 8);
    h.version = kVersion;
    h.nsec = static_cast<uint32_t>(items.size());
    h.part_count = a.vert_owner ? a.part_count : 0;
    std::vector<Section> secs(items.size());
    uint64_t off = align64(sizeof(Header) + sizeof(Section) * items.size());
    for (size_t i = 0; i < items.size(); i++) {
        Section& s = secs[i];
        std::memset(&s, 0, sizeof s);
        std::strncpy(s.tag, items[i].spec.tag, 8);
        s.dtype = items[i].spec.dtype; s.ncols = items[i].spec.ncols; s.nrows = items[i].nrows; s.offset = off;
        off = align64(off + 4ull * s.ncols * s.nrows);
    }
    h.file_bytes = off;
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return "cannot open " + tmp + " for writing";
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && (secs.empty() || std::fwrite(secs.data(), sizeof(Section), secs.size(), f) == secs.size());
    uint64_t at = sizeof(Header) + sizeof(Section) * secs.size();
    static const char zeros[64] = {0};
    for (size_t i = 0; ok && i < items.size(); i++) {
        ok = ok && std::fwrite(zeros, 1, secs[i].offset - at, f) == secs[i].offset - at;
        const uint64_t n = 4ull * secs[i].ncols * secs[i].nrows;
        ok = ok && (n == 0 || std::fwrite(items[i].data, 1, n, f) == n);
        at = secs[i].offset + n;
    }
    ok = ok && std::fwrite(zeros, 1, h.file_bytes - at, f) == h.file_bytes - at;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) { std::remove(tmp.c_str()); return std::string("write failed: ") + path; }
    return "";
}

std::string mesh_open(const char* path, MeshFile** out) {
    if (!path || !out) return "null argument";
    *out = nullptr;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) return std::string("cannot open ") + path;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < static_cast<off_t>(sizeof(Header))) { ::close(fd); return std::string(path) + ": not a .tetsim file (too short)"; }
    void* map = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map == MAP_FAILED) return std::string("mmap failed: ") + path;
    MeshFile* m = new MeshFile();
    m->map = map; m->bytes = st.st_size;
    auto bad = [&](const std::string& why) { mesh_close(m); return std::string(path) + ": " + why; };
    const Header* h = static_cast<const Header*>(map);
    if (std::memcmp(h->magic, kMagic, 8) != 0) return bad("not a .tetsim file (bad magic)");
    if (h->version != kVersion) return bad("unsupported .tetsim version " + std::to_string(h->version));
    if (h->file_bytes != static_cast<uint64_t>(st.st_size)) return bad("truncated or padded file (header says " + std::to_string(h->file_bytes) + " bytes)");
    if (h->nsec > 64 || sizeof(Header) + sizeof(Section) * uint64_t(h->nsec) > m->bytes) return bad("section table out of bounds");
    const Section* secs = reinterpret_cast<const Section*>(static_cast<const char*>(map) + sizeof(Header));
    TetSimMeshArrays& a = m->arrays;
    a.part_count = h->part_count;
    bool have_verts = false, have_tets = false;
    for (uint32_t i = 0; i < h->nsec; i++) {
        const Section& s = secs[i];
        char tag[9] = {0};
        std::memcpy(tag, s.tag, 8);
        const uint64_t n = 4ull * s.ncols * s.nrows;
        if ((s.offset & 63u) || s.offset > m->bytes || n > m->bytes - s.offset || s.nrows > 0xffffffffull) return bad(std::string("section '") + tag + "' out of bounds");
        const void* p = static_cast<const char*>(map) + s.offset;
        auto is = [&](const Spec& sp) {
            return std::strcmp(tag, sp.tag) == 0 && s.dtype == sp.dtype && s.ncols == sp.ncols;
        };
        const uint32_t rows = static_cast<uint32_t>(s.nrows);
        if (is(kVerts)) { a.verts = static_cast<const float*>(p); a.num_particles = rows; have_verts = true; }
        else if (is(kTets)) { a.tets = static_cast<const int32_t*>(p); a.num_elems = rows; have_tets = true; }
        else if (is(kEdges)) { a.edge_ids = static_cast<const int32_t*>(p); a.num_edges = rows; }
        else if (is(kVisVerts)) { a.vis_verts = static_cast<const float*>(p); a.num_vis_verts = rows; }
        else if (is(kVisTris)) { a.vis_tri_ids = static_cast<const int32_t*>(p); a.num_vis_tris = rows; }
        else if (is(kColour)) a.tet_colour = static_cast<const int32_t*>(p);
        else if (is(kOwner)) a.vert_owner = static_cast<const int32_t*>(p);
        else if (std::strcmp(tag, kVerts.tag) == 0 || std::strcmp(tag, kTets.tag) == 0 || std::strcmp(tag, kColour.tag) == 0 || std::strcmp(tag, kOwner.tag) == 0)
            return bad(std::string("section '") + tag + "' has the wrong type or shape");
        // unknown tags are skipped: newer writers may add sections
    }
    if (!have_verts || !have_tets || a.num_particles == 0) return bad("missing 'verts' / 'tets' section");
    for (uint32_t i = 0; i < h->nsec; i++) {  // per-row sections must match their parent's row count
        char tag[9] = {0};
        std::memcpy(tag, secs[i].tag, 8);
        if (std::strcmp(tag, kColour.tag) == 0 && secs[i].nrows != a.num_elems) return bad("'colour' must have one row per tet");
        if (std::strcmp(tag, kOwner.tag) == 0 && secs[i].nrows != a.num_particles) return bad("'owner' must have one row per particle");
    }
    for (uint64_t i = 0; i < 4ull * a.num_elems; i++)
        if (a.tets[i] < 0 || static_cast<uint32_t>(a.tets[i]) >= a.num_particles) return bad("tet " + std::to_string(i / 4) + " references a particle out of range");
    if (a.vert_owner) {
        if (a.part_count < 1) return bad("'owner' section without part_count");
        for (uint32_t v = 0; v < a.num_particles; v++)
            if (a.vert_owner[v] < 0 || static_cast<uint32_t>(a.vert_owner[v]) >= a.part_count) return bad("'owner' entry out of [0, part_count)");
    }
    if (a.vis_verts)
        for (uint32_t i = 0; i < a.num_vis_verts; i++) {
            const float t = a.vis_verts[4ull * i];
            if (!(t >= 0.0f) || t >= static_cast<float>(a.num_elems) || t != static_cast<float>(static_cast<uint32_t>(t))) return bad("'visverts' row " + std::to_string(i) + " references a tet out of range");
        }
    *out = m;
    return "";
}

const TetSimMeshArrays& mesh_arrays(const MeshFile* m) { return m->arrays; }

void mesh_close(MeshFile* m) {
    if (!m) return;
    if (m->map) munmap(m->map, m->bytes);
    delete m;
}

}  // namespace tetsim

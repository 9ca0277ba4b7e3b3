// tetsim_napi.cc -- thin N-API shim over the C ABI of libtetsim_hip.so (include/tetsim.h).
//
// Host orchestration stays in Node.js (the reference's driver is main.js:74-96); this addon only marshals
// Float32Array / Int32Array backing stores and plain numbers into the C entry points, 1:1, without copying on the
// JS side.  It links nothing: libtetsim_hip.so is dlopen()ed next to the package (or from TETSIM_HIP_LIB), so the
// addon also loads on hosts without ROCm -- every compute call then fails loudly (no CPU fallback).
//
// Build: g++ -std=c++17 -shared -fPIC -I/usr/include/node tetsim_napi.cc -o tetsim_napi.node -ldl
#define NAPI_VERSION 7   // napi_detach_arraybuffer (Node >= 12.17)
#include <dlfcn.h>
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tetsim.h"

namespace {

struct Api {
    void* lib = nullptr;
    decltype(&tetsim_default_options) default_options = nullptr;
    decltype(&tetsim_default_params) default_params = nullptr;
    decltype(&tetsim_create) create = nullptr;
    decltype(&tetsim_create_batch) create_batch = nullptr;
    decltype(&tetsim_get_batch_layout) get_batch_layout = nullptr;
    decltype(&tetsim_destroy) destroy = nullptr;
    decltype(&tetsim_last_error) last_error = nullptr;
    decltype(&tetsim_get_info) get_info = nullptr;
    decltype(&tetsim_step) step = nullptr;
    decltype(&tetsim_step_n) step_n = nullptr;
    decltype(&tetsim_sync) sync = nullptr;
    decltype(&tetsim_read_positions) read_positions = nullptr;
    decltype(&tetsim_read_positions_pinned) read_positions_pinned = nullptr;
    decltype(&tetsim_read_velocities) read_velocities = nullptr;
    decltype(&tetsim_read_quats) read_quats = nullptr;
    decltype(&tetsim_read_quats_pinned) read_quats_pinned = nullptr;
    decltype(&tetsim_state_size) state_size = nullptr;
    decltype(&tetsim_save_state) save_state = nullptr;
    decltype(&tetsim_load_state) load_state = nullptr;
    decltype(&tetsim_library_info) library_info = nullptr;
    decltype(&tetsim_read_vol_error) read_vol_error = nullptr;
    decltype(&tetsim_get_local_tets) get_local_tets = nullptr;
    decltype(&tetsim_set_visual_mesh) set_visual_mesh = nullptr;
    decltype(&tetsim_read_visual_mesh) read_visual_mesh = nullptr;
    decltype(&tetsim_get_visual_ids) get_visual_ids = nullptr;
    decltype(&tetsim_halo_refresh_final) halo_refresh_final = nullptr;
    decltype(&tetsim_group_refresh_final) group_refresh_final = nullptr;
    decltype(&tetsim_group_step_n) group_step_n = nullptr;
    decltype(&tetsim_set_visual_triangles) set_visual_triangles = nullptr;
    decltype(&tetsim_read_visual_vertex_normals) read_visual_vertex_normals = nullptr;
    decltype(&tetsim_visual_vertex_normals_from) visual_vertex_normals_from = nullptr;
    decltype(&tetsim_set_grab) set_grab = nullptr;
    decltype(&tetsim_start_grab) start_grab = nullptr;
    decltype(&tetsim_abi_version) abi_version = nullptr;
    decltype(&tetsim_comm_unique_id) comm_unique_id = nullptr;
    decltype(&tetsim_comm_init) comm_init = nullptr;
    decltype(&tetsim_get_owned_ids) get_owned_ids = nullptr;
    decltype(&tetsim_mesh_open) mesh_open = nullptr;
    decltype(&tetsim_mesh_arrays) mesh_arrays = nullptr;
    decltype(&tetsim_mesh_close) mesh_close = nullptr;
    decltype(&tetsim_create_from_file) create_from_file = nullptr;
    decltype(&tetsim_prep_partition) prep_partition = nullptr;
    decltype(&tetsim_prep_partition_quality) prep_partition_quality = nullptr;
    std::string err;
} g;

bool load_lib(const std::string& hint) {
    if (g.lib) return true;
    const char* env = getenv("TETSIM_HIP_LIB");
    std::string path = env ? env : hint;
    g.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g.lib) { g.err = std::string("cannot load ") + path + ": " + dlerror(); return false; }
#define SYM(field, name) g.field = reinterpret_cast<decltype(g.field)>(dlsym(g.lib, name)); if (!g.field) { g.err = std::string("libtetsim_hip lacks ") + name; return false; }
    // the ABI first: an older library lacks newer symbols, and "ABI version mismatch" is the message a maintainer can act on
    SYM(abi_version, "tetsim_abi_version")
    if (g.abi_version() != TETSIM_ABI_VERSION) { g.err = "libtetsim_hip ABI version mismatch: the library is ABI " + std::to_string(g.abi_version()) + ", this addon was built for ABI " + std::to_string(TETSIM_ABI_VERSION); return false; }
    SYM(default_options, "tetsim_default_options") SYM(default_params, "tetsim_default_params") SYM(create, "tetsim_create")
    SYM(destroy, "tetsim_destroy") SYM(last_error, "tetsim_last_error") SYM(get_info, "tetsim_get_info") SYM(step, "tetsim_step")
    SYM(step_n, "tetsim_step_n") SYM(sync, "tetsim_sync") SYM(read_positions, "tetsim_read_positions") SYM(read_positions_pinned, "tetsim_read_positions_pinned")
    SYM(read_velocities, "tetsim_read_velocities") SYM(read_quats, "tetsim_read_quats") SYM(read_vol_error, "tetsim_read_vol_error")
    SYM(get_local_tets, "tetsim_get_local_tets") SYM(set_grab, "tetsim_set_grab") SYM(start_grab, "tetsim_start_grab")
    SYM(set_visual_mesh, "tetsim_set_visual_mesh") SYM(read_visual_mesh, "tetsim_read_visual_mesh")
    SYM(set_visual_triangles, "tetsim_set_visual_triangles") SYM(read_visual_vertex_normals, "tetsim_read_visual_vertex_normals")
    SYM(create_batch, "tetsim_create_batch") SYM(get_batch_layout, "tetsim_get_batch_layout")
    SYM(get_visual_ids, "tetsim_get_visual_ids") SYM(read_quats_pinned, "tetsim_read_quats_pinned") SYM(state_size, "tetsim_state_size")
    SYM(save_state, "tetsim_save_state") SYM(load_state, "tetsim_load_state") SYM(library_info, "tetsim_library_info")
    SYM(comm_unique_id, "tetsim_comm_unique_id") SYM(comm_init, "tetsim_comm_init") SYM(get_owned_ids, "tetsim_get_owned_ids")
    SYM(mesh_open, "tetsim_mesh_open") SYM(mesh_arrays, "tetsim_mesh_arrays") SYM(mesh_close, "tetsim_mesh_close") SYM(create_from_file, "tetsim_create_from_file")
    SYM(prep_partition, "tetsim_prep_partition") SYM(prep_partition_quality, "tetsim_prep_partition_quality")
    SYM(visual_vertex_normals_from, "tetsim_visual_vertex_normals_from") SYM(halo_refresh_final, "tetsim_halo_refresh_final") SYM(group_refresh_final, "tetsim_group_refresh_final") SYM(group_step_n, "tetsim_group_step_n")
#undef SYM
    return true;
}

napi_value throw_err(napi_env env, const std::string& msg) {
    napi_throw_error(env, "TETSIM", msg.c_str());
    return nullptr;
}
bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
        throw_err(env, "wrong number of arguments");
        return false;
    }
    return true;
}
template <class T>
bool typed_array(napi_env env, napi_value v, napi_typedarray_type want, T** data, size_t* len) {
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return false;
    napi_typedarray_type type;
    void* p = nullptr;
    napi_value ab;
    size_t off;
    if (napi_get_typedarray_info(env, v, &type, len, &p, &ab, &off) != napi_ok || type != want) return false;
    *data = static_cast<T*>(p);
    return true;
}
bool get_double(napi_env env, napi_value obj, const char* key, double* out) {
    napi_value v;
    bool has = false;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return false;
    if (napi_get_named_property(env, obj, key, &v) != napi_ok) return false;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return false;
    return napi_get_value_double(env, v, out) == napi_ok;
}
// What the JS side holds as "the handle".  The pinned host buffers a body hands out (mapPositions / mapQuats) are wrapped as
// external ArrayBuffers whose memory belongs to the tetsim handle, so the handle must outlive every such buffer that can still
// be read: each mapped buffer holds a count on the box (released by the buffer's own finalizer), and an explicit destroy()
// DETACHES the buffers first -- a view kept by the caller then reads as empty instead of freed memory.
struct Box {
    tetsim_handle h = nullptr;
    int refs = 1;                     // the external + one per live mapped ArrayBuffer
    napi_ref pos_ab = nullptr, quat_ab = nullptr;
};
void box_release(Box* b) {
    if (--b->refs > 0) return;
    if (b->h) g.destroy(b->h);
    delete b;
}
Box* box_of(napi_env env, napi_value v) {
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) { throw_err(env, "not a tetsim handle"); return nullptr; }
    Box* b = static_cast<Box*>(p);
    if (!b->h) { throw_err(env, "tetsim handle already destroyed"); return nullptr; }
    return b;
}
tetsim_handle handle_of(napi_env env, napi_value v) {
    Box* b = box_of(env, v);
    return b ? b->h : nullptr;
}
// numbers: a failed conversion throws instead of leaving the value uninitialised
bool num_arg(napi_env env, napi_value v, double* out, const char* what) {
    if (napi_get_value_double(env, v, out) == napi_ok) return true;
    throw_err(env, std::string(what) + " must be a number");
    return false;
}
// physicsParams object (main.js:22-36) -> TetSimParams
void params_of(napi_env env, napi_value obj, TetSimParams* p) {
    g.default_params(p);
    get_double(env, obj, "gravity", &p->gravity);
    get_double(env, obj, "friction", &p->friction);
    get_double(env, obj, "devCompliance", &p->devCompliance);
    get_double(env, obj, "volCompliance", &p->volCompliance);
    napi_value wb;
    bool has = false, isarr = false;
    if (napi_has_named_property(env, obj, "worldBounds", &has) == napi_ok && has &&
        napi_get_named_property(env, obj, "worldBounds", &wb) == napi_ok && napi_is_array(env, wb, &isarr) == napi_ok && isarr)
        for (uint32_t i = 0; i < 6; i++) {
            napi_value e;
            if (napi_get_element(env, wb, i, &e) == napi_ok) napi_get_value_double(env, e, &p->worldBounds[i]);
        }
}
napi_value check(napi_env env, int rc, tetsim_handle h) {
    if (rc == TETSIM_OK) { napi_value u; napi_get_undefined(env, &u); return u; }
    return throw_err(env, std::string("tetsim error ") + std::to_string(rc) + ": " + g.last_error(h));
}
void finalize_handle(napi_env env, void* data, void*) {
    Box* b = static_cast<Box*>(data);
    if (b->pos_ab) { napi_delete_reference(env, b->pos_ab); b->pos_ab = nullptr; }
    if (b->quat_ab) { napi_delete_reference(env, b->quat_ab); b->quat_ab = nullptr; }
    box_release(b);
}
void finalize_mapped(napi_env, void*, void* hint) { box_release(static_cast<Box*>(hint)); }
napi_value wrap_handle(napi_env env, tetsim_handle h) {
    Box* b = new Box();
    b->h = h;
    napi_value ext;
    if (napi_create_external(env, b, finalize_handle, nullptr, &ext) != napi_ok) { g.destroy(h); delete b; return throw_err(env, "cannot wrap the handle"); }
    return ext;
}

// load(path) -> abi version
napi_value Load(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    char buf[4096];
    size_t n = 0;
    if (napi_get_value_string_utf8(env, a[0], buf, sizeof(buf), &n) != napi_ok) return throw_err(env, "load(path): path must be a string");
    if (!load_lib(buf)) return throw_err(env, g.err);
    napi_value v;
    napi_create_int32(env, g.abi_version(), &v);
    return v;
}

// create(Float32Array verts, Int32Array tets, {solver, precision, order, flags, device, density}) -> handle
void options_of(napi_env env, napi_value obj, TetSimOptions* o, size_t* owner_len);
napi_value Create(napi_env env, napi_callback_info info) {
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return nullptr;
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    float* verts; int32_t* tets; size_t nvf, ntf;
    if (!typed_array(env, a[0], napi_float32_array, &verts, &nvf)) return throw_err(env, "vertices must be a Float32Array");
    if (!typed_array(env, a[1], napi_int32_array, &tets, &ntf)) return throw_err(env, "tetIds must be an Int32Array");
    if (nvf % 3 || ntf % 4) return throw_err(env, "vertices need 3 floats per particle, tetIds 4 ids per tet");
    TetSimOptions o;
    size_t owner_len = 0;
    options_of(env, a[2], &o, &owner_len);
    if (o.vert_owner && owner_len != nvf / 3) return throw_err(env, "vertOwner must hold one partition index per particle");
    tetsim_handle h = nullptr;
    const int rc = g.create(verts, static_cast<uint32_t>(nvf / 3), tets, static_cast<uint32_t>(ntf / 4), &o, &h);
    if (rc != TETSIM_OK) return check(env, rc, nullptr);
    return wrap_handle(env, h);
}
// createBatch(Float32Array[] verts, Int32Array[] tets, options) -> handle of the concatenation (tetsim_create_batch)
napi_value CreateBatch(napi_env env, napi_callback_info info) {
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return nullptr;
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    bool isv = false, ist = false;
    uint32_t n = 0, nt_ = 0;
    if (napi_is_array(env, a[0], &isv) != napi_ok || !isv || napi_is_array(env, a[1], &ist) != napi_ok || !ist ||
        napi_get_array_length(env, a[0], &n) != napi_ok || napi_get_array_length(env, a[1], &nt_) != napi_ok || n == 0 || n != nt_)
        return throw_err(env, "createBatch(verts[], tets[], options): two arrays of equal, non-zero length");
    std::vector<const float*> vp(n);
    std::vector<const int32_t*> tp(n);
    std::vector<uint32_t> nv(n), nt(n);
    for (uint32_t b = 0; b < n; b++) {
        napi_value ev, et;
        float* v; int32_t* t; size_t lv, lt;
        if (napi_get_element(env, a[0], b, &ev) != napi_ok || !typed_array(env, ev, napi_float32_array, &v, &lv) || lv % 3)
            return throw_err(env, "every vertices entry must be a Float32Array of xyz triples");
        if (napi_get_element(env, a[1], b, &et) != napi_ok || !typed_array(env, et, napi_int32_array, &t, &lt) || lt % 4)
            return throw_err(env, "every tetIds entry must be an Int32Array of 4 ids per tet");
        vp[b] = v; tp[b] = t; nv[b] = static_cast<uint32_t>(lv / 3); nt[b] = static_cast<uint32_t>(lt / 4);
    }
    TetSimOptions o;
    size_t owner_len = 0;
    options_of(env, a[2], &o, &owner_len);
    o.vert_owner = nullptr;
    tetsim_handle h = nullptr;
    const int rc = g.create_batch(vp.data(), nv.data(), tp.data(), nt.data(), n, &o, &h);
    if (rc != TETSIM_OK) return check(env, rc, nullptr);
    return wrap_handle(env, h);
}
// batchLayout(handle) -> { firstParticle: Uint32Array[bodies+1], firstElem: Uint32Array[bodies+1] }
napi_value BatchLayout(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    TetSimInfo inf;
    g.get_info(h, &inf);
    const size_t n = static_cast<size_t>(inf.num_bodies) + 1;
    void *dp = nullptr, *de = nullptr; napi_value abp, abe, tp_, te_, o;
    napi_create_arraybuffer(env, 4 * n, &dp, &abp);
    napi_create_arraybuffer(env, 4 * n, &de, &abe);
    const int rc = g.get_batch_layout(h, static_cast<uint32_t*>(dp), static_cast<uint32_t*>(de));
    if (rc) return check(env, rc, h);
    napi_create_typedarray(env, napi_uint32_array, n, abp, 0, &tp_);
    napi_create_typedarray(env, napi_uint32_array, n, abe, 0, &te_);
    napi_create_object(env, &o);
    napi_set_named_property(env, o, "firstParticle", tp_);
    napi_set_named_property(env, o, "firstElem", te_);
    return o;
}
// options object -> TetSimOptions (shared by create / createFromFile)
void options_of(napi_env env, napi_value obj, TetSimOptions* o, size_t* owner_len) {
    g.default_options(o);
    *owner_len = 0;
    double d;
    if (get_double(env, obj, "solver", &d)) o->solver = static_cast<int32_t>(d);
    if (get_double(env, obj, "precision", &d)) o->precision = static_cast<int32_t>(d);
    if (get_double(env, obj, "order", &d)) o->order = static_cast<int32_t>(d);
    if (get_double(env, obj, "flags", &d)) o->flags = static_cast<uint32_t>(d);
    if (get_double(env, obj, "device", &d)) o->device = static_cast<int32_t>(d);
    if (get_double(env, obj, "density", &d)) o->density = d;
    // domain decomposition (one Node process per GPU): partCount / partIndex / vertOwner: Int32Array [numParticles]
    if (get_double(env, obj, "partCount", &d)) o->part_count = static_cast<int32_t>(d);
    if (get_double(env, obj, "partIndex", &d)) o->part_index = static_cast<int32_t>(d);
    napi_value ov; bool has = false;
    if (napi_has_named_property(env, obj, "vertOwner", &has) == napi_ok && has && napi_get_named_property(env, obj, "vertOwner", &ov) == napi_ok) {
        int32_t* own; size_t n;
        if (typed_array(env, ov, napi_int32_array, &own, &n)) { o->vert_owner = own; *owner_len = n; }   // read during create only; length checked by the caller
    }
}
// createFromFile(path, options) -> handle      (tetsim_create_from_file: the library maps the .tetsim container itself)
napi_value CreateFromFile(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return nullptr;
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    char path[4096]; size_t n = 0;
    if (napi_get_value_string_utf8(env, a[0], path, sizeof path, &n) != napi_ok) return throw_err(env, "path must be a string");
    TetSimOptions o;
    size_t owner_len = 0;
    options_of(env, a[1], &o, &owner_len);
    if (o.vert_owner) {   // its length must match the file's particle count: look before the library reads it
        tetsim_mesh m = nullptr;
        int mrc = g.mesh_open(path, &m);
        if (mrc != TETSIM_OK) return check(env, mrc, nullptr);
        TetSimMeshArrays A;
        g.mesh_arrays(m, &A);
        const bool ok = owner_len == A.num_particles;
        g.mesh_close(m);
        if (!ok) return throw_err(env, "vertOwner must hold one partition index per particle of the file");
    }
    tetsim_handle h = nullptr;
    const int rc = g.create_from_file(path, &o, &h);
    if (rc != TETSIM_OK) return check(env, rc, nullptr);
    return wrap_handle(env, h);
}
// readMesh(path) -> { vertices: Float32Array, tetIds: Int32Array, tetEdgeIds, visVerts, visTriIds, tetColour, vertOwner, partCount }
// (copies out of the mapping: the JS arrays outlive it; absent sections are null)
napi_value ReadMesh(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    char path[4096]; size_t n = 0;
    if (napi_get_value_string_utf8(env, a[0], path, sizeof path, &n) != napi_ok) return throw_err(env, "path must be a string");
    tetsim_mesh m = nullptr;
    int rc = g.mesh_open(path, &m);
    if (rc != TETSIM_OK) return check(env, rc, nullptr);
    TetSimMeshArrays A;
    g.mesh_arrays(m, &A);
    napi_value obj; napi_create_object(env, &obj);
    auto put = [&](const char* name, const void* src, size_t count, napi_typedarray_type type) {
        napi_value v;
        if (!src) napi_get_null(env, &v);
        else {
            void* dst = nullptr; napi_value ab;
            napi_create_arraybuffer(env, count * 4, &dst, &ab);
            if (count) std::memcpy(dst, src, count * 4);
            napi_create_typedarray(env, type, count, ab, 0, &v);
        }
        napi_set_named_property(env, obj, name, v);
    };
    put("vertices", A.verts, 3ull * A.num_particles, napi_float32_array);
    put("tetIds", A.tets, 4ull * A.num_elems, napi_int32_array);
    put("tetEdgeIds", A.edge_ids, 2ull * A.num_edges, napi_int32_array);
    put("visVerts", A.vis_verts, 4ull * A.num_vis_verts, napi_float32_array);
    put("visTriIds", A.vis_tri_ids, 3ull * A.num_vis_tris, napi_int32_array);
    put("tetColour", A.tet_colour, A.num_elems, napi_int32_array);
    put("vertOwner", A.vert_owner, A.num_particles, napi_int32_array);
    napi_value pc; napi_create_uint32(env, A.part_count, &pc);
    napi_set_named_property(env, obj, "partCount", pc);
    g.mesh_close(m);
    return obj;
}
napi_value Destroy(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    void* p = nullptr;
    if (napi_get_value_external(env, a[0], &p) == napi_ok && p) {
        Box* b = static_cast<Box*>(p);
        for (napi_ref* r : {&b->pos_ab, &b->quat_ab}) {   // the pinned memory goes away with the handle: detach what wraps it
            if (!*r) continue;
            napi_value ab;
            if (napi_get_reference_value(env, *r, &ab) == napi_ok && ab) napi_detach_arraybuffer(env, ab);
            napi_delete_reference(env, *r);
            *r = nullptr;
        }
        if (b->h) { g.destroy(b->h); b->h = nullptr; }
    }
    napi_value u; napi_get_undefined(env, &u); return u;
}
// step(handle, dt, physicsParams)  /  stepN(handle, n, dt, physicsParams)
napi_value Step(napi_env env, napi_callback_info info) {
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    double dt;
    if (!num_arg(env, a[1], &dt, "dt")) return nullptr;
    TetSimParams p; params_of(env, a[2], &p);
    return check(env, g.step(h, dt, &p), h);
}
napi_value StepN(napi_env env, napi_callback_info info) {
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    double nd, dt;
    if (!num_arg(env, a[1], &nd, "n") || !num_arg(env, a[2], &dt, "dt")) return nullptr;
    if (!(nd >= 0.0) || nd > 4294967295.0 || nd != static_cast<double>(static_cast<uint32_t>(nd))) return throw_err(env, "n must be a non-negative integer");
    const uint32_t n = static_cast<uint32_t>(nd);
    TetSimParams p; params_of(env, a[3], &p);
    return check(env, g.step_n(h, n, dt, &p), h);
}
napi_value Sync(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    return h ? check(env, g.sync(h), h) : nullptr;
}
template <int (*Api::*Fn)(tetsim_handle, float*), int Per, bool Tets>
napi_value ReadF32(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    float* out; size_t n;
    if (!typed_array(env, a[1], napi_float32_array, &out, &n)) return throw_err(env, "output must be a Float32Array");
    TetSimInfo inf;
    g.get_info(h, &inf);
    const size_t need = static_cast<size_t>(Per) * (Tets ? inf.local_elems : inf.owned_particles);
    if (n < need) return throw_err(env, "output array too small");
    return check(env, (g.*Fn)(h, out), h);
}
// mapPositions(handle) / mapQuats(handle) -> Float32Array over the handle's PINNED host buffer (zero copy);
// refreshPositions / refreshQuats re-fill it (one DMA).  destroy(handle) detaches the buffer (the view reads as empty
// afterwards); a view that outlives an undestroyed body keeps the handle alive until it is collected itself.
template <int (*Api::*Fn)(tetsim_handle, const float**), napi_ref Box::*Slot, int Per, bool Tets>
napi_value MapPinned(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    Box* b = box_of(env, a[0]);
    if (!b) return nullptr;
    const float* p = nullptr;
    const int rc = (g.*Fn)(b->h, &p);
    if (rc) return check(env, rc, b->h);
    TetSimInfo inf;
    g.get_info(b->h, &inf);
    const size_t count = static_cast<size_t>(Per) * (Tets ? inf.local_elems : inf.owned_particles);
    napi_value ab, ta;
    if (b->*Slot) {   // mapped before: hand out a view of the same ArrayBuffer
        if (napi_get_reference_value(env, b->*Slot, &ab) != napi_ok || !ab) return throw_err(env, "the mapped buffer is gone");
    } else {
        if (napi_create_external_arraybuffer(env, const_cast<float*>(p), sizeof(float) * count, finalize_mapped, b, &ab) != napi_ok)
            return throw_err(env, "cannot wrap the pinned buffer");
        b->refs++;   // released by finalize_mapped
        if (napi_create_reference(env, ab, 1, &(b->*Slot)) != napi_ok) return throw_err(env, "cannot reference the pinned buffer");
    }
    if (napi_create_typedarray(env, napi_float32_array, count, ab, 0, &ta) != napi_ok) return throw_err(env, "cannot wrap the pinned buffer");
    return ta;
}
template <int (*Api::*Fn)(tetsim_handle, const float**)>
napi_value RefreshPinned(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    const float* p = nullptr;
    return check(env, (g.*Fn)(h, &p), h);
}
// saveState(handle) -> Uint8Array (complete solver state)  /  loadState(handle, Uint8Array)
napi_value SaveState(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    uint64_t n = 0;
    int rc = g.state_size(h, &n);
    if (rc) return check(env, rc, h);
    void* data = nullptr; napi_value ab, out;
    if (napi_create_arraybuffer(env, n, &data, &ab) != napi_ok) return throw_err(env, "cannot allocate the state buffer");
    rc = g.save_state(h, data, n);
    if (rc) return check(env, rc, h);
    napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &out);
    return out;
}
napi_value LoadState(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    uint8_t* blob; size_t n;
    if (!typed_array(env, a[1], napi_uint8_array, &blob, &n)) return throw_err(env, "state must be the Uint8Array saveState() returned");
    return check(env, g.load_state(h, blob, n), h);
}
// partition(Float32Array verts | null, numParticles, Int32Array tets, parts) -> Int32Array owner   (include/tetsim.h: tetsim_prep_partition;
// host only -- what a Node host stores in its mesh container or passes as `vertOwner` to every rank's body)
napi_value Partition(napi_env env, napi_callback_info info) {
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return nullptr;
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    float* verts = nullptr; int32_t* tets = nullptr; size_t nvf = 0, ntl = 0;
    napi_valuetype vt;
    napi_typeof(env, a[0], &vt);
    const bool with_coords = vt != napi_null && vt != napi_undefined;
    if (with_coords && !typed_array(env, a[0], napi_float32_array, &verts, &nvf)) return throw_err(env, "partition: verts must be a Float32Array or null");
    double nvd = 0, partsd = 0;
    if (napi_get_value_double(env, a[1], &nvd) != napi_ok || napi_get_value_double(env, a[3], &partsd) != napi_ok || !(nvd >= 0) || nvd > 1073741823.0 || !(partsd >= 1) || partsd > 65536.0)
        return throw_err(env, "partition(verts | null, numParticles, tets, parts): numParticles / parts out of range");
    const uint32_t nv = static_cast<uint32_t>(nvd);
    if (!typed_array(env, a[2], napi_int32_array, &tets, &ntl) || ntl % 4) return throw_err(env, "partition: tets must be an Int32Array of 4 ids per tet");
    if (with_coords && nvf != 3ull * nv) return throw_err(env, "partition: verts must hold 3 floats per particle");
    napi_value ab, out; void* data = nullptr;
    if (napi_create_arraybuffer(env, sizeof(int32_t) * static_cast<size_t>(nv), &data, &ab) != napi_ok || napi_create_typedarray(env, napi_int32_array, nv, ab, 0, &out) != napi_ok)
        return throw_err(env, "partition: cannot allocate the result");
    const int rc = g.prep_partition(with_coords ? verts : nullptr, nv, tets, static_cast<uint32_t>(ntl / 4), static_cast<int32_t>(partsd), static_cast<int32_t*>(data));
    if (rc) return check(env, rc, nullptr);
    return out;
}
// partitionQuality(Int32Array tets, numParticles, parts, Int32Array owner | null) -> [{ownedParticles, ghostParticles, boundaryParticles, localElems, ownedElems, numNeighbours}]
napi_value PartitionQuality(napi_env env, napi_callback_info info) {
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return nullptr;
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    int32_t *tets = nullptr, *owner = nullptr; size_t ntl = 0, nol = 0;
    double nvd = 0, partsd = 0;
    if (!typed_array(env, a[0], napi_int32_array, &tets, &ntl) || ntl % 4) return throw_err(env, "partitionQuality: tets must be an Int32Array of 4 ids per tet");
    if (napi_get_value_double(env, a[1], &nvd) != napi_ok || napi_get_value_double(env, a[2], &partsd) != napi_ok || !(nvd >= 0) || nvd > 1073741823.0 || !(partsd >= 1) || partsd > 65536.0)
        return throw_err(env, "partitionQuality(tets, numParticles, parts, owner | null): numParticles / parts out of range");
    napi_valuetype vt;
    napi_typeof(env, a[3], &vt);
    if (vt != napi_null && vt != napi_undefined && (!typed_array(env, a[3], napi_int32_array, &owner, &nol) || nol != static_cast<size_t>(nvd)))
        return throw_err(env, "partitionQuality: owner must be an Int32Array of numParticles entries or null");
    std::vector<TetSimPartQuality> q(static_cast<size_t>(partsd));
    const int rc = g.prep_partition_quality(tets, static_cast<uint32_t>(ntl / 4), static_cast<uint32_t>(nvd), static_cast<int32_t>(partsd), owner, q.data());
    if (rc) return check(env, rc, nullptr);
    napi_value arr; napi_create_array_with_length(env, q.size(), &arr);
    for (size_t i = 0; i < q.size(); i++) {
        napi_value o, v; napi_create_object(env, &o);
        napi_create_uint32(env, q[i].owned_particles, &v); napi_set_named_property(env, o, "ownedParticles", v);
        napi_create_uint32(env, q[i].ghost_particles, &v); napi_set_named_property(env, o, "ghostParticles", v);
        napi_create_uint32(env, q[i].boundary_particles, &v); napi_set_named_property(env, o, "boundaryParticles", v);
        napi_create_uint32(env, q[i].local_elems, &v); napi_set_named_property(env, o, "localElems", v);
        napi_create_uint32(env, q[i].owned_elems, &v); napi_set_named_property(env, o, "ownedElems", v);
        napi_create_uint32(env, q[i].num_neighbours, &v); napi_set_named_property(env, o, "numNeighbours", v);
        napi_set_element(env, arr, static_cast<uint32_t>(i), o);
    }
    return arr;
}
// libraryInfo() -> { abi, ablation, sourceSha, kernelSha, debugEnv }
napi_value LibraryInfo(napi_env env, napi_callback_info) {
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    TetSimLibraryInfo li;
    const int rc = g.library_info(&li);
    if (rc) return check(env, rc, nullptr);
    napi_value o, v; napi_create_object(env, &o);
    napi_create_int32(env, li.abi, &v); napi_set_named_property(env, o, "abi", v);
    napi_get_boolean(env, li.ablation != 0, &v); napi_set_named_property(env, o, "ablation", v);
    napi_create_string_utf8(env, li.source_sha, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "sourceSha", v);
    napi_create_string_utf8(env, li.kernel_sha, NAPI_AUTO_LENGTH, &v); napi_set_named_property(env, o, "kernelSha", v);
    napi_create_uint32(env, li.debug_env, &v); napi_set_named_property(env, o, "debugEnv", v);
    return o;
}
napi_value ReadVolError(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    double v = 0.0;
    const int rc = g.read_vol_error(h, &v);
    if (rc) return check(env, rc, h);
    napi_value r; napi_create_double(env, v, &r); return r;
}
// setVisualMesh(handle, Float32Array visVerts [tetNr,b0,b1,b2]*, Float32Array restNormals | null)
napi_value SetVisualMesh(napi_env env, napi_callback_info info) {
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    float *vv, *nn = nullptr; size_t nv4, nn3 = 0;
    if (!typed_array(env, a[1], napi_float32_array, &vv, &nv4) || nv4 % 4) return throw_err(env, "visVerts must be a Float32Array of (tetNr,b0,b1,b2) rows");
    if (typed_array(env, a[2], napi_float32_array, &nn, &nn3) && nn3 != nv4 / 4 * 3) return throw_err(env, "restNormals must hold 3 floats per visual vertex");
    return check(env, g.set_visual_mesh(h, vv, static_cast<uint32_t>(nv4 / 4), nn3 ? nn : nullptr), h);
}
// readVisualMesh(handle, Float32Array positionsOut, Float32Array normalsOut | null)
napi_value ReadVisualMesh(napi_env env, napi_callback_info info) {
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    float *po, *no = nullptr; size_t np = 0, nn = 0;
    if (!typed_array(env, a[1], napi_float32_array, &po, &np)) return throw_err(env, "positions output must be a Float32Array");
    const bool want_normals = typed_array(env, a[2], napi_float32_array, &no, &nn);
    TetSimInfo inf;
    g.get_info(h, &inf);
    if (np < 3ull * inf.num_vis_verts) return throw_err(env, "positions output array too small (3 floats per visual vertex)");
    if (want_normals && nn < 3ull * inf.num_vis_verts) return throw_err(env, "normals output array too small (3 floats per visual vertex)");
    return check(env, g.read_visual_mesh(h, po, want_normals ? no : nullptr), h);
}
// visualIds(handle) -> Int32Array: row of the caller's visVerts behind each attached visual vertex (a partition keeps the rows of the tets it owns)
napi_value VisualIds(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    TetSimInfo inf;
    g.get_info(h, &inf);
    napi_value ab, out;
    void* data = nullptr;
    if (napi_create_arraybuffer(env, sizeof(int32_t) * inf.num_vis_verts, &data, &ab) != napi_ok) return throw_err(env, "cannot allocate");
    if (inf.num_vis_verts && g.get_visual_ids(h, static_cast<int32_t*>(data)) != 0) return throw_err(env, g.last_error(h));
    napi_create_typedarray(env, napi_int32_array, inf.num_vis_verts, ab, 0, &out);
    return out;
}
// setVisualTriangles(handle, Int32Array visTriIds)
napi_value SetVisualTriangles(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    int32_t* tri; size_t n3;
    if (!typed_array(env, a[1], napi_int32_array, &tri, &n3) || n3 % 3) return throw_err(env, "visTriIds must be an Int32Array of 3 ids per triangle");
    return check(env, g.set_visual_triangles(h, tri, static_cast<uint32_t>(n3 / 3)), h);
}
// readVisualVertexNormals(handle, Float32Array normalsOut): geometry.computeVertexNormals() of the skinned mesh, on the device
napi_value ReadVisualVertexNormals(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    float* no; size_t nn;
    if (!typed_array(env, a[1], napi_float32_array, &no, &nn)) return throw_err(env, "normals output must be a Float32Array");
    TetSimInfo inf;
    g.get_info(h, &inf);
    if (nn < 3ull * inf.num_vis_verts) return throw_err(env, "normals output array too small (3 floats per visual vertex)");
    return check(env, g.read_visual_vertex_normals(h, no), h);
}
// visualVertexNormalsFrom(handle, Float32Array allPositions [3 * rows of visVerts], Float32Array normalsOut [3 * numVisVerts])
napi_value VisualVertexNormalsFrom(napi_env env, napi_callback_info info) {
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    float *pi, *no; size_t np_, nn;
    if (!typed_array(env, a[1], napi_float32_array, &pi, &np_)) return throw_err(env, "positions must be a Float32Array");
    if (!typed_array(env, a[2], napi_float32_array, &no, &nn)) return throw_err(env, "normals output must be a Float32Array");
    TetSimInfo inf;
    g.get_info(h, &inf);
    if (np_ < 3ull * inf.total_vis_verts) return throw_err(env, "positions array too small (3 floats per row of visVerts)");
    if (nn < 3ull * inf.num_vis_verts) return throw_err(env, "normals output array too small (3 floats per visual vertex)");
    return check(env, g.visual_vertex_normals_from(h, pi, no), h);
}
// setGrab(handle, id, x, y, z)
napi_value SetGrab(napi_env env, napi_callback_info info) {
    napi_value a[5];
    if (!get_args(env, info, 5, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    double idd, x, y, z;
    if (!num_arg(env, a[1], &idd, "id") || !num_arg(env, a[2], &x, "x") || !num_arg(env, a[3], &y, "y") || !num_arg(env, a[4], &z, "z")) return nullptr;
    if (!(idd >= -1.0) || idd > 2147483647.0 || idd != static_cast<double>(static_cast<int32_t>(idd))) return throw_err(env, "id must be a particle index or -1");
    const int32_t id = static_cast<int32_t>(idd);
    const float p[3] = {static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)};
    return check(env, g.set_grab(h, id, p), h);
}
// startGrab(handle, x, y, z) -> particle id
napi_value StartGrab(napi_env env, napi_callback_info info) {
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    double x, y, z;
    if (!num_arg(env, a[1], &x, "x") || !num_arg(env, a[2], &y, "y") || !num_arg(env, a[3], &z, "z")) return nullptr;
    const float p[3] = {static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)};
    int32_t id = -1;
    const int rc = g.start_grab(h, p, &id);
    if (rc) return check(env, rc, h);
    napi_value r; napi_create_int32(env, id, &r); return r;
}
// commUniqueId() -> Uint8Array(128): rank 0 makes it, the host distributes it (any channel), every rank calls commInit
napi_value CommUniqueId(napi_env env, napi_callback_info) {
    if (!g.lib) return throw_err(env, "libtetsim_hip.so is not loaded (call load(path) first)");
    void* data = nullptr; napi_value ab, out;
    napi_create_arraybuffer(env, 128, &data, &ab);
    const int rc = g.comm_unique_id(data);
    if (rc != TETSIM_OK) return check(env, rc, nullptr);
    napi_create_typedarray(env, napi_uint8_array, 128, ab, 0, &out);
    return out;
}
// haloRefreshFinal(handle): every rank together, after the frame's last substep -- the ghost particles' end-of-substep positions from their
// owners, which readVisualMesh of a partition needs (tetsim_halo_refresh_final; the read itself never communicates)
napi_value HaloRefreshFinal(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    return h ? check(env, g.halo_refresh_final(h), h) : nullptr;
}
// the partitions of ONE process (in-process group: several GPUs -- or one -- driven by a single Node process), handles[i] = partition i
bool handles_of(napi_env env, napi_value arr, std::vector<tetsim_handle>* hs) {
    bool is = false;
    uint32_t n = 0;
    if (napi_is_array(env, arr, &is) != napi_ok || !is || napi_get_array_length(env, arr, &n) != napi_ok || n == 0) { throw_err(env, "handles must be a non-empty array of tetsim handles"); return false; }
    hs->resize(n);
    for (uint32_t i = 0; i < n; i++) {
        napi_value e;
        if (napi_get_element(env, arr, i, &e) != napi_ok) { throw_err(env, "handles must be a non-empty array of tetsim handles"); return false; }
        if (!((*hs)[i] = handle_of(env, e))) return false;
    }
    return true;
}
// groupStepN(handles[], n, dt, physicsParams)     (tetsim_group_step_n)
napi_value GroupStepN(napi_env env, napi_callback_info info) {
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return nullptr;
    std::vector<tetsim_handle> hs;
    if (!handles_of(env, a[0], &hs)) return nullptr;
    double nd, dt;
    if (!num_arg(env, a[1], &nd, "n") || !num_arg(env, a[2], &dt, "dt")) return nullptr;
    if (!(nd >= 0.0) || nd > 4294967295.0 || nd != static_cast<double>(static_cast<uint32_t>(nd))) return throw_err(env, "n must be a non-negative integer");
    TetSimParams p; params_of(env, a[3], &p);
    return check(env, g.group_step_n(hs.data(), static_cast<uint32_t>(hs.size()), static_cast<uint32_t>(nd), dt, &p), nullptr);
}
// groupRefreshFinal(handles[])                    (tetsim_group_refresh_final)
napi_value GroupRefreshFinal(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    std::vector<tetsim_handle> hs;
    if (!handles_of(env, a[0], &hs)) return nullptr;
    return check(env, g.group_refresh_final(hs.data(), static_cast<uint32_t>(hs.size())), nullptr);
}
// commInit(handle, Uint8Array(128) id, rank, nranks): RCCL communicator for this partition's ghost halo
napi_value CommInit(napi_env env, napi_callback_info info) {
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    uint8_t* id; size_t n;
    if (!typed_array(env, a[1], napi_uint8_array, &id, &n) || n < 128) return throw_err(env, "id must be the Uint8Array(128) of commUniqueId()");
    double rank, nranks;
    if (!num_arg(env, a[2], &rank, "rank") || !num_arg(env, a[3], &nranks, "nranks")) return nullptr;
    return check(env, g.comm_init(h, id, static_cast<int32_t>(rank), static_cast<int32_t>(nranks)), h);
}
// ownedIds(handle) -> Int32Array: global particle id of every row readPositions returns (partitioned bodies)
napi_value OwnedIds(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    TetSimInfo inf;
    g.get_info(h, &inf);
    void* data = nullptr; napi_value ab, out;
    napi_create_arraybuffer(env, 4ull * inf.owned_particles, &data, &ab);
    const int rc = g.get_owned_ids(h, static_cast<int32_t*>(data));
    if (rc != TETSIM_OK) return check(env, rc, h);
    napi_create_typedarray(env, napi_int32_array, inf.owned_particles, ab, 0, &out);
    return out;
}
napi_value Info(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    tetsim_handle h = handle_of(env, a[0]);
    if (!h) return nullptr;
    TetSimInfo inf;
    const int rc = g.get_info(h, &inf);
    if (rc) return check(env, rc, h);
    napi_value o; napi_create_object(env, &o);
    auto set = [&](const char* k, double v) { napi_value x; napi_create_double(env, v, &x); napi_set_named_property(env, o, k, x); };
    set("numParticles", inf.num_particles); set("numElems", inf.num_elems); set("ownedParticles", inf.owned_particles);
    set("localElems", inf.local_elems); set("numLevels", inf.num_levels); set("maxValence", inf.max_valence);
    set("droppedSlots", inf.dropped_slots); set("deviceBytes", static_cast<double>(inf.device_bytes));
    set("solver", inf.solver); set("precision", inf.precision); set("localParticles", inf.local_particles);
    set("ownedElems", inf.owned_elems); set("numNeighbours", inf.num_neighbours); set("numVisVerts", inf.num_vis_verts); set("numBodies", inf.num_bodies); set("totalVisVerts", inf.total_vis_verts);
    return o;
}

napi_value Init(napi_env env, napi_value exports) {
    const napi_property_descriptor props[] = {
        {"load", nullptr, Load, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"create", nullptr, Create, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"createFromFile", nullptr, CreateFromFile, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"createBatch", nullptr, CreateBatch, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"batchLayout", nullptr, BatchLayout, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readMesh", nullptr, ReadMesh, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"commUniqueId", nullptr, CommUniqueId, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"commInit", nullptr, CommInit, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"ownedIds", nullptr, OwnedIds, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"destroy", nullptr, Destroy, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"step", nullptr, Step, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"stepN", nullptr, StepN, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"sync", nullptr, Sync, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readPositions", nullptr, ReadF32<&Api::read_positions, 3, false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readVelocities", nullptr, ReadF32<&Api::read_velocities, 3, false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readQuats", nullptr, ReadF32<&Api::read_quats, 4, true>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"mapPositions", nullptr, MapPinned<&Api::read_positions_pinned, &Box::pos_ab, 3, false>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"refreshPositions", nullptr, RefreshPinned<&Api::read_positions_pinned>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"mapQuats", nullptr, MapPinned<&Api::read_quats_pinned, &Box::quat_ab, 4, true>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"refreshQuats", nullptr, RefreshPinned<&Api::read_quats_pinned>, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"saveState", nullptr, SaveState, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"loadState", nullptr, LoadState, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"libraryInfo", nullptr, LibraryInfo, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"partition", nullptr, Partition, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"partitionQuality", nullptr, PartitionQuality, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readVolError", nullptr, ReadVolError, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setVisualMesh", nullptr, SetVisualMesh, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readVisualMesh", nullptr, ReadVisualMesh, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"visualIds", nullptr, VisualIds, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"haloRefreshFinal", nullptr, HaloRefreshFinal, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupStepN", nullptr, GroupStepN, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"groupRefreshFinal", nullptr, GroupRefreshFinal, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setVisualTriangles", nullptr, SetVisualTriangles, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"readVisualVertexNormals", nullptr, ReadVisualVertexNormals, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"visualVertexNormalsFrom", nullptr, VisualVertexNormalsFrom, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setGrab", nullptr, SetGrab, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"startGrab", nullptr, StartGrab, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"info", nullptr, Info, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    };
    napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)

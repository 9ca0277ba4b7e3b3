// mesa_gl.cc -- TEST INFRASTRUCTURE (golden-vector generation only; never loaded by the product or by any test).
//
// A headless OpenGL ES 3.0 context for Node (N-API) on Mesa's software rasteriser (swrast_dri.so: softpipe is what the goldens are
// recorded with; llvmpipe was rejected, see tests/golden/make_golden_gpu.sh), exposed as RAW GL ES entry points.  On top of it
// webgl2_context.mjs provides a WebGL2RenderingContext-shaped object, and the reference's vendored, unmodified
// THREE.WebGLRenderer({canvas, context}) drives it: the reference's GLSL passes (SoftbodyGPU.js:59-376), its pass scheduler
// (MultiTargetGPUComputationRenderer.js) AND three.js's own renderer all run as they are in the build container, which has no GPU, no
// X server and no EGL.  The context is created straight through the DRI software-rasteriser interface that libGLX / libEGL use
// internally (swrast_dri.so, <GL/internal/dri_interface.h>, both shipped in the image).
//
// Nothing here knows about the simulation, three.js or WebGL: `call` invokes any GL entry point whose arguments are integers / floats,
// the rest are thin marshalling wrappers for the entry points that take pointers (strings, typed arrays, out-parameters).
//
// build: oracle/glsl_ref/build.sh  ->  oracle/_ref/mesa_gl.node + oracle/_ref/gl_constants.json   (git-ignored)
#include <node_api.h>

#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

namespace {

// ---- DRI software-rasteriser context -------------------------------------------------------------------------
void getDrawableInfo(__DRIdrawable*, int* x, int* y, int* w, int* h, void*) { *x = *y = 0; *w = *h = 16; }
void putImage(__DRIdrawable*, int, int, int, int, int, char*, void*) {}
void getImage(__DRIdrawable*, int, int, int, int, char*, void*) {}
const __DRIswrastLoaderExtension g_loader = {{__DRI_SWRAST_LOADER, 1}, getDrawableInfo, putImage, getImage,
                                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

void* (*g_gpa)(const char*) = nullptr;
bool g_ready = false;
std::string g_err;
std::map<std::string, void*> g_fn;

void* gl(const char* name) {
    auto it = g_fn.find(name);
    if (it != g_fn.end()) return it->second;
    void* p = g_gpa ? g_gpa(name) : nullptr;
    g_fn.emplace(name, p);
    return p;
}
template <typename T>
T glfn(const char* name) { return reinterpret_cast<T>(gl(name)); }

bool gl_init() {
    if (g_ready) return true;
    const char* paths[] = {"/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", "swrast_dri.so"};
    void* drv = nullptr;
    for (const char* p : paths) if ((drv = dlopen(p, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!drv) { g_err = std::string("cannot load Mesa swrast_dri.so: ") + dlerror(); return false; }
    auto get = reinterpret_cast<const __DRIextension** (*)()>(dlsym(drv, "__driDriverGetExtensions_swrast"));
    if (!get) { g_err = "swrast_dri.so has no __driDriverGetExtensions_swrast"; return false; }
    const __DRIextension** exts = get();
    const __DRIcoreExtension* core = nullptr;
    const __DRIswrastExtension* sw = nullptr;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = reinterpret_cast<const __DRIcoreExtension*>(exts[i]);
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = reinterpret_cast<const __DRIswrastExtension*>(exts[i]);
    }
    if (!core || !sw || sw->base.version < 4) { g_err = "DRI_Core / DRI_SWRast v4 not offered by the driver"; return false; }
    static const __DRIextension* loader_exts[] = {&g_loader.base, nullptr};
    const __DRIconfig** cfgs = nullptr;
    __DRIscreen* scr = sw->createNewScreen2(0, loader_exts, exts, &cfgs, nullptr);
    if (!scr || !cfgs || !cfgs[0]) { g_err = "createNewScreen2 failed"; return false; }
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 0};
    unsigned err = 0;
    __DRIcontext* ctx = sw->createContextAttribs(scr, __DRI_API_GLES3, cfgs[0], nullptr, 2, attribs, &err, nullptr);  // WebGL2 == ES 3.0
    if (!ctx) { g_err = "createContextAttribs(GLES 3.0) failed, error " + std::to_string(err); return false; }
    __DRIdrawable* dr = sw->createNewDrawable(scr, cfgs[0], nullptr);
    if (!dr || !core->bindContext(ctx, dr, dr)) { g_err = "bindContext failed"; return false; }
    void* ga = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!ga) { g_err = std::string("cannot load libglapi.so.0: ") + dlerror(); return false; }
    g_gpa = reinterpret_cast<void* (*)(const char*)>(dlsym(ga, "_glapi_get_proc_address"));
    if (!g_gpa) { g_err = "_glapi_get_proc_address missing"; return false; }
    auto getString = glfn<const GLubyte* (*)(GLenum)>("glGetString");
    if (!getString) { g_err = "glGetString missing"; return false; }
    const char* ext = reinterpret_cast<const char*>(getString(GL_EXTENSIONS));
    if (!ext || !strstr(ext, "GL_EXT_color_buffer_float")) { g_err = "GL_EXT_color_buffer_float missing (float render targets)"; return false; }
    g_ready = true;
    return true;
}

// ---- N-API helpers -------------------------------------------------------------------------------------------
napi_value fail(napi_env env, const std::string& m) { napi_throw_error(env, "MESA_GL", m.c_str()); return nullptr; }
constexpr size_t kMaxArgs = 12;
struct Args {
    napi_value v[kMaxArgs];
    size_t n = kMaxArgs;
    bool ok;
    Args(napi_env env, napi_callback_info info, size_t want) { ok = napi_get_cb_info(env, info, &n, v, nullptr, nullptr) == napi_ok && n >= want; if (!ok) napi_throw_error(env, "MESA_GL", "wrong number of arguments"); }
};
bool get_i64(napi_env env, napi_value v, int64_t* out) {
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok) { bool b; if (napi_get_value_bool(env, v, &b) != napi_ok) return false; d = b ? 1.0 : 0.0; }
    *out = static_cast<int64_t>(d);
    return true;
}
bool get_str(napi_env env, napi_value v, std::string* out) {
    size_t n = 0;
    if (napi_get_value_string_utf8(env, v, nullptr, 0, &n) != napi_ok) return false;
    out->resize(n);
    return napi_get_value_string_utf8(env, v, &(*out)[0], n + 1, &n) == napi_ok;
}
// typed array / DataView-free: pointer + byte length, or null for null / undefined
bool get_bytes(napi_env env, napi_value v, void** data, size_t* bytes) {
    *data = nullptr; *bytes = 0;
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return false;
    if (t == napi_null || t == napi_undefined) return true;
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return false;
    napi_typedarray_type ty; size_t len; napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, v, &ty, &len, data, &ab, &off) != napi_ok) return false;
    static const size_t width[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
    *bytes = len * width[ty];
    return true;
}
napi_value mk_num(napi_env env, double x) { napi_value v; napi_create_double(env, x, &v); return v; }
napi_value mk_str(napi_env env, const char* s) { napi_value v; napi_create_string_utf8(env, s ? s : "", NAPI_AUTO_LENGTH, &v); return v; }
#define NEED_GL() do { if (!g_ready) return fail(env, "init() first"); } while (0)
#define FN(var, type, name) auto var = glfn<type>(name); if (!var) return fail(env, std::string("GL entry point missing: ") + name)

// ---- exported functions --------------------------------------------------------------------------------------
napi_value Init(napi_env env, napi_callback_info) {
    if (!gl_init()) return fail(env, g_err);
    auto getString = glfn<const GLubyte* (*)(GLenum)>("glGetString");
    napi_value o; napi_create_object(env, &o);
    napi_set_named_property(env, o, "vendor", mk_str(env, reinterpret_cast<const char*>(getString(GL_VENDOR))));
    napi_set_named_property(env, o, "renderer", mk_str(env, reinterpret_cast<const char*>(getString(GL_RENDERER))));
    napi_set_named_property(env, o, "version", mk_str(env, reinterpret_cast<const char*>(getString(GL_VERSION))));
    napi_set_named_property(env, o, "glsl", mk_str(env, reinterpret_cast<const char*>(getString(GL_SHADING_LANGUAGE_VERSION))));
    return o;
}

// call(name, [ints...], [floats...]) -> integer result.  Any entry point whose parameters are integers (enums, names, sizes, booleans,
// byte offsets passed as pointers) and at most four floats; on x86-64 integers and floats travel in separate registers, each class in
// order, and a callee ignores what it does not declare.
napi_value Call(napi_env env, napi_callback_info info) {
    Args a(env, info, 2);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string name;
    if (!get_str(env, a.v[0], &name)) return fail(env, "call(name, ints, floats)");
    void* p = gl(name.c_str());
    if (!p) return fail(env, "GL entry point missing: " + name);
    intptr_t iv[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float fv[4] = {0, 0, 0, 0};
    uint32_t ni = 0, nf = 0;
    if (napi_get_array_length(env, a.v[1], &ni) != napi_ok || ni > 10) return fail(env, "call: at most ten integer arguments");
    for (uint32_t i = 0; i < ni; i++) { napi_value e; int64_t x; napi_get_element(env, a.v[1], i, &e); if (!get_i64(env, e, &x)) return fail(env, "call " + name + ": integer argument expected"); iv[i] = static_cast<intptr_t>(x); }
    if (a.n > 2) {
        if (napi_get_array_length(env, a.v[2], &nf) != napi_ok || nf > 4) return fail(env, "call: at most four float arguments");
        for (uint32_t i = 0; i < nf; i++) { napi_value e; double x; napi_get_element(env, a.v[2], i, &e); if (napi_get_value_double(env, e, &x) != napi_ok) return fail(env, "call " + name + ": float argument expected"); fv[i] = static_cast<float>(x); }
    }
    using Fn = intptr_t (*)(intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, float, float, float, float);
    const intptr_t r = reinterpret_cast<Fn>(p)(iv[0], iv[1], iv[2], iv[3], iv[4], iv[5], iv[6], iv[7], iv[8], iv[9], fv[0], fv[1], fv[2], fv[3]);
    return mk_num(env, static_cast<double>(static_cast<uint32_t>(r)));   // (every integer-returning entry point returns 32 bits)
}

// gen(kind) -> name;  kind: "Textures" | "Buffers" | "Framebuffers" | "Renderbuffers" | "VertexArrays" | "Samplers" | "Queries"
napi_value Gen(napi_env env, napi_callback_info info) {
    Args a(env, info, 1);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string kind;
    if (!get_str(env, a.v[0], &kind)) return fail(env, "gen(kind)");
    FN(f, void (*)(GLsizei, GLuint*), ("glGen" + kind).c_str());
    GLuint id = 0;
    f(1, &id);
    return mk_num(env, id);
}
napi_value Del(napi_env env, napi_callback_info info) {
    Args a(env, info, 2);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string kind; int64_t id;
    if (!get_str(env, a.v[0], &kind) || !get_i64(env, a.v[1], &id)) return fail(env, "del(kind, name)");
    FN(f, void (*)(GLsizei, const GLuint*), ("glDelete" + kind).c_str());
    const GLuint u = static_cast<GLuint>(id);
    f(1, &u);
    return nullptr;
}
napi_value GetString(napi_env env, napi_callback_info info) {
    Args a(env, info, 1);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t pname, index = -1;
    if (!get_i64(env, a.v[0], &pname)) return fail(env, "getString(pname[, index])");
    if (a.n > 1) get_i64(env, a.v[1], &index);
    if (index >= 0) { FN(f, const GLubyte* (*)(GLenum, GLuint), "glGetStringi"); return mk_str(env, reinterpret_cast<const char*>(f(pname, index))); }
    FN(f, const GLubyte* (*)(GLenum), "glGetString");
    return mk_str(env, reinterpret_cast<const char*>(f(pname)));
}
// getIntegerv(pname, count) / getFloatv(pname, count) -> [numbers]
template <typename T>
napi_value GetV(napi_env env, napi_callback_info info, const char* entry) {
    Args a(env, info, 2);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t pname, count;
    if (!get_i64(env, a.v[0], &pname) || !get_i64(env, a.v[1], &count) || count < 1 || count > 16) return fail(env, "getv(pname, count)");
    FN(f, void (*)(GLenum, T*), entry);
    T out[16] = {};
    f(pname, out);
    napi_value arr; napi_create_array_with_length(env, count, &arr);
    for (int64_t i = 0; i < count; i++) napi_set_element(env, arr, i, mk_num(env, out[i]));
    return arr;
}
napi_value GetIntegerv(napi_env env, napi_callback_info info) { return GetV<GLint>(env, info, "glGetIntegerv"); }
napi_value GetFloatv(napi_env env, napi_callback_info info) { return GetV<GLfloat>(env, info, "glGetFloatv"); }

napi_value ShaderSource(napi_env env, napi_callback_info info) {
    Args a(env, info, 2);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t id; std::string src;
    if (!get_i64(env, a.v[0], &id) || !get_str(env, a.v[1], &src)) return fail(env, "shaderSource(shader, text)");
    FN(f, void (*)(GLuint, GLsizei, const char* const*, const GLint*), "glShaderSource");
    const char* p = src.c_str();
    f(id, 1, &p, nullptr);
    return nullptr;
}
// objectiv("Shader" | "Program", name, pname) -> integer
napi_value Objectiv(napi_env env, napi_callback_info info) {
    Args a(env, info, 3);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string kind; int64_t id, pname;
    if (!get_str(env, a.v[0], &kind) || !get_i64(env, a.v[1], &id) || !get_i64(env, a.v[2], &pname)) return fail(env, "objectiv(kind, name, pname)");
    FN(f, void (*)(GLuint, GLenum, GLint*), ("glGet" + kind + "iv").c_str());
    GLint v = 0;
    f(id, pname, &v);
    return mk_num(env, v);
}
napi_value InfoLog(napi_env env, napi_callback_info info) {
    Args a(env, info, 2);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string kind; int64_t id;
    if (!get_str(env, a.v[0], &kind) || !get_i64(env, a.v[1], &id)) return fail(env, "infoLog(kind, name)");
    FN(f, void (*)(GLuint, GLsizei, GLsizei*, char*), ("glGet" + kind + "InfoLog").c_str());
    std::vector<char> buf(16384);
    GLsizei n = 0;
    f(id, static_cast<GLsizei>(buf.size()), &n, buf.data());
    return mk_str(env, std::string(buf.data(), n).c_str());
}
// active("Uniform" | "Attrib", program, index) -> {name, type, size}
napi_value Active(napi_env env, napi_callback_info info) {
    Args a(env, info, 3);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string kind; int64_t prog, index;
    if (!get_str(env, a.v[0], &kind) || !get_i64(env, a.v[1], &prog) || !get_i64(env, a.v[2], &index)) return fail(env, "active(kind, program, index)");
    FN(f, void (*)(GLuint, GLuint, GLsizei, GLsizei*, GLint*, GLenum*, char*), ("glGetActive" + kind).c_str());
    char name[512]; GLsizei len = 0; GLint size = 0; GLenum type = 0;
    f(prog, index, sizeof name, &len, &size, &type, name);
    napi_value o; napi_create_object(env, &o);
    napi_set_named_property(env, o, "name", mk_str(env, std::string(name, len).c_str()));
    napi_set_named_property(env, o, "type", mk_num(env, type));
    napi_set_named_property(env, o, "size", mk_num(env, size));
    return o;
}
// location("Uniform" | "Attrib", program, name) -> integer (-1: none)
napi_value Location(napi_env env, napi_callback_info info) {
    Args a(env, info, 3);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string kind, name; int64_t prog;
    if (!get_str(env, a.v[0], &kind) || !get_i64(env, a.v[1], &prog) || !get_str(env, a.v[2], &name)) return fail(env, "location(kind, program, name)");
    FN(f, GLint (*)(GLuint, const char*), ("glGet" + kind + "Location").c_str());
    return mk_num(env, f(prog, name.c_str()));
}
napi_value BindAttribLocation(napi_env env, napi_callback_info info) {
    Args a(env, info, 3);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t prog, index; std::string name;
    if (!get_i64(env, a.v[0], &prog) || !get_i64(env, a.v[1], &index) || !get_str(env, a.v[2], &name)) return fail(env, "bindAttribLocation(program, index, name)");
    FN(f, void (*)(GLuint, GLuint, const char*), "glBindAttribLocation");
    f(prog, index, name.c_str());
    return nullptr;
}
napi_value ShaderPrecisionFormat(napi_env env, napi_callback_info info) {
    Args a(env, info, 2);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t st, pt;
    if (!get_i64(env, a.v[0], &st) || !get_i64(env, a.v[1], &pt)) return fail(env, "shaderPrecisionFormat(shadertype, precisiontype)");
    FN(f, void (*)(GLenum, GLenum, GLint*, GLint*), "glGetShaderPrecisionFormat");
    GLint range[2] = {0, 0}, prec = 0;
    f(st, pt, range, &prec);
    napi_value arr; napi_create_array_with_length(env, 3, &arr);
    napi_set_element(env, arr, 0, mk_num(env, range[0])); napi_set_element(env, arr, 1, mk_num(env, range[1])); napi_set_element(env, arr, 2, mk_num(env, prec));
    return arr;
}
// texImage("glTexImage2D" | "glTexSubImage2D" | "glTexImage3D" | "glTexSubImage3D", [integer arguments in GL order without the pointer], data | null)
napi_value TexImage(napi_env env, napi_callback_info info) {
    Args a(env, info, 3);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string name; uint32_t ni = 0;
    if (!get_str(env, a.v[0], &name) || napi_get_array_length(env, a.v[1], &ni) != napi_ok || ni > 10) return fail(env, "texImage(entry, ints, data)");
    intptr_t iv[10] = {};
    for (uint32_t i = 0; i < ni; i++) { napi_value e; int64_t x; napi_get_element(env, a.v[1], i, &e); if (!get_i64(env, e, &x)) return fail(env, "texImage: integer expected"); iv[i] = static_cast<intptr_t>(x); }
    void* data; size_t bytes;
    if (!get_bytes(env, a.v[2], &data, &bytes)) return fail(env, "texImage: data must be a typed array or null");
    void* p = gl(name.c_str());
    if (!p) return fail(env, "GL entry point missing: " + name);
    if (ni == 8) reinterpret_cast<void (*)(intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, const void*)>(p)(iv[0], iv[1], iv[2], iv[3], iv[4], iv[5], iv[6], iv[7], data);
    else if (ni == 9) reinterpret_cast<void (*)(intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, const void*)>(p)(iv[0], iv[1], iv[2], iv[3], iv[4], iv[5], iv[6], iv[7], iv[8], data);
    else if (ni == 10) reinterpret_cast<void (*)(intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, intptr_t, const void*)>(p)(iv[0], iv[1], iv[2], iv[3], iv[4], iv[5], iv[6], iv[7], iv[8], iv[9], data);
    else return fail(env, "texImage: 8, 9 or 10 integer arguments");
    return nullptr;
}
// bufferData(target, typedArray | byteSize, usage);  bufferSubData(target, byteOffset, typedArray)
napi_value BufferData(napi_env env, napi_callback_info info) {
    Args a(env, info, 3);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t target, usage, size = 0;
    void* data; size_t bytes;
    if (!get_i64(env, a.v[0], &target) || !get_i64(env, a.v[2], &usage)) return fail(env, "bufferData(target, data | size, usage)");
    if (!get_bytes(env, a.v[1], &data, &bytes)) { if (!get_i64(env, a.v[1], &size)) return fail(env, "bufferData: data"); bytes = static_cast<size_t>(size); data = nullptr; }
    FN(f, void (*)(GLenum, GLsizeiptr, const void*, GLenum), "glBufferData");
    f(target, static_cast<GLsizeiptr>(bytes), data, usage);
    return nullptr;
}
napi_value BufferSubData(napi_env env, napi_callback_info info) {
    Args a(env, info, 3);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t target, offset;
    void* data; size_t bytes;
    if (!get_i64(env, a.v[0], &target) || !get_i64(env, a.v[1], &offset) || !get_bytes(env, a.v[2], &data, &bytes) || !data) return fail(env, "bufferSubData(target, offset, data)");
    FN(f, void (*)(GLenum, GLintptr, GLsizeiptr, const void*), "glBufferSubData");
    f(target, offset, static_cast<GLsizeiptr>(bytes), data);
    return nullptr;
}
// uniformv(entry, location, count, typedArray[, transpose]): glUniform{1234}{fi}v / glUniformMatrix{234}fv
napi_value Uniformv(napi_env env, napi_callback_info info) {
    Args a(env, info, 4);
    if (!a.ok) return nullptr;
    NEED_GL();
    std::string name; int64_t loc, count, transpose = 0;
    void* data; size_t bytes;
    if (!get_str(env, a.v[0], &name) || !get_i64(env, a.v[1], &loc) || !get_i64(env, a.v[2], &count) || !get_bytes(env, a.v[3], &data, &bytes) || !data) return fail(env, "uniformv(entry, location, count, data[, transpose])");
    void* p = gl(name.c_str());
    if (!p) return fail(env, "GL entry point missing: " + name);
    if (name.find("Matrix") != std::string::npos) {
        if (a.n > 4) get_i64(env, a.v[4], &transpose);
        reinterpret_cast<void (*)(GLint, GLsizei, GLboolean, const void*)>(p)(loc, count, transpose != 0, data);
    } else reinterpret_cast<void (*)(GLint, GLsizei, const void*)>(p)(loc, count, data);
    return nullptr;
}
napi_value DrawBuffers(napi_env env, napi_callback_info info) {
    Args a(env, info, 1);
    if (!a.ok) return nullptr;
    NEED_GL();
    uint32_t n = 0;
    if (napi_get_array_length(env, a.v[0], &n) != napi_ok || n > 16) return fail(env, "drawBuffers([enums])");
    GLenum bufs[16];
    for (uint32_t i = 0; i < n; i++) { napi_value e; int64_t x; napi_get_element(env, a.v[0], i, &e); if (!get_i64(env, e, &x)) return fail(env, "drawBuffers: enum expected"); bufs[i] = static_cast<GLenum>(x); }
    FN(f, void (*)(GLsizei, const GLenum*), "glDrawBuffers");
    f(n, bufs);
    return nullptr;
}
// readPixels(x, y, w, h, format, type, typedArray)
napi_value ReadPixels(napi_env env, napi_callback_info info) {
    Args a(env, info, 7);
    if (!a.ok) return nullptr;
    NEED_GL();
    int64_t v[6];
    for (int i = 0; i < 6; i++) if (!get_i64(env, a.v[i], &v[i])) return fail(env, "readPixels(x, y, w, h, format, type, data)");
    void* data; size_t bytes;
    if (!get_bytes(env, a.v[6], &data, &bytes) || !data) return fail(env, "readPixels: data must be a typed array");
    const size_t texel = (v[5] == GL_FLOAT ? 4u : v[5] == GL_UNSIGNED_BYTE ? 1u : 0u) * (v[4] == GL_RGBA ? 4u : v[4] == GL_RGB ? 3u : v[4] == GL_RED ? 1u : 0u);
    if (texel == 0 || bytes < static_cast<size_t>(v[2]) * static_cast<size_t>(v[3]) * texel) return fail(env, "readPixels: buffer too small or format not handled");
    FN(f, void (*)(GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void*), "glReadPixels");
    f(v[0], v[1], v[2], v[3], v[4], v[5], data);
    return nullptr;
}

napi_value ModuleInit(napi_env env, napi_value exports) {
    const napi_property_descriptor props[] = {
        {"init", nullptr, Init, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"call", nullptr, Call, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"gen", nullptr, Gen, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"del", nullptr, Del, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"getString", nullptr, GetString, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"getIntegerv", nullptr, GetIntegerv, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"getFloatv", nullptr, GetFloatv, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"shaderSource", nullptr, ShaderSource, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"objectiv", nullptr, Objectiv, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"infoLog", nullptr, InfoLog, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"active", nullptr, Active, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"location", nullptr, Location, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"bindAttribLocation", nullptr, BindAttribLocation, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"shaderPrecisionFormat", nullptr, ShaderPrecisionFormat, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"texImage", nullptr, TexImage, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"bufferData", nullptr, BufferData, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"bufferSubData", nullptr, BufferSubData, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"uniformv", nullptr, Uniformv, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"drawBuffers", nullptr, DrawBuffers, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"readPixels", nullptr, ReadPixels, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

}  // namespace

NAPI_MODULE(mesa_gl, ModuleInit)

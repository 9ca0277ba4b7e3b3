// mesa_gl.cc -- TEST INFRASTRUCTURE (golden-vector generation only; never loaded by the product or by any test).
//
// A minimal headless OpenGL ES 3.0 binding for Node (N-API) on top of Mesa's software rasteriser (swrast_dri.so: softpipe is what the goldens are recorded with; llvmpipe was rejected, see tests/golden/make_golden_gpu.sh), so that
// the REFERENCE's own GLSL passes (SoftbodyGPU.js:59-376) and its own pass scheduler
// (MultiTargetGPUComputationRenderer.js) can be executed in the build container, which has no GPU, no X server and
// no EGL: the context is created straight through the DRI software-rasteriser interface that libGLX/libEGL use
// internally (swrast_dri.so, <GL/internal/dri_interface.h>, both shipped in the image).
//
// It exposes exactly what a full-screen-quad "render to float texture" pass needs:
//   init() createTexture() uploadTexture() createProgram() activeUniforms() draw() readTexture()
// headless_renderer.mjs builds the subset of THREE.WebGLRenderer the reference calls on top of these.
//
// build: oracle/glsl_ref/build.sh  ->  oracle/_ref/mesa_gl.node   (git-ignored)
#include <node_api.h>

#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
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

#define GLFN(ret, name, ...) ret (*name)(__VA_ARGS__) = nullptr
GLFN(const GLubyte*, pGetString, GLenum);
GLFN(GLenum, pGetError, void);
GLFN(void, pGenTextures, GLsizei, GLuint*);
GLFN(void, pBindTexture, GLenum, GLuint);
GLFN(void, pTexParameteri, GLenum, GLenum, GLint);
GLFN(void, pTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void*);
GLFN(void, pTexSubImage2D, GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void*);
GLFN(void, pPixelStorei, GLenum, GLint);
GLFN(void, pActiveTexture, GLenum);
GLFN(void, pGenFramebuffers, GLsizei, GLuint*);
GLFN(void, pBindFramebuffer, GLenum, GLuint);
GLFN(void, pFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint);
GLFN(GLenum, pCheckFramebufferStatus, GLenum);
GLFN(void, pDrawBuffers, GLsizei, const GLenum*);
GLFN(void, pReadBuffer, GLenum);
GLFN(void, pReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void*);
GLFN(void, pViewport, GLint, GLint, GLsizei, GLsizei);
GLFN(void, pDisable, GLenum);
GLFN(void, pClearColor, GLfloat, GLfloat, GLfloat, GLfloat);
GLFN(void, pClear, GLbitfield);
GLFN(GLuint, pCreateShader, GLenum);
GLFN(void, pShaderSource, GLuint, GLsizei, const char* const*, const GLint*);
GLFN(void, pCompileShader, GLuint);
GLFN(void, pGetShaderiv, GLuint, GLenum, GLint*);
GLFN(void, pGetShaderInfoLog, GLuint, GLsizei, GLsizei*, char*);
GLFN(GLuint, pCreateProgram, void);
GLFN(void, pAttachShader, GLuint, GLuint);
GLFN(void, pBindAttribLocation, GLuint, GLuint, const char*);
GLFN(void, pLinkProgram, GLuint);
GLFN(void, pGetProgramiv, GLuint, GLenum, GLint*);
GLFN(void, pGetProgramInfoLog, GLuint, GLsizei, GLsizei*, char*);
GLFN(void, pUseProgram, GLuint);
GLFN(void, pGetActiveUniform, GLuint, GLuint, GLsizei, GLsizei*, GLint*, GLenum*, char*);
GLFN(GLint, pGetUniformLocation, GLuint, const char*);
GLFN(void, pUniform1fv, GLint, GLsizei, const GLfloat*);
GLFN(void, pUniform2fv, GLint, GLsizei, const GLfloat*);
GLFN(void, pUniform3fv, GLint, GLsizei, const GLfloat*);
GLFN(void, pUniform4fv, GLint, GLsizei, const GLfloat*);
GLFN(void, pUniform1iv, GLint, GLsizei, const GLint*);
GLFN(void, pUniformMatrix4fv, GLint, GLsizei, GLboolean, const GLfloat*);
GLFN(void, pGenBuffers, GLsizei, GLuint*);
GLFN(void, pBindBuffer, GLenum, GLuint);
GLFN(void, pBufferData, GLenum, GLsizeiptr, const void*, GLenum);
GLFN(void, pGenVertexArrays, GLsizei, GLuint*);
GLFN(void, pBindVertexArray, GLuint);
GLFN(void, pEnableVertexAttribArray, GLuint);
GLFN(void, pVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void*);
GLFN(void, pDrawElements, GLenum, GLsizei, GLenum, const void*);
GLFN(void, pFinish, void);

template <typename T>
bool load(T& fn, const char* name) {
    fn = reinterpret_cast<T>(g_gpa(name));
    if (!fn) g_err = std::string("GL entry point missing: ") + name;
    return fn != nullptr;
}

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
    bool ok = load(pGetString, "glGetString") && load(pGetError, "glGetError") && load(pGenTextures, "glGenTextures") &&
              load(pBindTexture, "glBindTexture") && load(pTexParameteri, "glTexParameteri") && load(pTexImage2D, "glTexImage2D") &&
              load(pTexSubImage2D, "glTexSubImage2D") && load(pPixelStorei, "glPixelStorei") && load(pActiveTexture, "glActiveTexture") &&
              load(pGenFramebuffers, "glGenFramebuffers") && load(pBindFramebuffer, "glBindFramebuffer") &&
              load(pFramebufferTexture2D, "glFramebufferTexture2D") && load(pCheckFramebufferStatus, "glCheckFramebufferStatus") &&
              load(pDrawBuffers, "glDrawBuffers") && load(pReadBuffer, "glReadBuffer") && load(pReadPixels, "glReadPixels") &&
              load(pViewport, "glViewport") && load(pDisable, "glDisable") && load(pClearColor, "glClearColor") && load(pClear, "glClear") &&
              load(pCreateShader, "glCreateShader") && load(pShaderSource, "glShaderSource") && load(pCompileShader, "glCompileShader") &&
              load(pGetShaderiv, "glGetShaderiv") && load(pGetShaderInfoLog, "glGetShaderInfoLog") && load(pCreateProgram, "glCreateProgram") &&
              load(pAttachShader, "glAttachShader") && load(pBindAttribLocation, "glBindAttribLocation") && load(pLinkProgram, "glLinkProgram") &&
              load(pGetProgramiv, "glGetProgramiv") && load(pGetProgramInfoLog, "glGetProgramInfoLog") && load(pUseProgram, "glUseProgram") &&
              load(pGetActiveUniform, "glGetActiveUniform") && load(pGetUniformLocation, "glGetUniformLocation") &&
              load(pUniform1fv, "glUniform1fv") && load(pUniform2fv, "glUniform2fv") && load(pUniform3fv, "glUniform3fv") &&
              load(pUniform4fv, "glUniform4fv") && load(pUniform1iv, "glUniform1iv") && load(pUniformMatrix4fv, "glUniformMatrix4fv") &&
              load(pGenBuffers, "glGenBuffers") && load(pBindBuffer, "glBindBuffer") && load(pBufferData, "glBufferData") &&
              load(pGenVertexArrays, "glGenVertexArrays") && load(pBindVertexArray, "glBindVertexArray") &&
              load(pEnableVertexAttribArray, "glEnableVertexAttribArray") && load(pVertexAttribPointer, "glVertexAttribPointer") &&
              load(pDrawElements, "glDrawElements") && load(pFinish, "glFinish");
    if (!ok) return false;
    const char* ext = reinterpret_cast<const char*>(pGetString(GL_EXTENSIONS));
    if (!ext || !strstr(ext, "GL_EXT_color_buffer_float")) { g_err = "GL_EXT_color_buffer_float missing (float render targets)"; return false; }
    g_ready = true;
    return true;
}

// ---- N-API helpers -------------------------------------------------------------------------------------------
#define NAPI_OK(call) do { if ((call) != napi_ok) { napi_throw_error(env, "MESA_GL", "N-API call failed: " #call); return nullptr; } } while (0)
napi_value fail(napi_env env, const std::string& m) { napi_throw_error(env, "MESA_GL", m.c_str()); return nullptr; }

bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
        napi_throw_error(env, "MESA_GL", "wrong number of arguments");
        return false;
    }
    return true;
}
bool get_u32(napi_env env, napi_value v, uint32_t* out) { return napi_get_value_uint32(env, v, out) == napi_ok; }
bool get_str(napi_env env, napi_value v, std::string* out) {
    size_t n = 0;
    if (napi_get_value_string_utf8(env, v, nullptr, 0, &n) != napi_ok) return false;
    out->resize(n);
    return napi_get_value_string_utf8(env, v, &(*out)[0], n + 1, &n) == napi_ok;
}
// typed array -> pointer + element count; *type receives the napi_typedarray_type
bool get_typed(napi_env env, napi_value v, napi_typedarray_type* type, void** data, size_t* len) {
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return false;
    napi_value ab; size_t off;
    return napi_get_typedarray_info(env, v, type, len, data, &ab, &off) == napi_ok;
}
napi_value mk_u32(napi_env env, uint32_t x) { napi_value v; napi_create_uint32(env, x, &v); return v; }
napi_value mk_str(napi_env env, const char* s) { napi_value v; napi_create_string_utf8(env, s ? s : "", NAPI_AUTO_LENGTH, &v); return v; }

GLuint g_fbo = 0, g_vao = 0, g_vbo = 0, g_ibo = 0;

// ---- exported functions --------------------------------------------------------------------------------------
napi_value Init(napi_env env, napi_callback_info) {
    if (!gl_init()) return fail(env, g_err);
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "vendor", mk_str(env, reinterpret_cast<const char*>(pGetString(GL_VENDOR))));
    napi_set_named_property(env, o, "renderer", mk_str(env, reinterpret_cast<const char*>(pGetString(GL_RENDERER))));
    napi_set_named_property(env, o, "version", mk_str(env, reinterpret_cast<const char*>(pGetString(GL_VERSION))));
    napi_set_named_property(env, o, "glsl", mk_str(env, reinterpret_cast<const char*>(pGetString(GL_SHADING_LANGUAGE_VERSION))));
    return o;
}

// createTexture(w, h, Float32Array | null) -> id     RGBA32F, NEAREST, CLAMP_TO_EDGE (DataTexture / render-target settings
// of MultiTargetGPUComputationRenderer.js:140-143,353-371,400-403)
napi_value CreateTexture(napi_env env, napi_callback_info info) {
    napi_value a[3];
    if (!get_args(env, info, 3, a)) return nullptr;
    if (!g_ready) return fail(env, "init() first");
    uint32_t w, h;
    if (!get_u32(env, a[0], &w) || !get_u32(env, a[1], &h)) return fail(env, "createTexture(w, h, data)");
    napi_typedarray_type t; void* data = nullptr; size_t len = 0;
    if (get_typed(env, a[2], &t, &data, &len)) {
        if (t != napi_float32_array || len != static_cast<size_t>(w) * h * 4) return fail(env, "data must be a Float32Array of w*h*4");
    } else data = nullptr;
    GLuint tex = 0;
    pGenTextures(1, &tex);
    pBindTexture(GL_TEXTURE_2D, tex);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    pPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    pTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, w, h, 0, GL_RGBA, GL_FLOAT, data);
    if (GLenum e = pGetError()) return fail(env, "glTexImage2D error " + std::to_string(e));
    return mk_u32(env, tex);
}

napi_value UploadTexture(napi_env env, napi_callback_info info) {
    napi_value a[4];
    if (!get_args(env, info, 4, a)) return nullptr;
    uint32_t tex, w, h;
    napi_typedarray_type t; void* data; size_t len;
    if (!get_u32(env, a[0], &tex) || !get_u32(env, a[1], &w) || !get_u32(env, a[2], &h) || !get_typed(env, a[3], &t, &data, &len) ||
        t != napi_float32_array || len != static_cast<size_t>(w) * h * 4)
        return fail(env, "uploadTexture(id, w, h, Float32Array)");
    pBindTexture(GL_TEXTURE_2D, tex);
    pPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    pTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, w, h, GL_RGBA, GL_FLOAT, data);
    if (GLenum e = pGetError()) return fail(env, "glTexSubImage2D error " + std::to_string(e));
    return nullptr;
}

GLuint compile(GLenum kind, const std::string& src, std::string* log) {
    GLuint s = pCreateShader(kind);
    const char* p = src.c_str();
    pShaderSource(s, 1, &p, nullptr);
    pCompileShader(s);
    GLint ok = 0;
    pGetShaderiv(s, GL_COMPILE_STATUS, &ok);
    if (!ok) {
        char buf[8192]; GLsizei n = 0;
        pGetShaderInfoLog(s, sizeof buf, &n, buf);
        *log = std::string(kind == GL_VERTEX_SHADER ? "vertex" : "fragment") + " shader: " + std::string(buf, n);
        return 0;
    }
    return s;
}

// createProgram(vertexGlsl, fragmentGlsl) -> id;  attribute 0 is "position" (three binds index0AttributeName the same way)
napi_value CreateProgram(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return nullptr;
    if (!g_ready) return fail(env, "init() first");
    std::string vs, fs, log;
    if (!get_str(env, a[0], &vs) || !get_str(env, a[1], &fs)) return fail(env, "createProgram(vs, fs)");
    GLuint v = compile(GL_VERTEX_SHADER, vs, &log);
    if (!v) return fail(env, log);
    GLuint f = compile(GL_FRAGMENT_SHADER, fs, &log);
    if (!f) return fail(env, log);
    GLuint p = pCreateProgram();
    pAttachShader(p, v);
    pAttachShader(p, f);
    pBindAttribLocation(p, 0, "position");
    pLinkProgram(p);
    GLint ok = 0;
    pGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) {
        char buf[8192]; GLsizei n = 0;
        pGetProgramInfoLog(p, sizeof buf, &n, buf);
        return fail(env, "link: " + std::string(buf, n));
    }
    return mk_u32(env, p);
}

// activeUniforms(program) -> [{name, type, size}]   (type is the GL enum, e.g. 0x1406 FLOAT, 0x8B5E SAMPLER_2D)
napi_value ActiveUniforms(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    uint32_t p;
    if (!get_u32(env, a[0], &p)) return fail(env, "activeUniforms(program)");
    GLint n = 0;
    pGetProgramiv(p, GL_ACTIVE_UNIFORMS, &n);
    napi_value arr; NAPI_OK(napi_create_array_with_length(env, n, &arr));
    for (GLint i = 0; i < n; i++) {
        char name[256]; GLsizei len = 0; GLint size = 0; GLenum type = 0;
        pGetActiveUniform(p, i, sizeof name, &len, &size, &type, name);
        napi_value o; NAPI_OK(napi_create_object(env, &o));
        napi_set_named_property(env, o, "name", mk_str(env, name));
        napi_set_named_property(env, o, "type", mk_u32(env, type));
        napi_set_named_property(env, o, "size", mk_u32(env, size));
        napi_set_element(env, arr, i, o);
    }
    return arr;
}

// draw({program, width, height, targets: [texId...], clear: [r,g,b,a] | null,
//       uniforms: [{name, kind: 'f1'|'f2'|'f3'|'f4'|'i1'|'m4'|'tex', data: Float32Array|Int32Array|Uint32Array(texture ids)}],
//       position: Float32Array(xyz...), index: Uint32Array})
napi_value Draw(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return nullptr;
    if (!g_ready) return fail(env, "init() first");
    napi_value v;
    uint32_t prog, w, h;
    if (napi_get_named_property(env, a[0], "program", &v) != napi_ok || !get_u32(env, v, &prog)) return fail(env, "draw: program");
    if (napi_get_named_property(env, a[0], "width", &v) != napi_ok || !get_u32(env, v, &w)) return fail(env, "draw: width");
    if (napi_get_named_property(env, a[0], "height", &v) != napi_ok || !get_u32(env, v, &h)) return fail(env, "draw: height");

    if (!g_fbo) { pGenFramebuffers(1, &g_fbo); pGenVertexArrays(1, &g_vao); pGenBuffers(1, &g_vbo); pGenBuffers(1, &g_ibo); }
    pBindFramebuffer(GL_FRAMEBUFFER, g_fbo);
    napi_value targets;
    uint32_t nt = 0;
    if (napi_get_named_property(env, a[0], "targets", &targets) != napi_ok || napi_get_array_length(env, targets, &nt) != napi_ok || nt < 1 || nt > 8)
        return fail(env, "draw: targets");
    GLenum bufs[8];
    for (uint32_t i = 0; i < 8; i++) {
        uint32_t tex = 0;
        if (i < nt) { napi_value e; napi_get_element(env, targets, i, &e); if (!get_u32(env, e, &tex)) return fail(env, "draw: target id"); }
        pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, tex, 0);
        bufs[i] = GL_COLOR_ATTACHMENT0 + i;
    }
    if (pCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail(env, "draw: framebuffer incomplete");
    pDrawBuffers(nt, bufs);
    pViewport(0, 0, w, h);
    pDisable(GL_BLEND); pDisable(GL_DEPTH_TEST); pDisable(GL_CULL_FACE); pDisable(GL_SCISSOR_TEST); pDisable(GL_DITHER);

    napi_value clear; napi_valuetype vt;
    if (napi_get_named_property(env, a[0], "clear", &clear) == napi_ok && napi_typeof(env, clear, &vt) == napi_ok && vt == napi_object) {
        double c[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < 4; i++) { napi_value e; napi_get_element(env, clear, i, &e); napi_get_value_double(env, e, &c[i]); }
        pClearColor(c[0], c[1], c[2], c[3]);
        pClear(GL_COLOR_BUFFER_BIT);
    }

    pUseProgram(prog);
    napi_value uniforms; uint32_t nu = 0;
    if (napi_get_named_property(env, a[0], "uniforms", &uniforms) != napi_ok || napi_get_array_length(env, uniforms, &nu) != napi_ok) return fail(env, "draw: uniforms");
    GLint unit = 0;
    for (uint32_t i = 0; i < nu; i++) {
        napi_value u, nv, kv, dv; std::string name, kind;
        napi_get_element(env, uniforms, i, &u);
        if (napi_get_named_property(env, u, "name", &nv) != napi_ok || !get_str(env, nv, &name) ||
            napi_get_named_property(env, u, "kind", &kv) != napi_ok || !get_str(env, kv, &kind) ||
            napi_get_named_property(env, u, "data", &dv) != napi_ok) return fail(env, "draw: uniform entry");
        napi_typedarray_type t; void* data; size_t len;
        if (!get_typed(env, dv, &t, &data, &len)) return fail(env, "draw: uniform data must be a typed array: " + name);
        const GLint loc = pGetUniformLocation(prog, name.c_str());
        if (loc < 0) return fail(env, "draw: no such active uniform: " + name);
        if (kind == "tex") {
            if (t != napi_uint32_array) return fail(env, "draw: tex uniform wants Uint32Array ids: " + name);
            std::vector<GLint> units(len);
            for (size_t k = 0; k < len; k++) {
                pActiveTexture(GL_TEXTURE0 + unit);
                pBindTexture(GL_TEXTURE_2D, static_cast<const uint32_t*>(data)[k]);
                units[k] = unit++;
            }
            pUniform1iv(loc, static_cast<GLsizei>(len), units.data());
        } else if (kind == "i1") {
            if (t != napi_int32_array) return fail(env, "draw: i1 wants Int32Array: " + name);
            pUniform1iv(loc, static_cast<GLsizei>(len), static_cast<const GLint*>(data));
        } else {
            if (t != napi_float32_array) return fail(env, "draw: float uniform wants Float32Array: " + name);
            const GLfloat* f = static_cast<const GLfloat*>(data);
            if (kind == "f1") pUniform1fv(loc, static_cast<GLsizei>(len), f);
            else if (kind == "f2") pUniform2fv(loc, static_cast<GLsizei>(len / 2), f);
            else if (kind == "f3") pUniform3fv(loc, static_cast<GLsizei>(len / 3), f);
            else if (kind == "f4") pUniform4fv(loc, static_cast<GLsizei>(len / 4), f);
            else if (kind == "m4") pUniformMatrix4fv(loc, static_cast<GLsizei>(len / 16), GL_FALSE, f);
            else return fail(env, "draw: unknown uniform kind " + kind);
        }
    }

    napi_value pv, iv; napi_typedarray_type t; void* pos; size_t npos; void* idx; size_t nidx;
    if (napi_get_named_property(env, a[0], "position", &pv) != napi_ok || !get_typed(env, pv, &t, &pos, &npos) || t != napi_float32_array) return fail(env, "draw: position");
    if (napi_get_named_property(env, a[0], "index", &iv) != napi_ok || !get_typed(env, iv, &t, &idx, &nidx) || t != napi_uint32_array) return fail(env, "draw: index");
    pBindVertexArray(g_vao);
    pBindBuffer(GL_ARRAY_BUFFER, g_vbo);
    pBufferData(GL_ARRAY_BUFFER, npos * sizeof(float), pos, GL_STREAM_DRAW);
    pEnableVertexAttribArray(0);
    pVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 0, nullptr);
    pBindBuffer(GL_ELEMENT_ARRAY_BUFFER, g_ibo);
    pBufferData(GL_ELEMENT_ARRAY_BUFFER, nidx * sizeof(uint32_t), idx, GL_STREAM_DRAW);
    pDrawElements(GL_TRIANGLES, static_cast<GLsizei>(nidx), GL_UNSIGNED_INT, nullptr);
    pFinish();
    if (GLenum e = pGetError()) return fail(env, "draw: GL error " + std::to_string(e));
    for (uint32_t i = 0; i < nt; i++) pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + i, GL_TEXTURE_2D, 0, 0);
    return nullptr;
}

// readTexture(id, x, y, w, h, Float32Array out)
napi_value ReadTexture(napi_env env, napi_callback_info info) {
    napi_value a[6];
    if (!get_args(env, info, 6, a)) return nullptr;
    uint32_t tex, x, y, w, h;
    napi_typedarray_type t; void* data; size_t len;
    if (!get_u32(env, a[0], &tex) || !get_u32(env, a[1], &x) || !get_u32(env, a[2], &y) || !get_u32(env, a[3], &w) || !get_u32(env, a[4], &h) ||
        !get_typed(env, a[5], &t, &data, &len) || t != napi_float32_array || len < static_cast<size_t>(w) * h * 4)
        return fail(env, "readTexture(id, x, y, w, h, Float32Array)");
    if (!g_fbo) { pGenFramebuffers(1, &g_fbo); pGenVertexArrays(1, &g_vao); pGenBuffers(1, &g_vbo); pGenBuffers(1, &g_ibo); }
    pBindFramebuffer(GL_FRAMEBUFFER, g_fbo);
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    if (pCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail(env, "readTexture: framebuffer incomplete");
    pReadBuffer(GL_COLOR_ATTACHMENT0);
    pPixelStorei(GL_PACK_ALIGNMENT, 1);
    pReadPixels(x, y, w, h, GL_RGBA, GL_FLOAT, data);
    if (GLenum e = pGetError()) return fail(env, "readTexture: GL error " + std::to_string(e));
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, 0, 0);
    return nullptr;
}

napi_value ModuleInit(napi_env env, napi_value exports) {
    const napi_property_descriptor props[] = {
        {"init", nullptr, Init, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"createTexture", nullptr, CreateTexture, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"uploadTexture", nullptr, UploadTexture, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"createProgram", nullptr, CreateProgram, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"activeUniforms", nullptr, ActiveUniforms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"draw", nullptr, Draw, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"readTexture", nullptr, ReadTexture, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

}  // namespace

NAPI_MODULE(mesa_gl, ModuleInit)

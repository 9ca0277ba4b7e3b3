// webgl2_context.mjs -- TEST INFRASTRUCTURE (golden-vector generation only).
//
// A WebGL2RenderingContext-shaped object on top of the raw OpenGL ES 3.0 entry points of oracle/_ref/mesa_gl.node (mesa_gl.cc; Mesa
// softpipe), and the canvas stand-in THREE.WebGLRenderer wants beside it.  With these two the reference's vendored, UNMODIFIED three.js
// renderer is what drives the GL: `new THREE.WebGLRenderer({ canvas, context })` -- program assembly, texture uploads, render-target
// set-up, uniform upload, state caching are all three r160's own code (/root/reference/node_modules/three/build/three.module.js), exactly
// as in the browser where the reference hands `world.renderer` to its solver (/root/reference/src/SoftbodyGPU.js:9,379-380;
// MultiTargetGPUComputationRenderer.js:468-475).
//
// What is here is the WebGL <-> OpenGL ES glue a browser keeps inside: object handles, the getParameter / getExtension tables, the three
// WebGL-only pixel-store parameters.  It holds no knowledge of three.js, of the reference or of the simulation.  Only the methods three
// r160 calls for this scene are provided; anything else THROWS by name (a Proxy), so nothing is silently ignored.
import { createRequire } from 'module';
import fs from 'fs';
import path from 'path';

class GLObject { constructor(kind, id) { this.kind = kind; this.id = id; } }
const idOf = o => (o === null || o === undefined ? 0 : o.id);

export function createWebGL2Context(addonPath, width = 16, height = 16) {
    const raw = createRequire(import.meta.url)(addonPath);
    const info = raw.init();
    const C = JSON.parse(fs.readFileSync(path.join(path.dirname(addonPath), 'gl_constants.json'), 'utf8'));
    // WebGL-only enumerants (WebGL 1.0 spec 5.14, WebGL 2.0 spec 3.7)
    const WEBGL = { UNPACK_FLIP_Y_WEBGL: 0x9240, UNPACK_PREMULTIPLY_ALPHA_WEBGL: 0x9241, CONTEXT_LOST_WEBGL: 0x9242, UNPACK_COLORSPACE_CONVERSION_WEBGL: 0x9243,
        BROWSER_DEFAULT_WEBGL: 0x9244, MAX_CLIENT_WAIT_TIMEOUT_WEBGL: 0x9247 };
    const call = (name, ints = [], floats = []) => raw.call(name, ints, floats);
    const stats = { draws: 0, programs: 0, texUploads: 0 };

    // what getParameter returns by kind (everything else: one integer)
    const STRINGS = new Set([C.VENDOR, C.RENDERER, C.VERSION, C.SHADING_LANGUAGE_VERSION]);
    const INT4 = new Set([C.VIEWPORT, C.SCISSOR_BOX]);
    const INT2 = new Set([C.MAX_VIEWPORT_DIMS]);
    const FLOAT2 = new Set([C.ALIASED_LINE_WIDTH_RANGE, C.ALIASED_POINT_SIZE_RANGE, C.DEPTH_RANGE]);
    const FLOAT4 = new Set([C.COLOR_CLEAR_VALUE, C.BLEND_COLOR]);
    const FLOAT1 = new Set([C.DEPTH_CLEAR_VALUE, C.LINE_WIDTH, C.POLYGON_OFFSET_FACTOR, C.POLYGON_OFFSET_UNITS, C.SAMPLE_COVERAGE_VALUE, C.MAX_TEXTURE_MAX_ANISOTROPY_EXT, C.MAX_TEXTURE_LOD_BIAS]);
    const glExtensions = new Set();
    { const n = raw.getIntegerv(C.NUM_EXTENSIONS, 1)[0]; for (let i = 0; i < n; i++) glExtensions.add(raw.getString(C.EXTENSIONS, i)); }
    // WebGL extension name -> the GL ES extension that backs it; the object handed out carries the extension's enumerants
    const EXTENSIONS = {
        EXT_color_buffer_float: ['GL_EXT_color_buffer_float', {}],
        OES_texture_float_linear: ['GL_OES_texture_float_linear', {}],
        EXT_texture_filter_anisotropic: ['GL_EXT_texture_filter_anisotropic', { TEXTURE_MAX_ANISOTROPY_EXT: C.TEXTURE_MAX_ANISOTROPY_EXT, MAX_TEXTURE_MAX_ANISOTROPY_EXT: C.MAX_TEXTURE_MAX_ANISOTROPY_EXT }],
    };

    const methods = {
        // ---- context --------------------------------------------------------------------------------------------------------------------------
        getContextAttributes: () => ({ alpha: true, antialias: false, depth: true, stencil: false, premultipliedAlpha: true, preserveDrawingBuffer: false,
            powerPreference: 'default', failIfMajorPerformanceCaveat: false, desynchronized: false }),
        isContextLost: () => false,
        getSupportedExtensions: () => Object.keys(EXTENSIONS).filter(n => glExtensions.has(EXTENSIONS[n][0])),
        getExtension(name) { const e = EXTENSIONS[name]; return e && glExtensions.has(e[0]) ? e[1] : null; },
        getParameter(p) {
            if (STRINGS.has(p)) {   // WebGL 2.0 spec 3.7.2: VERSION / SHADING_LANGUAGE_VERSION start with "WebGL 2.0" / "WebGL GLSL ES 3.00"
                const s = raw.getString(p);
                return p === C.VERSION ? 'WebGL 2.0 (' + s + ')' : p === C.SHADING_LANGUAGE_VERSION ? 'WebGL GLSL ES 3.00 (' + s + ')' : s;
            }
            if (INT4.has(p)) return Int32Array.from(raw.getIntegerv(p, 4));
            if (INT2.has(p)) return Int32Array.from(raw.getIntegerv(p, 2));
            if (FLOAT4.has(p)) return Float32Array.from(raw.getFloatv(p, 4));
            if (FLOAT2.has(p)) return Float32Array.from(raw.getFloatv(p, 2));
            if (FLOAT1.has(p)) return raw.getFloatv(p, 1)[0];
            if (p === WEBGL.UNPACK_FLIP_Y_WEBGL || p === WEBGL.UNPACK_PREMULTIPLY_ALPHA_WEBGL) return false;
            if (p === WEBGL.UNPACK_COLORSPACE_CONVERSION_WEBGL) return WEBGL.BROWSER_DEFAULT_WEBGL;
            if (p === undefined) throw new Error('getParameter(undefined)');
            return raw.getIntegerv(p, 1)[0];
        },
        getShaderPrecisionFormat(st, pt) { const [rangeMin, rangeMax, precision] = raw.shaderPrecisionFormat(st, pt); return { rangeMin, rangeMax, precision }; },
        getError: () => call('glGetError'),
        finish: () => { call('glFinish'); },
        flush: () => { call('glFlush'); },

        // ---- state ----------------------------------------------------------------------------------------------------------------------------
        enable: cap => { call('glEnable', [cap]); },
        disable: cap => { call('glDisable', [cap]); },
        isEnabled: cap => call('glIsEnabled', [cap]) !== 0,
        viewport: (x, y, w, h) => { call('glViewport', [x, y, w, h]); },
        scissor: (x, y, w, h) => { call('glScissor', [x, y, w, h]); },
        clearColor: (r, g, b, a) => { call('glClearColor', [], [r, g, b, a]); },
        clearDepth: d => { call('glClearDepthf', [], [d]); },
        clearStencil: s => { call('glClearStencil', [s]); },
        clear: mask => { call('glClear', [mask]); },
        colorMask: (r, g, b, a) => { call('glColorMask', [r, g, b, a]); },
        depthMask: f => { call('glDepthMask', [f]); },
        depthFunc: f => { call('glDepthFunc', [f]); },
        depthRange: (n, f) => { call('glDepthRangef', [], [n, f]); },
        stencilMask: m => { call('glStencilMask', [m]); },
        stencilFunc: (f, r, m) => { call('glStencilFunc', [f, r, m]); },
        stencilOp: (a, b, c) => { call('glStencilOp', [a, b, c]); },
        frontFace: m => { call('glFrontFace', [m]); },
        cullFace: m => { call('glCullFace', [m]); },
        lineWidth: w => { call('glLineWidth', [], [w]); },
        polygonOffset: (f, u) => { call('glPolygonOffset', [], [f, u]); },
        blendEquation: m => { call('glBlendEquation', [m]); },
        blendEquationSeparate: (a, b) => { call('glBlendEquationSeparate', [a, b]); },
        blendFunc: (s, d) => { call('glBlendFunc', [s, d]); },
        blendFuncSeparate: (a, b, c, d) => { call('glBlendFuncSeparate', [a, b, c, d]); },
        blendColor: (r, g, b, a) => { call('glBlendColor', [], [r, g, b, a]); },
        pixelStorei(p, v) {
            // the three WebGL-only parameters have no GL counterpart: a browser applies them while it converts DOM sources.  Typed-array
            // uploads are all this context sees; flipping or premultiplying them would have to be done here -- not needed, so refused.
            if (p === WEBGL.UNPACK_FLIP_Y_WEBGL || p === WEBGL.UNPACK_PREMULTIPLY_ALPHA_WEBGL) { if (v) throw new Error('pixelStorei: flipY / premultiplyAlpha uploads are not implemented'); return; }
            if (p === WEBGL.UNPACK_COLORSPACE_CONVERSION_WEBGL) return;
            call('glPixelStorei', [p, v]);
        },

        // ---- textures -------------------------------------------------------------------------------------------------------------------------
        createTexture: () => new GLObject('Textures', raw.gen('Textures')),
        deleteTexture: t => { if (t) raw.del('Textures', t.id); },
        bindTexture: (target, t) => { call('glBindTexture', [target, idOf(t)]); },
        activeTexture: u => { call('glActiveTexture', [u]); },
        texParameteri: (t, p, v) => { call('glTexParameteri', [t, p, v]); },
        texParameterf: (t, p, v) => { call('glTexParameterf', [t, p], [v]); },
        generateMipmap: t => { call('glGenerateMipmap', [t]); },
        texStorage2D: (t, l, f, w, h) => { call('glTexStorage2D', [t, l, f, w, h]); },
        texStorage3D: (t, l, f, w, h, d) => { call('glTexStorage3D', [t, l, f, w, h, d]); },
        texImage2D(target, level, ifmt, w, h, border, fmt, type, data) {
            if (arguments.length !== 9) throw new Error('texImage2D: only the typed-array form is implemented');
            raw.texImage('glTexImage2D', [target, level, ifmt, w, h, border, fmt, type], data === undefined ? null : data); stats.texUploads++;
        },
        texSubImage2D(target, level, x, y, w, h, fmt, type, data) {
            if (arguments.length !== 9) throw new Error('texSubImage2D: only the typed-array form is implemented');
            raw.texImage('glTexSubImage2D', [target, level, x, y, w, h, fmt, type], data); stats.texUploads++;
        },
        texImage3D(target, level, ifmt, w, h, d, border, fmt, type, data) { raw.texImage('glTexImage3D', [target, level, ifmt, w, h, d, border, fmt, type], data === undefined ? null : data); },

        // ---- framebuffers ---------------------------------------------------------------------------------------------------------------------
        createFramebuffer: () => new GLObject('Framebuffers', raw.gen('Framebuffers')),
        deleteFramebuffer: f => { if (f) raw.del('Framebuffers', f.id); },
        bindFramebuffer: (target, f) => { call('glBindFramebuffer', [target, idOf(f)]); },
        framebufferTexture2D: (target, att, textarget, t, level) => { call('glFramebufferTexture2D', [target, att, textarget, idOf(t), level]); },
        createRenderbuffer: () => new GLObject('Renderbuffers', raw.gen('Renderbuffers')),
        deleteRenderbuffer: r => { if (r) raw.del('Renderbuffers', r.id); },
        bindRenderbuffer: (target, r) => { call('glBindRenderbuffer', [target, idOf(r)]); },
        renderbufferStorage: (t, f, w, h) => { call('glRenderbufferStorage', [t, f, w, h]); },
        framebufferRenderbuffer: (t, a, rt, r) => { call('glFramebufferRenderbuffer', [t, a, rt, idOf(r)]); },
        checkFramebufferStatus: t => call('glCheckFramebufferStatus', [t]),
        drawBuffers: list => { raw.drawBuffers(Array.from(list)); },
        readBuffer: b => { call('glReadBuffer', [b]); },
        readPixels(x, y, w, h, fmt, type, data) { if (!ArrayBuffer.isView(data)) throw new Error('readPixels: only the typed-array form is implemented'); raw.readPixels(x, y, w, h, fmt, type, data); },

        // ---- shaders and programs -------------------------------------------------------------------------------------------------------------
        createShader: type => new GLObject('Shader', call('glCreateShader', [type])),
        shaderSource: (s, text) => { raw.shaderSource(s.id, text); },
        compileShader: s => { call('glCompileShader', [s.id]); },
        getShaderParameter(s, p) { const v = raw.objectiv('Shader', s.id, p); return p === C.SHADER_TYPE ? v : v !== 0; },
        getShaderInfoLog: s => raw.infoLog('Shader', s.id),
        deleteShader: s => { if (s) call('glDeleteShader', [s.id]); },
        createProgram: () => { stats.programs++; return new GLObject('Program', call('glCreateProgram')); },
        attachShader: (p, s) => { call('glAttachShader', [p.id, s.id]); },
        bindAttribLocation: (p, i, name) => { raw.bindAttribLocation(p.id, i, name); },
        linkProgram: p => { call('glLinkProgram', [p.id]); },
        getProgramParameter(p, pname) {
            const v = raw.objectiv('Program', p.id, pname);
            return (pname === C.LINK_STATUS || pname === C.DELETE_STATUS || pname === C.VALIDATE_STATUS) ? v !== 0 : v;
        },
        getProgramInfoLog: p => raw.infoLog('Program', p.id),
        useProgram: p => { call('glUseProgram', [idOf(p)]); },
        deleteProgram: p => { if (p) call('glDeleteProgram', [p.id]); },
        getActiveUniform: (p, i) => raw.active('Uniform', p.id, i),
        getActiveAttrib: (p, i) => raw.active('Attrib', p.id, i),
        getUniformLocation(p, name) { const l = raw.location('Uniform', p.id, name); return l < 0 ? null : new GLObject('UniformLocation', l); },
        getAttribLocation: (p, name) => raw.location('Attrib', p.id, name),
        uniform1f: (l, x) => { if (l) call('glUniform1f', [l.id], [x]); },
        uniform2f: (l, x, y) => { if (l) call('glUniform2f', [l.id], [x, y]); },
        uniform3f: (l, x, y, z) => { if (l) call('glUniform3f', [l.id], [x, y, z]); },
        uniform4f: (l, x, y, z, w) => { if (l) call('glUniform4f', [l.id], [x, y, z, w]); },
        uniform1i: (l, x) => { if (l) call('glUniform1i', [l.id, x]); },
        uniform1ui: (l, x) => { if (l) call('glUniform1ui', [l.id, x]); },
        uniform1fv: (l, v) => { if (l) raw.uniformv('glUniform1fv', l.id, v.length, Float32Array.from(v)); },
        uniform2fv: (l, v) => { if (l) raw.uniformv('glUniform2fv', l.id, v.length / 2, Float32Array.from(v)); },
        uniform3fv: (l, v) => { if (l) raw.uniformv('glUniform3fv', l.id, v.length / 3, Float32Array.from(v)); },
        uniform4fv: (l, v) => { if (l) raw.uniformv('glUniform4fv', l.id, v.length / 4, Float32Array.from(v)); },
        uniform1iv: (l, v) => { if (l) raw.uniformv('glUniform1iv', l.id, v.length, Int32Array.from(v)); },
        uniformMatrix3fv: (l, tr, v) => { if (l) raw.uniformv('glUniformMatrix3fv', l.id, v.length / 9, Float32Array.from(v), tr ? 1 : 0); },
        uniformMatrix4fv: (l, tr, v) => { if (l) raw.uniformv('glUniformMatrix4fv', l.id, v.length / 16, Float32Array.from(v), tr ? 1 : 0); },

        // ---- buffers, vertex arrays, draws ----------------------------------------------------------------------------------------------------
        createBuffer: () => new GLObject('Buffers', raw.gen('Buffers')),
        deleteBuffer: b => { if (b) raw.del('Buffers', b.id); },
        bindBuffer: (target, b) => { call('glBindBuffer', [target, idOf(b)]); },
        bufferData(target, data, usage) { raw.bufferData(target, typeof data === 'number' ? data : ArrayBuffer.isView(data) ? data : new Uint8Array(data), usage); },
        bufferSubData(target, offset, data) { raw.bufferSubData(target, offset, ArrayBuffer.isView(data) ? data : new Uint8Array(data)); },
        createVertexArray: () => new GLObject('VertexArrays', raw.gen('VertexArrays')),
        deleteVertexArray: v => { if (v) raw.del('VertexArrays', v.id); },
        bindVertexArray: v => { call('glBindVertexArray', [idOf(v)]); },
        enableVertexAttribArray: i => { call('glEnableVertexAttribArray', [i]); },
        disableVertexAttribArray: i => { call('glDisableVertexAttribArray', [i]); },
        vertexAttribPointer: (i, size, type, norm, stride, offset) => { call('glVertexAttribPointer', [i, size, type, norm, stride, offset]); },
        vertexAttribIPointer: (i, size, type, stride, offset) => { call('glVertexAttribIPointer', [i, size, type, stride, offset]); },
        vertexAttribDivisor: (i, d) => { call('glVertexAttribDivisor', [i, d]); },
        drawElements: (mode, count, type, offset) => { stats.draws++; call('glDrawElements', [mode, count, type, offset]); },
        drawArrays: (mode, first, count) => { stats.draws++; call('glDrawArrays', [mode, first, count]); },
    };

    // three.js asks `gl.constructor.name === 'WebGL2RenderingContext'` (three.module.js WebGLCapabilities) and wants the global to exist
    class WebGL2RenderingContext {}
    if (typeof globalThis.WebGL2RenderingContext === 'undefined') globalThis.WebGL2RenderingContext = WebGL2RenderingContext;
    const target = new WebGL2RenderingContext();
    for (const [k, v] of Object.entries(C)) Object.defineProperty(target, k, { value: v, enumerable: false });
    for (const [k, v] of Object.entries(WEBGL)) Object.defineProperty(target, k, { value: v, enumerable: false });
    for (const [k, v] of Object.entries(methods)) Object.defineProperty(target, k, { value: v, enumerable: false });
    Object.defineProperty(target, 'drawingBufferWidth', { value: width });
    Object.defineProperty(target, 'drawingBufferHeight', { value: height });
    const canvas = {   // what THREE.WebGLRenderer touches of an HTMLCanvasElement when it is handed a context
        width, height, style: {},
        addEventListener() {}, removeEventListener() {},
        getContext() { return gl; },
    };
    Object.defineProperty(target, 'canvas', { value: canvas });
    const gl = new Proxy(target, {
        get(t, prop) {
            if (prop in t || typeof prop === 'symbol') return t[prop];
            if (prop === 'then' || prop === 'toJSON') return undefined;
            throw new Error('webgl2_context: WebGL2RenderingContext.' + String(prop) + ' is not provided (three.js was not expected to use it for this scene)');
        },
    });
    return { gl, canvas, info, stats };
}

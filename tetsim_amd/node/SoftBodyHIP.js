'use strict';
// SoftBodyHIP.js -- Node.js twin of the reference's solver objects, backed by libtetsim_hip.so through the N-API
// shim (tetsim_napi.cc -> include/tetsim.h).  Drop-in for
//     new SoftBody(vertices, tetIds, tetEdgeIds, physicsParams, visVerts, visTriIds, visMaterial)          (Softbody.js:4-5)
//     new SoftBodyGPU(vertices, tetIds, tetEdgeIds, physicsParams, visVerts, visTriIds, visMaterial, world)  (SoftbodyGPU.js:5-6)
// in the driver loop of main.js:52-96: `.simulate(dt, physicsParams)` per substep, `.endFrame()` per frame,
// `.startGrab/.moveGrabbed/.endGrab`, fields `.edgeMesh .visMesh .pos .grabId .grabPos .numParticles .numElems .volError`.
//
// three.js is INJECTED, never bundled: pass it as `world.THREE` (or set SoftBodyHIP.THREE = require('three')).  Without it
// the body is headless (physics only; edgeMesh / visMesh stay null).
//
// Which reference solver is mirrored is chosen by `physicsParams.tetsim` (optional):
//     { solver: 'polar' | 'neohookean', precision: 'precise' | 'fast', order: 'original' | 'coloured' | 'clustered', device: 0,
//       refSlotTable: true, refFixedBounds: true, refGrabTexel: false, gather: false, constantRestShape: false, leanState: false,   (include/tetsim.h flags)
//       partCount: 1, partIndex: 0, vertOwner: Int32Array,    (one Node process per GPU: see commUniqueId / commInit below)
//       refStartGrab: false }   (true: startGrab searches the edge-mesh copy of the positions, exactly as SoftbodyGPU.js:692-704)
// default: polar + precise, i.e. SoftBodyGPU's algorithm with reference-order arithmetic.
const path = require('path');

const SOLVER = { polar: 0, neohookean: 1 };
const PRECISION = { precise: 0, fast: 1 };
const ORDER = { original: 0, coloured: 1, clustered: 2 };
const FLAG_REF_SLOT_TABLE = 1, FLAG_REF_FIXED_BOUNDS = 2, FLAG_GATHER_FORMULATION = 4, FLAG_CONSTANT_REST_SHAPE = 8, FLAG_REF_GRAB_TEXEL = 16,
    FLAG_REF_ROTATION_EXIT = 64, FLAG_LEAN_STATE = 128;

let addon = null;
function loadTetSim(libPath) {
    if (addon) return addon;
    addon = require(path.join(__dirname, 'tetsim_napi.node'));
    addon.load(libPath || process.env.TETSIM_HIP_LIB || path.join(__dirname, '..', 'libtetsim_hip.so'));
    return addon;
}

class SoftBodyHIP {
    constructor(vertices, tetIds, tetEdgeIds, physicsParams, visVerts, visTriIds, visMaterial, world, _meshFile) {
        const api = loadTetSim();
        this.physicsParams = physicsParams || {};
        const opt = this.physicsParams.tetsim || {};
        this.numParticles = vertices.length / 3;   // Softbody.js:9
        this.numElems = tetIds.length / 4;         // Softbody.js:10
        this.tetIds = tetIds;                      // held by reference, as the reference does (Softbody.js:19)
        this.pos = Float32Array.from(vertices);    // vertices.slice(0), Softbody.js:12
        this.grabPos = new Float32Array(3);
        this.grabId = -1;
        this.volError = 0.0;
        this._api = api;
        this._solver = opt.solver || 'polar';
        const verts32 = vertices instanceof Float32Array ? vertices : Float32Array.from(vertices);
        const tets32 = tetIds instanceof Int32Array ? tetIds : Int32Array.from(tetIds);
        const createOptions = {
            solver: SOLVER[this._solver], precision: PRECISION[opt.precision || 'precise'], order: ORDER[opt.order || 'original'],
            flags: (opt.refSlotTable === false ? 0 : FLAG_REF_SLOT_TABLE) | (opt.refFixedBounds === false ? 0 : FLAG_REF_FIXED_BOUNDS) |
                   (opt.gather ? FLAG_GATHER_FORMULATION : 0) | (opt.constantRestShape ? FLAG_CONSTANT_REST_SHAPE : 0) |
                   (opt.refGrabTexel ? FLAG_REF_GRAB_TEXEL : 0) | (opt.refRotationExit ? FLAG_REF_ROTATION_EXIT : 0) | (opt.leanState ? FLAG_LEAN_STATE : 0),
            device: opt.device || 0,
            partCount: opt.partCount || 1, partIndex: opt.partIndex || 0,
            density: this.physicsParams.density === undefined ? 1000.0 : this.physicsParams.density,
        };
        if (opt.vertOwner) createOptions.vertOwner = opt.vertOwner instanceof Int32Array ? opt.vertOwner : Int32Array.from(opt.vertOwner);
        // fromFile: the library maps the .tetsim container itself (and picks up a stored colouring / visual mesh);
        // batch: `_meshFile` carries the list of bodies, vertices / tetIds are their concatenation
        if (_meshFile && _meshFile.batch) {
            this._h = api.createBatch(_meshFile.batch.map(b => b.vertices instanceof Float32Array ? b.vertices : Float32Array.from(b.vertices)),
                                      _meshFile.batch.map(b => b.tetIds instanceof Int32Array ? b.tetIds : Int32Array.from(b.tetIds)), createOptions);
            _meshFile = null;
        } else this._h = _meshFile ? api.createFromFile(_meshFile, createOptions) : api.create(verts32, tets32, createOptions);
        this._dirty = false;
        this._visOnDevice = false;

        // display objects (Softbody.js:36-57 / SoftbodyGPU.js:415-461), only when three.js was injected
        const THREE = (world && world.THREE) || SoftBodyHIP.THREE || null;
        this.edgeMesh = null;
        this.visMesh = null;
        this.visVerts = visVerts || new Float32Array(0);
        this.numVisVerts = this.visVerts.length / 4;
        this._partitioned = createOptions.partCount > 1;
        // The embedded mesh is skinned on the device (SURVEY.md §8(f)-1; createFromFile attached it already).  A PARTITION takes the same
        // list and keeps the rows whose tet it owns (visualIds()): readVisualPositions() then returns ITS rows, and the ranks' outputs,
        // scattered by row number (scatterVisualPositions), are the whole mesh -- SoftbodyGPU.js:424-448 skins every vertex every frame.
        if (this.numVisVerts > 0) {
            if (!_meshFile) api.setVisualMesh(this._h, this.visVerts instanceof Float32Array ? this.visVerts : Float32Array.from(this.visVerts), null);
            this._visOnDevice = true;
            this._visIds = this._partitioned ? api.visualIds(this._h) : null;
            // ... and its vertex normals too: Softbody.js:273 runs geometry.computeVertexNormals() every frame (37 ms of the
            // CPU path's frame); the device reproduces three.js's result bit for bit (tetsim_read_visual_vertex_normals; a triangle's
            // corners may be skinned by different partitions, so unpartitioned bodies only)
            // (a PARTITION takes the same global triangle list and computes ITS rows' normals from the ranks' skins put together:
            // readVisualVertexNormalsFrom below)
            this._visTris = visTriIds && visTriIds.length ? (visTriIds instanceof Int32Array ? visTriIds : Int32Array.from(visTriIds)) : null;
            if (this._visTris) api.setVisualTriangles(this._h, this._visTris);
        }
        if (THREE) {
            let geometry = new THREE.BufferGeometry();
            // the reference aliases the caller's `vertices` array here and overwrites it every frame (Softbody.js:37,252)
            geometry.setAttribute('position', new THREE.BufferAttribute(vertices, 3));
            geometry.setIndex(tetEdgeIds);
            this.edgeMesh = new THREE.LineSegments(geometry);
            this.edgeMesh.userData = this;   // for raycasting
            this.edgeMesh.layers.enable(1);
            this.edgeMesh.visible = true;
            geometry = new THREE.BufferGeometry();
            geometry.setAttribute('position', new THREE.BufferAttribute(new Float32Array(3 * this.numVisVerts), 3));
            geometry.setIndex(visTriIds);
            this.visMesh = new THREE.Mesh(geometry, visMaterial);
            this.visMesh.castShadow = true;
            this.visMesh.userData = this;
            this.visMesh.layers.enable(1);
            geometry.computeVertexNormals();
            this.updateVisMesh();
        }
    }

    // Build the body from a .tetsim container (SURVEY.md §8(f)-3: the five Dragon.js arrays as raw sections of one
    // mmap-able file, see tetsim_amd/meshfile.py / include/tetsim.h) instead of parsing 2.4 MB of array literals.
    static fromFile(path, physicsParams, visMaterial, world) {
        const m = loadTetSim().readMesh(path);
        return new SoftBodyHIP(m.vertices, m.tetIds, m.tetEdgeIds || [], physicsParams, m.visVerts || new Float32Array(0),
                               m.visTriIds || [], visMaterial, world, path);
    }

    // Several INDEPENDENT bodies [{vertices, tetIds}, ...] behind one handle: main.js:80-84 steps softBodies[] one after the other,
    // here ONE simulate() steps them all with one launch per kernel (a Dragon alone fills 15 of the chip's 2,048 workgroup
    // slots).  `.pos` is the concatenation; bodyRanges() gives each body's particle / tet range.  Every body's result equals
    // its solo run bit for bit.  Physics only (no display meshes).
    static batch(bodies, physicsParams) {
        const nv = bodies.reduce((s, b) => s + b.vertices.length, 0), nt = bodies.reduce((s, b) => s + b.tetIds.length, 0);
        const vertices = new Float32Array(nv), tetIds = new Int32Array(nt);
        let ov = 0, ot = 0;
        for (const b of bodies) {
            vertices.set(b.vertices, ov);
            for (let i = 0; i < b.tetIds.length; i++) tetIds[ot + i] = b.tetIds[i] + ov / 3;
            ov += b.vertices.length; ot += b.tetIds.length;
        }
        return new SoftBodyHIP(vertices, tetIds, [], physicsParams, new Float32Array(0), [], null, null, { batch: bodies });
    }
    bodyRanges() { return this._api.batchLayout(this._h); }

    // ---- multi-GPU: one Node process per GPU, this body = partition partIndex of partCount (INTEGRATION.md §4) -------------
    // rank 0: id = SoftBodyHIP.commUniqueId(); ship the 128 bytes to every process (any channel); all: body.commInit(id, rank, n).
    // Afterwards simulate()/simulateSubsteps() exchange ghost positions with the neighbouring partitions over RCCL; `pos`
    // then holds the OWNED particles only, whose global ids are body.ownedIds().
    static commUniqueId() { return loadTetSim().commUniqueId(); }
    // The library's own vertex partitioner (include/tetsim.h: tetsim_prep_partition): owner of every particle for `parts` partitions, what
    // goes into `physicsParams.tetsim.vertOwner` of every rank's body (a body created with partCount > 1 and no vertOwner computes the
    // same map without coordinates).  partitionQuality: per partition { ownedParticles, ghostParticles, boundaryParticles, localElems, ... }.
    static partition(vertices, tetIds, parts, useCoordinates = true) {
        const tets32 = tetIds instanceof Int32Array ? tetIds : Int32Array.from(tetIds);
        const verts32 = vertices instanceof Float32Array ? vertices : Float32Array.from(vertices);
        return loadTetSim().partition(useCoordinates ? verts32 : null, verts32.length / 3, tets32, parts);
    }
    static partitionQuality(tetIds, numParticles, parts, vertOwner = null) {
        const tets32 = tetIds instanceof Int32Array ? tetIds : Int32Array.from(tetIds);
        return loadTetSim().partitionQuality(tets32, numParticles, parts, vertOwner);
    }
    commInit(id, rank, nranks) { this._api.commInit(this._h, id, rank, nranks); }
    ownedIds() { return this._api.ownedIds(this._h); }

    // ---- the hot path ------------------------------------------------------------------------------------------
    simulate(dt, physicsParams) {                       // ONE substep, asynchronous (Softbody.js:195 / SoftbodyGPU.js:610)
        const pp = physicsParams || this.physicsParams;
        if (this._solver === 'polar') pp.dt = dt;       // SoftbodyGPU.js:611 writes dt back into the caller's object
        this._api.step(this._h, dt, pp);
        this._dirty = true;
    }
    simulateSubsteps(n, dt, physicsParams) {            // the whole loop of main.js:79-84 as one FFI crossing / graph launch
        this._api.stepN(this._h, n, dt, physicsParams || this.physicsParams);
        this._dirty = true;
    }
    endFrame() {                                        // Softbody.js:244-247
        this.readToCPU();
        if (this.edgeMesh) {
            this.updateEdgeMesh();
            // the GUI's 'ShowTetMesh' switch (main.js:34,42) is honoured by SoftBodyGPU.endFrame only (SoftbodyGPU.js:646)
            if (this._solver === 'polar' && this.physicsParams.ShowTetMesh !== undefined) this.edgeMesh.visible = !!this.physicsParams.ShowTetMesh;
        }
        if (this.visMesh) this.updateVisMesh();
    }
    readToCPU() {                                       // SoftbodyGPU.js:649-653
        if (this._dirty) {
            // zero copy: `pos` becomes a view of the handle's pinned host buffer on first use; afterwards a read-back is
            // one device pack kernel + one DMA into that same memory
            if (!this._mapped) { this.pos = this._api.mapPositions(this._h); this._mapped = true; }
            else this._api.refreshPositions(this._h);
            if (this._solver === 'neohookean') this.volError = this._api.readVolError(this._h);
            this._dirty = false;
        }
        return this.pos;
    }
    updateEdgeMesh() {                                  // Softbody.js:249-257
        const positions = this.edgeMesh.geometry.attributes.position.array;
        positions.set(this.pos);
        this.edgeMesh.geometry.attributes.position.needsUpdate = true;
        this.edgeMesh.geometry.computeBoundingSphere();
    }
    updateVisMesh() {                                   // Softbody.js:259-277: barycentric skinning of the embedded mesh,
        const positions = this.visMesh.geometry.attributes.position.array;   // done by the device kernel
        this.readVisualPositions(positions);
        // Softbody.js:273 always recomputes the normals; SoftbodyGPU.js:687 only when physicsParams.computeNormals is set
        if (this._solver !== 'polar' || this.physicsParams.computeNormals) {
            const normal = this.visMesh.geometry.attributes.normal;
            if (this._visTris && normal && !this._partitioned) { this.readVisualVertexNormals(normal.array); normal.needsUpdate = true; }   // = computeVertexNormals()
            else this.visMesh.geometry.computeVertexNormals();
        }
        this.visMesh.geometry.attributes.position.needsUpdate = true;
        this.visMesh.geometry.computeBoundingSphere();
    }

    readVisualVertexNormals(out) {                      // Float32Array [3*numVisVerts]: three.js computeVertexNormals on the GPU
        out = out || new Float32Array(3 * this.numVisVerts);
        if (!this._visTris) throw new Error('no visual triangles (visTriIds) were given');
        this._api.readVisualVertexNormals(this._h, out);
        return out;
    }
    // partitions: the normals of THIS rank's rows (visualIds() order) from the ranks' skins put together -- `allPositions` is the full
    // [3 * rows of visVerts] array every rank's scatterVisualPositions() has filled (gathered over the ranks by the host); scattered by
    // row the ranks' normals equal the unpartitioned computeVertexNormals() bit for bit
    readVisualVertexNormalsFrom(allPositions, out) {
        out = out || new Float32Array(3 * (this._visIds ? this._visIds.length : this.numVisVerts));
        if (!this._visTris) throw new Error('no visual triangles (visTriIds) were given');
        this._api.visualVertexNormalsFrom(this._h, allPositions, out);
        return out;
    }
    readVisualPositions(out) {                          // Float32Array [3*numVisVerts], skinned on the GPU
        out = out || new Float32Array(3 * this.numVisVerts);
        if (!this._visOnDevice) return out;
        if (this._visIds) return this.scatterVisualPositions(out);
        this._api.readVisualMesh(this._h, out, null);
        return out;
    }
    // partitions: which rows of visVerts this rank skins, and its rows written into a full-size [3*numVisVerts] array (the other
    // ranks' rows are left as they are: a host that gathers the ranks' arrays row by row has the whole mesh).  Corners this rank does
    // not own need their owners' end-of-substep positions: with an RCCL transport EVERY rank calls refreshFinalGhosts() after the
    // frame's last substep, before endFrame() / readVisualPositions() (a collective; the read itself never communicates, and fails
    // with "stale" if the refresh was left out).  A freshly built partition is fresh (its ghosts hold the rest pose).
    refreshFinalGhosts() { this._api.haloRefreshFinal(this._h); }
    // the partitions of ONE Node process (handles[i] = partition i): step them together / refresh their ghosts together
    static groupStepN(bodies, n, dt, physicsParams) {
        loadTetSim().groupStepN(bodies.map(b => b._h), n, dt, physicsParams || bodies[0].physicsParams);
        for (const b of bodies) b._dirty = true;
    }
    static groupRefreshFinal(bodies) { loadTetSim().groupRefreshFinal(bodies.map(b => b._h)); }
    visualIds() { return this._visIds || Int32Array.from({ length: this.numVisVerts }, (_, i) => i); }
    scatterVisualPositions(full) {
        const ids = this._visIds, own = new Float32Array(3 * ids.length);
        this._api.readVisualMesh(this._h, own, null);
        for (let i = 0; i < ids.length; i++) { const r = 3 * ids[i]; full[r] = own[3 * i]; full[r + 1] = own[3 * i + 1]; full[r + 2] = own[3 * i + 2]; }
        return full;
    }

    // ---- grab (Softbody.js:279-298) ------------------------------------------------------------------------------
    startGrab(pos) {
        const opt = this.physicsParams.tetsim || {};
        if (opt.refStartGrab && this.edgeMesh) {
            // SoftbodyGPU.js:692-704 to the letter: the search runs over the edge mesh's COPY of the positions, i.e. whatever
            // the last updateEdgeMesh() left there (GPUGrabber.start refreshes it first, :790-795; a direct call does not)
            const particles = this.edgeMesh.geometry.attributes.position.array;
            let minD2 = Number.MAX_VALUE;
            this.grabId = -1;
            for (let i = 0; i < this.numParticles; i++) {
                const dx = pos.x - particles[3 * i], dy = pos.y - particles[3 * i + 1], dz = pos.z - particles[3 * i + 2];
                const d2 = dx * dx + dy * dy + dz * dz;
                if (d2 < minD2) { minD2 = d2; this.grabId = i; }
            }
            this._api.setGrab(this._h, this.grabId, pos.x, pos.y, pos.z);
        } else {
            this.grabId = this._api.startGrab(this._h, pos.x, pos.y, pos.z);   // nearest particle among the CURRENT device positions (Softbody.js:279-291)
        }
        this.grabPos[0] = pos.x; this.grabPos[1] = pos.y; this.grabPos[2] = pos.z;
    }
    moveGrabbed(pos) {
        this.grabPos[0] = pos.x; this.grabPos[1] = pos.y; this.grabPos[2] = pos.z;
        this._api.setGrab(this._h, this.grabId, pos.x, pos.y, pos.z);
    }
    endGrab() {
        this.grabId = -1;
        this._api.setGrab(this._h, -1, 0, 0, 0);
    }

    // per-tet rotation quaternions (textureQuat, SoftbodyGPU.js:55; consumed by the normal path :440) -- a zero-copy view of the
    // handle's pinned host buffer, [4 * localElems] in the body's local tet order, refreshed by every call
    readQuats() {
        if (!this._quats) this._quats = this._api.mapQuats(this._h);
        else this._api.refreshQuats(this._h);
        return this._quats;
    }
    // checkpoint / resume of the complete solver state (positions, velocities, per-tet quaternions and carried rest shape)
    saveState() { return this._api.saveState(this._h); }
    loadState(blob) { this._api.loadState(this._h, blob); this._dirty = true; }

    info() { return this._api.info(this._h); }
    dispose() {
        if (!this._h) return;
        // destroy() frees the pinned buffers `pos` / the quaternion view alias (the addon detaches them): keep plain copies
        if (this._mapped) { this.pos = Float32Array.from(this.pos); this._mapped = false; }
        if (this._quats) { this._quats = Float32Array.from(this._quats); }
        this._api.destroy(this._h);
        this._h = null;
    }
}
SoftBodyHIP.THREE = null;

module.exports = { SoftBodyHIP, loadTetSim };

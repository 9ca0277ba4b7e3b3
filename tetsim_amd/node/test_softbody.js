'use strict';
// Node-side parity check of the N-API boundary (run on a GPU host):  node tetsim_amd/node/test_softbody.js
// Neo-Hookean PRECISE through SoftBodyHIP must reproduce, bit for bit, golden vectors recorded from the reference's
// Softbody.js (tests/golden/make_golden.mjs); the polar path must stay finite and honour grab/endFrame.
const fs = require('fs');
const path = require('path');
const assert = require('assert');
const { SoftBodyHIP } = require('./SoftBodyHIP.js');

const G = path.join(__dirname, '..', '..', 'tests', 'golden');
const f32 = n => { const b = fs.readFileSync(path.join(G, n)); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
const i32 = n => { const b = fs.readFileSync(path.join(G, n)); return new Int32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
const bitsEqual = (a, b) => { const x = new Uint32Array(a.buffer, a.byteOffset, a.length), y = new Uint32Array(b.buffer, b.byteOffset, b.length); for (let i = 0; i < x.length; i++) if (x[i] !== y[i]) return i; return -1; };

const verts = f32('dragon_verts.f32'), tets = Array.from(i32('dragon_tets.i32'));   // the reference passes a plain Array
const pp = { gravity: -9.81, timeScale: 1.0, timeStep: 1.0 / 60.0, numSubsteps: 10, friction: 1000.0, density: 1000.0,
             devCompliance: 1.0 / 100000.0, volCompliance: 0.0, worldBounds: [-2.5, -1.0, -2.5, 2.5, 10.0, 2.5] };
const dt = (pp.timeScale * pp.timeStep) / pp.numSubsteps;   // main.js:79

// 1. Softbody.js mirror, exact
pp.tetsim = { solver: 'neohookean', precision: 'precise' };
let body = new SoftBodyHIP(verts.slice(0), tets, [], pp, new Float32Array(0), [], null);
assert.strictEqual(body.numParticles, 1234); assert.strictEqual(body.numElems, 3840);
for (let step = 1; step <= 100; step++) {
    body.simulate(dt, pp);
    if (step === 1 || step === 10 || step === 100) {
        body.endFrame();
        assert.strictEqual(bitsEqual(body.pos, f32(`dragon_pos_${step}.f32`)), -1, `positions differ from Softbody.js at substep ${step}`);
    }
}
const golden = JSON.parse(fs.readFileSync(path.join(G, 'golden.json'))).cases.dragon.steps['100'];
assert.strictEqual(body.volError, golden.volError);
// embedded visual mesh: device skinning == the reference's updateVisMesh output (29,800 vertices, after 10 substeps)
{
    const b2 = new SoftBodyHIP(verts.slice(0), tets, [], pp, f32('dragon_vis.f32'), [], null);
    for (let step = 0; step < 10; step++) b2.simulate(dt, pp);
    assert.strictEqual(bitsEqual(b2.readVisualPositions(), f32('dragon_vispos_10.f32')), -1, 'visual mesh differs from updateVisMesh');
    b2.dispose();
}
body.dispose();
console.log('neohookean/precise: bit-exact vs Softbody.js goldens at substeps 1, 10, 100 (volError', golden.volError + ')');

// 2. SoftbodyGPU.js mirror: frame loop, grab, dt write-back
pp.tetsim = { solver: 'polar', precision: 'fast' };
pp.numSubsteps = 20;
body = new SoftBodyHIP(verts.slice(0), tets, [], pp, new Float32Array(0), [], null, {});
const dt20 = (pp.timeScale * pp.timeStep) / pp.numSubsteps;
for (let frame = 0; frame < 5; frame++) { body.simulateSubsteps(pp.numSubsteps, dt20, pp); body.endFrame(); }
body.startGrab({ x: 0.0, y: 1.2, z: 0.0 });
assert.ok(body.grabId >= 0 && body.grabId < body.numParticles);
body.moveGrabbed({ x: 0.05, y: 1.25, z: 0.0 });
body.simulate(dt20, pp); body.endFrame();
assert.strictEqual(pp.dt, dt20);
const g = body.grabId;
assert.ok(Math.abs(body.pos[3 * g] - 0.05) < 1e-7 && Math.abs(body.pos[3 * g + 1] - 1.25) < 1e-6);
body.endGrab();
for (let i = 0; i < body.pos.length; i++) assert.ok(Number.isFinite(body.pos[i]));
console.log('polar/fast: 5 frames x 20 substeps + grab ok; info', JSON.stringify(body.info()));
body.dispose();

// 3. SoftbodyGPU.js mirror against golden vectors recorded from the reference's own GLSL passes (tests/golden/make_golden_gpu.*):
//    reference-faithful flags, the mouse-drag case, tolerances of tests/test_gpu_polar_reference.py
{
    const cases = JSON.parse(fs.readFileSync(path.join(G, 'cases_gpu.json')));
    const gold = JSON.parse(fs.readFileSync(path.join(G, 'golden_gpu.json'))).cases;
    const c = cases.find(x => x.name === 'dragon_grab'), gc = gold.dragon_grab;
    const tol = { 10: 1e-5, 60: 1e-4 };
    const p2 = Object.assign({ timeScale: c.timeScale, timeStep: c.timeStep, numSubsteps: c.numSubsteps }, c.params);
    p2.tetsim = { solver: 'polar', precision: 'precise', refGrabTexel: true };
    const b = new SoftBodyHIP(verts.slice(0), tets, [], p2, new Float32Array(0), [], null, {});
    let start = null;
    for (let step = 1; step <= c.nsteps; step++) {
        for (const ev of c.grab) {
            if (ev.at !== step) continue;
            if (ev.op === 'start_id') { start = gc.grabStartPos; b.grabId = ev.id; b.moveGrabbed({ x: start[0], y: start[1], z: start[2] }); }
            else if (ev.op === 'move_rel') b.moveGrabbed({ x: start[0] + ev.d[0], y: start[1] + ev.d[1], z: start[2] + ev.d[2] });
            else if (ev.op === 'end') b.endGrab();
        }
        b.simulate(gc.dt, p2);
        if (c.dumps.includes(step)) {
            b.endFrame();
            const want = f32(`dragon_grab_gpu_pos_${step}.f32`);
            let err = 0;
            for (let i = 0; i < want.length; i++) err = Math.max(err, Math.abs(b.pos[i] - want[i]));
            assert.ok(err <= tol[step], `polar/precise vs reference GLSL at substep ${step}: ${err}`);
            console.log(`polar/precise vs reference GLSL goldens, substep ${step}: max |dx| = ${err.toExponential(2)} m`);
        }
    }
    b.dispose();
}
// 4. SoftBodyHIP.fromFile: the same body from a .tetsim container (written by tests/test_node_boundary.py)
if (process.env.TETSIM_TEST_MESH) {
    const p4 = Object.assign({}, pp, { numSubsteps: 10, tetsim: { solver: 'neohookean', precision: 'precise' } });
    const b = SoftBodyHIP.fromFile(process.env.TETSIM_TEST_MESH, p4, null, null);
    assert.strictEqual(b.numParticles, 1234); assert.strictEqual(b.numElems, 3840); assert.strictEqual(b.numVisVerts, 29800);
    for (let step = 1; step <= 10; step++) b.simulate(dt, p4);
    b.endFrame();
    assert.strictEqual(bitsEqual(b.pos, f32('dragon_pos_10.f32')), -1, 'fromFile body differs from Softbody.js at substep 10');
    assert.strictEqual(bitsEqual(b.readVisualPositions(), f32('dragon_vispos_10.f32')), -1, 'fromFile visual mesh differs');
    b.dispose();
    console.log('fromFile: .tetsim container -> bit-exact vs Softbody.js goldens (positions + 29,800 skinned vertices)');
}
// 4b. the clustered Gauss-Seidel schedule from Node: a permuted sequential sweep, deterministic, finite, close to the
//     coloured one after a frame (both are valid Gauss-Seidel orders of the same system)
{
    const mk = order => new SoftBodyHIP(verts.slice(0), tets, [], Object.assign({}, pp, { numSubsteps: 10, tetsim: { solver: 'neohookean', precision: 'precise', order } }), new Float32Array(0), [], null, {});
    const a = mk('clustered'), b = mk('clustered'), c = mk('coloured');
    assert.ok(a.info().numLevels < c.info().numLevels, 'clustered schedule must need fewer launches than the coloured one');
    for (const body of [a, b, c]) { body.simulateSubsteps(10, dt); body.endFrame(); }
    assert.strictEqual(bitsEqual(a.pos, b.pos), -1, 'clustered schedule is not deterministic');
    let worst = 0; for (let i = 0; i < a.pos.length; i++) worst = Math.max(worst, Math.abs(a.pos[i] - c.pos[i]));
    assert.ok(Number.isFinite(worst) && worst < 5e-2, `clustered vs coloured after one frame: ${worst}`);
    console.log(`neohookean clustered: ${a.info().numLevels} launches/substep (coloured ${c.info().numLevels}), max |dx| vs coloured after 10 substeps ${worst.toExponential(2)} m`);
    a.dispose(); b.dispose(); c.dispose();
}
// 4c. geometry.computeVertexNormals() of the embedded mesh on the device: bit-exact with three.js inside the reference
{
    const b = fs.readFileSync(path.join(G, 'dragon_vistris.u16'));
    const tris = Int32Array.from(new Uint16Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)));
    const p6 = Object.assign({}, pp, { numSubsteps: 10, tetsim: { solver: 'neohookean', precision: 'precise' } });
    const body6 = new SoftBodyHIP(verts.slice(0), tets, [], p6, f32('dragon_vis.f32'), tris, null, {});
    for (let step = 1; step <= 10; step++) body6.simulate(dt, p6);
    assert.strictEqual(bitsEqual(body6.readVisualVertexNormals(), f32('dragon_visnormal_10.f32')), -1, 'device vertex normals differ from three.js computeVertexNormals');
    console.log('visual mesh: 29,800 vertex normals over 59,657 triangles bit-exact vs three.js computeVertexNormals');
    body6.dispose();
}
// 4d. the display objects main.js:67-68 adds to the scene (SURVEY.md 8(b)): built when three.js is injected.  three.js itself cannot
//     travel to the GPU box, so a stand-in with the handful of members SoftBodyHIP.js touches is injected here; it counts what a
//     renderer would need to see (needsUpdate flags, bounding spheres) and keeps its own computeVertexNormals OUT of the way.
{
    const calls = { computeVertexNormals: 0, computeBoundingSphere: 0 };
    class BufferAttribute { constructor(array, itemSize) { this.array = array; this.itemSize = itemSize; this.needsUpdate = false; } }
    class BufferGeometry {
        constructor() { this.attributes = {}; this.index = null; }
        setAttribute(name, a) { this.attributes[name] = a; return this; }
        setIndex(ids) { this.index = ids; return this; }
        computeVertexNormals() { calls.computeVertexNormals++; if (!this.attributes.normal) this.attributes.normal = new BufferAttribute(new Float32Array(this.attributes.position.array.length), 3); }
        computeBoundingSphere() { calls.computeBoundingSphere++; }
    }
    class Layers { constructor() { this.mask = 1; } enable(l) { this.mask |= 1 << l; } }
    class Object3D { constructor(geometry, material) { this.geometry = geometry; this.material = material; this.layers = new Layers(); this.userData = {}; this.visible = true; } }
    const THREE = { BufferAttribute, BufferGeometry, LineSegments: Object3D, Mesh: Object3D };
    const b = fs.readFileSync(path.join(G, 'dragon_vistris.u16'));
    const tris = Array.from(new Uint16Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)));   // a plain Array, as Dragon.js has it
    const caller = verts.slice(0);
    const p7 = Object.assign({}, pp, { numSubsteps: 10, tetsim: { solver: 'neohookean', precision: 'precise' } });
    const material = { name: 'visMaterial' };
    const body7 = new SoftBodyHIP(caller, tets, [0, 1, 1, 2], p7, f32('dragon_vis.f32'), tris, material, { THREE });
    assert.ok(body7.edgeMesh && body7.visMesh, 'display objects missing');
    assert.strictEqual(body7.edgeMesh.userData, body7); assert.strictEqual(body7.visMesh.userData, body7);      // for the grabber's raycast
    assert.strictEqual(body7.edgeMesh.layers.mask & 2, 2); assert.strictEqual(body7.visMesh.layers.mask & 2, 2); // layer 1 (Softbody.js:40,54)
    assert.strictEqual(body7.visMesh.material, material); assert.strictEqual(body7.visMesh.castShadow, true);
    assert.strictEqual(body7.edgeMesh.geometry.attributes.position.array, caller, 'the edge mesh aliases the caller\'s vertices (Softbody.js:37)');
    const cvn = calls.computeVertexNormals;
    for (let step = 1; step <= 10; step++) body7.simulate(dt, p7);
    body7.endFrame();
    assert.strictEqual(bitsEqual(caller, f32('dragon_pos_10.f32')), -1, 'edge mesh / caller vertices after endFrame (Softbody.js:252)');
    assert.strictEqual(bitsEqual(body7.visMesh.geometry.attributes.position.array, f32('dragon_vispos_10.f32')), -1);
    assert.strictEqual(bitsEqual(body7.visMesh.geometry.attributes.normal.array, f32('dragon_visnormal_10.f32')), -1, 'normal attribute after endFrame');
    assert.strictEqual(calls.computeVertexNormals, cvn, 'the per-frame computeVertexNormals must come from the device');
    assert.ok(body7.visMesh.geometry.attributes.position.needsUpdate && body7.visMesh.geometry.attributes.normal.needsUpdate && body7.edgeMesh.geometry.attributes.position.needsUpdate);
    assert.ok(calls.computeBoundingSphere >= 2);
    console.log('display objects: edgeMesh/visMesh with userData + layer 1, positions and normals refreshed by endFrame() from the device');
    body7.dispose();
}
// 5. partitioned bodies and the RCCL communicator from Node (one rank here: the entry points and the bookkeeping)
{
    const nv = verts.length / 3;
    const owner = new Int32Array(nv); for (let i = 0; i < nv; i++) owner[i] = i < nv / 2 ? 0 : 1;
    const p5 = Object.assign({}, pp, { numSubsteps: 20, tetsim: { solver: 'polar', precision: 'precise', partCount: 2, partIndex: 1, vertOwner: owner } });
    const part = new SoftBodyHIP(verts.slice(0), tets, [], p5, new Float32Array(0), [], null, {});
    const inf = part.info(), ids = part.ownedIds();
    assert.strictEqual(inf.ownedParticles, nv - Math.ceil(nv / 2)); assert.strictEqual(ids.length, inf.ownedParticles);
    assert.ok(inf.localParticles > inf.ownedParticles && inf.numNeighbours === 1);
    for (let i = 0; i < ids.length; i++) assert.strictEqual(owner[ids[i]], 1);
    part.dispose();
    {   // the library's own partitioner through N-API: one owner per particle, both parts used, fewer ghosts than the file's index ranges
        const own = SoftBodyHIP.partition(verts, tets, 3);
        assert.ok(own instanceof Int32Array && own.length === verts.length / 3 && own.every(r => r >= 0 && r < 3) && new Set(own).size === 3);
        const ghosts = q => q.reduce((s, p) => s + p.ghostParticles, 0);
        const ranges = Int32Array.from(own, (_, i) => Math.min(2, Math.floor(i * 3 / own.length)));
        const qa = SoftBodyHIP.partitionQuality(tets, own.length, 3, own), qb = SoftBodyHIP.partitionQuality(tets, own.length, 3, ranges);
        assert.ok(qa.length === 3 && ghosts(qa) < ghosts(qb) && qa.reduce((s, p) => s + p.ownedParticles, 0) === own.length);
        assert.throws(() => SoftBodyHIP.partition(verts, tets, 0), /parts/);
        assert.throws(() => SoftBodyHIP.partition(verts, Int32Array.of(0, 1, 2), 2), /4 ids/);
    }
    const id = SoftBodyHIP.commUniqueId();
    assert.ok(id instanceof Uint8Array && id.length === 128);
    const solo = new SoftBodyHIP(verts.slice(0), tets, [], Object.assign({}, pp, { tetsim: { solver: 'polar', precision: 'fast' } }), new Float32Array(0), [], null, {});
    solo.commInit(id, 0, 1);
    solo.simulateSubsteps(20, dt20, pp); solo.endFrame();
    for (let i = 0; i < solo.pos.length; i++) assert.ok(Number.isFinite(solo.pos[i]));
    solo.dispose();
    console.log('partition options, ownedIds, commUniqueId/commInit ok');
}
// 6. hardening of the shim (round-1 review): typed-array lengths are checked before the C side writes, views of the pinned
//    buffers do not dangle after dispose(), partitioned bodies ignore the visual mesh instead of throwing
{
    const p8 = Object.assign({}, pp, { numSubsteps: 20, tetsim: { solver: 'polar', precision: 'fast' } });
    const b8 = new SoftBodyHIP(verts.slice(0), tets, [], p8, f32('dragon_vis.f32'), [], null, {});
    assert.strictEqual(b8.info().numVisVerts, 29800);
    assert.throws(() => b8.readVisualPositions(new Float32Array(10)), /too small/);
    assert.throws(() => b8._api.readVisualMesh(b8._h, new Float32Array(3 * 29800), new Float32Array(5)), /too small/);
    assert.throws(() => b8._api.step(b8._h, 'soon', p8), /must be a number/);
    assert.throws(() => b8._api.stepN(b8._h, -1, dt20, p8), /non-negative integer/);
    assert.throws(() => b8._api.setGrab(b8._h, {}, 0, 0, 0), /must be a number/);
    const nvv = verts.length / 3;
    assert.throws(() => new SoftBodyHIP(verts.slice(0), tets, [], Object.assign({}, pp, { tetsim: { solver: 'polar', partCount: 2, partIndex: 0, vertOwner: new Int32Array(nvv - 1) } }), new Float32Array(0), [], null, {}), /vertOwner/);
    // quaternions: zero-copy view == the copying read, unit length; save / load of the complete state continues bit for bit
    b8.simulateSubsteps(20, dt20, p8);
    const q = b8.readQuats(), qc = new Float32Array(4 * b8.info().localElems);
    b8._api.readQuats(b8._h, qc);
    assert.strictEqual(q.length, qc.length); assert.strictEqual(bitsEqual(q, qc), -1, 'pinned quaternion view differs from readQuats');
    for (let e = 0; e < q.length; e += 4) assert.ok(Math.abs(Math.hypot(q[e], q[e + 1], q[e + 2], q[e + 3]) - 1) < 1e-5);
    const blob = b8.saveState();
    b8.simulateSubsteps(20, dt20, p8); b8.endFrame();
    const want = Float32Array.from(b8.pos);
    const b9 = new SoftBodyHIP(verts.slice(0), tets, [], p8, new Float32Array(0), [], null, {});
    b9.loadState(blob);
    b9.simulateSubsteps(20, dt20, p8); b9.endFrame();
    assert.strictEqual(bitsEqual(b9.pos, want), -1, 'a body restored with loadState must continue the trajectory bit for bit');
    // dispose(): `pos` stays readable (a plain copy), the raw pinned views are detached, the handle is unusable
    const rawView = b9._api.mapPositions(b9._h), kept = b9.pos;
    b9.dispose();
    assert.strictEqual(bitsEqual(b9.pos, want), -1); assert.strictEqual(rawView.length, 0, 'pinned view must be detached by destroy'); assert.strictEqual(kept.length, 0);
    assert.throws(() => b9._api.sync(b9._h));
    b8.dispose();
    console.log('shim hardening: length checks, number checks, detached views after dispose, quaternion view, save/load state ok');
}
if (process.env.TETSIM_TEST_MESH) {   // partitioned bodies from a file that carries a visual mesh: each keeps the rows of the tets it owns
    const nvv = verts.length / 3;
    const owner = new Int32Array(nvv); for (let i = 0; i < nvv; i++) owner[i] = i < nvv / 2 ? 0 : 1;
    const seen = new Uint8Array(29800);
    let total = 0;
    for (let r = 0; r < 2; r++) {
        const p10 = Object.assign({}, pp, { numSubsteps: 20, tetsim: { solver: 'polar', precision: 'fast', partCount: 2, partIndex: r, vertOwner: owner } });
        const part = SoftBodyHIP.fromFile(process.env.TETSIM_TEST_MESH, p10, null, null);
        const ids = part.visualIds();
        assert.strictEqual(part.info().numVisVerts, ids.length); assert.ok(ids.length > 0 && ids.length < 29800); assert.ok(part.info().ownedParticles < nvv);
        for (const i of ids) { assert.strictEqual(seen[i], 0); seen[i] = 1; }
        total += ids.length;
        part.simulate(dt20, p10); part.endFrame();
        part.dispose();
    }
    assert.strictEqual(total, 29800);   // every row of the Dragon's visVerts in exactly one partition
    console.log('partitioned fromFile bodies with a stored visual mesh: the two partitions keep ' + total + ' visual vertices between them, each once');
}
// 7. startGrab exactly as SoftbodyGPU.js:692-704 (tetsim.refStartGrab): the search runs over the edge mesh's copy of the positions
{
    class BufferAttribute { constructor(array, itemSize) { this.array = array; this.itemSize = itemSize; this.needsUpdate = false; } }
    class BufferGeometry { constructor() { this.attributes = {}; } setAttribute(n, a) { this.attributes[n] = a; return this; } setIndex() { return this; } computeVertexNormals() {} computeBoundingSphere() {} }
    class Layers { enable() {} }
    class Object3D { constructor(g, m) { this.geometry = g; this.material = m; this.layers = new Layers(); this.userData = {}; this.visible = true; } }
    const THREE = { BufferAttribute, BufferGeometry, LineSegments: Object3D, Mesh: Object3D };
    const p11 = Object.assign({}, pp, { numSubsteps: 20, tetsim: { solver: 'polar', precision: 'precise', refStartGrab: true } });
    const b = new SoftBodyHIP(verts.slice(0), tets, [], p11, new Float32Array(0), [], null, { THREE });
    const probe = { x: verts[3 * 700] + 1e-4, y: verts[3 * 700 + 1] - 0.3, z: verts[3 * 700 + 2] };   // where particle 700 will be after falling ~0.3 m
    for (let f = 0; f < 15; f++) b.simulateSubsteps(20, dt20, p11);
    b.readToCPU();                       // positions read back, edge mesh NOT refreshed (SoftbodyGPU.js:643-647 leaves it commented out)
    b.startGrab(probe);
    let best = -1, bd = Infinity;        // the reference's loop over the stale copy (= the rest positions here)
    for (let i = 0; i < b.numParticles; i++) { const d = (probe.x - verts[3 * i]) ** 2 + (probe.y - verts[3 * i + 1]) ** 2 + (probe.z - verts[3 * i + 2]) ** 2; if (d < bd) { bd = d; best = i; } }
    assert.strictEqual(b.grabId, best, 'refStartGrab must search the edge-mesh copy');
    b.endGrab();
    b.updateEdgeMesh();                  // what GPUGrabber.start does first (:790-795): now the copy is current
    b.startGrab(probe);
    const cur = b.pos; best = -1; bd = Infinity;
    for (let i = 0; i < b.numParticles; i++) { const d = (probe.x - cur[3 * i]) ** 2 + (probe.y - cur[3 * i + 1]) ** 2 + (probe.z - cur[3 * i + 2]) ** 2; if (d < bd) { bd = d; best = i; } }
    assert.strictEqual(b.grabId, best);
    b.dispose();
    console.log('refStartGrab: searches the edge-mesh copy (stale until updateEdgeMesh), like SoftbodyGPU.js:692-704');
}
// 8. a batch of independent bodies behind one handle: each equals its solo run bit for bit (Neo-Hookean PRECISE == Softbody.js goldens)
{
    const p12 = Object.assign({}, pp, { numSubsteps: 10, tetsim: { solver: 'neohookean', precision: 'precise' } });
    const lv = f32('lat4_verts.f32'), lt = i32('lat4_tets.i32');
    const batch = SoftBodyHIP.batch([{ vertices: verts, tetIds: tets }, { vertices: lv, tetIds: lt }, { vertices: verts, tetIds: tets }], p12);
    assert.strictEqual(batch.info().numBodies, 3); assert.strictEqual(batch.numParticles, 2 * 1234 + lv.length / 3);
    for (let step = 1; step <= 10; step++) batch.simulate(dt, p12);
    batch.endFrame();
    const r = batch.bodyRanges().firstParticle, want = f32('dragon_pos_10.f32');
    for (const b of [0, 2]) assert.strictEqual(bitsEqual(batch.pos.subarray(3 * r[b], 3 * r[b + 1]), want), -1, `batched Dragon ${b} differs from Softbody.js at substep 10`);
    batch.dispose();
    console.log('batch: 2 Dragons + a lattice behind one handle, both Dragons bit-exact vs Softbody.js goldens');
}
// 9. partitions WITH display meshes (advisor, round 5): the constructor's first updateVisMesh() must work on a fresh partition (its ghosts
//    hold the rest pose), a read with stale ghosts says so instead of communicating, the in-process group entry points are reachable
//    from Node; and the lean tet record (tetsim.leanState) through the same surface
{
    class BufferAttribute { constructor(array, itemSize) { this.array = array; this.itemSize = itemSize; this.needsUpdate = false; } }
    class BufferGeometry { constructor() { this.attributes = {}; } setAttribute(n, a) { this.attributes[n] = a; return this; } setIndex() { return this; }
                           computeVertexNormals() { if (!this.attributes.normal) this.attributes.normal = new BufferAttribute(new Float32Array(this.attributes.position.array.length), 3); } computeBoundingSphere() {} }
    class Layers { enable() {} }
    class Object3D { constructor(g, m) { this.geometry = g; this.material = m; this.layers = new Layers(); this.userData = {}; this.visible = true; } }
    const THREE = { BufferAttribute, BufferGeometry, LineSegments: Object3D, Mesh: Object3D };
    const nvv = verts.length / 3, vis = f32('dragon_vis.f32');
    const owner = new Int32Array(nvv); for (let i = 0; i < nvv; i++) owner[i] = i < nvv / 2 ? 0 : 1;
    const mk = (r, extra) => new SoftBodyHIP(verts.slice(0), tets, [0, 1], Object.assign({}, pp, { numSubsteps: 20, tetsim: Object.assign({ solver: 'polar', precision: 'fast', partCount: 2, partIndex: r, vertOwner: owner }, extra || {}) }),
                                             vis, [], null, { THREE });
    const parts = [mk(0), mk(1)];                                  // (threw TETSIM_ESTATE before: the constructor skins the visual mesh)
    const p13 = parts[0].physicsParams;
    const rest = new Float32Array(3 * 29800);
    for (const b of parts) b.scatterVisualPositions(rest);
    const mono = new SoftBodyHIP(verts.slice(0), tets, [], Object.assign({}, pp, { tetsim: { solver: 'polar', precision: 'fast' } }), vis, [], null, {});
    assert.strictEqual(bitsEqual(rest, mono.readVisualPositions()), -1, 'the partitions\' rest-pose skins add up to the whole visual mesh');
    SoftBodyHIP.groupStepN(parts, 20, dt20, p13);
    assert.throws(() => parts[0].readVisualPositions(), /stale/);
    SoftBodyHIP.groupRefreshFinal(parts);
    const full = new Float32Array(3 * 29800);
    for (const b of parts) { b.endFrame(); b.scatterVisualPositions(full); }
    for (let i = 0; i < full.length; i++) assert.ok(Number.isFinite(full[i]));
    assert.ok(full.some((x, i) => x !== rest[i]));
    for (const b of parts) b.dispose();
    mono.dispose();
    const lean = new SoftBodyHIP(verts.slice(0), tets, [], Object.assign({}, pp, { tetsim: { solver: 'polar', precision: 'fast', leanState: true } }), new Float32Array(0), [], null, {});
    const dflt = new SoftBodyHIP(verts.slice(0), tets, [], Object.assign({}, pp, { tetsim: { solver: 'polar', precision: 'fast' } }), new Float32Array(0), [], null, {});
    for (const b of [lean, dflt]) { b.simulateSubsteps(20, dt20, p13); b.endFrame(); }
    let worst = 0;
    for (let i = 0; i < lean.pos.length; i++) worst = Math.max(worst, Math.abs(lean.pos[i] - dflt.pos[i]));
    assert.ok(worst < 5e-5, 'lean state vs default FAST after 20 substeps: ' + worst);
    const ql = lean.readQuats();
    for (let e = 0; e < ql.length; e += 4) assert.ok(Math.abs(Math.hypot(ql[e], ql[e + 1], ql[e + 2], ql[e + 3]) - 1) < 1e-5);
    lean.dispose(); dflt.dispose();
    console.log('partitions with display meshes build and skin (fresh ghosts), stale reads say so, group entry points + leanState reachable from Node');
}
console.log('node boundary ok');

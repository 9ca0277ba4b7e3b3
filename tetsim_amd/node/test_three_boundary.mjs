// SoftBodyHIP's display objects against the REAL three.js r160 the reference vendors (build container only: the module is imported from
// a scratch copy of /root/reference/node_modules/three/build/three.module.js whose path is argv[2]; nothing of three.js is committed).
//     node tetsim_amd/node/test_three_boundary.mjs <path to three.module.js>
// The native addon is replaced by a stand-in with a 2-tet body (there is no GPU where this runs): what is under test is the JavaScript
// boundary itself -- what main.js:60-68 and Grabber (Softbody.js:36-57, 440-456) need from `.edgeMesh` / `.visMesh`.
import { createRequire } from 'module';
import { pathToFileURL, fileURLToPath } from 'url';
import path from 'path';
import assert from 'assert';

const require = createRequire(import.meta.url);
const here = path.dirname(fileURLToPath(import.meta.url));

async function main() {
    const THREE = await import(pathToFileURL(path.resolve(process.argv[2])).href);
    assert.ok(THREE.REVISION, 'not three.js');

    // ---- the stand-in addon: the entry points SoftBodyHIP.js calls, over plain arrays -----------------------------------------
    const calls = { startGrab: [], setGrab: [], step: 0, destroyed: 0 };
    const fake = {
        load() { return 5; },
        create(verts, tets) { return { pos: Float32Array.from(verts), tets: Int32Array.from(tets), vis: null }; },
        destroy() { calls.destroyed++; },
        step(h) { for (let i = 1; i < h.pos.length; i += 3) h.pos[i] -= 0.01; calls.step++; },
        stepN(h, n) { for (let s = 0; s < n; s++) fake.step(h); },
        sync() {},
        mapPositions(h) { h.mapped = new Float32Array(h.pos.length); h.mapped.set(h.pos); return h.mapped; },
        refreshPositions(h) { h.mapped.set(h.pos); },
        setVisualMesh(h, vis) { h.vis = Float32Array.from(vis); },
        visualIds() { return null; },
        readVisualMesh(h, out) {           // Softbody.js:259-277 / SoftbodyGPU.js:429-435
            const nv = h.vis.length / 4;
            for (let i = 0; i < nv; i++) {
                const e = h.vis[4 * i], b = [h.vis[4 * i + 1], h.vis[4 * i + 2], h.vis[4 * i + 3]];
                b.push(1 - b[0] - b[1] - b[2]);
                for (let c = 0; c < 3; c++) { let a = 0; for (let k = 0; k < 4; k++) a += h.pos[3 * h.tets[4 * e + k] + c] * b[k]; out[3 * i + c] = a; }
            }
        },
        setVisualTriangles(h, t) { h.tris = t; },
        readVisualVertexNormals(h, out) { out.fill(0); for (let i = 2; i < out.length; i += 3) out[i] = 1; },
        startGrab(h, x, y, z) {            // Softbody.js:279-291: nearest particle
            calls.startGrab.push([x, y, z]);
            let best = -1, bd = Infinity;
            for (let i = 0; i < h.pos.length / 3; i++) { const d = (x - h.pos[3 * i]) ** 2 + (y - h.pos[3 * i + 1]) ** 2 + (z - h.pos[3 * i + 2]) ** 2; if (d < bd) { bd = d; best = i; } }
            return best;
        },
        setGrab(h, id, x, y, z) { calls.setGrab.push([id, x, y, z]); },
        readVolError() { return 0; },
        info() { return { numVisVerts: 4 }; },
    };
    const addonPath = path.join(here, 'tetsim_napi.node');
    require.cache[addonPath] = { id: addonPath, filename: addonPath, loaded: true, exports: fake, children: [], paths: [] };
    const { SoftBodyHIP } = require('./SoftBodyHIP.js');
    SoftBodyHIP.THREE = THREE;             // injected, as INTEGRATION.md shows (never bundled)

    // ---- a 2-tet body with a 4-vertex, 2-triangle embedded mesh ---------------------------------------------------------------
    const vertices = new Float32Array([0, 1, 0, 1, 1, 0, 0, 2, 0, 0, 1, 1, 1, 2, 1]);
    const tetIds = [0, 1, 2, 3, 1, 2, 3, 4], tetEdgeIds = [0, 1, 0, 2, 0, 3, 1, 2, 1, 3, 2, 3, 1, 4, 2, 4, 3, 4];
    const visVerts = new Float32Array([0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.25, 0.25, 0.25]);   // (tetNr, b0, b1, b2): the corners 0, 1, 2 of tet 0 and an inner point
    const visTriIds = [0, 1, 2, 2, 1, 3];
    const material = new THREE.MeshPhongMaterial({ color: 0xf78a1d });
    const pp = { gravity: -10, density: 1000, tetsim: { solver: 'polar', precision: 'fast' } };
    const body = new SoftBodyHIP(vertices, tetIds, tetEdgeIds, pp, visVerts, visTriIds, material);

    // what main.js:67-68 adds to the scene, and what the Grabber's raycast reads off it (Softbody.js:36-57)
    assert.ok(body.edgeMesh instanceof THREE.LineSegments && body.edgeMesh.isLineSegments, 'edgeMesh must be a THREE.LineSegments');
    assert.ok(body.visMesh instanceof THREE.Mesh && body.visMesh.isMesh, 'visMesh must be a THREE.Mesh');
    assert.strictEqual(body.edgeMesh.userData, body); assert.strictEqual(body.visMesh.userData, body);
    assert.strictEqual(body.visMesh.material, material); assert.strictEqual(body.visMesh.castShadow, true);
    const raycaster = new THREE.Raycaster();
    raycaster.layers.set(1);               // main.js / Grabber: this.raycaster.layers.set(1)
    assert.ok(body.visMesh.layers.test(raycaster.layers) && body.edgeMesh.layers.test(raycaster.layers), 'layer 1 (Softbody.js:40,54)');
    assert.strictEqual(body.edgeMesh.geometry.attributes.position.array, vertices, 'the edge mesh aliases the caller\'s vertices (Softbody.js:37)');
    assert.strictEqual(body.edgeMesh.geometry.index.count, tetEdgeIds.length);
    assert.strictEqual(body.visMesh.geometry.index.count, visTriIds.length);
    const vp = body.visMesh.geometry.attributes.position, vn = body.visMesh.geometry.attributes.normal;
    assert.ok(vp instanceof THREE.BufferAttribute && vp.count === 4 && vn && vn.count === 4, 'position + computeVertexNormals() in the constructor (Softbody.js:55-56)');
    assert.deepStrictEqual(Array.from(vp.array.slice(0, 9)), [0, 1, 0, 1, 1, 0, 0, 2, 0]);   // skinned at construction: the rest pose

    // a frame: two substeps, endFrame() -- real BufferAttributes count their updates in `version` (needsUpdate is a setter)
    const v0 = [body.edgeMesh.geometry.attributes.position.version, vp.version];
    body.simulate(1 / 600, pp); body.simulate(1 / 600, pp);
    body.endFrame();
    assert.strictEqual(calls.step, 2);
    assert.ok(Math.abs(vertices[1] - 0.98) < 1e-6, 'endFrame() writes the positions into the caller\'s array through the edge mesh (Softbody.js:252)');
    assert.ok(body.edgeMesh.geometry.attributes.position.version > v0[0] && vp.version > v0[1], 'needsUpdate = true must reach three\'s version counters');
    assert.ok(Math.abs(vp.array[1] - 0.98) < 1e-6);
    assert.ok(body.edgeMesh.geometry.boundingSphere && body.visMesh.geometry.boundingSphere && body.visMesh.geometry.boundingSphere.radius > 0);

    // Grabber.start (Softbody.js:440-456): a ray through the embedded mesh -> intersects[0].object.userData -> startGrab(hit)
    raycaster.set(new THREE.Vector3(0.3, 1.3, 5), new THREE.Vector3(0, 0, -1));
    const scene = new THREE.Scene();
    scene.add(body.edgeMesh); scene.add(body.visMesh);
    raycaster.params.Line.threshold = 0.001;   // (three's default of 1 world unit lets every tet edge within a metre of the ray "hit" first)
    const hits = raycaster.intersectObjects(scene.children);
    assert.strictEqual(hits[0].object, body.visMesh);
    assert.ok(hits.length > 0, 'the ray must hit the embedded mesh');
    const obj = hits[0].object.userData;
    assert.ok(obj instanceof SoftBodyHIP && obj === body);
    const hit = raycaster.ray.origin.clone();
    hit.addScaledVector(raycaster.ray.direction, hits[0].distance);
    body.startGrab(hit);
    assert.ok(Math.abs(hit.x - 0.3) < 1e-6 && Math.abs(hit.y - 1.3) < 1e-6 && hit.z >= -1e-6 && hit.z <= 0.25 + 1e-6, 'the hit lies on the embedded mesh');
    assert.deepStrictEqual(calls.startGrab[0], [hit.x, hit.y, hit.z]);
    let best = -1, bd = Infinity;          // Softbody.js:279-291 over the positions endFrame() left in the caller's array
    for (let i = 0; i < 5; i++) { const d = (hit.x - vertices[3 * i]) ** 2 + (hit.y - vertices[3 * i + 1]) ** 2 + (hit.z - vertices[3 * i + 2]) ** 2; if (d < bd) { bd = d; best = i; } }
    assert.strictEqual(body.grabId, best);
    assert.ok(Math.abs(body.grabPos[0] - 0.3) < 1e-6);
    body.moveGrabbed(new THREE.Vector3(0.4, 1.4, 0.1));
    assert.deepStrictEqual(calls.setGrab[calls.setGrab.length - 1].map(x => Math.round(x * 1e6) / 1e6), [best, 0.4, 1.4, 0.1]);
    body.endGrab();
    assert.strictEqual(body.grabId, -1); assert.strictEqual(calls.setGrab[calls.setGrab.length - 1][0], -1);
    // the edge mesh alone is raycastable too (the reference's commented-out layers.enable(1) on it is enabled here: SoftbodyGPU.js:419)
    raycaster.params.Line.threshold = 0.05;
    raycaster.set(new THREE.Vector3(0.5, 0.98, 5), new THREE.Vector3(0, 0, -1));
    assert.ok(raycaster.intersectObject(body.edgeMesh).length > 0);
    body.dispose();
    assert.strictEqual(calls.destroyed, 1);
    console.log('THREE_BOUNDARY_OK r' + THREE.REVISION);
}
main().catch(e => { console.error(e); process.exit(1); });

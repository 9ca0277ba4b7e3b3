// Golden-vector generator: imports the REFERENCE CPU solver (Softbody.js) from a scratch copy and
// records its outputs on the meshes/cases listed in cases.json.  Only data is written to the repo.
// usage: node make_golden.mjs <scratch-dir-with-reference> <output-dir>
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';

const [scratch, outDir] = process.argv.slice(2);
const sha = a => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex').slice(0, 16);
const sum = a => { let s = 0.0; for (let i = 0; i < a.length; i++) s += a[i]; return s; };
const writeF32 = (name, a) => fs.writeFileSync(path.join(outDir, name), Buffer.from(a.buffer, a.byteOffset, a.byteLength));
const readF32 = name => { const b = fs.readFileSync(path.join(outDir, name)); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
const readI32 = name => { const b = fs.readFileSync(path.join(outDir, name)); return new Int32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };

async function main() {
    const { SoftBody } = await import(path.join(scratch, 'src/Softbody.js'));
    const D = await import(path.join(scratch, 'src/Dragon.js'));

    // Export the demo mesh (input data of BASELINE configs 1/2) as binary fixtures.
    writeF32('dragon_verts.f32', D.dragonTetVerts);
    const dragonTets = Int32Array.from(D.dragonTetIds);
    fs.writeFileSync(path.join(outDir, 'dragon_tets.i32'), Buffer.from(dragonTets.buffer));
    writeF32('dragon_vis.f32', D.dragonAttachedVerts);  // [tetNr,b0,b1,b2] per embedded visual vertex

    const cases = JSON.parse(fs.readFileSync(path.join(outDir, 'cases.json')));
    const golden = { generator: 'tests/golden/make_golden.mjs', node: process.version, cases: {} };
    for (const c of cases) {
        const verts = readF32(c.mesh + '_verts.f32');
        // Dragon ids go in as the reference passes them (a plain Array); lattices as plain Arrays too.
        const tets = Array.from(readI32(c.mesh + '_tets.i32'));
        const pp = Object.assign({ timeScale: c.timeScale, timeStep: c.timeStep, numSubsteps: c.numSubsteps }, c.params);
        // vertices.slice(0): the constructor aliases and later overwrites its input (Softbody.js:12,37,252)
        const body = new SoftBody(verts.slice(0), tets, [], pp, new Float32Array(0), [], null);
        const out = { init: { invMass: sha(body.invMass), invRestPose: sha(body.invRestPose), invRestVolume: sha(body.invRestVolume) }, steps: {} };
        if (!fs.existsSync(path.join(outDir, c.mesh + '_invMass.f32')) || !golden.meshDone) {
            writeF32(c.mesh + '_invMass.f32', body.invMass);             // initPhysics outputs, once per mesh
            writeF32(c.mesh + '_invRestPose.f32', body.invRestPose);
            writeF32(c.mesh + '_invRestVolume.f32', body.invRestVolume);
        }
        const dt = (pp.timeScale * pp.timeStep) / pp.numSubsteps;   // main.js:79, formed in f64
        out.dt = dt;
        out.grabIds = [];
        for (let step = 1; step <= c.nsteps; step++) {
            for (const g of c.grab) {
                if (g.at !== step) continue;
                if (g.op === 'start') { body.startGrab({ x: g.p[0], y: g.p[1], z: g.p[2] }); out.grabIds.push(body.grabId); }
                else if (g.op === 'move') body.moveGrabbed({ x: g.p[0], y: g.p[1], z: g.p[2] });
                else if (g.op === 'end') body.endGrab();
            }
            body.simulate(dt, pp);
            const dump = c.dumps.includes(step), hash = c.hashes.includes(step);
            if (dump || hash) {
                let ymin = Infinity, vmax = 0;
                for (let i = 0; i < body.numParticles; i++) {
                    ymin = Math.min(ymin, body.pos[3 * i + 1]);
                    vmax = Math.max(vmax, Math.hypot(body.vel[3 * i], body.vel[3 * i + 1], body.vel[3 * i + 2]));
                }
                out.steps[step] = { pos: sha(body.pos), vel: sha(body.vel), prev: sha(body.prevPos), sumPos: sum(body.pos), volError: body.volError, ymin, vmax };
            }
            if (dump) {
                writeF32(`${c.name}_pos_${step}.f32`, body.pos);
                writeF32(`${c.name}_vel_${step}.f32`, body.vel);
            }
        }
        golden.cases[c.name] = out;
        console.log(c.name, JSON.stringify(out.steps[Object.keys(out.steps).pop()]));
    }
    // Embedded visual mesh (row (f)-1): the reference's updateVisMesh (Softbody.js:259-277) on the Dragon after 10 substeps.
    {
        const pp = Object.assign({ timeScale: 1.0, timeStep: 1.0 / 60.0, numSubsteps: 10 }, cases[0].params);
        const body = new SoftBody(D.dragonTetVerts.slice(0), D.dragonTetIds, [], pp, D.dragonAttachedVerts, D.dragonAttachedTriIds, null);
        const dt = (pp.timeScale * pp.timeStep) / pp.numSubsteps;
        for (let i = 0; i < 10; i++) body.simulate(dt, pp);
        body.endFrame();
        const vis = body.visMesh.geometry.attributes.position.array;
        writeF32('dragon_vispos_10.f32', vis);
        // ... and the vertex normals three.js derives from them every frame (Softbody.js:273 -> BufferGeometry.computeVertexNormals)
        const nrm = body.visMesh.geometry.attributes.normal.array;
        writeF32('dragon_visnormal_10.f32', nrm);
        const tris = Uint16Array.from(D.dragonAttachedTriIds);   // input data: the visual mesh's triangle list (Dragon.js; 29,800 vertices fit u16)
        fs.writeFileSync(path.join(outDir, 'dragon_vistris.u16'), Buffer.from(tris.buffer));
        golden.vis = { dragon_vispos_10: sha(vis), dragon_visnormal_10: sha(nrm), numVisVerts: body.numVisVerts, numVisTris: tris.length / 3 };
        console.log('vis', golden.vis);
    }
    fs.writeFileSync(path.join(outDir, 'golden.json'), JSON.stringify(golden, null, 1));
}
main().catch(e => { console.error(e); process.exit(1); });

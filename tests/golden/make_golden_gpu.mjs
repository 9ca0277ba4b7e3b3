// Golden-vector generator for the polar-decomposition (WebGL) solver: imports the REFERENCE SoftBodyGPU from a scratch
// copy and gives it what the reference's own World.js gives it in the browser -- a THREE.WebGLRenderer, the reference's vendored,
// unmodified one -- created on a WebGL2 context whose GL is Mesa softpipe (oracle/glsl_ref: webgl2_context.mjs over mesa_gl.cc), and
// records texturePos / textureVel / textureQuat after the substeps listed in cases_gpu.json.  Only data is written to the repo.
// usage: node make_golden_gpu.mjs <scratch-dir-with-reference> <output-dir> <mesa_gl.node>
import fs from 'fs';
import path from 'path';
import crypto from 'crypto';
import { createWebGL2Context } from '../../oracle/glsl_ref/webgl2_context.mjs';

const [scratch, outDir, addon] = process.argv.slice(2);
const sha = a => crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex').slice(0, 16);
const write = (name, a) => fs.writeFileSync(path.join(outDir, name), Buffer.from(a.buffer, a.byteOffset, a.byteLength));
const readF32 = name => { const b = fs.readFileSync(path.join(outDir, name)); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
const readI32 = name => { const b = fs.readFileSync(path.join(outDir, name)); return new Int32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };

async function main() {
    const THREE = await import(path.join(scratch, 'node_modules/three/build/three.module.js'));
    const { SoftBodyGPU } = await import(path.join(scratch, 'src/SoftbodyGPU.js'));
    const ctx = createWebGL2Context(addon);
    const renderer = new THREE.WebGLRenderer({ canvas: ctx.canvas, context: ctx.gl });   // three r160's own renderer (SoftbodyGPU.js:9 takes world.renderer)
    console.log('GL:', JSON.stringify(ctx.info), ' renderer: THREE.WebGLRenderer r' + THREE.REVISION, 'isWebGL2', renderer.capabilities.isWebGL2);

    const cases = JSON.parse(fs.readFileSync(path.join(outDir, 'cases_gpu.json')));
    const golden = { generator: 'tests/golden/make_golden_gpu.mjs', node: process.version, three: THREE.REVISION, gl: ctx.info, cases: {} };
    const quietLog = console.log;
    for (const c of cases) {
        const verts = readF32(c.mesh + '_verts.f32');
        const tets = Array.from(readI32(c.mesh + '_tets.i32'));
        const pp = Object.assign({ timeScale: c.timeScale, timeStep: c.timeStep, numSubsteps: c.numSubsteps }, c.params);
        const world = { renderer, scene: new THREE.Scene() };
        console.log = () => {};           // initPhysics prints biggestT
        const body = new SoftBodyGPU(verts.slice(0), tets, [], pp, new Float32Array(0), [], new THREE.MeshPhongMaterial(), world);
        console.log = quietLog;
        const nv = body.numParticles, nt = body.numElems, dim = body.texDim;
        const texel = new Float32Array(dim * dim * 4);
        const grab3 = (variable, n) => {   // xyz of the first n texels of the variable's CURRENT target
            renderer.readRenderTargetPixels(body.gpuCompute.getCurrentRenderTarget(variable), 0, 0, dim, dim, texel);   // as readToCPU, SoftbodyGPU.js:649-668
            const out = new Float32Array(3 * n);
            for (let i = 0; i < n; i++) { out[3 * i] = texel[4 * i]; out[3 * i + 1] = texel[4 * i + 1]; out[3 * i + 2] = texel[4 * i + 2]; }
            return out;
        };
        const grab4 = (variable, n) => {
            renderer.readRenderTargetPixels(body.gpuCompute.getCurrentRenderTarget(variable), 0, 0, dim, dim, texel);
            return texel.slice(0, 4 * n);
        };

        // host-side tables the constructor built (SoftbodyGPU.js:487-608): scatter table and rest volumes
        const slots = new Int32Array(nv * 36);
        for (let v = 0; v < nv; v++)
            for (let t = 0; t < 9; t++)
                for (let ch = 0; ch < 4; ch++) slots[v * 36 + 4 * t + ch] = body.particleToElemVertsTable[t].image.data[4 * v + ch];
        const invVol = new Float32Array(nt);
        for (let e = 0; e < nt; e++) invVol[e] = body.invRestVolumeAndColor.image.data[4 * e];
        const out = { numParticles: nv, numElems: nt, texDim: dim, slots: sha(slots), invRestVolume: sha(invVol), steps: {}, grabIds: [] };
        if (c.dumpTables) { write(`${c.name}_gpu_slots.i32`, slots); write(`${c.name}_gpu_invRestVolume.f32`, invVol); }

        const dt = (pp.timeScale * pp.timeStep) / pp.numSubsteps;   // main.js:79
        out.dt = dt;
        let grabStart = null;
        for (let step = 1; step <= c.nsteps; step++) {
            for (const g of c.grab) {
                if (g.at !== step) continue;
                if (g.op === 'start') { body.startGrab({ x: g.p[0], y: g.p[1], z: g.p[2] }); out.grabIds.push(body.grabId); }
                else if (g.op === 'move') body.moveGrabbed({ x: g.p[0], y: g.p[1], z: g.p[2] });
                else if (g.op === 'start_id') {
                    // a GENTLE grab: the collision pass pins the particle its indexFromUV selects for grabId (not particle
                    // grabId), so set grabId directly and start from the current position of the particle that will be
                    // pinned (`follow`), read back from the GPU; recorded so that replays use the same numbers
                    const cur = grab3(body.pos, nv);
                    grabStart = [cur[3 * g.follow], cur[3 * g.follow + 1], cur[3 * g.follow + 2]];
                    body.grabId = g.id;
                    body.moveGrabbed({ x: grabStart[0], y: grabStart[1], z: grabStart[2] });
                    out.grabIds.push(g.id);
                    out.grabStartPos = grabStart;
                } else if (g.op === 'move_rel') body.moveGrabbed({ x: grabStart[0] + g.d[0], y: grabStart[1] + g.d[1], z: grabStart[2] + g.d[2] });
                else if (g.op === 'end') body.endGrab();
            }
            body.simulate(dt, pp);
            if (c.dumps.includes(step)) {
                const pos = grab3(body.pos, nv), vel = grab3(body.vel, nv), prev = grab3(body.prevPos, nv), quat = grab4(body.quats, nt);
                write(`${c.name}_gpu_pos_${step}.f32`, pos);
                write(`${c.name}_gpu_vel_${step}.f32`, vel);
                if (!c.quatDumps || c.quatDumps.includes(step)) write(`${c.name}_gpu_quat_${step}.f32`, quat);
                let ymin = Infinity;
                for (let i = 0; i < nv; i++) ymin = Math.min(ymin, pos[3 * i + 1]);
                out.steps[step] = { pos: sha(pos), vel: sha(vel), prev: sha(prev), quat: sha(quat), ymin };
            }
        }
        golden.cases[c.name] = out;
        console.log(c.name, JSON.stringify(out.steps[Object.keys(out.steps).pop()]), 'draws', ctx.stats.draws);
    }
    fs.writeFileSync(path.join(outDir, 'golden_gpu.json'), JSON.stringify(golden, null, 1));
}
main().catch(e => { console.error(e); process.exit(1); });

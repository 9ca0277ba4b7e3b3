'use strict';
// nh_port.js -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.  The Neo-Hookean XPBD Gauss-Seidel substep of the reference's CPU
// solver (/root/reference/src/Softbody.js:60-87 initPhysics, :91-166 solveElem, :168-193 applyToElem, :195-240 simulate)
// restated in plain JavaScript from this repo's C restatement (oracle/tetsim_oracle.c section A), so that the "JS CPU path"
// can be timed on the GPU box's own host cores, where the reference's source must not travel.  Single thread, like the
// reference.  No three.js, no classes, no shared vector helpers: state lives in flat Float32Arrays and every expression
// is written out, so JavaScript's own number semantics (f64 arithmetic, f32 rounding at each typed-array store) reproduce
// the reference bit for bit -- tests/test_oracle_golden.py::test_js_port_bit_exact checks it against the golden vectors
// recorded from Softbody.js.
//
//   node oracle/nh_port.js --verts V.f32 --tets T.i32 --substeps N [--per-frame 10] [--warmup W] [--params '{"gravity":-9.81,...}']
// prints one JSON line: { substeps, seconds, m_tet_solves_per_s, pos_sha16 }
const fs = require('fs');
const crypto = require('crypto');

function createBody(verts, tets, density) {
    const nv = verts.length / 3, nt = tets.length / 4;
    const s = {
        nv, nt, tets,
        pos: Float32Array.from(verts), prev: Float32Array.from(verts), vel: new Float32Array(3 * nv),
        invMass: new Float32Array(nv), invRest: new Float32Array(9 * nt), invVol: new Float32Array(nt),
        F: new Float32Array(9), dF: new Float32Array(9), P: new Float32Array(9), g: new Float32Array(12),
        volError: 0.0, grabId: -1, grabPos: new Float32Array(3),
    };
    const p = s.pos, A = s.invRest;
    for (let e = 0; e < nt; e++) {
        const i0 = 3 * tets[4 * e], i1 = 3 * tets[4 * e + 1], i2 = 3 * tets[4 * e + 2], i3 = 3 * tets[4 * e + 3], o = 9 * e;
        // rest edge matrix, column-major (:66-68)
        A[o] = p[i1] - p[i0]; A[o + 1] = p[i1 + 1] - p[i0 + 1]; A[o + 2] = p[i1 + 2] - p[i0 + 2];
        A[o + 3] = p[i2] - p[i0]; A[o + 4] = p[i2 + 1] - p[i0 + 1]; A[o + 5] = p[i2 + 2] - p[i0 + 2];
        A[o + 6] = p[i3] - p[i0]; A[o + 7] = p[i3 + 1] - p[i0 + 1]; A[o + 8] = p[i3 + 2] - p[i0 + 2];
        const a11 = A[o], a12 = A[o + 3], a13 = A[o + 6], a21 = A[o + 1], a22 = A[o + 4], a23 = A[o + 7], a31 = A[o + 2], a32 = A[o + 5], a33 = A[o + 8];
        const det = a11 * a22 * a33 + a12 * a23 * a31 + a13 * a21 * a32 - a13 * a22 * a31 - a12 * a21 * a33 - a11 * a23 * a32;
        const V = det / 6.0;
        if (det === 0.0) {
            // the reference's zero-determinant branch clears elements [e, e+9) of the WHOLE array (:391-394)
            for (let k = 0; k < 9; k++) if (e + k < A.length) A[e + k] = 0.0;
        } else {
            const r = 1.0 / det;
            A[o] = (a22 * a33 - a23 * a32) * r; A[o + 3] = -(a12 * a33 - a13 * a32) * r; A[o + 6] = (a12 * a23 - a13 * a22) * r;
            A[o + 1] = -(a21 * a33 - a23 * a31) * r; A[o + 4] = (a11 * a33 - a13 * a31) * r; A[o + 7] = -(a11 * a23 - a13 * a21) * r;
            A[o + 2] = (a21 * a32 - a22 * a31) * r; A[o + 5] = -(a11 * a32 - a12 * a31) * r; A[o + 8] = (a11 * a22 - a12 * a21) * r;
        }
        const pm = V / 4.0 * density;   // lumped mass, accumulated with an f32 rounding per add (:74-78)
        s.invMass[tets[4 * e]] += pm; s.invMass[tets[4 * e + 1]] += pm; s.invMass[tets[4 * e + 2]] += pm; s.invMass[tets[4 * e + 3]] += pm;
        s.invVol[e] = 1.0 / V;
    }
    for (let i = 0; i < nv; i++) if (s.invMass[i] !== 0.0) s.invMass[i] = 1.0 / s.invMass[i];
    return s;
}

// F = Ds * Dm^-1 for tet e, each column accumulated with an f32 store after every add (:363-379)
function deformationGradient(s, e) {
    const t = s.tets, p = s.pos, P = s.P, F = s.F, A = s.invRest;
    const i0 = 3 * t[4 * e], i1 = 3 * t[4 * e + 1], i2 = 3 * t[4 * e + 2], i3 = 3 * t[4 * e + 3];
    P[0] = p[i1] - p[i0]; P[1] = p[i1 + 1] - p[i0 + 1]; P[2] = p[i1 + 2] - p[i0 + 2];
    P[3] = p[i2] - p[i0]; P[4] = p[i2 + 1] - p[i0 + 1]; P[5] = p[i2 + 2] - p[i0 + 2];
    P[6] = p[i3] - p[i0]; P[7] = p[i3 + 1] - p[i0 + 1]; P[8] = p[i3 + 2] - p[i0 + 2];
    for (let j = 0; j < 3; j++) {
        const b0 = A[9 * e + 3 * j], b1 = A[9 * e + 3 * j + 1], b2 = A[9 * e + 3 * j + 2], c = 3 * j;
        F[c] = 0.0; F[c + 1] = 0.0; F[c + 2] = 0.0;
        F[c] += P[0] * b0; F[c + 1] += P[1] * b0; F[c + 2] += P[2] * b0;
        F[c] += P[3] * b1; F[c + 1] += P[4] * b1; F[c + 2] += P[5] * b1;
        F[c] += P[6] * b2; F[c + 1] += P[7] * b2; F[c + 2] += P[8] * b2;
    }
}

// gradients 1..3 = M * Dm^-T scaled, accumulated like the reference's vecAdd chain (:112-124, :146-158)
function gradients(s, e, M, scale) {
    const g = s.g, A = s.invRest, o = 9 * e;
    for (let k = 0; k < 3; k++) {
        const c = 3 * (k + 1);
        const w0 = scale * A[o + k], w1 = scale * A[o + 3 + k], w2 = scale * A[o + 6 + k];
        g[c] = 0.0; g[c + 1] = 0.0; g[c + 2] = 0.0;
        g[c] += M[0] * w0; g[c + 1] += M[1] * w0; g[c + 2] += M[2] * w0;
        g[c] += M[3] * w1; g[c + 1] += M[4] * w1; g[c + 2] += M[5] * w1;
        g[c] += M[6] * w2; g[c + 1] += M[7] * w2; g[c + 2] += M[8] * w2;
    }
}

function project(s, e, C, compliance, dt) {   // applyToElem (:168-193)
    if (C === 0.0) return;
    const g = s.g, t = s.tets, p = s.pos, im = s.invMass;
    g[0] = 0.0; g[1] = 0.0; g[2] = 0.0;
    for (let k = 1; k <= 3; k++) { g[0] += g[3 * k] * -1.0; g[1] += g[3 * k + 1] * -1.0; g[2] += g[3 * k + 2] * -1.0; }
    let w = 0.0;
    for (let i = 0; i < 4; i++) w += (g[3 * i] * g[3 * i] + g[3 * i + 1] * g[3 * i + 1] + g[3 * i + 2] * g[3 * i + 2]) * im[t[4 * e + i]];
    if (w === 0.0) return;
    const alpha = compliance / dt / dt * s.invVol[e];
    const dlambda = -C / (w + alpha);
    for (let i = 0; i < 4; i++) {
        const id = t[4 * e + i], k = dlambda * im[id];
        p[3 * id] += g[3 * i] * k; p[3 * id + 1] += g[3 * i + 1] * k; p[3 * id + 2] += g[3 * i + 2] * k;
    }
}

function simulate(s, dt, pp) {
    const { nv, nt, pos, prev, vel, F, dF } = s;
    for (let i = 0; i < nv; i++) {   // predict (:198-202); gravity is an f64 triple
        vel[3 * i] += 0.0 * dt; vel[3 * i + 1] += pp.gravity * dt; vel[3 * i + 2] += 0.0 * dt;
        prev[3 * i] = pos[3 * i]; prev[3 * i + 1] = pos[3 * i + 1]; prev[3 * i + 2] = pos[3 * i + 2];
        pos[3 * i] += vel[3 * i] * dt; pos[3 * i + 1] += vel[3 * i + 1] * dt; pos[3 * i + 2] += vel[3 * i + 2] * dt;
    }
    s.volError = 0.0;
    for (let e = 0; e < nt; e++) {   // sequential Gauss-Seidel over tets (:207-208)
        deformationGradient(s, e);   // deviatoric: C = sqrt(tr F^T F)
        const rs = Math.sqrt((F[0] * F[0] + F[1] * F[1] + F[2] * F[2]) + (F[3] * F[3] + F[4] * F[4] + F[5] * F[5]) + (F[6] * F[6] + F[7] * F[7] + F[8] * F[8]));
        gradients(s, e, F, 1.0 / rs);
        project(s, e, rs, pp.devCompliance, dt);
        deformationGradient(s, e);   // hydrostatic: C = det F - 1 - volCompliance / devCompliance, on the updated positions
        dF[0] = F[4] * F[8] - F[5] * F[7]; dF[1] = F[5] * F[6] - F[3] * F[8]; dF[2] = F[3] * F[7] - F[4] * F[6];
        dF[3] = F[7] * F[2] - F[8] * F[1]; dF[4] = F[8] * F[0] - F[6] * F[2]; dF[5] = F[6] * F[1] - F[7] * F[0];
        dF[6] = F[1] * F[5] - F[2] * F[4]; dF[7] = F[2] * F[3] - F[0] * F[5]; dF[8] = F[0] * F[4] - F[1] * F[3];
        gradients(s, e, dF, 1.0);
        const vol = F[0] * F[4] * F[8] + F[3] * F[7] * F[2] + F[6] * F[1] * F[5] - F[6] * F[4] * F[2] - F[3] * F[1] * F[8] - F[0] * F[7] * F[5];
        s.volError += vol - 1.0;
        project(s, e, vol - 1.0 - pp.volCompliance / pp.devCompliance, pp.volCompliance, dt);
    }
    s.volError /= nt;
    const wb = pp.worldBounds, fr = Math.min(1.0, dt * pp.friction);
    for (let i = 0; i < nv; i++) {   // bounds, floor with friction (:213-231)
        for (let c = 0; c < 3; c++) pos[3 * i + c] = Math.max(wb[c], Math.min(wb[3 + c], pos[3 * i + c]));
        if (pos[3 * i + 1] < 0.0) {
            pos[3 * i + 1] = 0.0;
            F[0] = prev[3 * i] - pos[3 * i]; F[2] = prev[3 * i + 2] - pos[3 * i + 2];
            pos[3 * i] += F[0] * fr; pos[3 * i + 2] += F[2] * fr;
        }
    }
    if (s.grabId >= 0 && s.grabId < nv) { pos[3 * s.grabId] = s.grabPos[0]; pos[3 * s.grabId + 1] = s.grabPos[1]; pos[3 * s.grabId + 2] = s.grabPos[2]; }
    const inv = 1.0 / dt;
    for (let i = 0; i < 3 * nv; i++) vel[i] = (pos[i] - prev[i]) * inv;   // (:238-239)
}

function sha16(a) { return crypto.createHash('sha256').update(Buffer.from(a.buffer, a.byteOffset, a.byteLength)).digest('hex').slice(0, 16); }

function main(argv) {
    const opt = { perFrame: 10, warmup: 0, substeps: 1, reps: 1 };
    for (let i = 0; i < argv.length; i++) {
        if (argv[i] === '--verts') opt.verts = argv[++i];
        else if (argv[i] === '--tets') opt.tets = argv[++i];
        else if (argv[i] === '--substeps') opt.substeps = parseInt(argv[++i], 10);
        else if (argv[i] === '--per-frame') opt.perFrame = parseInt(argv[++i], 10);
        else if (argv[i] === '--warmup') opt.warmup = parseInt(argv[++i], 10);
        else if (argv[i] === '--reps') opt.reps = parseInt(argv[++i], 10);   // timed repetitions of `substeps` substeps each; the median is reported
        else if (argv[i] === '--params') opt.params = JSON.parse(argv[++i]);   // physicsParams keys (main.js:22-36) + timeScale/timeStep
    }
    const rd = (f, T) => { const b = fs.readFileSync(f); return new T(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
    const verts = rd(opt.verts, Float32Array), tets = rd(opt.tets, Int32Array);
    const pp = Object.assign({ gravity: -9.81, friction: 1000.0, density: 1000.0, devCompliance: 1.0 / 100000.0, volCompliance: 0.0,
                               worldBounds: [-2.5, -1.0, -2.5, 2.5, 10.0, 2.5], timeScale: 1.0, timeStep: 1.0 / 60.0 }, opt.params || {});
    const dt = (pp.timeScale * pp.timeStep) / opt.perFrame;   // main.js:79
    const s = createBody(verts, tets, pp.density);
    for (let i = 0; i < opt.warmup; i++) simulate(s, dt, pp);
    const rates = [];
    let sec = 0.0;
    for (let r = 0; r < Math.max(1, opt.reps); r++) {
        const t0 = process.hrtime.bigint();
        for (let i = 0; i < opt.substeps; i++) simulate(s, dt, pp);
        const el = Number(process.hrtime.bigint() - t0) / 1e9;
        sec += el;
        rates.push(s.nt * opt.substeps / el / 1e6);
    }
    const sorted = rates.slice().sort((a, b) => a - b);
    console.log(JSON.stringify({ substeps: opt.substeps, warmup: opt.warmup, reps: rates.length, seconds: sec, m_tet_solves_per_s: sorted[(sorted.length - 1) >> 1],
                                 rates: rates, pos_sha16: sha16(s.pos), vol_error: s.volError, node: process.version }));
}

if (require.main === module) main(process.argv.slice(2));
module.exports = { createBody, simulate };

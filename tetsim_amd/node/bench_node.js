'use strict';
// bench_node.js -- the headline workload driven from Node.js through the N-API shim (the reference's host language):
//   python -c "from tetsim_amd import make_lattice; from tetsim_amd.meshfile import write_mesh; v,t=make_lattice(55); write_mesh('/tmp/lat55.tetsim', v, t)"
//   node tetsim_amd/node/bench_node.js /tmp/lat55.tetsim [frames=50] [warmup=5]
// One frame = 20 substeps issued as ONE simulateSubsteps() call (one FFI crossing, one HIP-graph launch), as bench.py does.
const { SoftBodyHIP } = require('./SoftBodyHIP.js');
const file = process.argv[2], frames = parseInt(process.argv[3] || '50', 10), warmup = parseInt(process.argv[4] || '5', 10);
const pp = { gravity: -9.81, timeScale: 1.0, timeStep: 1.0 / 60.0, numSubsteps: 20, friction: 1000.0, density: 1000.0,
             devCompliance: 1.0 / 100000.0, volCompliance: 0.0, worldBounds: [-2.5, -1.0, -2.5, 2.5, 10.0, 2.5],
             tetsim: { solver: 'polar', precision: 'fast' } };
const dt = (pp.timeScale * pp.timeStep) / pp.numSubsteps;
const body = SoftBodyHIP.fromFile(file, pp, null, null);
for (let i = 0; i < warmup; i++) body.simulateSubsteps(pp.numSubsteps, dt, pp);
body.endFrame();
const t0 = process.hrtime.bigint();
for (let i = 0; i < frames; i++) body.simulateSubsteps(pp.numSubsteps, dt, pp);
body.endFrame();   // syncs (reads positions back once)
const sec = Number(process.hrtime.bigint() - t0) / 1e9;
let finite = true;
for (let i = 0; i < body.pos.length; i++) if (!Number.isFinite(body.pos[i])) { finite = false; break; }
console.log(JSON.stringify({ host: 'node ' + process.version, tets: body.numElems, particles: body.numParticles, frames, substeps_per_frame: pp.numSubsteps,
                             ms_per_frame: sec / frames * 1e3, m_tet_solves_per_s: body.numElems * pp.numSubsteps * frames / sec / 1e6, finite }));
body.dispose();

"""Soak: the persistent launches (polar four-lane, polar one-lane over all XCDs incl. an `exclusive` body, Neo-Hookean single workgroup) stepped
for minutes without a pause, interleaved, with a checkpoint comparison at the end: no wait may ever give up, every state stays finite, and two
bodies fed the same calls stay bit-equal.  python tools/soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_mesh
from tetsim_amd import SoftBodyHIP
from tetsim_amd.lattice import make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dv, dt_ = load_mesh("dragon")
lv, lt = make_lattice(28, y0=0.05)
bv, bt = make_lattice(46, y0=0.05)
mk = lambda: [SoftBodyHIP(dv, dt_, None, dict(PP), solver="polar", precision="fast"),
              SoftBodyHIP(lv, lt, None, dict(PP), solver="polar", precision="fast"),
              SoftBodyHIP(dv, dt_, None, dict(PP), solver="neohookean", precision="precise", order="coloured"),
              SoftBodyHIP(dv, dt_, None, dict(PP), solver="neohookean", precision="fast", order="coloured"),
              # round 6: calls as ONE launch with stamped hand-overs -- a large polar body (2,282 tiles) with the reference's and with the lean
              # record, the clustered FAST Gauss-Seidel sweeps of a lattice
              SoftBodyHIP(bv, bt, None, dict(PP), solver="polar", precision="fast"),
              SoftBodyHIP(bv, bt, None, dict(PP), solver="polar", precision="fast", lean_state=True),
              SoftBodyHIP(lv, lt, None, dict(PP), solver="neohookean", precision="fast", order="clustered")]
A, B = mk(), mk()
print("paths", [int(b.info.fused_particle_pass) for b in A], flush=True)
DT = (1 / 60) / 20
t0 = time.time(); calls = 0; sub = 0
rng = np.random.default_rng(1)
while time.time() - t0 < budget:
    for _ in range(50):
        n = int(rng.integers(1, 41))
        grab = rng.random() < 0.05
        for S in (A, B):
            for b in S:
                if grab: b.setGrab(int(n) % 200, [0.1, 0.8, 0.0])
                elif rng.random() < 0.1 and S is A: pass
                b.simulateSubsteps(n, DT, PP)
        if grab:
            for S in (A, B):
                for b in S: b.endGrab()
        calls += 1; sub += n
    for b in A + B: b.sync()          # raises if a bounded wait ever gave up
ok = all(np.isfinite(b.pos).all() for b in A + B)
same = all(np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32)) for a, b in zip(A, B))
print("calls per body %d, substeps per body %d in %.0f s; finite %s; twins bit-equal %s; paths still %s" % (calls, sub, time.time() - t0, ok, same, [int(b.info.fused_particle_pass) for b in A]))
sys.exit(0 if ok and same else 1)

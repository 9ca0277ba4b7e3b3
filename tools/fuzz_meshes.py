"""Fuzz on the GPU box: seeded random Delaunay meshes of many sizes through every small- and mid-size path against the CPU oracle and against
each other.  Neo-Hookean PRECISE (original / coloured, n-substep launches) == oracle bit for bit; Neo-Hookean FAST launch == its stepwise twin
(tetsim_profile) bit for bit; polar FAST n-substep launch == one-substep launches == stepwise kernels bit for bit, and within 5e-4 m of the
oracle after 40 substeps; polar PRECISE == oracle exactly.  python tools/fuzz_meshes.py [first_seed] [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_random_meshes import random_mesh, PP, DT
from oracle import OracleNH, OraclePJ
from tetsim_amd import SoftBodyHIP
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
same = lambda a, b: np.array_equal(a.view(np.uint32), b.view(np.uint32))
bad = 0
for seed in range(first, first + count):
    npts = int(np.random.default_rng(seed).choice([40, 90, 200, 450, 900, 1800, 3500, 5200]))
    v, t = random_mesh(seed, npts)
    v = v - np.float32([0.0, v[:, 1].min() - 0.004, 0.0])       # a few millimetres above the floor: contact within the run
    msgs = []
    for order in ("original", "coloured"):
        b = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="precise", order=order)
        o = OracleNH(v, t[b.tetOrder], PP)
        for n in (7, 1, 12):
            b.simulateSubsteps(n, DT * 2, PP)
            for _ in range(n): o.simulate(DT * 2, PP)
        if not (same(b.pos, o.pos) and b.volError == o.volError): msgs.append("NH precise %s != oracle (mode %d)" % (order, b.info.fused_particle_pass))
    a, c = [SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="fast", order="coloured") for _ in range(2)]
    a.simulateSubsteps(9, DT * 2, PP); a.simulate(DT * 2, PP); c.profile(10, DT * 2, PP)
    if not (same(a.pos, c.pos) and np.isfinite(a.pos).all()): msgs.append("NH fast launch != stepwise twin (mode %d)" % a.info.fused_particle_pass)
    a, b, c = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast") for _ in range(3)]
    o = OraclePJ(v, t, PP, slot_quirk=True)
    for n in (20, 1, 19):
        a.simulateSubsteps(n, DT, PP)
        for _ in range(n): b.simulate(DT, PP); o.simulate(DT, PP)
        c.profile(n, DT, PP)
    if not (same(a.pos, b.pos) and same(a.pos, c.pos) and same(a.quats, c.quats)): msgs.append("polar fast launch / steps / stepwise kernels differ (mode %d)" % a.info.fused_particle_pass)
    err = float(np.abs(a.pos - o.pos).max())
    if not err < 5e-4: msgs.append("polar fast vs oracle %.3g" % err)
    # round 6: the lean tet record (call == steps == stepwise kernels, and inside the FAST envelope), the kernel pair through the one-launch
    # call (a loose particle puts a small body on path 5), the clustered FAST sweep as one launch per substep /
    # per call against one launch per colour
    a, b, c = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", lean_state=True) for _ in range(3)]
    for n in (20, 1, 19):
        a.simulateSubsteps(n, DT, PP)
        for _ in range(n): b.simulate(DT, PP)
        c.profile(n, DT, PP)
    if not (same(a.pos, b.pos) and same(a.pos, c.pos) and same(a.quats, c.quats)): msgs.append("polar lean launch / steps / stepwise kernels differ (mode %d)" % a.info.fused_particle_pass)
    err_l = float(np.abs(a.pos - o.pos).max())
    if not err_l < 5e-4: msgs.append("polar lean vs oracle %.3g" % err_l)
    v2 = np.concatenate([v, [[3.0, 3.0, 3.0]]]).astype(np.float32)      # a loose particle: no tile sums it, the body keeps the kernel pair -- path 5 inside tetsim_step_n
    a, b = [SoftBodyHIP(v2, t, None, dict(PP), solver="polar", precision="fast", lean_state=bool(seed & 1)) for _ in range(2)]
    for n in (20, 1, 19):
        a.simulateSubsteps(n, DT, PP)
        for _ in range(n): b.simulate(DT, PP)
    if not (a.info.fused_particle_pass == 5 and same(a.pos[:-1], b.pos[:-1]) and same(a.quats, b.quats)): msgs.append("one-launch call != kernel pair (mode %d)" % a.info.fused_particle_pass)
    a = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="fast", order="clustered")
    os.environ["TETSIM_NH_ONE_LAUNCH"] = "0"
    try:
        b = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="fast", order="clustered")
    finally:
        del os.environ["TETSIM_NH_ONE_LAUNCH"]
    for n in (7, 1, 12):
        a.simulateSubsteps(n, DT * 2, PP)
        for _ in range(n): b.simulate(DT * 2, PP)
    a.simulate(DT * 2, PP); b.simulateSubsteps(1, DT * 2, PP)
    if not (same(a.pos, b.pos) and a.volError == b.volError and np.isfinite(a.pos).all()): msgs.append("NH clustered one-launch != one launch per colour")
    p = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise")
    p.simulateSubsteps(40, DT, PP)
    if not same(p.pos, o.pos): msgs.append("polar precise != oracle (max %.3g)" % float(np.abs(p.pos - o.pos).max()))
    print("seed %d: %d particles, %d tets, polar path %d: %s" % (seed, len(v), len(t), a.info.fused_particle_pass, "ok (fast vs oracle %.2g m)" % err if not msgs else "; ".join(msgs)), flush=True)
    bad += bool(msgs)
print("meshes with a finding: %d of %d" % (bad, count))
sys.exit(1 if bad else 0)

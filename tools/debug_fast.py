import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice
from oracle import OraclePJ
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1/60)/20
for name in ("dragon", "lat"):
    if name == "dragon":
        v = np.fromfile("tests/golden/dragon_verts.f32", dtype="<f4").reshape(-1, 3); t = np.fromfile("tests/golden/dragon_tets.i32", dtype="<i4").reshape(-1, 4)
    else:
        v, t = make_lattice(6, y0=0.05)
    for prec in ("precise", "fast"):
        b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=prec); o = OraclePJ(v, t, PP)
        for step in range(1, 4):
            b.simulate(DT, PP); o.simulate(DT, PP)
            p, q = b.pos, b.quats
            bad = np.where(~np.isfinite(p).all(axis=1))[0]
            e = np.abs(p - o.pos).max(axis=1)
            print(name, prec, "step", step, "nan verts", len(bad), bad[:8], "nan quats", int((~np.isfinite(q)).any(axis=1).sum()),
                  "maxerr", np.nanmax(e), "argmax", int(np.nanargmax(e)), "p", p[int(np.nanargmax(e))], "ref", o.pos[int(np.nanargmax(e))])

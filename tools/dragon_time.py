"""BASELINE configs 1/2 on the GPU: Dragon demo mesh, both solvers, frame time through one graph launch per frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
v = np.fromfile(os.path.join(G, "dragon_verts.f32"), dtype="<f4").reshape(-1, 3); t = np.fromfile(os.path.join(G, "dragon_tets.i32"), dtype="<i4").reshape(-1, 4)
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
for solver, prec, order, n in [("polar", "precise", "original", 20), ("polar", "fast", "original", 20), ("neohookean", "precise", "original", 10),
                                ("neohookean", "precise", "coloured", 10), ("neohookean", "fast", "coloured", 10),
                                ("neohookean", "precise", "clustered", 10), ("neohookean", "fast", "clustered", 10)]:
    b = SoftBodyHIP(v, t, None, dict(pp), solver=solver, precision=prec, order=order)
    dt = (1 / 60) / n
    b.simulateSubsteps(n, dt, pp); b.sync()
    ms = min(b.timeSubsteps(n, dt, pp) for _ in range(10))
    print("dragon %-10s %-7s %-8s levels %4d  %2d substeps/frame: %.3f ms/frame -> %.1f M tet-solves/s" %
          (solver, prec, order, b.info.num_levels, n, ms, len(t) * n / ms / 1e3))

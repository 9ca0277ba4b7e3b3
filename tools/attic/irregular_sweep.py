"""Polar Jacobi FAST on IRREGULAR meshes (seeded random Delaunay, ragged valence, slivers dropped) against Kuhn lattices of the same size:
the path the library picks, the longest partial-sum list, us per substep.  python tools/attic/irregular_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_random_meshes import random_mesh, PP
from tetsim_amd import SoftBodyHIP
PATH = {0: "two kernels", 1: "fused kernel", 2: "frame, 1 lane", 3: "frame, 4 lanes"}
n_sub, dt = 20, (1 / 60) / 20
print("%8s %9s %9s %-15s %12s %12s" % ("points", "particles", "tets", "path", "us/substep", "M tet-s/s"))
for npts in (200, 800, 3500, 10000, 30000, 80000):
    v, t = random_mesh(11, npts, min_vol=0.02 * 0.336 / (6.7 * npts))   # slivers: below 2% of the mean tet volume
    b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    for _ in range(3):
        b.simulateSubsteps(n_sub, dt, PP)
    b.sync()
    ms = min(b.timeSubsteps(n_sub, dt, PP) for _ in range(6))
    print("%8d %9d %9d %-15s %12.2f %12.1f   max valence %d, dropped slots %d" % (npts, len(v), len(t), PATH.get(b.info.fused_particle_pass, "?"), ms * 1e3 / n_sub, len(t) * n_sub / ms / 1e3,
                                                                               b.info.max_valence, b.info.dropped_slots), flush=True)
    assert np.isfinite(b.pos).all()
    del b

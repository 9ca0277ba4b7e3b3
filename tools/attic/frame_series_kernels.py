"""Per-frame kernel times of a fresh 1 M-tet body (tetsim_profile: the kernels' own begin/end events): which kernel is slower in
the first ~30 frames of the simulation, and does it depend on the data (free fall from rest) or on the GPU's state?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tetsim_amd import SoftBodyHIP, make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(55); DT = (1 / 60) / 20
for label, pp in (("free fall from rest", PP), ("same, second body", PP)):
    b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
    rows = []
    for f in range(48):
        pr = b.profile(20, DT, pp)
        rows.append((pr["tet_ms"] / pr["tet_launches"] * 1e3, pr["vertex_ms"] / pr["vertex_launches"] * 1e3))
    print(label, "tet us:", " ".join("%.1f" % r[0] for r in rows[:16]), "...", " ".join("%.1f" % r[0] for r in rows[-4:]))
    print(label, "vertex us:", " ".join("%.1f" % r[1] for r in rows[:16]), "...", " ".join("%.1f" % r[1] for r in rows[-4:]))
    b.close()

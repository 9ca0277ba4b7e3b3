import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, group_step_n, make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0)
DT = (1/60)/20
n = 10
v, t = make_lattice(n, nz=2*n, y0=0.05)
plane = (n+1)*(n+1)
parts = 2
owner = np.minimum((np.arange(len(v)) // plane) * parts // (2*n+1), parts-1).astype(np.int32)
mk = lambda: [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise", part_count=parts, part_index=p, vert_owner=owner) for p in range(parts)]
for trial in range(6):
    m1 = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise")
    m2 = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise")
    g1, g2 = mk(), mk()
    out = []
    for it in range(5):
        m1.simulateSubsteps(10, DT, PP)
        group_step_n(g1, 10, DT, PP)
        for _ in range(10): m2.simulate(DT, PP)      # eager monolithic
        group_step_n(g2, 10, DT, PP)
        a, b = m1.pos, m2.pos
        e_mm = float(np.abs(a - b).max())
        e_g1 = max(float(np.abs(x.pos - a[x.ownedIds]).max()) for x in g1)
        e_g2 = max(float(np.abs(x.pos - a[x.ownedIds]).max()) for x in g2)
        e_gg = max(float(np.abs(x.pos - y.pos).max()) for x, y in zip(g1, g2))
        out.append("mono(graph) vs mono(eager) %.2g | g1 vs mono %.2g | g2 vs mono %.2g | g1 vs g2 %.2g" % (e_mm, e_g1, e_g2, e_gg))
    print("trial", trial, out[-1])

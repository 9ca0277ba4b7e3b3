import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, group_step_n, make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0)
DT = (1/60)/20
n = 10
v, t = make_lattice(n, nz=2*n, y0=0.05)
plane = (n+1)*(n+1)
for parts in (2, 3):
    owner = np.minimum((np.arange(len(v)) // plane) * parts // (2*n+1), parts-1).astype(np.int32)
    for trial in range(3):
        mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise")
        bodies = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise", part_count=parts, part_index=p, vert_owner=owner) for p in range(parts)]
        errs = []
        for it in range(6):
            mono.simulateSubsteps(10, DT, PP)
            group_step_n(bodies, 10, DT, PP)
            ref = mono.pos
            errs.append(max(float(np.abs(b.pos - ref[b.ownedIds]).max()) for b in bodies))
        print("parts", parts, "trial", trial, "err after each 10-substep call", ["%.2g" % e for e in errs])

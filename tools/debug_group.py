import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, group_step_n, halo_exchange_local, make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0)
DT = (1/60)/20
n = 10
v, t = make_lattice(n, nz=2*n, y0=0.05)
plane = (n+1)*(n+1)
for parts in (2, 3, 4):
    owner = np.minimum((np.arange(len(v)) // plane) * parts // (2*n+1), parts-1).astype(np.int32)
    for prec in ("precise",):
        mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=prec)
        bodies = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=prec, part_count=parts, part_index=p, vert_owner=owner) for p in range(parts)]
        sync = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=prec, part_count=parts, part_index=p, vert_owner=owner) for p in range(parts)]
        for it in range(12):
            mono.simulateSubsteps(1, DT, PP)
            group_step_n(bodies, 1, DT, PP)
            for b in sync: b.simulate(DT, PP)
            halo_exchange_local(sync)
            ref = mono.pos
            e1 = max(np.abs(b.pos - ref[b.ownedIds]).max() for b in bodies)
            e2 = max(np.abs(b.pos - ref[b.ownedIds]).max() for b in sync)
            bad = [(i, int((np.abs(b.pos - ref[b.ownedIds]).max(axis=1) > 0).sum()), b.info.owned_particles) for i, b in enumerate(bodies)]
            if it in (0, 1, 2, 5, 11): print("parts", parts, prec, "substep", it+1, "group err", e1, "sync-path err", e2, "bad verts per part", bad, "nbound", [b.info.num_neighbours for b in bodies])

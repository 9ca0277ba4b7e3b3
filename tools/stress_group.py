"""Stress the asynchronous halo choreography (in-process group transport): many cold and warm trials, deep queues."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, group_step_n, make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0)
DT = (1/60)/20
bad = 0; total = 0
for n, nzf in ((10, 2), (16, 3)):
    v, t = make_lattice(n, nz=nzf*n, y0=0.05)
    plane = (n+1)*(n+1)
    for parts in (2, 3, 5):
        owner = np.minimum((np.arange(len(v)) // plane) * parts // (nzf*n+1), parts-1).astype(np.int32)
        for prec in ("precise", "fast"):
            for trial in range(4):
                mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=prec)
                g = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=prec, part_count=parts, part_index=p, vert_owner=owner) for p in range(parts)]
                for it in range(4):
                    mono.simulateSubsteps(20, DT, PP)
                    group_step_n(g, 20, DT, PP)
                ref = mono.pos
                err = max(float(np.abs(x.pos - ref[x.ownedIds]).max()) for x in g)
                ok = err == 0.0 if prec == "precise" else err < 1e-4
                total += 1; bad += (not ok)
                if not ok: print("FAIL n", n, "parts", parts, prec, "trial", trial, "err", err)
print("stress: %d/%d configurations ok" % (total - bad, total))

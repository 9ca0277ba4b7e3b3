"""Host enqueue cost vs GPU time of the eager halo choreography (in-process group as a stand-in for the RCCL path)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tetsim_amd import SoftBodyHIP, group_step_n, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0)
dt = (1/60)/20
for parts in (1, 2):
    v, t = make_lattice(55, nz=55 * parts)
    plane = 56 * 56
    owner = np.minimum((np.arange(len(v)) // plane) // 55, parts - 1).astype(np.int32)
    if parts == 1:
        b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
        step = lambda: b.simulateSubsteps(20, dt, pp); sync = b.sync
    else:
        g = [SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", part_count=parts, part_index=p, vert_owner=owner) for p in range(parts)]
        step = lambda: group_step_n(g, 20, dt, pp); sync = lambda: [x.sync() for x in g]
    step(); sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20): step()
        th = time.perf_counter() - t0
        sync(); tt = time.perf_counter() - t0
        print("parts %d: host enqueue %.1f us per substep (all partitions), wall %.1f us per substep; %d tets total -> %.1f M tet-solves/s" %
              (parts, th / 400 * 1e6, tt / 400 * 1e6, len(t), len(t) * 400 / tt / 1e6))

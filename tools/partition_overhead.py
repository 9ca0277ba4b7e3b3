"""What one rank of a slab decomposition costs per substep, measured on ONE GPU: a 1 M-tet slab stepped in an in-process
group together with a thin (1-cell) neighbour slab, so the big partition runs the full halo choreography (interior /
boundary tiles, ghost tets, transfers) while the neighbour takes almost no GPU time.  Compare with the monolithic body."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, group_step_n, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0)
dt = (1 / 60) / 20
thin = int(os.environ.get("THIN", "1"))
v, t = make_lattice(55, nz=55 + thin)
plane = 56 * 56
owner = ((np.arange(len(v)) // plane) >= 56 - (0 if thin else 0)).astype(np.int32)   # z-planes 0..55 -> part 0 (55 cells), the rest -> part 1
mono_v, mono_t = make_lattice(55)
b = SoftBodyHIP(mono_v, mono_t, None, dict(pp), solver="polar", precision="fast")
g = [SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", part_count=2, part_index=p, vert_owner=owner, ref_fixed_bounds=False) for p in range(2)]
print("partition 0: %d local tets (%d owned), partition 1: %d local tets" % (g[0].info.local_elems, g[0].info.owned_elems, g[1].info.local_elems))
for name, step, sync in (("monolithic", lambda: b.simulateSubsteps(20, dt, pp), b.sync),
                         ("slab + thin neighbour", lambda: group_step_n(g, 20, dt, pp), lambda: [x.sync() for x in g])):
    step(); sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20): step()
        th = time.perf_counter() - t0
        sync(); tt = time.perf_counter() - t0
        print("%-24s host enqueue %.1f us, wall %.1f us per substep" % (name, th / 400 * 1e6, tt / 400 * 1e6), flush=True)

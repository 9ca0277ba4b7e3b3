import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import SoftBodyHIP, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0)
v, t = make_lattice(55); dt = (1/60)/20
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
b.simulateSubsteps(20, dt, pp); b.sync()
for mode in ("graph", "eager", "graph", "eager"):
    b.sync(); t0 = time.perf_counter()
    for _ in range(30):
        if mode == "graph": b.simulateSubsteps(20, dt, pp)
        else:
            for _ in range(20): b.simulate(dt, pp)
    th = time.perf_counter() - t0
    b.sync(); tt = time.perf_counter() - t0
    print("%s: host enqueue %.3f ms/frame, wall %.3f ms/frame (%.1f us/substep)" % (mode, th/30*1e3, tt/30*1e3, tt/30/20*1e6))

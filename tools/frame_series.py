import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(55); DT = (1/60)/20
b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
b.sync()
ts = []
for f in range(120):
    t0 = time.perf_counter(); b.simulateSubsteps(20, DT, PP); b.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-frame ms (each frame synced):", " ".join("%.3f" % x for x in ts[:12]), "...", " ".join("%.3f" % x for x in ts[40:46]), "...", " ".join("%.3f" % x for x in ts[-6:]))
# batches without intermediate sync
for n in (5, 10, 20, 40, 80, 160):
    b.sync(); t0 = time.perf_counter()
    for _ in range(n): b.simulateSubsteps(20, DT, PP)
    b.sync(); print("batch of %3d frames: %.4f ms/frame" % (n, (time.perf_counter() - t0) / n * 1e3))
# is the slow phase of frames ~3-30 the governor or the data?  A second, fresh body right after the first one (GPU warm):
b2 = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
b2.sync()
ts = []
for f in range(60):
    t0 = time.perf_counter(); b2.simulateSubsteps(20, DT, PP); b2.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print("fresh body on a warm GPU:", " ".join("%.3f" % x for x in ts[:14]), "...", " ".join("%.3f" % x for x in ts[-4:]))

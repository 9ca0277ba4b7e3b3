"""Development: is the frame kernel of the loaded library (TETSIM_HIP_LIB) bit-equal to the stepwise kernels, and how fast?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tetsim_amd import SoftBodyHIP
G = os.path.join(ROOT, "tests", "golden")
v = np.fromfile(os.path.join(G, "dragon_verts.f32"), dtype="<f4").reshape(-1, 3); t = np.fromfile(os.path.join(G, "dragon_tets.i32"), dtype="<i4").reshape(-1, 4)
v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
dt = (1 / 60) / 20
a = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
ok = True
t0 = time.time()
for n in (20, 3, 20, 20):
    a.simulateSubsteps(n, dt, pp)
    for _ in range(n):
        b.simulate(dt, pp)
    try:
        a.sync()
    except Exception as e:
        print("SYNC ERROR:", str(e)[:120]); ok = False; break
    ok = ok and np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32))
    if time.time() - t0 > 20:
        print("too slow: waits are timing out"); ok = False; break
ms = min(a.timeSubsteps(20, dt, pp) for _ in range(10))
print("%-28s mode %d  bit-equal %s  %.2f us per substep" % (os.path.basename(os.environ.get("TETSIM_HIP_LIB", "libtetsim_hip.so")), a.info.fused_particle_pass, ok, ms / 20 * 1e3))

"""Quick timing probe (development aid): per-substep time of both solvers on the 1M-tet lattice."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice, measure_copy_bandwidth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(n)
dt = (1.0 / 60.0) / 20
print("lattice", n, "tets", len(t), "verts", len(v))
for nbytes in (64 << 20, 1 << 30):
    print("copy bw %5d MiB: %.0f GB/s" % (nbytes >> 20, measure_copy_bandwidth(nbytes, 20)))
fastonly = len(sys.argv) > 2
for prec in (("fast",) if fastonly else ("precise", "fast")):
    t0 = time.time()
    b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision=prec)
    t1 = time.time()
    b.simulateSubsteps(20, dt, pp); b.sync()
    ms = min(b.timeSubsteps(20, dt, pp) for _ in range(5))
    pr = b.profile(20, dt, pp)
    print("polar %-7s create %.2fs  frame(20) %.3f ms  -> %.1f M tet-solves/s | tet %.1f us vertex %.1f us per substep"
          % (prec, t1 - t0, ms, len(t) * 20 / ms / 1e3, pr["tet_ms"] / 20 * 1e3, pr["vertex_ms"] / 20 * 1e3))
    b.close()
for prec, order in (() if fastonly else [(p, o) for o in ("coloured", "clustered") for p in ("precise", "fast")]):
    t0 = time.time()
    b = SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision=prec, order=order)
    t1 = time.time()
    b.simulateSubsteps(20, dt, pp); b.sync()
    ms = min(b.timeSubsteps(20, dt, pp) for _ in range(5))
    print("neohk %-9s %-7s create %.2fs  launches/substep %d  frame(20) %.3f ms -> %.1f M tet-solves/s"
          % (order, prec, t1 - t0, b.info.num_levels, ms, len(t) * 20 / ms / 1e3))
    b.close()

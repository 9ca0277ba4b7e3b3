"""Neo-Hookean GS timing on the 1 M-tet lattice (development aid): ms per 20-substep frame for each order / precision."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import SoftBodyHIP, make_lattice

n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
orders = sys.argv[2].split(",") if len(sys.argv) > 2 else ["coloured", "clustered"]
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(n)
dt = (1.0 / 60.0) / 20
for order in orders:
    for prec in ("precise", "fast"):
        b = SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision=prec, order=order)
        b.simulateSubsteps(20, dt, pp); b.sync()
        ms = min(b.timeSubsteps(20, dt, pp) for _ in range(7))
        print("neohookean %-9s %-7s launches/substep %2d  frame(20) %.3f ms = %.1f us/substep -> %.1f M tet-solves/s"
              % (order, prec, b.info.num_levels, ms, ms * 50, len(t) * 20 / ms / 1e3), flush=True)
        b.close()

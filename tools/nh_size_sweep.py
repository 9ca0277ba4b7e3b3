"""Neo-Hookean Gauss-Seidel over body sizes (Kuhn lattices of n^3 cells): us per substep and M tet-solves/s by order and arithmetic, and
which bodies take the single-workgroup launch (TetSimInfo.fused_particle_pass == 4).  python tools/nh_size_sweep.py [n ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import SoftBodyHIP
from tetsim_amd.lattice import make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
ns = [int(a) for a in sys.argv[1:]] or [5, 8, 12, 15, 16, 20, 32, 55]
n_sub, dt = 10, (1 / 60) / 10
print("%4s %9s %9s  %s" % ("n", "tets", "particles", "us per substep (M tet-solves/s): coloured precise | coloured fast | clustered precise | clustered fast"))
for n in ns:
    v, t = make_lattice(n, y0=0.02)
    cells = []
    for order, prec in (("coloured", "precise"), ("coloured", "fast"), ("clustered", "precise"), ("clustered", "fast")):
        b = SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision=prec, order=order)
        for _ in range(3):
            b.simulateSubsteps(n_sub, dt, pp)
        b.sync()
        ms = min(b.timeSubsteps(n_sub, dt, pp) for _ in range(5))
        cells.append("%8.1f (%7.1f)%s" % (ms * 1e3 / n_sub, len(t) * n_sub / ms / 1e3, "*" if b.info.fused_particle_pass == 4 else " "))
        del b
    print("%4d %9d %9d  %s" % (n, len(t), len(v), " | ".join(cells)), flush=True)
print("* = one single-workgroup launch per call, every particle in LDS")

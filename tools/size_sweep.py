"""Polar Jacobi FAST over body sizes: Kuhn lattices of n^3 cells from 750 tets to 2 M, which path the library picks for each
(TetSimInfo.fused_particle_pass: 0 two kernels per substep (5: and the whole tetsim_step_n call as one launch of them), 1 one fused kernel, 2 one persistent launch per call with one lane per tet,
3 the same with four lanes per tet) and what a substep costs, falling and lying on the floor.  python tools/size_sweep.py [n ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import SoftBodyHIP
from tetsim_amd.lattice import make_lattice
PATH = {0: "two kernels", 1: "fused kernel", 2: "frame, 1 lane", 3: "frame, 4 lanes", 5: "call in 1 launch"}
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
ns = [int(a) for a in sys.argv[1:]] or [5, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 55, 70]
n_sub, dt = 20, (1 / 60) / 20
print("%4s %9s %-15s %28s %28s" % ("n", "tets", "path", "falling: us/substep  M/s", "floor: us/substep  M/s"))
for n in ns:
    row = []
    for y0 in (0.5, 0.01):          # 0.5 m: falls for ~19 frames; 1 cm: on the floor within the warm-up
        v, t = make_lattice(n, y0=y0)
        b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
        for _ in range(8 if y0 < 0.1 else 2):
            b.simulateSubsteps(n_sub, dt, pp)
        b.sync()
        ms = min(b.timeSubsteps(n_sub, dt, pp) for _ in range(8))
        row.append((ms * 1e3 / n_sub, len(t) * n_sub / ms / 1e3))
        path = PATH.get(b.info.fused_particle_pass, "?")
        del b
    print("%4d %9d %-15s %14.2f %13.1f %14.2f %13.1f" % (n, len(t), path, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)

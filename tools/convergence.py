"""BASELINE config 4: graph-coloured Neo-Hookean Gauss-Seidel vs polar-decomposition Jacobi on the same 1 M-tet lattice,
same start (dropped from rest onto the floor), same substep count.  Residuals are evaluated on the host from the
positions both solvers return: mean |det F - 1| (the reference's volError analogue, Softbody.js:163) and the
deviatoric residual | sqrt(tr(F^T F)) - sqrt(3) |."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice

n = int(sys.argv[1]) if len(sys.argv) > 1 else 55
v, t = make_lattice(n, y0=0.02)  # 2 cm above the floor: contact within the first frames
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
dt, sub = (1.0 / 60.0) / 20, 20
Dm = (v[t[:, 1:]] - v[t[:, :1]]).astype(np.float64).transpose(0, 2, 1)  # columns = rest edges
Dm_inv = np.linalg.inv(Dm)


def residuals(pos):
    Ds = (pos[t[:, 1:]] - pos[t[:, :1]]).astype(np.float64).transpose(0, 2, 1)
    F = Ds @ Dm_inv
    det = np.linalg.det(F)
    dev = np.sqrt((F * F).sum(axis=(1, 2)))
    return np.abs(det - 1).mean(), np.abs(det - 1).max(), np.abs(dev - np.sqrt(3.0)).mean()


bodies = {
    "polar Jacobi (FAST, blocked)": SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast"),
    "Neo-Hookean coloured GS (PRECISE)": SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision="precise", order="coloured"),
    "Neo-Hookean coloured GS (FAST)": SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision="fast", order="coloured"),
    "Neo-Hookean clustered GS (PRECISE)": SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision="precise", order="clustered"),
    "Neo-Hookean clustered GS (FAST)": SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision="fast", order="clustered"),
}
print("lattice %d^3 cells: %d tets, %d particles; %d substeps/frame, dt = 1/%d s" % (n, len(t), len(v), sub, round(1 / dt)))
print("%-36s %7s %12s %12s %12s %10s" % ("solver", "frames", "mean|detF-1|", "max|detF-1|", "mean dev res", "ms/frame"))
for name, b in bodies.items():
    for frames in (1, 5, 30):
        done = getattr(b, "_frames", 0)
        t0 = time.perf_counter()
        for _ in range(frames - done):
            b.simulateSubsteps(sub, dt, pp)
        b.sync()
        ms = (time.perf_counter() - t0) / max(1, frames - done) * 1e3
        b._frames = frames
        m, mx, dv = residuals(b.pos)
        print("%-36s %7d %12.3e %12.3e %12.3e %10.3f" % (name, frames, m, mx, dv, ms))

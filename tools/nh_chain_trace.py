"""Where a colour of the persistent clustered sweep spends its time (TETSIM_NH_CHAIN_TRACE=1: s_memtime stamps of workgroup 0,
printed by tetsim_destroy; 100 MHz ticks x 24 = shader cycles at 2.4 GHz -- s_memtime counts the constant 100 MHz clock on gfx950)."""
import os, sys
os.environ["TETSIM_NH_CHAIN_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import SoftBodyHIP, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(55)
dt = (1.0 / 60.0) / 20
for prec in sys.argv[1:] or ["fast"]:
    b = SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision=prec, order="clustered")
    for _ in range(3): b.simulateSubsteps(20, dt, pp)
    b.sync()
    print(prec, "ms per 20 substeps:", min(b.timeSubsteps(20, dt, pp) for _ in range(5)), flush=True)
    b.close()

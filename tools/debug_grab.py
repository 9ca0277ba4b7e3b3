import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice
from oracle import OraclePJ
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1/60)/20
v, t = make_lattice(4, y0=0.02)
for prec, kw in (("precise", {}), ("fast", {}), ("fast", dict(gather=True))):
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=prec, **kw); orc = OraclePJ(v, t, PP)
    gid = 7
    out = []
    for step in range(150):
        if step == 30:
            body.setGrab(gid, [0.3, 0.8, 0.1]); orc.setGrab(gid, [0.3, 0.8, 0.1])
        if 30 < step < 90:
            p = [0.3 + 0.002 * step, 0.8, 0.1]
            body.moveGrabbed(p); orc.setGrab(gid, p)
        if step == 90:
            body.endGrab(); orc.endGrab()
        body.simulate(DT, PP); orc.simulate(DT, PP)
        if step % 10 == 9 or step in (30, 31, 32):
            out.append("%d:%.2g" % (step, np.abs(body.pos - orc.pos).max()))
    print(prec, kw, " ".join(out))

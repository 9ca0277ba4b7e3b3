#!/usr/bin/env python3
"""A/B of the clustered FAST Gauss-Seidel sweep on the 1 M-tet lattice: ONE launch per substep (particles handed on with their stamp,
nh_kernels.inc: nh_sweep1_kernel) against one launch per colour (TETSIM_NH_ONE_LAUNCH=0), alternating, best of 7 frames each.

    python tools/nh_one_launch_ab.py [cells] [rounds]     ->  profiles/r06_nh_one_launch.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import SoftBodyHIP, make_lattice  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 55
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(cells)
dt = (1.0 / 60.0) / 20
print("lattice %d^3 cells, %d tets; clustered FAST, frames of 20 substeps as graph replays, best of 7" % (cells, len(t)))
end = {}
for r in range(rounds):
    for mode in ("one launch", "per colour"):
        if mode == "per colour":
            os.environ["TETSIM_NH_ONE_LAUNCH"] = "0"
        else:
            os.environ.pop("TETSIM_NH_ONE_LAUNCH", None)
        b = SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision="fast", order="clustered")
        b.simulateSubsteps(20, dt, pp)
        b.sync()
        ms = sorted(b.timeSubsteps(20, dt, pp) for _ in range(7))
        print("%-11s colours %2d  frame(20) best %.3f ms median %.3f ms = %.1f us/substep -> %.1f M tet-solves/s" % (
            mode, b.info.num_levels, ms[0], ms[3], ms[0] * 50, len(t) * 20 / ms[0] / 1e3), flush=True)
        end[mode] = b.pos
        b.close()
print("bit-equal after 160 substeps:", bool(np.array_equal(end["one launch"].view(np.uint32), end["per colour"].view(np.uint32))))

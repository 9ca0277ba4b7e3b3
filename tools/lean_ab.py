#!/usr/bin/env python3
"""In-run A/B of the tet record formulations on the 1 M-tet lattice (polar Jacobi FAST): the reference's (148 B/tet), the constant
rest shape (100 B/tet) and the lean state (TETSIM_FLAG_LEAN_STATE, 92 B/tet) -- wall clock of the driver's frames (falling), wall clock
and per-launch events on the floor, and how far the lean trajectory / its recovered quaternions sit from the default's.

    python tools/lean_ab.py [cells] [rounds]      ->  profiles/r06_lean_state_ab.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import SoftBodyHIP, make_lattice  # noqa: E402

PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20
SUB = 20
BYTES = {"reference": 148.0, "constant-rest": 100.0, "lean": 92.0}
KW = {"reference": {}, "constant-rest": dict(constant_rest_shape=True), "lean": dict(lean_state=True)}


def frames(body, n):
    body.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        body.simulateSubsteps(SUB, DT, PP)
    body.sync()
    return (time.perf_counter() - t0) / (n * SUB) * 1e6


def main():
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 55
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    v, t = make_lattice(cells)
    nt = len(t)
    print("lattice %d^3 cells, %d tets, %d particles; %d rounds, modes alternate inside a round" % (cells, nt, len(v), rounds))
    print("(/2k = TETSIM_PJ_ONE_LAUNCH=0: two kernels per substep in the graphs; else tiles + particles in ONE launch per substep)")
    print("%-14s %-6s %9s %9s %9s %9s %9s %9s" % ("mode", "exit", "fall us", "G/s", "floor us", "tet us", "vert us", "tet TB/s"))
    end = {}
    for r in range(rounds):
        for mode, ref_exit, one in [(m, x, o) for m in ("reference", "lean", "constant-rest") for x in (False, True) for o in (True, False)]:
            if True:
                if one:
                    os.environ.pop("TETSIM_PJ_ONE_LAUNCH", None)
                else:
                    os.environ["TETSIM_PJ_ONE_LAUNCH"] = "0"     # two kernels per substep inside tetsim_step_n (rounds 1-5)
                b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", ref_rotation_exit=ref_exit, **KW[mode])
                frames(b, 5)
                fall = frames(b, 20)
                frames(b, 15)           # the body reaches the floor around frame 19 and settles
                floor = frames(b, 10)
                p = b.profile(SUB * 3, DT, PP)
                tet_us = p["tet_ms"] / p["tet_launches"] * 1e3
                vert_us = p["vertex_ms"] / max(p["vertex_launches"], 1) * 1e3
                print("%-14s %-6s %9.2f %9.2f %9.2f %9.2f %9.2f %9.3f" % (mode + ("" if one else " /2k"), "1e-9" if ref_exit else "1e-6", fall, nt / fall / 1e3, floor, tet_us, vert_us,
                                                                         BYTES[mode] * nt / (tet_us * 1e-6) / 1e12), flush=True)
                if r == 0 and not ref_exit and one:
                    q = b.quats
                    qq = np.empty_like(q)
                    qq[b.localTets] = q
                    end[mode] = (b.pos, qq)
                b.close()
    for mode in ("lean", "constant-rest"):
        dp = np.abs(end[mode][0] - end["reference"][0]).max()
        dq = np.abs(end[mode][1] - end["reference"][1]).max()
        print("%s vs reference after %d substeps (contact from ~380 on): max |dpos| %.3g m, max |dquat| %.3g, |q| - 1: %.3g" % (
            mode, 53 * SUB, dp, dq, np.abs(np.linalg.norm(end[mode][1], axis=1) - 1.0).max()))


if __name__ == "__main__":
    main()

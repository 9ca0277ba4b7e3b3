#!/usr/bin/env python3
"""Step 1 of golden generation: write the synthetic mesh fixtures and cases.json.

Run from the repo root:  python tests/golden/make_cases.py
Then:                    bash tests/golden/make_golden.sh      (needs /root/reference + node)

Meshes are DATA produced by this repo's own generator (tetsim_amd/lattice.py); the Dragon arrays
are exported from the reference's data file by make_golden.mjs (data, not code).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tetsim_amd.lattice import make_lattice, save_mesh  # noqa: E402


def greedy_colour_sort(tets, nv):
    """Greedy vertex-disjoint colouring; returns tets stably sorted by colour."""
    used = [set() for _ in range(nv)]
    colour = np.zeros(len(tets), dtype=np.int64)
    for e, t in enumerate(tets):
        c = 0
        while any(c in used[v] for v in t):
            c += 1
        colour[e] = c
        for v in t:
            used[v].add(c)
    order = np.argsort(colour, kind="stable")
    return tets[order], int(colour.max()) + 1


DEFAULT = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1.0 / 100000.0,
               volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])


def main():
    cases = []
    # --- lattice 4^3 cells dropped from just above the floor, original cell-major order
    v, t = make_lattice(4, y0=0.05)
    save_mesh(os.path.join(HERE, "lat4"), v, t)
    tc, ncol = greedy_colour_sort(t, len(v))
    save_mesh(os.path.join(HERE, "lat4c"), v, tc)
    # --- lattice 2^3 with an appended coplanar (zero-volume) tet: exercises Softbody.js:391-394
    v2, t2 = make_lattice(2, y0=0.3)
    flat = np.array([[0, 1, 3, 4]], dtype=np.int32)  # four corners of one cell face: det == 0 exactly
    save_mesh(os.path.join(HERE, "lat2degen"), v2, np.concatenate([t2, flat]))
    # --- particles only (no tets): free fall + floor
    save_mesh(os.path.join(HERE, "notets"), v2[:5], np.zeros((0, 4), dtype=np.int32))

    ts = dict(timeScale=1.0, timeStep=1.0 / 60.0)
    cases.append(dict(name="dragon", mesh="dragon", params=DEFAULT, numSubsteps=10, **ts, nsteps=1200,
                      dumps=[1, 10, 100], hashes=[600, 1200], grab=[]))
    cases.append(dict(name="dragon_sub5", mesh="dragon", params=DEFAULT, numSubsteps=5, **ts, nsteps=60,
                      dumps=[60], hashes=[], grab=[]))
    cases.append(dict(name="dragon_grab", mesh="dragon", params=DEFAULT, numSubsteps=10, **ts, nsteps=60,
                      dumps=[20, 60], hashes=[],
                      grab=[dict(at=5, op="start", p=[0.1, 1.2, -0.1])] +
                           [dict(at=6 + i, op="move", p=[0.1 + 0.01 * i, 1.2 + 0.02 * i, -0.1]) for i in range(25)] +
                           [dict(at=40, op="end")]))
    soft = dict(DEFAULT, devCompliance=1.0 / 10000.0, volCompliance=1.0e-6, gravity=-5.0, friction=200.0,
                worldBounds=[-0.3, -1.0, -0.4, 0.35, 2.0, 0.3])
    cases.append(dict(name="dragon_soft", mesh="dragon", params=soft, numSubsteps=7, **ts, nsteps=80,
                      dumps=[80], hashes=[], grab=[]))
    cases.append(dict(name="lat4", mesh="lat4", params=DEFAULT, numSubsteps=10, **ts, nsteps=300,
                      dumps=[1, 50, 300], hashes=[], grab=[]))
    cases.append(dict(name="lat4c", mesh="lat4c", params=DEFAULT, numSubsteps=10, **ts, nsteps=300,
                      dumps=[1, 50, 300], hashes=[], grab=[], ncolours=ncol))
    cases.append(dict(name="lat2degen", mesh="lat2degen", params=dict(DEFAULT, volCompliance=1.0e-6),
                      numSubsteps=10, **ts, nsteps=40, dumps=[1, 40], hashes=[], grab=[]))
    cases.append(dict(name="notets", mesh="notets", params=DEFAULT, numSubsteps=10, **ts, nsteps=200,
                      dumps=[200], hashes=[], grab=[]))
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()

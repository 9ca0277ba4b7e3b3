#!/usr/bin/env python3
"""Step 1 of the POLAR (WebGL solver) golden generation: writes cases_gpu.json and the `hub` mesh fixture.

    python tests/golden/make_cases_gpu.py        (needs scipy; the lat4 / dragon meshes come from make_cases.py / make_golden.sh)
    bash tests/golden/make_golden_gpu.sh          (needs /root/reference, node and Mesa's swrast_dri.so)

Grab scripts use two event kinds of make_golden_gpu.mjs: `start_id` (set grabId directly and start from the current
position of the particle the reference's collision pass will actually pin, `follow`) and `move_rel` (offsets from that
start), so that the drag is gentle -- a teleporting grab turns last-ulp differences between GLSL implementations into
centimetres within a few substeps and pins nothing down.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

PARAMS = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-05, volCompliance=0.0,
              worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])


def load(name):
    v = np.fromfile(os.path.join(HERE, name + "_verts.f32"), dtype="<f4").reshape(-1, 3)
    t = np.fromfile(os.path.join(HERE, name + "_tets.i32"), dtype="<i4").reshape(-1, 4)
    return v, t


def pinned_by(grab_id, nt, nv):
    """Particles the reference pins for grab_id (SoftbodyGPU.js:335-338), via the library's GPU-free helper."""
    from tetsim_amd import _capi
    out = (C.c_int32 * 2)()
    _capi.lib().tetsim_prep_ref_grab_texels(int(grab_id), nt, nv, out)
    return [x for x in out if x >= 0]


def grab_id_for(particle, nt, nv):
    """A grabId whose (single) pinned particle is `particle`."""
    for g in range(nv):
        if pinned_by(g, nt, nv) == [particle]:
            return g
    raise SystemExit("no grabId pins particle %d alone" % particle)


def make_hub():
    """A Delaunay ball around a hub particle: its valence (44) exceeds the reference's 36 scatter slots."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(77)
    n = 70
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    shell = d * [0.35, 0.3, 0.33] * (0.85 + 0.3 * rng.random((n, 1)))
    pts = np.vstack([[0, 0, 0], shell, rng.normal(size=(12, 3)) * 0.08]).astype(np.float32) + np.float32([0.0, 0.6, 0.0])
    tets = Delaunay(pts.astype(np.float64)).simplices.astype(np.int32)
    dd = pts[tets[:, 1:]].astype(np.float64) - pts[tets[:, :1]].astype(np.float64)
    vol = np.linalg.det(dd) / 6
    flip = vol < 0
    tets[flip] = tets[flip][:, [0, 1, 3, 2]]
    tets = tets[np.abs(vol) > 2e-6]
    used = np.unique(tets)
    remap = np.full(len(pts), -1, np.int32)
    remap[used] = np.arange(len(used), dtype=np.int32)
    v, t = np.ascontiguousarray(pts[used]), np.ascontiguousarray(remap[tets])
    v.astype("<f4").tofile(os.path.join(HERE, "hub_verts.f32"))
    t.astype("<i4").tofile(os.path.join(HERE, "hub_tets.i32"))
    return v, t


def main():
    base = dict(params=PARAMS, numSubsteps=20, timeScale=1.0, timeStep=1 / 60, grab=[], dumpTables=False)
    dv, dt = load("dragon")
    lv, lt = load("lat4")
    top = int(np.argsort(-dv[:, 1])[0])                       # the Dragon's topmost particle (1071)
    g_dragon = grab_id_for(top, len(dt), len(dv))
    g_lat = grab_id_for(99, len(lt), len(lv))                  # lattice top-face particle 99
    drag = dict(PARAMS, friction=100.0, gravity=-20.0)
    cases = [
        dict(base, name="lat4", mesh="lat4", nsteps=300, dumps=[1, 2, 20, 100, 300], dumpTables=True),
        dict(base, name="dragon", mesh="dragon", nsteps=600, dumps=[1, 20, 200, 600]),
        dict(base, name="dragon_grab", mesh="dragon", nsteps=60, dumps=[10, 60],
             grab=[{"at": 3, "op": "start_id", "id": g_dragon, "follow": top}] +
                  [{"at": s, "op": "move_rel", "d": [0.002 * (s - 3), 0.002 * (s - 3), 0.0]} for s in range(4, 41)] + [{"at": 41, "op": "end"}]),
        dict(name="lat4_drag", mesh="lat4", params=drag, numSubsteps=10, timeScale=1.0, timeStep=1 / 60, nsteps=200, dumps=[60, 200], dumpTables=False,
             grab=[{"at": 20, "op": "start_id", "id": g_lat, "follow": 99}] +
                  [{"at": s, "op": "move_rel", "d": [0.003 * (s - 20), 0.001 * (s - 20), 0.0]} for s in range(21, 121)] + [{"at": 121, "op": "end"}]),
    ]
    make_hub()
    cases.append(dict(name="hub", mesh="hub", params=PARAMS, numSubsteps=20, timeScale=1.0, timeStep=1 / 60, nsteps=150, dumps=[1, 20, 150],
                      grab=[], dumpTables=True))
    # a lattice big enough for the blocked formulation to cut it into 41 workgroup tiles (12^3 cells = 10,368 tets, 2,197
    # particles), dropped 2 mm onto the floor: contact from substep ~24 on.  Quaternions are kept for the last dump only (size).
    from tetsim_amd import make_lattice
    v12, t12 = make_lattice(12, y0=0.002)
    v12.astype("<f4").tofile(os.path.join(HERE, "lat12_verts.f32"))
    t12.astype("<i4").tofile(os.path.join(HERE, "lat12_tets.i32"))
    cases.append(dict(base, name="lat12", mesh="lat12", nsteps=40, dumps=[1, 20, 40], quatDumps=[40]))
    with open(os.path.join(HERE, "cases_gpu.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote cases_gpu.json (%d cases), the hub mesh and the lat12 mesh" % len(cases))


if __name__ == "__main__":
    main()

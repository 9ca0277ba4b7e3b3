"""Pins the POLAR oracle (oracle/tetsim_oracle.c section G) to the REFERENCE's WebGL solver.

The golden vectors were recorded by running the reference's own SoftbodyGPU.js + MultiTargetGPUComputationRenderer.js +
vendored three.js under Node, with the GL calls executed by Mesa's software rasteriser (tests/golden/make_golden_gpu.sh,
oracle/glsl_ref/).  The GLSL runs in plain IEEE f32 there (softpipe), as on desktop WebGL; sin(), inversesqrt() and
division may differ from glibc in the last ulp, so: bit-exact where no such operation has fed back yet (host-built
tables, the first substep), a stated absolute position tolerance in metres after that.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_f32, load_mesh, sha16
from oracle import OraclePJ

# absolute position error (m) allowed at each recorded substep; measured values are ~3x below
TOL = {
    "lat4": {1: 0.0, 2: 2.5e-7, 20: 2e-6, 100: 2e-5, 300: 1e-4},
    "dragon": {1: 0.0, 20: 1e-5, 200: 1e-4, 600: 1e-3},
    "dragon_grab": {10: 5e-6, 60: 5e-5},
    "lat4_drag": {60: 5e-5, 200: 2e-4},
    "hub": {1: 0.0, 20: 1e-5, 150: 2e-4},     # Delaunay ball around a hub particle of valence >> 36 (slots silently dropped)
    "lat12": {1: 0.0, 20: 4e-6, 40: 5e-6},    # 10,368 tets (41 workgroup tiles of the blocked kernel), floor contact from substep ~24; observed 1.3e-6 / 1.7e-6
}


@pytest.fixture(scope="module")
def glsl_golden():
    with open(os.path.join(GOLDEN, "golden_gpu.json")) as f:
        g = json.load(f)
    with open(os.path.join(GOLDEN, "cases_gpu.json")) as f:
        cases = {c["name"]: c for c in json.load(f)}
    return g, cases


def replay(body, c, g, on_dump, set_grab, end_grab):
    """Drive `body` through case c exactly as make_golden_gpu.mjs drove the reference (grab ids come from the golden:
    the reference's startGrab searches the never-refreshed edge-mesh positions, SoftbodyGPU.js:646,692-706)."""
    gids = iter(g["grabIds"])
    gid, start = -1, None
    for step in range(1, c["nsteps"] + 1):
        for ev in c["grab"]:
            if ev["at"] != step:
                continue
            if ev["op"] == "start":
                gid = next(gids)
                set_grab(gid, ev["p"])
            elif ev["op"] == "move":
                set_grab(gid, ev["p"])
            elif ev["op"] == "start_id":      # gentle grab: see make_golden_gpu.mjs
                gid = next(gids)
                assert gid == ev["id"]
                start = g["grabStartPos"]
                set_grab(gid, start)
            elif ev["op"] == "move_rel":
                set_grab(gid, [start[k] + ev["d"][k] for k in range(3)])
            else:
                end_grab()
        body.simulate(g["dt"], c["params"])
        if step in c["dumps"]:
            on_dump(step)


def test_generated_by_the_reference(glsl_golden):
    g, _ = glsl_golden
    assert g["three"] == "160" and "softpipe" in g["gl"]["renderer"]


@pytest.mark.parametrize("mesh", ["lat4", "dragon", "hub"])
def test_host_tables_bit_exact(mesh, glsl_golden):
    """Scatter table (incl. the `<= 0.0` quirk, SoftbodyGPU.js:563-577) and 1/V (:579-589) as the reference built them."""
    g, _ = glsl_golden
    v, t = load_mesh(mesh)
    o = OraclePJ(v, t, {"density": 1000.0}, slot_quirk=True)
    gc = g["cases"][mesh]
    assert (gc["numParticles"], gc["numElems"]) == (len(v), len(t))
    slots = np.ascontiguousarray(o.slots, dtype="<i4")
    import hashlib
    assert hashlib.sha256(slots.tobytes()).hexdigest()[:16] == gc["slots"]
    assert sha16(o.invRestVolume) == gc["invRestVolume"]
    if mesh in ("lat4", "hub"):
        assert np.array_equal(slots.ravel(), np.fromfile(os.path.join(GOLDEN, mesh + "_gpu_slots.i32"), dtype="<i4"))
        assert np.array_equal(o.invRestVolume.view(np.uint32), load_f32(mesh + "_gpu_invRestVolume.f32").view(np.uint32))
    if mesh == "hub":   # the point of this mesh: a particle with far more incident tets than the 36 slots
        valence = np.bincount(t.ravel(), minlength=len(v))
        assert valence.max() > 36 and (slots >= 0).sum(axis=1).max() == 36
    # the library's own host prep builds the same table
    from tetsim_amd import _capi
    import ctypes as C
    L = _capi.lib()
    out = np.empty(len(v) * 36, dtype=np.int32)
    dropped = C.c_uint32()
    tt = np.ascontiguousarray(t, dtype=np.int32)
    assert L.tetsim_prep_slot_table(tt.ctypes.data_as(C.POINTER(C.c_int32)), len(t), len(v), 1, out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(dropped)) == 0
    assert np.array_equal(out, slots.ravel())


@pytest.mark.parametrize("name", ["lat4", "dragon", "dragon_grab", "lat4_drag", "hub", "lat12"])
def test_trajectory_tracks_the_reference_glsl(name, glsl_golden):
    g, cases = glsl_golden
    c, gc = cases[name], g["cases"][name]
    v, t = load_mesh(c["mesh"])
    o = OraclePJ(v, t, c["params"], slot_quirk=True, ref_grab_texel=True)
    seen = []

    def on_dump(step):
        gp = load_f32(f"{name}_gpu_pos_{step}.f32").reshape(-1, 3)
        gv = load_f32(f"{name}_gpu_vel_{step}.f32").reshape(-1, 3)
        has_q = step in c.get("quatDumps", c["dumps"])   # big cases keep the quaternion dump of the last step only (its hash always)
        gq = load_f32(f"{name}_gpu_quat_{step}.f32").reshape(-1, 4) if has_q else None
        assert sha16(gp) == gc["steps"][str(step)]["pos"]
        tol = TOL[name][step]
        if tol == 0.0:
            assert np.array_equal(o.pos.view(np.uint32), gp.view(np.uint32)), (name, step)
            assert np.array_equal(o.vel.view(np.uint32), gv.view(np.uint32)), (name, step)
            assert sha16(o.quats) == gc["steps"][str(step)]["quat"], (name, step)
        else:
            err = np.abs(o.pos - gp).max()
            assert err <= tol, "%s step %d: observed %.3g m, allowed %.3g m" % (name, step, err, tol)
            assert np.abs(o.vel - gv).max() <= 2.0 * tol / gc["dt"], (name, step)   # v = dx/dt: position ulps / dt
            if has_q:
                assert np.abs(o.quats - gq).max() <= max(50 * tol, 1e-5), (name, step)
        seen.append(step)

    replay(o, c, gc, on_dump, o.setGrab, o.endGrab)
    assert seen == c["dumps"]


def test_grab_texel_quirk_is_real(glsl_golden):
    """Without the reference's indexFromUV mapping the grabbed trajectory is a different one (0.3 m away)."""
    g, cases = glsl_golden
    c, gc = cases["dragon_grab"], g["cases"]["dragon_grab"]
    v, t = load_mesh("dragon")
    o = OraclePJ(v, t, c["params"], slot_quirk=True, ref_grab_texel=False)
    errs = {}
    replay(o, c, gc, lambda s: errs.__setitem__(s, np.abs(o.pos - load_f32(f"dragon_grab_gpu_pos_{s}.f32").reshape(-1, 3)).max()), o.setGrab, o.endGrab)
    assert errs[10] > 0.1


def test_ref_grab_texels_host_mapping():
    """tetsim_prep_ref_grab_texels against a brute-force f32 evaluation of SoftbodyGPU.js:335-338 over every texel."""
    from tetsim_amd import _capi
    import ctypes as C
    L = _capi.lib()
    f = np.float32
    for nt in (1, 2, 5, 17, 384, 1000, 3840, 250000, 998250):
        R = int(np.ceil(np.sqrt(nt)))
        nv = min(R * R, max(1, int(nt / 5.5)))
        i = np.arange(R * R)
        ux = ((i % R).astype(f) + f(0.5)) / f(R)
        uy = ((i // R).astype(f) + f(0.5)) / f(R)
        idx = ((ux * f(R - 1)).astype(np.int32) + ((uy * f(R - 1)) * f(R)).astype(np.int32)).astype(f)
        ids = list(range(min(nv, 200))) + [int(x) for x in np.random.RandomState(nt).randint(0, nv, 100)]
        for gid in ids:
            want = [int(k) for k in np.nonzero(idx == f(gid))[0] if k < nv][:2]
            out = (C.c_int32 * 2)()
            assert L.tetsim_prep_ref_grab_texels(gid, nt, nv, out) == 0
            assert sorted(x for x in out if x >= 0) == sorted(want), (nt, gid)

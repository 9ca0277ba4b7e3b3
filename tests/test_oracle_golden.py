"""Pins the CPU oracle (oracle/tetsim_oracle.c, section A) to the REFERENCE: bit-exact equality with
golden vectors recorded by importing /root/reference/src/Softbody.js under Node (tests/golden/make_golden.*)."""
import numpy as np
import pytest

from conftest import case_dt, load_f32, load_mesh, sha16
from oracle import OracleNH

CASES = ["dragon", "dragon_sub5", "dragon_grab", "dragon_soft", "lat4", "lat4c", "lat2degen", "notets"]


@pytest.mark.parametrize("mesh", ["dragon", "lat4", "lat4c", "lat2degen"])
def test_init_physics_bit_exact(mesh, golden):
    v, t = load_mesh(mesh)
    o = OracleNH(v, t, {"density": 1000.0})
    assert np.array_equal(o.invMass.view(np.uint32), load_f32(mesh + "_invMass.f32").view(np.uint32))
    assert np.array_equal(o.invRestPose.view(np.uint32), load_f32(mesh + "_invRestPose.f32").view(np.uint32))
    assert np.array_equal(o.invRestVolume.view(np.uint32), load_f32(mesh + "_invRestVolume.f32").view(np.uint32))


@pytest.mark.parametrize("name", CASES)
def test_trajectory_bit_exact(name, golden):
    gold, cases = golden
    c, g = cases[name], gold[name]
    if name == "dragon":
        nsteps = 600  # the 1200-substep hash is covered by test_dragon_long
    else:
        nsteps = c["nsteps"]
    v, t = load_mesh(c["mesh"])
    o = OracleNH(v, t, c["params"])
    dt = case_dt(c)
    assert dt == g["dt"]
    grab_ids = []
    for step in range(1, nsteps + 1):
        for ev in c["grab"]:
            if ev["at"] != step:
                continue
            if ev["op"] == "start":
                grab_ids.append(o.startGrab(*ev["p"]))
            elif ev["op"] == "move":
                o.moveGrabbed(*ev["p"])
            else:
                o.endGrab()
        o.simulate(dt, c["params"])
        key = str(step)
        if key in g["steps"]:
            gs = g["steps"][key]
            assert sha16(o.pos) == gs["pos"], (name, step)
            assert sha16(o.vel) == gs["vel"], (name, step)
            assert sha16(o.prevPos) == gs["prev"], (name, step)
            if gs["volError"] is not None:
                assert o.volError == gs["volError"], (name, step)
            else:
                assert np.isnan(o.volError)
        if step in c["dumps"]:
            assert np.array_equal(o.pos.ravel().view(np.uint32), load_f32(f"{name}_pos_{step}.f32").view(np.uint32))
            assert np.array_equal(o.vel.ravel().view(np.uint32), load_f32(f"{name}_vel_{step}.f32").view(np.uint32))
    assert grab_ids == g["grabIds"]


def test_dragon_long(golden):
    """1200 substeps (2 s of simulated time incl. floor contact): SURVEY.md §8(c) known answer."""
    gold, cases = golden
    c, g = cases["dragon"], gold["dragon"]
    v, t = load_mesh("dragon")
    o = OracleNH(v, t, c["params"])
    dt = case_dt(c)
    for _ in range(1200):
        o.simulate(dt, c["params"])
    assert sha16(o.pos) == g["steps"]["1200"]["pos"] == "9f52c76cdd7223f0"
    assert float(np.sum(o.pos.ravel().astype(np.float64))) == pytest.approx(g["steps"]["1200"]["sumPos"], rel=1e-12)


@pytest.mark.parametrize("name,step", [("dragon", 100), ("dragon_sub5", 60), ("dragon_soft", 80), ("lat4", 300), ("lat4c", 300),
                                       ("lat2degen", 40), ("notets", 200)])
def test_js_port_bit_exact(name, step, golden):
    """oracle/nh_port.js (the JavaScript restatement timed as the single-thread "JS CPU path" on the GPU box) reproduces the
    golden vectors recorded from the reference's Softbody.js bit for bit -- same cases as the C oracle, minus the grab script."""
    import json
    import shutil
    import subprocess
    from conftest import GOLDEN, ROOT
    node = shutil.which("node")
    if node is None:
        pytest.skip("node is not installed on this host")
    gold, cases = golden
    c, g = cases[name], gold[name]
    assert not c["grab"] and str(step) in g["steps"]
    import os
    pp = dict(c["params"], timeScale=c["timeScale"], timeStep=c["timeStep"])
    out = subprocess.run([node, os.path.join(ROOT, "oracle", "nh_port.js"), "--verts", os.path.join(GOLDEN, c["mesh"] + "_verts.f32"),
                          "--tets", os.path.join(GOLDEN, c["mesh"] + "_tets.i32"), "--substeps", str(step),
                          "--per-frame", str(c["numSubsteps"]), "--params", json.dumps(pp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert r["pos_sha16"] == g["steps"][str(step)]["pos"], (name, step)
    want = g["steps"][str(step)]["volError"]
    if want is None:
        assert r["vol_error"] is None          # NaN serialises as null
    else:
        assert r["vol_error"] == want


def test_vertex_normals_golden_is_threejs_computeVertexNormals():
    """The golden normals (three.js r160 inside the reference, tests/golden/make_golden.mjs) equal a plain restatement of
    BufferGeometry.computeVertexNormals + normalizeNormals: f64 face normals from f32 positions, accumulated in triangle order
    with an f32 store per add, normalised in f64, stored f32.  This pins what the device kernel (skin_kernels.hip) follows."""
    import os
    import numpy as np
    from conftest import GOLDEN, load_f32
    pos = load_f32("dragon_vispos_10.f32").reshape(-1, 3)
    tri = np.fromfile(os.path.join(GOLDEN, "dragon_vistris.u16"), dtype="<u2").astype(np.int64).reshape(-1, 3)
    ref = load_f32("dragon_visnormal_10.f32").reshape(-1, 3)
    P = pos.astype(np.float64)
    cb, ab = P[tri[:, 2]] - P[tri[:, 1]], P[tri[:, 0]] - P[tri[:, 1]]
    face = np.stack([cb[:, 1] * ab[:, 2] - cb[:, 2] * ab[:, 1], cb[:, 2] * ab[:, 0] - cb[:, 0] * ab[:, 2],
                     cb[:, 0] * ab[:, 1] - cb[:, 1] * ab[:, 0]], axis=1)
    n = np.zeros_like(pos)
    for t in range(len(tri)):
        for v in tri[t]:
            n[v] = (n[v].astype(np.float64) + face[t]).astype(np.float32)
    N = n.astype(np.float64)
    length = np.sqrt(N[:, 0] * N[:, 0] + N[:, 1] * N[:, 1] + N[:, 2] * N[:, 2])
    length[length == 0] = 1.0
    out = (N * (1.0 / length)[:, None]).astype(np.float32)
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))

"""Row (f)-1: embedded visual-mesh skinning on the device vs the reference's updateVisMesh (Softbody.js:259-277)."""
import numpy as np
import pytest

from conftest import load_f32, load_mesh, sha16, within
from oracle import OraclePJ
from tetsim_amd import SoftBodyHIP

pytestmark = pytest.mark.gpu
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])


def test_neohookean_skinning_bit_exact_vs_reference(golden):
    """29,800 embedded vertices of the Dragon after 10 substeps: bit-identical to the reference's visMesh positions."""
    v, t = load_mesh("dragon")
    vis = load_f32("dragon_vis.f32").reshape(-1, 4)
    body = SoftBodyHIP(v, t, None, dict(PP), vis, solver="neohookean", precision="precise")
    assert body.numVisVerts == 29800
    dt = (1.0 * (1.0 / 60.0)) / 10
    for _ in range(10):
        body.simulate(dt, PP)
    got = body.visualPositions()
    ref = load_f32("dragon_vispos_10.f32").reshape(-1, 3)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert sha16(got) == "8df79c236ba69d61"


def test_vertex_normals_bit_exact_vs_threejs(golden):
    """`visMesh.geometry.computeVertexNormals()` (Softbody.js:273: the bulk of the CPU path's frame once the solve is fast) on
    the device: 29,800 normals over 59,657 triangles after 10 substeps, bit-identical to what three.js r160 produced inside
    the reference (golden recorded by tests/golden/make_golden.mjs)."""
    import os
    from conftest import GOLDEN
    v, t = load_mesh("dragon")
    vis = load_f32("dragon_vis.f32").reshape(-1, 4)
    tris = np.fromfile(os.path.join(GOLDEN, "dragon_vistris.u16"), dtype="<u2").astype(np.int32).reshape(-1, 3)
    assert len(tris) == 59657
    body = SoftBodyHIP(v, t, None, dict(PP), vis, tris, solver="neohookean", precision="precise")   # visTriIds as the reference's 6th argument
    dt = (1.0 * (1.0 / 60.0)) / 10
    for _ in range(10):
        body.simulate(dt, PP)
    got = body.visualVertexNormals()
    ref = load_f32("dragon_visnormal_10.f32").reshape(-1, 3)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert sha16(got) == "77a1f9768ab27ed3"   # golden.json: vis.dragon_visnormal_10
    # degenerate input: a triangle that lists a vertex twice has a zero face normal; a vertex without triangles keeps (0,0,0)
    small = SoftBodyHIP(v, t, None, dict(PP), vis[:8], solver="neohookean", precision="precise")
    small.setVisualTriangles(np.array([[0, 1, 2], [2, 1, 3], [4, 4, 5]], np.int32))
    n = small.visualVertexNormals()
    assert np.isfinite(n).all() and np.all(n[6] == 0) and np.all(n[7] == 0) and np.all(n[4] == 0)
    assert np.abs(np.linalg.norm(n[:4], axis=1) - 1.0).max() < 1e-6
    with pytest.raises(Exception):
        small.setVisualTriangles(np.array([[0, 1, 2]], np.int32))        # attached once
    other = SoftBodyHIP(v, t, None, dict(PP), vis[:8], solver="neohookean", precision="precise")
    with pytest.raises(Exception):
        other.setVisualTriangles(np.array([[0, 1, 8]], np.int32))        # id outside the visual mesh


def _threejs_vertex_normals(pos, tri):
    """BufferGeometry.computeVertexNormals + normalizeNormals of three.js r160, restated (pinned against the reference's own
    output by tests/test_oracle_golden.py::test_vertex_normals_golden_is_threejs_computeVertexNormals)."""
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
    return (N * (1.0 / length)[:, None]).astype(np.float32)


def test_vertex_normals_on_the_polar_path():
    """SoftbodyGPU.js:687 (`physicsParams.computeNormals`): the same device routine on the polar solver's skinned positions
    (internally renumbered particles, FAST), bit-identical to three.js's algorithm applied to those positions."""
    import os
    from conftest import GOLDEN
    v, t = load_mesh("dragon")
    vis = load_f32("dragon_vis.f32").reshape(-1, 4)
    tris = np.fromfile(os.path.join(GOLDEN, "dragon_vistris.u16"), dtype="<u2").astype(np.int32).reshape(-1, 3)
    body = SoftBodyHIP(v, t, None, dict(PP), vis, tris, solver="polar", precision="fast")
    body.simulateSubsteps(40, (1.0 / 60.0) / 20, PP)
    pos = body.visualPositions()
    got = body.visualVertexNormals()
    ref = _threejs_vertex_normals(pos, tris.astype(np.int64))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("precision", ["precise", "fast", "fast-lean"])
def test_polar_skinning_and_normals(precision):
    """Vertex-shader formula of SoftbodyGPU.js:429-440 evaluated on the device: positions from the (internally
    renumbered) particles, normals rotated by the quaternion of the vertex's tet (tile-ordered in FAST)."""
    v, t = load_mesh("dragon")
    vis = load_f32("dragon_vis.f32").reshape(-1, 4)
    rng = np.random.default_rng(7)
    n0 = rng.standard_normal((len(vis), 3)).astype(np.float32)
    n0 /= np.linalg.norm(n0, axis=1, keepdims=True)
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision.split("-")[0], lean_state=precision == "fast-lean")   # lean: the quaternions are recovered from the carried shape
    body.setVisualMesh(vis, n0)
    orc = OraclePJ(v, t, PP)
    dt = (1.0 / 60.0) / 20
    for _ in range(40):
        body.simulate(dt, PP)
        orc.simulate(dt, PP)
    pos, nrm = body.visualPositions(with_normals=True)
    p, q = orc.pos, orc.quats
    tn = vis[:, 0].astype(np.int64)
    b = vis[:, 1:4]
    b3 = np.float32(1.0) - ((b[:, 0] + b[:, 1]) + b[:, 2])
    c = t[tn]
    ref = ((p[c[:, 0]] * b[:, :1] + p[c[:, 1]] * b[:, 1:2]) + p[c[:, 2]] * b[:, 2:3]) + p[c[:, 3]] * b3[:, None]
    within("skinning %s vs oracle positions" % precision, np.abs(pos - ref).max(), 1e-6 if precision == "precise" else 1e-4)
    qq = q[tn]
    qv, w = qq[:, :3], qq[:, 3:]
    refn = n0 + 2.0 * np.cross(qv, np.cross(qv, n0) + w * n0)
    within("skinning %s vs oracle normals" % precision, np.abs(nrm - refn).max(), 1e-5 if precision == "precise" else 1e-3)
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1.0).max() < 1e-4
